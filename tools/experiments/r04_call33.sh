#!/bin/bash
# final binary (LN backward grid changed after the evidence run): goldens, determinism screen, headline line + kernel stats again
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r4c33; mkdir -p $O
timeout 600 python -m pytest tests/test_planner_gpu.py tests/test_ops_gpu.py -q --tb=short -k "golden or layer_norm or issue_order or layer_ranges" 2>&1 | tail -3
timeout 300 python tools/determinism_screen.py > $O/determinism_screen.txt 2>&1; tail -6 $O/determinism_screen.txt
python __graft_entry__.py smoke 2>&1 | grep "^smoke" | tee $O/smoke.log
(cd /tmp && timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o r -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-optimizer --no-roofline > $R/$O/bench_under_rocprof.json 2> $R/$O/prof.err)
cp $O/prof/r_kernel_stats.csv $O/bench_kernel_stats.csv; cp $O/bench_kernel_stats.csv profiles/r04_bench_kernel_stats.csv; rm -rf $O/prof
timeout 420 python bench.py > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().split("\n")[-1]); r = d["roofline"]
print("bench", d["value"], d["ms_per_step"], r["achieved"], r["frac"], r["achieved_isolated"], r["device_span_us"], r.get("rocprof_avg_launch_us"), r["traffic"], d["optimizer"]["train_iteration"]["ms"])
PY
