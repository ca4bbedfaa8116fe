#!/bin/bash
# round 6, GPU call 42: the shipped library with BOTH sets of epilogue specialisations (NN products: call 39; TN weight gradients in overwrite mode: call 41): operator, neighbour,
# planner-golden, issue-order and determinism-screen tests, then the kernel trace of the bench command and the bench lines (the full suite ran on the NN-only library in call 40;
# the GPU budget of the round does not hold another full run).
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c42; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
python __graft_entry__.py smoke 2>&1 | grep "^smoke" | tee $O/smoke.log
( timeout 900 python -m pytest tests/test_mm32_gpu.py tests/test_ops_gpu.py tests/test_neighbours_gpu.py tests/test_optim_gpu.py -q -x 2>&1 | grep -v amdgpu.ids | tail -2 ) | tee $O/ops.log
( timeout 900 python -m pytest tests/test_planner_gpu.py tests/test_variants_gpu.py -q -x -k "golden or graph_replay or issue_order or three_stream or layer_ranges" 2>&1 | grep -v amdgpu.ids | tail -2 ) | tee $O/planner.log
T="timeout 420"
(cd /tmp && $T rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o r -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-optimizer --no-roofline > $R/$O/bench_under_rocprof.json 2> $R/$O/prof.err)
cp $O/prof/r_kernel_stats.csv $O/bench_kernel_stats.csv; cp $O/bench_kernel_stats.csv profiles/r06_bench_kernel_stats.csv
python tools/timeline.py $O/prof/r_kernel_trace.csv --steps 20 > $O/timeline.txt 2>&1
rm -rf $O/prof
$T python tools/gemm_phase_probe.py > $O/gemm_phases.txt 2> $O/gemm_phases.err
$T python bench.py > $O/bench.json 2> $O/bench.err
for wl in c4 c5; do $T python bench.py --workload $wl --no-cpu-baseline --no-optimizer > $O/bench_$wl.json 2> $O/bench_$wl.err; done
for f in bench bench_c4 bench_c5; do python - <<PY
import json
try:
    d = json.loads(open("$O/$f.json").read().strip().split("\n")[-1])
    r = d.get("roofline") or {}
    print("$f", d["value"], d["ms_per_step"], "roofline", r.get("kernel"), r.get("achieved"), r.get("frac"), r.get("avg_launch_us"), "rocprof", r.get("rocprof_avg_launch_us"), "iso", r.get("achieved_isolated"))
except Exception as e:
    print("$f FAILED", e)
PY
done
