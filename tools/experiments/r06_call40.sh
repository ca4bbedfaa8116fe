#!/bin/bash
# round 6, GPU call 40: the library with the compile-time epilogue specialisations of the NN products (calls 36-39) as the shipped default: the full GPU suite in the
# driver's form, then the evidence that changes with it -- kernel trace + stats of the bench command, GEMM phases, chain stamps, the bench lines of every workload.
# (PMC traffic of the dominant kernel: unchanged kernel, files of call 15 stand.)
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c40; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
( timeout 1500 python -m pytest tests -m gpu -x -q --durations=10 2>&1 | grep -v "amdgpu.ids\|Gloo\|socket.cpp" ) > $O/gpu_suite.log
tail -16 $O/gpu_suite.log | cut -c1-200
python __graft_entry__.py smoke 2>&1 | grep "^smoke" | tee $O/smoke.log
T="timeout 420"
(cd /tmp && $T rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o r -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-optimizer --no-roofline > $R/$O/bench_under_rocprof.json 2> $R/$O/prof.err)
python tools/timeline.py $O/prof/r_kernel_trace.csv --steps 20 > $O/timeline.txt 2>&1
cp $O/prof/r_kernel_stats.csv $O/bench_kernel_stats.csv; cp $O/bench_kernel_stats.csv profiles/r06_bench_kernel_stats.csv
rm -rf $O/prof
$T python tools/gemm_phase_probe.py > $O/gemm_phases.txt 2> $O/gemm_phases.err
$T python tools/chain_waits.py --steps 24 --out $O/chain_waits.txt > /dev/null 2> $O/chain_waits.err
$T python bench.py > $O/bench.json 2> $O/bench.err
for wl in c4 c5 sap; do $T python bench.py --workload $wl --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err; done
$T python bench.py --dtype fp32 --no-cpu-baseline > $O/bench_fp32.json 2> $O/bench_fp32.err
for f in bench bench_c4 bench_c5 bench_sap bench_fp32; do python - <<PY
import json
try:
    d = json.loads(open("$O/$f.json").read().strip().split("\n")[-1])
    r = d.get("roofline") or {}
    print("$f", d["value"], d["ms_per_step"], "roofline", r.get("kernel"), r.get("achieved"), r.get("frac"), "traffic", r.get("traffic"), "rocprof", r.get("rocprof_avg_launch_us"), (d.get("optimizer") or {}).get("train_iteration"))
except Exception as e:
    print("$f FAILED", e)
PY
done
