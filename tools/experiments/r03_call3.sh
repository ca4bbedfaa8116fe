#!/bin/bash
# Round-3 GPU call 3: warp-specialised GEMM classes (4 loader + 4 compute wavefronts), Z prefetch, LN backward row pairs,
# text weight cast off the chain: correctness, tile sweep, same-box A/B.
set -x
O=gpurun_out/c3; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
OLD=$R/etpnav_amd/libetpnav_hip_r02.so
T="timeout 600"
$T python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gemm or layer_norm" --tb=short > $O/ops.log 2>&1; echo "rc ops $?"; tail -3 $O/ops.log
B="--steps 100 --warmup 20 --no-cpu-baseline --no-optimizer"
$T python bench.py $B > $O/bench_new.json 2> $O/bench_new.err; echo "rc bench $?"
ETP_GEMM_WS=0 $T python bench.py $B > $O/bench_nows.json 2> $O/bench_nows.err
ETP_TXT_CAST_SPLIT=0 $T python bench.py $B > $O/bench_nosplit.json 2> $O/bench_nosplit.err
ETP_LIB=$OLD $T python bench.py $B > $O/bench_r02.json 2> $O/bench_r02.err
for f in new nows nosplit r02; do python -c "import json,sys; d=json.load(open('$O/bench_$f.json')); print('$f', d['value'], d['ms_per_step'])"; done
$T python tools/chain_budget.py --seq > $O/chain_budget_new.txt 2>&1
$T python tools/gemm_phase_probe.py > $O/gemm_phases.txt 2>&1
SWEEP_SPEC=1 $T python tools/gemm_sweep.py > $O/gemm_sweep_spec.json 2> $O/gemm_sweep_spec.err
timeout 900 python -m pytest tests/test_planner_gpu.py -m gpu -q -x --tb=short -k "golden or fresh or train_mode or ranges" > $O/planner.log 2>&1; echo "rc planner $?"; tail -3 $O/planner.log
