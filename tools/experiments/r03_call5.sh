#!/bin/bash
# Round-3 GPU call 5: the 256x128 eight-wavefront tile class (chain GEMMs with N >= 2304 and the grouped weight gradients)
set -x
O=gpurun_out/c5; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
T="timeout 600"
$T python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gemm" --tb=short > $O/ops.log 2>&1; echo "rc ops $?"; tail -3 $O/ops.log
B="--steps 100 --warmup 20 --no-cpu-baseline --no-optimizer"
run() { name=$1; shift; env "$@" $T python bench.py $B > $O/bench_$name.json 2> $O/bench_$name.err; python -c "import json; d=json.load(open('$O/bench_$name.json')); print('RESULT $name', d['value'], d['ms_per_step'])"; }
run default A=1
run no256 ETP_GEMM_256=0
run group256only ETP_GEMM_TILE_DUMMY=1 ETP_GROUP_TILE=256s2 ETP_GEMM_256=1
run group256s3 ETP_GROUP_TILE=256s3
run skipwgrad ETP_SKIP_WGRAD=1
run default2 A=1
$T python tools/chain_budget.py --seq > $O/chain_budget.txt 2>&1
$T python tools/gemm_phase_probe.py > $O/gemm_phases.txt 2>&1
SWEEP_256=1 $T python tools/gemm_sweep.py > $O/gemm_sweep_256.json 2> $O/gemm_sweep_256.err
GEMM_GROUP_ONLY=1 GEMM_GROUP_TABLE=1 $T python tools/gemm_bench.py > $O/gemm_group_table.txt 2>&1
timeout 900 python -m pytest tests/test_planner_gpu.py tests/test_baseline_shapes_gpu.py -m gpu -q -x --tb=short -k "golden or benchmarked_shape_b32 or train_mode_step" > $O/planner.log 2>&1; echo "rc planner $?"; tail -3 $O/planner.log
