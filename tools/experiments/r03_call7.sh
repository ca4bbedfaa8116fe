#!/bin/bash
# Round-3 GPU call 7: persistent (grid-capped) grouped weight-gradient kernel
set -x
O=gpurun_out/c7; mkdir -p $O
export TMPDIR=/tmp
T="timeout 300"
$T python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gemm_group" --tb=short > $O/ops.log 2>&1; echo "rc ops $?"; tail -3 $O/ops.log
B="--steps 100 --warmup 20 --no-cpu-baseline --no-optimizer --no-roofline"
run() { name=$1; shift; env "$@" $T python bench.py $B > $O/bench_$name.json 2> $O/bench_$name.err; python -c "import json; d=json.load(open('$O/bench_$name.json')); print('RESULT $name', d['value'], d['ms_per_step'])"; }
run default A=1
run persist256 ETP_GROUP_PERSIST=256
run persist192 ETP_GROUP_PERSIST=192
run persist128 ETP_GROUP_PERSIST=128
run persist256_noprio ETP_GROUP_PERSIST=256 ETP_STREAM_PRIO=0
run default2 A=1
