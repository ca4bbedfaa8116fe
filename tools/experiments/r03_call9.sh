#!/bin/bash
# Round-3 GPU call 9: co-scheduling experiments on the final default (64x64 class for N = 768)
set -x
O=gpurun_out/c9; mkdir -p $O
export TMPDIR=/tmp
T="timeout 300"
B="--steps 100 --warmup 20 --no-cpu-baseline --no-optimizer --no-roofline"
run() { name=$1; shift; env "$@" $T python bench.py $B > $O/bench_$name.json 2> $O/bench_$name.err; python -c "import json; d=json.load(open('$O/bench_$name.json')); print('RESULT $name', d['value'], d['ms_per_step'])"; }
run default_a A=1
run all64s3 ETP_GEMM_TILE=64s3
run all64s4 ETP_GEMM_TILE=64s4
run flushmid ETP_FLUSH_MID=1
run noprio ETP_STREAM_PRIO=0
run group128s3 ETP_GROUP_TILE=128s3
run default_b A=1
run skipwgrad ETP_SKIP_WGRAD=1
