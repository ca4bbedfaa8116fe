#!/bin/bash
# round 3, call 17: hardware-queue count sensitivity (GPU_MAX_HW_QUEUES), same-box A/B
set -x
O=gpurun_out/c17; mkdir -p $O
export TMPDIR=/tmp
B="--steps 100 --warmup 20 --no-cpu-baseline --no-optimizer --no-roofline"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B > $O/bench_$name.json 2> $O/bench_$name.err; python -c "import json; d=json.load(open('$O/bench_$name.json')); print('RESULT $name', d['value'], d['ms_per_step'], d['loss'])"; }
run default_a A=1
run q8 GPU_MAX_HW_QUEUES=8
run q2 GPU_MAX_HW_QUEUES=2
run default_b A=1
run q16 GPU_MAX_HW_QUEUES=16
