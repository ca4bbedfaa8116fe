#!/bin/bash
# Round-3 GPU call 6: leaf-work politeness -- shorter-lived weight-gradient workgroups, earlier flushes
set -x
O=gpurun_out/c6; mkdir -p $O
export TMPDIR=/tmp
T="timeout 300"
B="--steps 100 --warmup 20 --no-cpu-baseline --no-optimizer --no-roofline"
run() { name=$1; shift; env "$@" $T python bench.py $B > $O/bench_$name.json 2> $O/bench_$name.err; python -c "import json; d=json.load(open('$O/bench_$name.json')); print('RESULT $name', d['value'], d['ms_per_step'])"; }
run default A=1
run group64s3 ETP_GROUP_TILE=64s3
run group64s4 ETP_GROUP_TILE=64s4
run flush2 ETP_WGRAD_FLUSH=2
run flush2_64s3 ETP_WGRAD_FLUSH=2 ETP_GROUP_TILE=64s3
run group128s3 ETP_GROUP_TILE=128s3
run nosplitcast ETP_TXT_CAST_SPLIT=0
run nogroup ETP_WGRAD_GROUP=0
run default2 A=1
