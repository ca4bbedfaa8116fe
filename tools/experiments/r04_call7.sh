#!/bin/bash
set -x
O=gpurun_out/r4c7; mkdir -p $O
export TMPDIR=/tmp
B="--steps 100 --warmup 20 --no-cpu-baseline --no-optimizer --no-roofline"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B > $O/bench_$name.json 2> $O/bench_$name.err; python -c "import json; d=json.load(open('$O/bench_$name.json')); print('RESULT $name', d['value'], d['ms_per_step'], d['loss'])"; }
run xcd1 A=1
run xcd0 ETP_GEMM_XCD=0
run xcd1_b A=1
run xcd0_b ETP_GEMM_XCD=0
ETP_GEMM_XCD=0 timeout 200 python tools/mm32_probe.py > $O/probe_xcd0.json 2> $O/probe.err; cat $O/probe_xcd0.json
