#!/bin/bash
# round 3, call 15: weight-gradient launches of 2 / 3 text layers merged (ETP_FLUSH_EVERY), same-box A/B
set -x
O=gpurun_out/c15; mkdir -p $O
export TMPDIR=/tmp
B="--steps 100 --warmup 20 --no-cpu-baseline --no-optimizer --no-roofline"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B > $O/bench_$name.json 2> $O/bench_$name.err; python -c "import json; d=json.load(open('$O/bench_$name.json')); print('RESULT $name', d['value'], d['ms_per_step'], d['loss'])"; }
run default_a A=1
run every2 ETP_FLUSH_EVERY=2
run every3 ETP_FLUSH_EVERY=3
run default_b A=1
run every2_b ETP_FLUSH_EVERY=2
run every9 ETP_FLUSH_EVERY=9
