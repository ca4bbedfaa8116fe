#!/bin/bash
# Round-4 GPU call 1: un-profiled chain waits (device stamps) in both host issue orders + same-box baseline bench line
set -x
O=gpurun_out/r4c1; mkdir -p $O
export TMPDIR=/tmp
B="--steps 100 --warmup 20 --no-cpu-baseline --no-optimizer --no-roofline"
timeout 300 python bench.py $B > $O/bench_base.json 2> $O/bench_base.err; tail -c 400 $O/bench_base.json | head -c 400; echo
timeout 300 python tools/chain_waits.py --steps 24 --out $O/chain_waits.txt > $O/chain_waits.log 2>&1; echo "rc $?"; cat $O/chain_waits.txt
ETP_CHAIN_FIRST=1 timeout 300 python tools/chain_waits.py --steps 24 --out $O/chain_waits_chainfirst.txt > $O/chain_waits_cf.log 2>&1; echo "rc $?"; cat $O/chain_waits_chainfirst.txt
ETP_CHAIN_FIRST=1 timeout 300 python bench.py $B > $O/bench_chainfirst.json 2> $O/bench_cf.err; python -c "import json; d=json.load(open('$O/bench_chainfirst.json')); print('RESULT chainfirst', d['value'], d['ms_per_step'])"
python -c "import json; d=json.load(open('$O/bench_base.json')); print('RESULT base', d['value'], d['ms_per_step'])"
