#!/bin/bash
# round 3, call 13: balanced (stream-K) grouped weight-gradient launches -- parity, then A/B in the step
set -x
O=gpurun_out/c13; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_planner_gpu.py -m gpu -q --tb=short -x -k "balanced" > $O/sk_tests.log 2>&1; echo "rc sktests $?"; tail -15 $O/sk_tests.log
B="--steps 100 --warmup 20 --no-cpu-baseline --no-optimizer --no-roofline"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B > $O/bench_$name.json 2> $O/bench_$name.err; python -c "import json; d=json.load(open('$O/bench_$name.json')); print('RESULT $name', d['value'], d['ms_per_step'], d['loss'])"; }
run default_a A=1
run sk_a ETP_GROUP_SK=1
run default_b A=1
run sk_b ETP_GROUP_SK=1
ETP_GROUP_SK=1 timeout 300 python tools/gemm_phase_probe.py > $O/gemm_phases_sk.txt 2>&1; head -12 $O/gemm_phases_sk.txt
