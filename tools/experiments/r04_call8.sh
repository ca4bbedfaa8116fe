#!/bin/bash
set -x
O=gpurun_out/r4c8; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_planner_gpu.py tests/test_baseline_shapes_gpu.py tests/test_mm32_gpu.py -x -q --tb=short > $O/tests.log 2>&1; echo "rc tests $?"; tail -15 $O/tests.log
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q --tb=short > $O/ops_tests.log 2>&1; echo "rc ops $?"; tail -5 $O/ops_tests.log
B="--steps 100 --warmup 20 --no-cpu-baseline --no-optimizer --no-roofline"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B > $O/bench_$name.json 2> $O/bench_$name.err; python -c "import json; d=json.load(open('$O/bench_$name.json')); print('RESULT $name', d['value'], d['ms_per_step'], d['loss'])"; }
run new A=1
run new_b A=1
