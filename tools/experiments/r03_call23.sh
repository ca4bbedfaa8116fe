#!/bin/bash
# round 3, call 23: the free-running data-parallel issue order -- parity, then what it costs on one GPU without collectives
set -x
O=gpurun_out/c23; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_planner_gpu.py tests/test_dp_gpu.py -m gpu -q --tb=short -x -k "data_parallel_issue or two_rank_planner or self_launches" > $O/tests.log 2>&1; echo "rc tests $?"; tail -12 $O/tests.log
B="--steps 100 --warmup 20 --no-cpu-baseline --no-optimizer --no-roofline"
run() { name=$1; shift; timeout 300 python bench.py $B "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python -c "import json; d=json.load(open('$O/bench_$name.json')); print('RESULT $name', d['value'], d['ms_per_step'], d['loss'])"; }
run default_a
run dp_overlapped_a --dp-schedule overlapped
run dp_joined_a --dp-schedule joined
run default_b
run dp_overlapped_b --dp-schedule overlapped
