#!/bin/bash
# round 3, call 11: the new GPU tests, the rollout bench (N1), the default bench line with the committed PMC / rocprof cross-checks
set -x
O=gpurun_out/c11; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_baseline_shapes_gpu.py -m gpu -q --tb=short -k "race_screen or batched_rollout or tile_classes" > $O/new_tests.log 2>&1; echo "rc newtests $?"; tail -5 $O/new_tests.log
timeout 600 python tools/rollout_bench.py > $O/rollout_bench.json 2> $O/rollout_bench.err; echo "rc rollout $?"; cat $O/rollout_bench.json; tail -3 $O/rollout_bench.err
timeout 600 python tools/rollout_bench.py --B 32 --T 5,15 > $O/rollout_bench_b32.json 2> $O/rollout_bench_b32.err; cat $O/rollout_bench_b32.json
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc bench $?"; python -c "import json; d=json.load(open('$O/bench.json')); print('RESULT', d['value'], d['ms_per_step']); print(json.dumps(d['roofline'], indent=1))"
