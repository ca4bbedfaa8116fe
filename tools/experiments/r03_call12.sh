#!/bin/bash
# round 3, call 12: communicator self-test on the real RCCL binding (world 1), bench line with the device-span cross-check
set -x
O=gpurun_out/c12; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dp_gpu.py -m gpu -q --tb=short > $O/dp_tests.log 2>&1; echo "rc dptests $?"; tail -5 $O/dp_tests.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc bench $?"; tail -3 $O/bench.err; python -c "import json; d=json.load(open('$O/bench.json')); print('RESULT', d['value'], d['ms_per_step']); r=d['roofline']; print({k:v for k,v in r.items() if 'note' not in k and 'source' not in k})"
