#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5c20; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_variants_gpu.py tests/test_planner_gpu.py tests/test_ops_gpu.py -m gpu -q -s -k "three_stream or fp32_step_matches or bf16_step_close or pano or embed" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; grep "tensors above\|passed\|failed" $O/tests.log | cut -c1-300
B="--steps 100 --warmup 20 --no-cpu-baseline --no-optimizer --no-roofline"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B > $O/bench_$name.json 2> $O/bench_$name.err; python -c "import json; d=json.load(open('$O/bench_$name.json')); print('RESULT $name', d['value'], d['ms_per_step'])"; }
run new A=1
run base ETP_LIB=$PWD/etpnav_amd/build/libetp_base.so
run new2 A=1
