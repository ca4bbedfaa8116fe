#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5c19; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_variants_gpu.py tests/test_planner_gpu.py tests/test_baseline_shapes_gpu.py -m gpu -q -k "three_stream or bf16 or rollout" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -6 $O/tests.log
NOISE_B=3 timeout 300 python tools/experiments/r05_b1_noise.py 2>&1 | grep -v amdgpu > $O/noise_b3_new.txt; cat $O/noise_b3_new.txt | cut -c1-200
B="--steps 100 --warmup 20 --no-cpu-baseline --no-optimizer --no-roofline"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B > $O/bench_$name.json 2> $O/bench_$name.err; python -c "import json; d=json.load(open('$O/bench_$name.json')); print('RESULT $name', d['value'], d['ms_per_step'])"; }
run new A=1
run base ETP_LIB=$PWD/etpnav_amd/build/libetp_base.so
