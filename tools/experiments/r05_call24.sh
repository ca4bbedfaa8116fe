#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5c24; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_variants_gpu.py tests/test_planner_gpu.py -m gpu -q -k "three_stream or fp32_step_matches or bf16_step_close or issue_order or layer_ranges or b32" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -4 $O/tests.log
B="--steps 100 --warmup 20 --no-cpu-baseline --no-optimizer --no-roofline"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B > $O/bench_$name.json 2> $O/bench_$name.err; python -c "import json; d=json.load(open('$O/bench_$name.json')); print('RESULT $name', d['value'], d['ms_per_step'])"; }
run new A=1
run base ETP_LIB=$PWD/etpnav_amd/build/libetp_base.so
B="--workload c5 $B"; run c5_new A=1; run c5_base ETP_LIB=$PWD/etpnav_amd/build/libetp_base.so
B="--workload c4 --steps 40 --warmup 10 --no-cpu-baseline --no-optimizer --no-roofline"; run c4_new A=1; run c4_nodelay ETP_FLUSH_DELAY=0
