#!/bin/bash
# round 5, GPU call 5: parity of (a) the GELU pair that saves the derivative (+ v_rcp instead of the IEEE division), (b) the
# pano_embed_bwd rewrite (no AGPRs, LDS accumulators); then same-box A/B against the round-4-equivalent library (HEAD of the
# parity commit, etpnav_amd/build/libetp_base.so)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5c5; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_mm32_gpu.py tests/test_planner_gpu.py tests/test_baseline_shapes_gpu.py tests/test_variants_gpu.py tests/test_optim_gpu.py -m gpu -q -x > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -5 $O/tests.log
B="--steps 100 --warmup 20 --no-cpu-baseline --no-optimizer --no-roofline"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B > $O/bench_$name.json 2> $O/bench_$name.err; python -c "import json; d=json.load(open('$O/bench_$name.json')); print('RESULT $name', d['value'], d['ms_per_step'])"; }
run new A=1
run base ETP_LIB=$PWD/etpnav_amd/build/libetp_base.so
run new2 A=1
run base2 ETP_LIB=$PWD/etpnav_amd/build/libetp_base.so
B2="$B --mode eval"; (B="$B2"; run new_eval A=1)
B="--workload c5 $B"
run c5_new A=1
run c5_base ETP_LIB=$PWD/etpnav_amd/build/libetp_base.so
