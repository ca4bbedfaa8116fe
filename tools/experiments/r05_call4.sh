#!/bin/bash
# round 5, GPU call 4: (a) determinism screen tests; (b) is the GELU / GELU' epilogue VALU-bound?  phase probe of the step with the
# erf arithmetic compiled out (-DETP_EPI_NOMATH variant) against the default; (c) staggered second workgroup per CU (128x128 classes)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5c4; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_variants_gpu.py -m gpu -q -x -k "three_stream" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -4 $O/tests.log
timeout 300 python tools/gemm_phase_probe.py > $O/phases_default.txt 2> $O/phases_default.err
ETP_LIB=$PWD/etpnav_amd/build/libetp_nomath.so timeout 300 python tools/gemm_phase_probe.py > $O/phases_nomath.txt 2> $O/phases_nomath.err
grep "2560x3072x768\|2560x2304x768" $O/phases_default.txt $O/phases_nomath.txt
B="--steps 100 --warmup 20 --no-cpu-baseline --no-optimizer --no-roofline"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B > $O/bench_$name.json 2> $O/bench_$name.err; python -c "import json; d=json.load(open('$O/bench_$name.json')); print('RESULT $name', d['value'], d['ms_per_step'])"; }
run default ETP_LN_FOLD=0
run stag1 ETP_LN_FOLD=0 ETP_LIB=$PWD/etpnav_amd/build/libetp_stag1.so
run stag2 ETP_LN_FOLD=0 ETP_LIB=$PWD/etpnav_amd/build/libetp_stag2.so
run nomath ETP_LN_FOLD=0 ETP_LIB=$PWD/etpnav_amd/build/libetp_nomath.so
run default2 ETP_LN_FOLD=0
