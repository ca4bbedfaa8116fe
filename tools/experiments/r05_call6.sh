#!/bin/bash
# round 5, GPU call 6: parity with the fp16-stored GELU derivative (the bf16-stored one failed one B = 1 fixture bound) + phases
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5c6; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_mm32_gpu.py tests/test_planner_gpu.py tests/test_baseline_shapes_gpu.py tests/test_variants_gpu.py tests/test_optim_gpu.py -m gpu -q -x > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -5 $O/tests.log
timeout 300 python tools/gemm_phase_probe.py > $O/phases.txt 2> $O/phases.err; grep "2560x3072x768" $O/phases.txt
B="--steps 100 --warmup 20 --no-cpu-baseline --no-optimizer --no-roofline"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B > $O/bench_$name.json 2> $O/bench_$name.err; python -c "import json; d=json.load(open('$O/bench_$name.json')); print('RESULT $name', d['value'], d['ms_per_step'])"; }
run new A=1
run base ETP_LIB=$PWD/etpnav_amd/build/libetp_base.so
run new2 A=1
