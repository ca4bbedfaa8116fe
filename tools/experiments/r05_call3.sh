#!/bin/bash
# round 5, GPU call 3: (a) parity of the carried LayerNorm second stage (op test + whole-step tests), the rest of the new tests;
# (b) same-box A/B of the step: default (LN fold on) / ETP_LN_FOLD=0 / ETP_FLUSH_SPLIT=1 / ETP_FLUSH_SPLIT=2
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5c3; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "layer_norm" > $O/tests_ln.log 2>&1; echo "rc=$?" >> $O/tests_ln.log; tail -3 $O/tests_ln.log
timeout 1500 python -m pytest tests/test_planner_gpu.py tests/test_variants_gpu.py tests/test_mm32_gpu.py tests/test_optim_gpu.py -m gpu -q -x > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -5 $O/tests.log
B="--steps 100 --warmup 20 --no-cpu-baseline --no-optimizer --no-roofline"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B > $O/bench_$name.json 2> $O/bench_$name.err; python -c "import json; d=json.load(open('$O/bench_$name.json')); print('RESULT $name', d['value'], d['ms_per_step'])"; }
run default A=1
run nofold ETP_LN_FOLD=0
run split1 ETP_FLUSH_SPLIT=1
run split2 ETP_FLUSH_SPLIT=2
run split1_nofold ETP_FLUSH_SPLIT=1 ETP_LN_FOLD=0
run default2 A=1
B="--workload c5 $B"; run c5_default A=1; run c5_nofold ETP_LN_FOLD=0
