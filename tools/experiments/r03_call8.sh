#!/bin/bash
# Round-3 GPU call 8: final A/B of the 128x64 class in-step, then the whole GPU test suite on the pruned build
set -x
O=gpurun_out/c8; mkdir -p $O
export TMPDIR=/tmp
T="timeout 300"
B="--steps 100 --warmup 20 --no-cpu-baseline --no-optimizer --no-roofline"
run() { name=$1; shift; env "$@" $T python bench.py $B > $O/bench_$name.json 2> $O/bench_$name.err; python -c "import json; d=json.load(open('$O/bench_$name.json')); print('RESULT $name', d['value'], d['ms_per_step'])"; }
run wide_a A=1
run nowide_a ETP_GEMM_WIDE=0
run wide_b A=1
run nowide_b ETP_GEMM_WIDE=0
run nowide_s4 ETP_GEMM_WIDE=0 ETP_GEMM_64S=4
timeout 1500 python -m pytest tests -m gpu -q --tb=short -rP > $O/gpu_tests.log 2>&1; echo "rc tests $?"; tail -4 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc smoke $?"; tail -3 $O/smoke.log
