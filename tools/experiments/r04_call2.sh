#!/bin/bash
# Round-4 GPU call 2: first contact of the mm32 GEMM family -- parity tests, old gemm tests, A/B bench, phase probe
set -x
O=gpurun_out/r4c2; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_mm32_gpu.py -x -q --tb=short > $O/mm32_tests.log 2>&1; echo "rc mm32 $?"; tail -30 $O/mm32_tests.log
timeout 600 python -m pytest tests/test_ops_gpu.py -q --tb=short -k "gemm" > $O/gemm_tests.log 2>&1; echo "rc gemm $?"; tail -5 $O/gemm_tests.log
B="--steps 100 --warmup 20 --no-cpu-baseline --no-optimizer --no-roofline"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B > $O/bench_$name.json 2> $O/bench_$name.err; python -c "import json; d=json.load(open('$O/bench_$name.json')); print('RESULT $name', d['value'], d['ms_per_step'], d['loss'])"; }
run new A=1
run old ETP_MM32=0
run new_b A=1
run old_b ETP_MM32=0
timeout 300 python tools/gemm_phase_probe.py > $O/gemm_phases.txt 2> $O/gemm_phases.err; echo "rc probe $?"; head -50 $O/gemm_phases.txt
