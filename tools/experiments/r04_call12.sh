#!/bin/bash
O=gpurun_out/r4c12; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_mm32_gpu.py tests/test_ops_gpu.py -x -q --tb=short -k "gemm or mm32" > $O/gemm_tests.log 2>&1; echo "rc gemm $?"; tail -4 $O/gemm_tests.log
timeout 900 python -m pytest tests/test_planner_gpu.py tests/test_baseline_shapes_gpu.py -x -q --tb=short > $O/planner_tests.log 2>&1; echo "rc planner $?"; tail -4 $O/planner_tests.log
B="--steps 100 --warmup 20 --no-cpu-baseline --no-optimizer --no-roofline"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B > $O/bench_$name.json 2> $O/bench_$name.err; python -c "import json; d=json.load(open('$O/bench_$name.json')); print('RESULT $name', d['value'], d['ms_per_step'], d['loss'])"; }
run w8 A=1
run w4 ETP_GEMM_W8=0
run w8_b A=1
run w4_b ETP_GEMM_W8=0
run c5_w8 A=1 --workload c5
timeout 300 python bench.py $B --workload c5 > $O/bench_c5_w8.json 2>/dev/null; python -c "import json; d=json.load(open('$O/bench_c5_w8.json')); print('RESULT c5 w8', d['value'], d['ms_per_step'])"
ETP_GEMM_W8=0 timeout 300 python bench.py $B --workload c5 > $O/bench_c5_w4.json 2>/dev/null; python -c "import json; d=json.load(open('$O/bench_c5_w4.json')); print('RESULT c5 w4', d['value'], d['ms_per_step'])"
timeout 300 python tools/gemm_phase_probe.py > $O/gemm_phases.txt 2> $O/gemm_phases.err; grep "gemm_dma8\|sum of" $O/gemm_phases.txt
