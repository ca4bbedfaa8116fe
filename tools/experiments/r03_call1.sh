#!/bin/bash
# Round-3 GPU call 1: correctness of the pipelined GEMM main loop + same-box A/B against the round-2 binary
# (etpnav_amd/libetpnav_hip_r02.so, built from commit d19790c) + phase probe + SQ counters.
set -x
O=gpurun_out/c1; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
OLD=$R/etpnav_amd/libetpnav_hip_r02.so
T="timeout 600"
$T python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gemm" --tb=short > $O/ops_gemm.log 2>&1; echo "rc gemm tests $?"
B="--steps 100 --warmup 20 --no-cpu-baseline --no-optimizer"
$T python bench.py $B > $O/bench_new.json 2> $O/bench_new.err; echo "rc bench $?"
ETP_LIB=$OLD $T python bench.py $B > $O/bench_r02.json 2> $O/bench_r02.err
ETP_GEMM_WIDE=0 $T python bench.py $B > $O/bench_new_nowide.json 2> $O/bench_new_nowide.err
$T python tools/chain_budget.py --seq > $O/chain_budget_new.txt 2>&1
ETP_LIB=$OLD $T python tools/chain_budget.py > $O/chain_budget_r02.txt 2>&1
$T python tools/gemm_phase_probe.py > $O/gemm_phases.txt 2>&1
$T python tools/gemm_sweep.py > $O/gemm_sweep_new.json 2> $O/gemm_sweep_new.err
ETP_LIB=$OLD $T python tools/gemm_sweep.py > $O/gemm_sweep_r02.json 2> $O/gemm_sweep_r02.err
(cd /tmp && $T rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/pmc_sq -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-optimizer > /dev/null 2> $R/$O/pmc_sq.err)
python tools/pmc_sq.py $O/pmc_sq/p_counter_collection.csv --out $O/gemm_counters_new.json > $O/gemm_counters_new.txt 2>&1
(cd /tmp && ETP_LIB=$OLD $T rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/pmc_sq_r02 -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-optimizer > /dev/null 2> $R/$O/pmc_sq_r02.err)
python tools/pmc_sq.py $O/pmc_sq_r02/p_counter_collection.csv --out $O/gemm_counters_r02.json > $O/gemm_counters_r02.txt 2>&1
rm -rf $O/pmc_sq/*kernel_trace* $O/pmc_sq_r02/*kernel_trace* $O/pmc_sq/p_counter_collection.csv $O/pmc_sq_r02/p_counter_collection.csv
# parity of the whole path with the tightened bf16 bounds (prints the observed worst ratios)
timeout 900 python -m pytest tests/test_planner_gpu.py tests/test_baseline_shapes_gpu.py tests/test_ops_gpu.py -m gpu -q -rP --tb=short > $O/parity.log 2>&1; echo "rc parity $?"
tail -5 $O/ops_gemm.log; cat $O/bench_new.json | head -c 600; echo; cat $O/bench_r02.json | head -c 300; echo; tail -15 $O/parity.log
