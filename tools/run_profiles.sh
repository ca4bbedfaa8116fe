#!/bin/bash
# Reproduces the round-3 measurements committed under profiles/ on an MI355X box (run through gpurun from the repo root):
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/run_profiles.sh'
# Outputs land in gpurun_out/final/ (scratch); copy what should be judged into profiles/ (see profiles/README.md).
set -x
O=gpurun_out/final; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
T="timeout 900"
# 1. bench lines: headline config 2 with all legs, fp32 parity mode, the other BASELINE configs
$T python bench.py > $O/bench.json 2> $O/bench.err; head -c 400 $O/bench.json; echo
$T python bench.py --dtype fp32 --steps 50 --warmup 10 --no-cpu-baseline --no-optimizer > $O/bench_fp32.json 2> $O/bench_fp32.err
$T python bench.py --workload c4 --steps 30 --warmup 5 --no-cpu-baseline --no-optimizer > $O/bench_c4.json 2> $O/bench_c4.err
$T python bench.py --workload c5 --no-cpu-baseline --no-optimizer > $O/bench_c5.json 2> $O/bench_c5.err
$T python bench.py --workload sap --steps 100 --no-cpu-baseline --no-optimizer > $O/bench_sap.json 2> $O/bench_sap.err
for f in bench bench_fp32 bench_c4 bench_c5 bench_sap; do python -c "import json; d=json.load(open('$O/$f.json')); print('RESULT $f', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('frac'))"; done
# 2. rocprofv3: kernel stats + timeline (trace ends with timed steps: --no-roofline)
(cd /tmp && $T rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o r -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-optimizer --no-roofline > $R/$O/bench_under_rocprof.json 2> $R/$O/prof.err)
python tools/timeline.py $O/prof/r_kernel_trace.csv --steps 20 > $O/timeline.txt 2>&1
cp $O/prof/r_kernel_stats.csv $O/bench_kernel_stats.csv
rm -f $O/prof/r_kernel_trace.csv
# 3. PMC: HBM traffic (separate passes per counter), then the SQ / GRBM set; never combined with other trace domains
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && ETP_TXT_CAST_SPLIT=0 $T rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$O/pmc_$c -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-optimizer --no-roofline > /dev/null 2> $R/$O/pmc_$c.err)
done
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE/p_counter_collection.csv $O/pmc_WRITE_SIZE/p_counter_collection.csv --cast-elems 38961152 --out $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
(cd /tmp && $T rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/pmc_sq -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-optimizer --no-roofline > /dev/null 2> $R/$O/pmc_sq.err)
python tools/pmc_sq.py $O/pmc_sq/p_counter_collection.csv --out $O/gemm_counters.json > $O/gemm_counters.txt 2>&1
rm -rf $O/pmc_*/*kernel_trace* $O/pmc_*/p_counter_collection.csv
# 4. per-kernel budget and the in-kernel phase probe
$T python tools/chain_budget.py --seq > $O/chain_budget.txt 2>&1
$T python tools/gemm_phase_probe.py > $O/gemm_phases.txt 2>&1
# 4b. rollout episode: per-step calls vs text K/V cache vs one batched call
$T python tools/rollout_bench.py > $O/rollout_bench.json 2> $O/rollout_bench.err
$T python tools/rollout_bench.py --B 32 --T 5,15 > $O/rollout_bench_b32.json 2> $O/rollout_bench_b32.err
# 5. parity + smoke on the same build
timeout 1500 python -m pytest tests -m gpu -q --tb=short > $O/gpu_tests.log 2>&1; echo "rc tests $?"; tail -3 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc smoke $?"; tail -2 $O/smoke.log
