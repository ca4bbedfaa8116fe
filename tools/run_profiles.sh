#!/bin/bash
# Reproduces the measurements committed under profiles/ on an MI355X box (run through gpurun from the repo root):
#   /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash tools/run_profiles.sh'
# Outputs land in gpurun_out/profiles_run/ (scratch); copy what should be judged into profiles/ (see profiles/README.md).
set -x
O=gpurun_out/profiles_run; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
# 1. parity + smoke
python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
# 2. bench lines (headline config 2 with all legs; the other BASELINE configs; A/B switches)
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --workload c4 --steps 30 --warmup 5 --no-cpu-baseline --no-optimizer > $O/bench_c4.json 2> $O/bench_c4.err
ETP_ATTN_FLASH=0 python bench.py --workload c4 --steps 30 --warmup 5 --no-cpu-baseline --no-optimizer > $O/bench_c4_noflash.json 2>> $O/bench_c4.err
python bench.py --workload c5 --no-cpu-baseline --no-optimizer > $O/bench_c5.json 2> $O/bench_c5.err
python bench.py --workload sap --steps 100 --no-cpu-baseline --no-optimizer > $O/bench_sap.json 2> $O/bench_sap.err
for v in "ETP_WGRAD_GROUP=0 ETP_GRAD_OVERWRITE=0 ETP_LNBWD_TWO_STAGE=0" "ETP_STREAM_PRIO=0" "ETP_ATTN_ROWS=0" "ETP_DTXT_STREAM=0" "ETP_CHAIN_FIRST=1"; do
  env $v python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-optimizer > "$O/bench_ab_$(echo $v | tr ' =' '__').json" 2>> $O/bench_ab.err
done
python bench.py --graph --steps 100 --warmup 10 --no-cpu-baseline --no-optimizer > $O/bench_graph.json 2> $O/bench_graph.err
# 3. rocprofv3: kernel stats + timeline, then PMC traffic (separate passes per counter; never combined with other traces)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o r -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-optimizer > $R/$O/bench_under_rocprof.json 2> $R/$O/prof.err)
python tools/timeline.py $O/prof/r_kernel_trace.csv --steps 20 > $O/timeline.txt 2>&1
rm -f $O/prof/r_kernel_trace.csv
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$O/pmc_$c -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-optimizer > /dev/null 2> $R/$O/pmc_$c.err)
done
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE/p_counter_collection.csv $O/pmc_WRITE_SIZE/p_counter_collection.csv --cast-elems 38961152 --out $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
rm -rf $O/pmc_*/*kernel_trace*
# 4. micro-benchmarks behind DESIGN.md §4
python tools/chain_budget.py --seq > $O/chain_budget.txt 2>&1
python tools/chain_budget.py --workload c4 --steps 2 > $O/chain_budget_c4.txt 2>&1
python tools/gemm_sweep.py > $O/gemm_sweep.json 2> $O/gemm_sweep.err
KSWEEP_ONLY=1 python tools/gemm_sweep.py > $O/ksweep_cold_vs_warm.json 2>> $O/gemm_sweep.err
GEMM_GROUP_ONLY=1 GEMM_GROUP_TABLE=1 python tools/gemm_bench.py > $O/gemm_group_table.txt 2>&1
python tools/host_timing.py > $O/host_timing.txt 2>&1
python tools/attn_bench.py > $O/attn_bench_rows_kernels.json 2> $O/attn_bench.err
ETP_ATTN_ROWS=0 python tools/attn_bench.py > $O/attn_bench_tile_kernels.json 2>> $O/attn_bench.err
for m in 2 4; do python bench.py --micro $m --steps 100 --warmup 20 --no-cpu-baseline --no-optimizer > $O/bench_micro$m.json 2>> $O/bench_ab.err; done
