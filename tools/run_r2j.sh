set -x
O=gpurun_out/r2j; mkdir -p $O
KSWEEP_ONLY=1 python tools/gemm_sweep.py > $O/ksweep_cold_warm.json 2> $O/ksweep.err
python tools/chain_budget.py --workload c4 --steps 2 > $O/chain_budget_c4.txt 2>&1
python tools/chain_budget.py --workload c5 --steps 2 > $O/chain_budget_c5.txt 2>&1
cat $O/ksweep_cold_warm.json; head -24 $O/chain_budget_c4.txt
