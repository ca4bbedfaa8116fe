set -x
O=gpurun_out/r2k; mkdir -p $O
python -m pytest tests/test_ops_gpu.py tests/test_baseline_shapes_gpu.py -q -k "streaming or attention or long_instruction or c4_rxr" 2>&1 | tail -12 > $O/tests.log
python bench.py --workload c4 --steps 30 --warmup 5 --no-cpu-baseline --no-optimizer > $O/bench_c4.json 2> $O/bench_c4.err
python tools/chain_budget.py --workload c4 --steps 2 2>&1 | head -16 > $O/chain_budget_c4.txt
tail -3 $O/tests.log; cut -c1-150 $O/bench_c4.json; cat $O/chain_budget_c4.txt
