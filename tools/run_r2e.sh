set -x
O=gpurun_out/r2e; mkdir -p $O
python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $O/tests_all.log
B="python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-optimizer"
$B > $O/bench_c2.json 2> $O/bench_c2.err
python bench.py --workload c4 --steps 30 --warmup 5 --no-cpu-baseline --no-optimizer > $O/bench_c4.json 2> $O/bench_c4.err
python bench.py --workload c5 --steps 100 --warmup 10 --no-cpu-baseline --no-optimizer > $O/bench_c5.json 2> $O/bench_c5.err
python tools/chain_budget.py > $O/chain_budget.txt 2>&1
tail -4 $O/tests_all.log; cut -c1-160 $O/bench_c2.json $O/bench_c4.json $O/bench_c5.json
