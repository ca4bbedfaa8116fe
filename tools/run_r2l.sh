set -x
O=gpurun_out/r2l; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/tests_all.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
python bench.py --workload c4 --steps 30 --warmup 5 --no-cpu-baseline --no-optimizer > $O/bench_c4.json 2> $O/bench_c4.err
ETP_ATTN_FLASH=0 python bench.py --workload c4 --steps 30 --warmup 5 --no-cpu-baseline --no-optimizer > $O/bench_c4_noflash.json 2> $O/bench_c4_noflash.err
python bench.py --workload c5 --no-cpu-baseline --no-optimizer > $O/bench_c5.json 2> $O/bench_c5.err
python bench.py --workload sap --steps 100 --no-cpu-baseline --no-optimizer > $O/bench_sap.json 2> $O/bench_sap.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --gpus 2 --dist-backend gloo --same-device --steps 20 --warmup 3 --no-cpu-baseline --no-optimizer > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err || true
tail -3 $O/tests_all.log; tail -2 $O/smoke.log; cut -c1-150 $O/bench_*.json
