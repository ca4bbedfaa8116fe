set -x
O=gpurun_out/r2c; mkdir -p $O
python -m pytest tests/test_dp_gpu.py tests/test_optim_gpu.py tests/test_ops_gpu.py -q -x -k "dp or native or cast or fused_adamw or two_stage or cross_entropy or rank" 2>&1 | tail -15 > $O/tests_new.log
python -m pytest tests/test_planner_gpu.py tests/test_baseline_shapes_gpu.py -q -x 2>&1 | tail -15 > $O/tests_planner.log
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-optimizer"
$B > $O/bench_new.json 2> $O/bench_new.err
ETP_DTXT_STREAM=0 $B > $O/bench_nodtxt.json 2> $O/bench_nodtxt.err
python tools/chain_budget.py --seq > $O/chain_budget.txt 2>&1
python tools/gemm_sweep.py > $O/gemm_sweep.json 2> $O/gemm_sweep.err
tail -3 $O/tests_new.log $O/tests_planner.log; cut -c1-200 $O/bench_new.json
