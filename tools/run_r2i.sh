set -x
O=gpurun_out/r2i; mkdir -p $O
python -m pytest tests/test_planner_gpu.py tests/test_baseline_shapes_gpu.py tests/test_dp_gpu.py -q -k "pretrain_driver or same_seed or recorded or graph_replay or layer_ranges or two_rank or bf16_train or fp32_train_mode_step" 2>&1 | tail -12 > $O/tests.log
python tools/host_timing.py > $O/host_timing.txt 2>&1
ETP_CHAIN_FIRST=0 python tools/host_timing.py > $O/host_timing_oldorder.txt 2>&1
B="python bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-optimizer"
$B > $O/bench_c2.json 2> $O/bench_c2.err
ETP_CHAIN_FIRST=0 $B > $O/bench_c2_oldorder.json 2> $O/bench_c2_oldorder.err
$B > $O/bench_c2_b.json 2> $O/bench_c2_b.err
tail -3 $O/tests.log; cat $O/host_timing*.txt | grep wall; cut -c1-150 $O/bench_c2*.json
