set -x
O=gpurun_out/r2d; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -q -x -k "streaming or attention or gemm_group" 2>&1 | tail -25 > $O/tests_flash.log
python -m pytest tests/test_baseline_shapes_gpu.py tests/test_planner_gpu.py -q -x -k "long_instruction or c4_rxr or mlm or sap_pretraining or recorded or b32" 2>&1 | tail -25 > $O/tests_step.log
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-optimizer"
$B > $O/bench_c2.json 2> $O/bench_c2.err
python bench.py --workload c4 --steps 30 --warmup 5 --no-cpu-baseline --no-optimizer > $O/bench_c4.json 2> $O/bench_c4.err
ETP_ATTN_FLASH=0 python bench.py --workload c4 --steps 30 --warmup 5 --no-cpu-baseline --no-optimizer > $O/bench_c4_noflash.json 2> $O/bench_c4_noflash.err
python bench.py --workload c5 --steps 100 --warmup 10 --no-cpu-baseline --no-optimizer > $O/bench_c5.json 2> $O/bench_c5.err
tail -4 $O/tests_flash.log $O/tests_step.log; cut -c1-160 $O/bench_c2.json $O/bench_c4.json $O/bench_c4_noflash.json $O/bench_c5.json
