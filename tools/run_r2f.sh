set -x
O=gpurun_out/r2f; mkdir -p $O
T="python -m pytest tests/test_baseline_shapes_gpu.py -q -x -k same_seed"
$T 2>&1 | grep -E "differ|passed|failed" | cut -c1-900 > $O/det_default.log
ETP_LNBWD_TWO_STAGE=0 $T 2>&1 | grep -E "differ|passed|failed" | cut -c1-900 > $O/det_ln1.log
ETP_DTXT_STREAM=0 $T 2>&1 | grep -E "differ|passed|failed" | cut -c1-900 > $O/det_nodtxt.log
ETP_WGRAD_GROUP=0 $T 2>&1 | grep -E "differ|passed|failed" | cut -c1-900 > $O/det_nogroup.log
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-optimizer"
$B > $O/bench_c2.json 2> $O/bench_c2.err
ETP_STREAM_PRIO=0 $B > $O/bench_c2_noprio.json 2> $O/bench_c2_noprio.err
ETP_ATTN_Q96=0 $B > $O/bench_c2_noq96.json 2> $O/bench_c2_noq96.err
rocm-smi --showclocks --showpower --showperflevel > $O/smi.txt 2>&1
cat $O/det_*.log; cut -c1-150 $O/bench_c2*.json
