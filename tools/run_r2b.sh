set -x
O=gpurun_out/r2b; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -q -x -k "group or two_stage or cross_entropy" 2>&1 | tail -15 > $O/tests_ops.log
python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $O/tests_all.log
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-optimizer"
$B > $O/bench_new.json 2> $O/bench_new.err
ETP_WGRAD_GROUP=0 ETP_GRAD_OVERWRITE=0 ETP_LNBWD_TWO_STAGE=0 $B > $O/bench_old.json 2> $O/bench_old.err
ETP_GROUP_TILE=64s3 $B > $O/bench_g64.json 2> $O/bench_g64.err
ETP_GROUP_TILE=128s3 $B > $O/bench_g128s3.json 2> $O/bench_g128s3.err
ETP_LNBWD_TWO_STAGE=0 $B > $O/bench_ln1.json 2> $O/bench_ln1.err
timeout 300 $B --graph > $O/bench_graph.json 2> $O/bench_graph.err
GEMM_GROUP_ONLY=1 GEMM_GROUP_TABLE=1 timeout 300 python tools/gemm_bench.py > $O/gemm_group.txt 2>&1
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o r2b -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-optimizer > $R/$O/bench_prof.json 2> $R/$O/prof.err)
T=$(ls $O/prof/*/*kernel_trace.csv $O/prof/*kernel_trace.csv 2>/dev/null | head -1)
python tools/timeline.py $T --steps 20 > $O/timeline.txt 2>&1
cp $(ls $O/prof/*/*kernel_stats.csv $O/prof/*kernel_stats.csv 2>/dev/null | head -1) $O/kernel_stats.csv
rm -f $O/prof/*/*kernel_trace.csv $O/prof/*kernel_trace.csv
tail -3 $O/tests_all.log; cat $O/bench_new.json | cut -c1-300
