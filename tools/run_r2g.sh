set -x
O=gpurun_out/r2g; mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
ETP_DET_RUNS=6 python -m pytest tests/test_baseline_shapes_gpu.py -q -x -k same_seed 2>&1 | grep -E "differ|passed|failed" | cut -c1-600 > $O/det.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o r2 -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-optimizer > $R/$O/bench_prof.json 2> $R/$O/prof.err)
python tools/timeline.py $O/prof/r2_kernel_trace.csv --steps 20 > $O/timeline.txt 2>&1
rm -f $O/prof/r2_kernel_trace.csv
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_fetch -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-optimizer > /dev/null 2> $R/$O/pmc_fetch.err)
(cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_write -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-optimizer > /dev/null 2> $R/$O/pmc_write.err)
python tools/pmc_traffic.py $O/pmc_fetch/p_counter_collection.csv $O/pmc_write/p_counter_collection.csv --cast-elems 38961152 --out $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
rm -rf $O/pmc_fetch/*kernel_trace* $O/pmc_write/*kernel_trace*
ls -la $O/pmc_fetch $O/pmc_write | head; du -sh $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
cat $O/det.log; head -20 $O/pmc_traffic.txt; cut -c1-200 $O/bench_default.json
