#!/bin/bash
# The round's evidence on the committed binary (run on the GPU box: gpurun -- bash tools/run_profiles.sh): observed bf16 parity,
# kernel trace, PMC passes, GEMM phases, chain intervals, bench lines of every workload.  Outputs under gpurun_out/evidence/;
# copy what is to be judged into profiles/ (names per round).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
RND=${ETP_ROUND:-r05}            # prefix of the files copied into profiles/ (bench.py reads the newest round present)
O=gpurun_out/evidence; mkdir -p $O
T="timeout 420"
# 1. observed bf16 parity of the benchmarked shapes and the fixtures (prints the worst tensors).  SKIP_PARITY=1: taken from the -s log
#    of the full GPU suite instead (the same tests run there)
[ -n "$SKIP_PARITY" ] || $T python -m pytest tests/test_baseline_shapes_gpu.py tests/test_planner_gpu.py -q -s --tb=short -k "bf16" 2>&1 | grep -E "bf16 worst|passed|failed|Error|^E " | cut -c1-260 > $O/parity_bf16_observed.txt
[ -n "$SKIP_PARITY" ] || tail -3 $O/parity_bf16_observed.txt
[ -n "$SKIP_PARITY" ] || $T python -m pytest tests/test_mm32_gpu.py -q --tb=short 2>&1 | tail -2
python __graft_entry__.py smoke 2>&1 | grep "^smoke" | tee $O/smoke.log
# 2. kernel trace + stats of the bench command
(cd /tmp && $T rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o r -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-optimizer --no-roofline > $R/$O/bench_under_rocprof.json 2> $R/$O/prof.err)
python tools/timeline.py $O/prof/r_kernel_trace.csv --steps 20 > $O/timeline.txt 2>&1
cp $O/prof/r_kernel_stats.csv $O/bench_kernel_stats.csv; cp $O/bench_kernel_stats.csv profiles/${RND}_bench_kernel_stats.csv
rm -rf $O/prof
# 3. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes
P="--steps 2 --warmup 1 --no-cpu-baseline --no-optimizer --no-roofline"
(cd /tmp && $T rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pf -o p -- python $R/bench.py $P > /dev/null 2> $R/$O/pf.err)
(cd /tmp && $T rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pw -o p -- python $R/bench.py $P > /dev/null 2> $R/$O/pw.err)
python tools/pmc_traffic.py $O/pf/p_counter_collection.csv $O/pw/p_counter_collection.csv --cast-elems 36601856 --out $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
cp $O/pmc_traffic.json profiles/${RND}_pmc_traffic.json
rm -rf $O/pf $O/pw
# 4. SQ counters
(cd /tmp && $T rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/ps -o p -- python $R/bench.py $P > /dev/null 2> $R/$O/ps.err)
python tools/pmc_sq.py $O/ps/p_counter_collection.csv --out $O/gemm_counters.json > $O/gemm_counters.txt 2>&1
rm -rf $O/ps
# 5. device-side phases and the chain intervals
$T python tools/gemm_phase_probe.py > $O/gemm_phases.txt 2> $O/gemm_phases.err
[ -n "$SKIP_CHAIN" ] || $T python tools/chain_waits.py --steps 24 --out $O/chain_waits.txt > /dev/null 2> $O/chain_waits.err
# 6. the bench line (reads the two files copied into profiles/ above) and the other workloads
$T python bench.py > $O/bench.json 2> $O/bench.err
for wl in c4 c5 sap; do $T python bench.py --workload $wl --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err; done
$T python bench.py --dtype fp32 --no-cpu-baseline > $O/bench_fp32.json 2> $O/bench_fp32.err
for f in bench bench_c4 bench_c5 bench_sap bench_fp32; do python - <<PY
import json
try:
    d = json.loads(open("$O/$f.json").read().strip().split("\n")[-1])
    r = d.get("roofline") or {}
    print("$f", d["value"], d["ms_per_step"], "roofline", r.get("kernel"), r.get("achieved"), r.get("frac"), "traffic", r.get("traffic"), "rocprof", r.get("rocprof_avg_launch_us"), (d.get("optimizer") or {}).get("train_iteration"))
except Exception as e:
    print("$f FAILED", e)
PY
done
ls -la $O
