#!/bin/bash
# Round-3 GPU call 4: what does the leaf work (grouped weight gradients, panorama branch) cost the dependent chain?
#   ETP_SKIP_WGRAD=1          chain + panorama branch only (gradients wrong: measurement only)
#   ETP_GROUP_WG_PER_CU=1     the grouped weight-gradient kernel occupies at most one workgroup per CU
#   ETP_AUX_CU_MASK=a/b       weight-gradient stream confined to a/b of the CUs (hipExtStreamCreateWithCUMask)
#   ETP_S2_CU_MASK=a/b        panorama stream confined likewise
set -x
O=gpurun_out/c4; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
T="timeout 300"
B="--steps 100 --warmup 20 --no-cpu-baseline --no-optimizer"
run() { name=$1; shift; env "$@" $T python bench.py $B > $O/bench_$name.json 2> $O/bench_$name.err; python -c "import json; d=json.load(open('$O/bench_$name.json')); print('RESULT $name', d['value'], d['ms_per_step'])"; }
run default A=1
run skipwgrad ETP_SKIP_WGRAD=1
run group1 ETP_GROUP_WG_PER_CU=1
run aux_1_4 ETP_AUX_CU_MASK=1/4
run aux_3_8 ETP_AUX_CU_MASK=3/8
run aux_1_2 ETP_AUX_CU_MASK=1/2
run aux_1_2_s2_1_4 ETP_AUX_CU_MASK=1/2 ETP_S2_CU_MASK=1/4
run aux_3_8_s2_1_4 ETP_AUX_CU_MASK=3/8 ETP_S2_CU_MASK=1/4
run group1_aux_1_2 ETP_GROUP_WG_PER_CU=1 ETP_AUX_CU_MASK=1/2
run noprio ETP_STREAM_PRIO=0
run default2 A=1
# kernel trace of the default three-stream step -> timeline (which kernels wait, how long chain kernels take in-step)
(cd /tmp && $T rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o r -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-optimizer > $R/$O/bench_under_rocprof.json 2> $R/$O/prof.err)
python tools/timeline.py $O/prof/r_kernel_trace.csv --steps 20 > $O/timeline.txt 2>&1
cp $O/prof/r_kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null
rm -f $O/prof/r_kernel_trace.csv
grep RESULT -r $O/*.err | head -0
