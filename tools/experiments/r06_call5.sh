#!/bin/bash
# round 6, GPU call 5: out-projection dgrad fused into the register-resident attention backward (attn_rows.hip PROJ): parity, isolated
# timing against the launch pair it replaces, same-box A/B of the step (ATTN_PROJ = 0 / default, two Q/K/V fetch positions), c5.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c5; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "fused_out_projection or proj_refuses or attention_fwd_bwd" 2>&1 | grep -v amdgpu.ids | tail -15 ) > $O/proj_tests.log
( timeout 300 python tools/experiments/r06_attn_proj_bench.py 2>&1 | grep -v amdgpu.ids ) > $O/proj_bench.txt
for v in fetch3 fetch6; do ( ETP_LIB=$PWD/etpnav_amd/build/libetp_r6_$v.so timeout 300 python tools/experiments/r06_attn_proj_bench.py 2>&1 | grep -v amdgpu.ids | head -6 | sed "s/^/$v /" ) >> $O/proj_bench.txt; done
( timeout 900 python -m pytest tests/test_planner_gpu.py tests/test_variants_gpu.py tests/test_neighbours_gpu.py -q -x 2>&1 | grep -v "amdgpu.ids" | tail -15 ) > $O/planner_tests.log
B="--steps 100 --warmup 30 --no-cpu-baseline --no-optimizer --no-roofline"
run() { # label, env...
  lbl=$1; shift
  env "$@" timeout 300 python bench.py $B $WL 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$lbl', j['ms_per_step'], j['value'], j['config'].get('env_overrides'))"
}
for i in 1 2; do
  run proj0 ETP_ATTN_PROJ=0; run proj1 X=1; run fetch3 ETP_LIB=$PWD/etpnav_amd/build/libetp_r6_fetch3.so; run fetch6 ETP_LIB=$PWD/etpnav_amd/build/libetp_r6_fetch6.so
done > $O/ab_proj.log
WL="--workload c5"; for i in 1 2; do run c5_proj0 ETP_ATTN_PROJ=0; run c5_proj1 X=1; done > $O/ab_proj_c5.log
WL="--workload c4"; for i in 1; do run c4_proj0 ETP_ATTN_PROJ=0; run c4_proj1 X=1; done > $O/ab_proj_c4.log
cat $O/proj_tests.log $O/proj_bench.txt $O/planner_tests.log $O/ab_proj.log $O/ab_proj_c5.log $O/ab_proj_c4.log
