#!/bin/bash
# round 6, GPU call 6: QKV projection fused into the register-resident self-attention forward (attn_rows.hip QKV) + the grid-size rule
# for both folds: parity, isolated timing (incl. one-workgroup-per-CU grids), planner parity suite, same-box A/B per workload.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c6; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "fused_qkv or fused_out_projection or proj_refuses" 2>&1 | grep -v amdgpu.ids | tail -15 ) > $O/fused_tests.log
( timeout 300 python tools/experiments/r06_attn_qkv_bench.py 2>&1 | grep -v "amdgpu.ids\|^{" ) > $O/qkv_bench.txt
( timeout 300 python tools/experiments/r06_attn_proj_bench.py 2>&1 | grep -v "amdgpu.ids\|^{" ) > $O/proj_bench.txt
B="--steps 100 --warmup 30 --no-cpu-baseline --no-optimizer --no-roofline"
run() { # label, env...
  lbl=$1; shift
  env "$@" timeout 300 python bench.py $B $WL 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$lbl', j['ms_per_step'], j['value'], j['config'].get('env_overrides'))"
}
for i in 1 2; do run c2_auto X=1; run c2_qkv1 ETP_ATTN_QKV=1; run c2_proj1 ETP_ATTN_PROJ=1; done > $O/ab_c2.log
WL="--workload c5"; for i in 1 2; do run c5_off ETP_ATTN_PROJ=0 ETP_ATTN_QKV=0; run c5_proj ETP_ATTN_QKV=0; run c5_qkv ETP_ATTN_PROJ=0; run c5_auto X=1; done > $O/ab_c5.log
WL="--workload c4"; for i in 1 2; do run c4_off ETP_ATTN_PROJ=0 ETP_ATTN_QKV=0; run c4_auto X=1; done > $O/ab_c4.log
WL="--workload sap"; for i in 1; do run sap_off ETP_ATTN_PROJ=0 ETP_ATTN_QKV=0; run sap_auto X=1; done > $O/ab_sap.log
cat $O/fused_tests.log $O/qkv_bench.txt $O/proj_bench.txt $O/ab_c2.log $O/ab_c5.log $O/ab_c4.log $O/ab_sap.log
( timeout 1200 python -m pytest tests/test_planner_gpu.py tests/test_variants_gpu.py tests/test_baseline_shapes_gpu.py -q -x 2>&1 | grep -v "amdgpu.ids" | tail -15 ) > $O/planner_tests.log
cat $O/planner_tests.log
