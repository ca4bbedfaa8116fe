#!/bin/bash
# round 6, GPU call 18: the fp32 residual of the 128x64 stream tiles fetched before the reduction (gemm_mm32.hip RPre): parity, phase probe,
# same-box A/B against the same library built with -DETP_MM32_NO_RPRE.
# (RPre / -DETP_MM32_NO_RPRE were reverted after this call: profiles/r06_ab_runs.json r6c18)
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c18; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_mm32_gpu.py tests/test_ops_gpu.py -q -x -k "mm32 or gemm" 2>&1 | grep -v amdgpu.ids | tail -4 ) > $O/gemm_tests.log
cat $O/gemm_tests.log
B="--steps 100 --warmup 30 --no-cpu-baseline --no-optimizer --no-roofline"
run() { # label, env...
  lbl=$1; shift
  env "$@" timeout 300 python bench.py $B $WL 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$lbl', j['ms_per_step'], j['value'], j['config'].get('env_overrides'))"
}
for i in 1 2 3 4; do
  run rpre X=1
  run no_rpre ETP_LIB=$PWD/etpnav_amd/build/libetp_r6_norpre.so
done > $O/ab_rpre.log
cat $O/ab_rpre.log
WL="--workload c4"; for i in 1 2; do run c4_rpre X=1; run c4_no_rpre ETP_LIB=$PWD/etpnav_amd/build/libetp_r6_norpre.so; done > $O/ab_rpre_c4.log
cat $O/ab_rpre_c4.log
( timeout 300 python tools/gemm_phase_probe.py 2>/dev/null | head -14 ) > $O/gemm_phases_rpre.txt
( ETP_LIB=$PWD/etpnav_amd/build/libetp_r6_norpre.so timeout 300 python tools/gemm_phase_probe.py 2>/dev/null | head -14 ) > $O/gemm_phases_norpre.txt
grep "128x64" $O/gemm_phases_rpre.txt | cut -c1-150; echo; grep "128x64" $O/gemm_phases_norpre.txt | cut -c1-150
( timeout 900 python -m pytest tests/test_planner_gpu.py -q -x -k "golden or same_masks" 2>&1 | grep -v "amdgpu.ids" | tail -3 ) > $O/parity.log
cat $O/parity.log
