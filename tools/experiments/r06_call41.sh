#!/bin/bash
# round 6, GPU call 41: the plain-store epilogue of the weight-gradient products (TN, fp32 C, overwrite mode: key 0) as a compile-time specialisation too
# (variant library -DETP_EPI_SPECIAL_TN; DESIGN.md 3.2c left it open): parity, phase probe, six alternating pairs against the shipped library.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c41; mkdir -p $O
export TMPDIR=/tmp
V=$PWD/etpnav_amd/build/libetp_r6_epispec_tn.so
( ETP_LIB=$V timeout 600 python -m pytest tests/test_mm32_gpu.py -q -x 2>&1 | grep -v "amdgpu.ids" | tail -2 ) | tee $O/parity_mm32.log
( ETP_LIB=$V timeout 600 python -m pytest tests/test_planner_gpu.py -q -x -k "golden" 2>&1 | grep -v "amdgpu.ids" | tail -2 ) | tee $O/parity_planner.log
sel() { grep "span\|mm32_group\|TN,128x64\|sum of" | head -9; }
echo "== shipped"; ( timeout 300 python tools/gemm_phase_probe.py 2>/dev/null | sel ) | tee $O/phases_base.txt
echo "== TN specialised"; ( ETP_LIB=$V timeout 300 python tools/gemm_phase_probe.py 2>/dev/null | sel ) | tee $O/phases_tn.txt
B="--steps 100 --warmup 30 --no-cpu-baseline --no-optimizer --no-roofline"
run() { # label, env...
  lbl=$1; shift
  env "$@" timeout 300 python bench.py $B $WL 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$lbl', j['ms_per_step'], j['value'])"
}
for i in 1 2 3 4 5 6; do
  run base X=1
  run tn ETP_LIB=$V
done > $O/ab_tn.log
cat $O/ab_tn.log
