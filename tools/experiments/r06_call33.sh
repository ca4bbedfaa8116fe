#!/bin/bash
# round 6, GPU call 33: ETP_MM32_ZEARLY variant library (the FFN dgrad's gelu' operand fetched behind the ring fill instead of behind the reduction; gemm_mm32.hip)
# against the shipped library, same box: parity of the epilogues and the planner goldens with the variant, the GEMM phase probe, eight alternating pairs.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c33; mkdir -p $O
export TMPDIR=/tmp
V=$PWD/etpnav_amd/build/libetp_r6_zearly.so
( ETP_LIB=$V timeout 600 python -m pytest tests/test_mm32_gpu.py -q -x 2>&1 | grep -v "amdgpu.ids" | tail -2 ) | tee $O/parity_mm32.log
( ETP_LIB=$V timeout 600 python -m pytest tests/test_planner_gpu.py tests/test_baseline_shapes_gpu.py -q -x -k "golden or b32_bf16" 2>&1 | grep -v "amdgpu.ids" | tail -2 ) | tee $O/parity_planner.log
( timeout 300 python tools/gemm_phase_probe.py 2>/dev/null | grep "NN,128x128\|span" | head -4 ) | tee $O/phases_base.txt
( ETP_LIB=$V timeout 300 python tools/gemm_phase_probe.py 2>/dev/null | grep "NN,128x128\|span" | head -4 ) | tee $O/phases_zearly.txt
B="--steps 100 --warmup 30 --no-cpu-baseline --no-optimizer --no-roofline"
run() { # label, env...
  lbl=$1; shift
  env "$@" timeout 300 python bench.py $B $WL 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$lbl', j['ms_per_step'], j['value'])"
}
for i in 1 2 3 4 5 6 7 8; do
  run base X=1
  run zearly ETP_LIB=$V
done > $O/ab_zearly.log
cat $O/ab_zearly.log
