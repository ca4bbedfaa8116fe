#!/bin/bash
# round 6, GPU call 35: VERDICT r5 #4a on the CURRENT kernel family -- the grouped weight-gradient launch as 256 persistent walkers (one 67.6-KB leaf workgroup per CU, so a
# chain workgroup always fits beside it; variant library -DETP_MM32_GROUP_WALK=256, gemm_mm32.hip) against the one-tile-per-workgroup launch, same box.
# Round 3 measured persistent grids of 256 / 192 / 128 at +3.4 ... +12 % with the older 16x16x32 family.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c35; mkdir -p $O
export TMPDIR=/tmp
V=$PWD/etpnav_amd/build/libetp_r6_walk256.so
( ETP_LIB=$V timeout 600 python -m pytest tests/test_mm32_gpu.py -q -x 2>&1 | grep -v "amdgpu.ids" | tail -2 ) | tee $O/parity_mm32.log
( ETP_LIB=$V timeout 600 python -m pytest tests/test_planner_gpu.py -q -x -k "golden" 2>&1 | grep -v "amdgpu.ids" | tail -2 ) | tee $O/parity_planner.log
( timeout 300 python tools/gemm_phase_probe.py 2>/dev/null | grep "mm32_group\|span" | head -5 ) | tee $O/phases_base.txt
( ETP_LIB=$V timeout 300 python tools/gemm_phase_probe.py 2>/dev/null | grep "mm32_group\|span" | head -5 ) | tee $O/phases_walk.txt
B="--steps 100 --warmup 30 --no-cpu-baseline --no-optimizer --no-roofline"
run() { # label, env...
  lbl=$1; shift
  env "$@" timeout 300 python bench.py $B $WL 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$lbl', j['ms_per_step'], j['value'])"
}
for i in 1 2 3 4 5; do
  run base X=1
  run walk256 ETP_LIB=$V
done > $O/ab_walk.log
cat $O/ab_walk.log
( ETP_LIB=$V timeout 300 python tools/chain_waits.py --steps 24 --out $O/chain_waits_walk.txt > /dev/null 2>&1 ); grep "txt_bwd layer [3210]\|embeddings done\|step end" $O/chain_waits_walk.txt | cut -c1-125
