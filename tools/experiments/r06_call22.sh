#!/bin/bash
# round 6, GPU call 22: the panorama branch made "polite" -- its products forced onto small workgroups (32x64 ring 2 = 24 KB of LDS, which fits
# as a THIRD workgroup beside two 67.6-KB 128x128 workgroups of the dependent chain; 32x64 ring 4; 64x64 ring 3) through the new tile hint
# (common.h gemm_tile_hint, switches PANO_TILE_FWD / PANO_TILE_BWD).  The branch has 0.68 ms (forward) / 1.3 ms (backward) of slack on its stream
# (profiles/r06_chain_waits_tails.txt) while text layers 0-2 run 79 / 44 / 35 us longer beside it and the text backward's layers 8-6 ~40-50 us each.
# Parity of the hinted classes first (planner goldens with the hint on), then the A/B; + the dependent chain on a highest-priority stream.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c22; mkdir -p $O
export TMPDIR=/tmp
( ETP_PANO_TILE_FWD=322 ETP_PANO_TILE_BWD=322 timeout 600 python -m pytest tests/test_planner_gpu.py -q -x -k "golden" 2>&1 | grep -v "amdgpu.ids" | tail -3 ) > $O/parity_322.log
cat $O/parity_322.log
( ETP_PANO_TILE_FWD=324 ETP_PANO_TILE_BWD=643 timeout 600 python -m pytest tests/test_planner_gpu.py -q -x -k "golden" 2>&1 | grep -v "amdgpu.ids" | tail -3 ) > $O/parity_324_643.log
cat $O/parity_324_643.log
B="--steps 100 --warmup 30 --no-cpu-baseline --no-optimizer --no-roofline"
run() { # label, env...
  lbl=$1; shift
  env "$@" timeout 300 python bench.py $B $WL 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$lbl', j['ms_per_step'], j['value'], j['config'].get('env_overrides'))"
}
for i in 1 2 3; do
  run base X=1
  run fwd322 ETP_PANO_TILE_FWD=322
  run fwd324 ETP_PANO_TILE_FWD=324
  run fwd643 ETP_PANO_TILE_FWD=643
  run bwd322 ETP_PANO_TILE_BWD=322
  run bwd324 ETP_PANO_TILE_BWD=324
  run both322 ETP_PANO_TILE_FWD=322 ETP_PANO_TILE_BWD=322
  run chain_high ETP_BENCH_CHAIN_PRIO=high
done > $O/ab_polite.log
cat $O/ab_polite.log
( ETP_PANO_TILE_FWD=322 ETP_PANO_TILE_BWD=322 timeout 300 python tools/chain_waits.py --steps 24 --out $O/chain_waits_polite322.txt > /dev/null 2>&1 ); grep "txt_fwd layer [0-3] \|pano_\|txt_bwd layer [876] " $O/chain_waits_polite322.txt | head -24 | cut -c1-125
