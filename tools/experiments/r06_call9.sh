#!/bin/bash
# round 6, GPU call 9: node-side schedule experiments on config 2 -- the text K/V projections of the four x-layers as ONE grouped launch (side
# stream / on the chain), the node assembly behind the panorama branch -- parity of the grouped form, same-box A/B, chain stamps.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c9; mkdir -p $O
export TMPDIR=/tmp
( ETP_NAV_KV_GROUP=1 ETP_ASSEMBLE_ON_S2=1 timeout 900 python -m pytest tests/test_planner_gpu.py -q -x -k "golden or oracle" 2>&1 | grep -v amdgpu.ids | tail -5 ) > $O/parity_kvgroup.log
cat $O/parity_kvgroup.log
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
  run kvgroup1 ETP_NAV_KV_GROUP=1
  run kvgroup2 ETP_NAV_KV_GROUP=2
  run asm_s2 ETP_ASSEMBLE_ON_S2=1
  run both ETP_NAV_KV_GROUP=1 ETP_ASSEMBLE_ON_S2=1
done > $O/ab_nav.log
cat $O/ab_nav.log
WL="--workload c5"; for i in 1 2; do run c5_base X=1; run c5_both ETP_NAV_KV_GROUP=1 ETP_ASSEMBLE_ON_S2=1; done > $O/ab_nav_c5.log
cat $O/ab_nav_c5.log
( ETP_NAV_KV_GROUP=1 ETP_ASSEMBLE_ON_S2=1 timeout 300 python tools/chain_waits.py --steps 24 --out $O/chain_waits_kvgroup.txt > /dev/null 2>&1 ); grep "nav_fwd\|join panorama" $O/chain_waits_kvgroup.txt | head -8
