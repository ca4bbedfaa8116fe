#!/bin/bash
# round 6, GPU call 20: the K/V weight gradients of the four x-layers held back to the end of etp_nav_bwd (NAV_TAIL bit 2): A/B + stamps + one parity test.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c20; mkdir -p $O
export TMPDIR=/tmp
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
  run base X=1
  run hold_kv ETP_NAV_TAIL=7
done > $O/ab_hold_kv.log
cat $O/ab_hold_kv.log
( ETP_NAV_TAIL=7 timeout 300 python tools/chain_waits.py --steps 24 --out $O/chain_waits_hold_kv.txt > /dev/null 2>&1 ); grep "nav_bwd\|txt_bwd layer 8\|txt_bwd layer 7 " $O/chain_waits_hold_kv.txt | head -12 | cut -c1-125
( ETP_NAV_TAIL=7 timeout 600 python -m pytest tests/test_planner_gpu.py -q -x -k "golden" 2>&1 | grep -v "amdgpu.ids" | tail -3 ) > $O/parity.log
cat $O/parity.log
