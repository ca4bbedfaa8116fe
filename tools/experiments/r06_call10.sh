#!/bin/bash
# round 6, GPU call 10: leaves taken off the chain's tails -- the node-embedding backward and the text-embedding backward produce parameter
# gradients only; the d txt_embeds join moves to its consumers -- parity, determinism, data-parallel order tests, same-box A/B, chain stamps.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c10; mkdir -p $O
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
for i in 1 2 3; do
  run new X=1
  run old ETP_NAV_TAIL=0 ETP_TXT_TAIL=0
  run nav_only ETP_TXT_TAIL=0
  run txt_only ETP_NAV_TAIL=0
  run nav_leaf_only ETP_NAV_TAIL=1 ETP_TXT_TAIL=0
done > $O/ab_tails.log
cat $O/ab_tails.log
WL="--workload c5"; for i in 1 2; do run c5_new X=1; run c5_old ETP_NAV_TAIL=0 ETP_TXT_TAIL=0; done > $O/ab_tails_c5.log
WL="--workload c4"; for i in 1 2; do run c4_new X=1; run c4_old ETP_NAV_TAIL=0 ETP_TXT_TAIL=0; done >> $O/ab_tails_c5.log
cat $O/ab_tails_c5.log
( timeout 300 python tools/chain_waits.py --steps 24 --out $O/chain_waits_tails.txt > /dev/null 2>&1 ); grep "nav_bwd\|embeddings done\|step end\|node assembly bwd\|txt_bwd begin" $O/chain_waits_tails.txt | head -14
( timeout 1500 python -m pytest tests/test_planner_gpu.py tests/test_variants_gpu.py tests/test_dp_gpu.py -q -x 2>&1 | grep -v "amdgpu.ids\|Gloo\|socket" | tail -6 ) > $O/parity.log
cat $O/parity.log
