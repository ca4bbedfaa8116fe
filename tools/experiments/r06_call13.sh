#!/bin/bash
# round 6, GPU call 13: text layer 0's attention weight gradients forked as soon as their operands exist (TXT_LAST_SPLIT): A/B, chain stamps, parity.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c13; mkdir -p $O
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
  run split X=1
  run nosplit ETP_TXT_LAST_SPLIT=0
  run split_txttail ETP_TXT_TAIL=1
done > $O/ab_split.log
cat $O/ab_split.log
WL="--workload c5"; for i in 1 2; do run c5_split X=1; run c5_nosplit ETP_TXT_LAST_SPLIT=0; done > $O/ab_split_c5.log
WL="--workload c4"; for i in 1 2; do run c4_split X=1; run c4_nosplit ETP_TXT_LAST_SPLIT=0; done >> $O/ab_split_c5.log
cat $O/ab_split_c5.log
( timeout 300 python tools/chain_waits.py --steps 24 --out $O/chain_waits_split.txt > /dev/null 2>&1 ); grep "embeddings done\|step end\|layer 0" $O/chain_waits_split.txt | head -8
( timeout 1200 python -m pytest tests/test_planner_gpu.py tests/test_variants_gpu.py -q -x 2>&1 | grep -v "amdgpu.ids\|Gloo\|socket" | tail -5 ) > $O/parity.log
cat $O/parity.log
