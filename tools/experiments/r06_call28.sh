#!/bin/bash
# round 6, GPU call 28: the dependent chain on a highest-priority stream (bench.py --chain-priority high) against torch's default stream: call 22 saw
# 4.012 against 4.021 ms over three pairs, inside the spread -- eight alternating pairs to decide it; configs 5 and 4 two pairs each.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c28; mkdir -p $O
export TMPDIR=/tmp
B="--steps 100 --warmup 30 --no-cpu-baseline --no-optimizer --no-roofline"
run() { # label, args...
  lbl=$1; shift
  timeout 300 python bench.py $B $WL "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$lbl', j['ms_per_step'], j['value'], j['config'].get('chain_priority'))"
}
for i in 1 2 3 4 5 6 7 8; do
  run default --chain-priority default
  run high --chain-priority high
done > $O/ab_prio.log
cat $O/ab_prio.log
WL="--workload c5"; for i in 1 2 3; do run c5_default --chain-priority default; run c5_high --chain-priority high; done > $O/ab_prio_c5.log; cat $O/ab_prio_c5.log
WL="--workload c4"; for i in 1 2 3; do run c4_default --chain-priority default; run c4_high --chain-priority high; done > $O/ab_prio_c4.log; cat $O/ab_prio_c4.log
