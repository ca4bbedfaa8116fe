#!/bin/bash
# round 6, GPU call 25: WARM -- read-only sweeps of the NEXT backward layer's forward stash (bit 0) and bf16 weights (bit 1) on the aux2 stream,
# one layer ahead of the text backward: every backward kernel starts with a cold HBM read (stash written > 2 ms / > 1 GB of traffic earlier);
# the sweep moves those reads off the chain into the memory-side Infinity Cache.  A/B + stamps + one golden run.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c25; mkdir -p $O
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
  run base X=1
  run warm1 ETP_WARM=1
  run warm3 ETP_WARM=3
  run warm2 ETP_WARM=2
done > $O/ab_warm.log
cat $O/ab_warm.log
( ETP_WARM=3 timeout 300 python tools/chain_waits.py --steps 24 --out $O/chain_waits_warm3.txt > /dev/null 2>&1 ); grep "txt_bwd layer" $O/chain_waits_warm3.txt | cut -c1-125
( ETP_WARM=3 timeout 600 python -m pytest tests/test_planner_gpu.py -q -x -k "golden" 2>&1 | grep -v "amdgpu.ids" | tail -3 ) > $O/parity.log
cat $O/parity.log
WL="--workload c5"; for i in 1 2; do run c5_base X=1; run c5_warm3 ETP_WARM=3; done > $O/ab_warm_c5.log; cat $O/ab_warm_c5.log
WL="--workload c4"; for i in 1 2; do run c4_base X=1; run c4_warm3 ETP_WARM=3; done > $O/ab_warm_c4.log; cat $O/ab_warm_c4.log
