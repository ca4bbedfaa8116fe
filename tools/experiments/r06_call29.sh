#!/bin/bash
# round 6, GPU call 29: the split-reduction 128x64 class (two wave groups, two rings of three) for the FORWARD one-per-CU products only (MM32_K2 = 1264: out-projection
# and FFN-down of the text forward, where no weight-gradient workgroup shares the CU -- call 1 measured +0.3 % with the class in BOTH directions, its 144-KB workgroup
# displacing the leaf workgroup in the backward; alone the K = 3072 forward product is 20.7 -> 19.5 us): eight alternating pairs + one parity run.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c29; mkdir -p $O
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
for i in 1 2 3 4 5 6 7 8; do
  run base X=1
  run k2fwd ETP_MM32_K2=1264
done > $O/ab_k2fwd.log
cat $O/ab_k2fwd.log
( ETP_MM32_K2=1264 timeout 600 python -m pytest tests/test_planner_gpu.py -q -x -k "golden" 2>&1 | grep -v "amdgpu.ids" | tail -3 ) > $O/parity.log
cat $O/parity.log
