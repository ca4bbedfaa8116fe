#!/bin/bash
# round 6, GPU call 16: the 96-MB gradient memset beside forward_navigation instead of beside text layer 0 (ETP_LATE_ZERO=1): A/B, parity of one step test.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c16; mkdir -p $O
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
  run late_zero ETP_LATE_ZERO=1
done > $O/ab_late_zero.log
cat $O/ab_late_zero.log
WL="--workload c5"; for i in 1 2; do run c5_base X=1; run c5_late_zero ETP_LATE_ZERO=1; done > $O/ab_late_zero_c5.log
WL="--workload c4"; for i in 1 2; do run c4_base X=1; run c4_late_zero ETP_LATE_ZERO=1; done >> $O/ab_late_zero_c5.log
cat $O/ab_late_zero_c5.log
( ETP_LATE_ZERO=1 timeout 900 python -m pytest tests/test_planner_gpu.py -q -x -k "golden or same_masks or issue_order" 2>&1 | grep -v "amdgpu.ids" | tail -4 ) > $O/parity.log
cat $O/parity.log
