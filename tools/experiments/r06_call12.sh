#!/bin/bash
# round 6, GPU call 12: parity / determinism / data-parallel-order / graph tests after the schedule changes of calls 9-11 (node assembly on the
# panorama stream, node-embedding backward as a leaf, d txt_embeds joined by its consumers with per-stream debts), then config-2 A/B against all off.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c12; mkdir -p $O
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests/test_planner_gpu.py tests/test_variants_gpu.py tests/test_dp_gpu.py tests/test_baseline_shapes_gpu.py tests/test_graph_gpu.py -q -x 2>&1 | grep -v "amdgpu.ids\|Gloo\|socket" | tail -25 ) > $O/parity.log
tail -25 $O/parity.log | cut -c1-250
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
  run r6_final X=1
  run r6_sched_off ETP_NAV_TAIL=0 ETP_ASSEMBLE_ON_S2=0 ETP_ATTN_PROJ=0
done > $O/ab_final.log
cat $O/ab_final.log
