#!/bin/bash
# round 6, GPU call 11: the graph-replay test that fails since call 9 (full traceback), then A/B of the layer-0 weight cast beside the
# embedding kernel and of the late navigation cast / gradient memset.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c11; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_planner_gpu.py -q -x -k "graph_replay" 2>&1 | grep -v amdgpu.ids | tail -60 ) > $O/graph_replay.log
grep -n "Error\|error\|^E " $O/graph_replay.log | head -20
( ETP_ASSEMBLE_ON_S2=0 timeout 600 python -m pytest tests/test_planner_gpu.py -q -x -k "graph_replay" 2>&1 | grep -v amdgpu.ids | tail -3 ) > $O/graph_replay_asm0.log
cat $O/graph_replay_asm0.log
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
  run cast0_main ETP_TXT_CAST0_SIDE=0
  run late_nav ETP_LATE_NAV_CAST=1
done > $O/ab_casts.log
cat $O/ab_casts.log
