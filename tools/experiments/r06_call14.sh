#!/bin/bash
# round 6, GPU call 14: node-assembly backward on the panorama stream, the text operand cast of etp_nav_fwd on the side stream: A/B + stamps + parity.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c14; mkdir -p $O
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
  run new X=1
  run asm_bwd_chain ETP_ASSEMBLE_BWD_ON_S2=0
done > $O/ab_asm_bwd.log
cat $O/ab_asm_bwd.log
( timeout 300 python tools/chain_waits.py --steps 24 --out $O/chain_waits.txt > /dev/null 2>&1 ); grep "join panorama\|nav_bwd returned\|node assembly bwd\|txt_bwd begin\|step time" $O/chain_waits.txt | head -8 | cut -c1-130
( timeout 1500 python -m pytest tests/test_planner_gpu.py tests/test_variants_gpu.py tests/test_dp_gpu.py tests/test_baseline_shapes_gpu.py -q -x 2>&1 | grep -v "amdgpu.ids\|Gloo\|socket" | tail -5 ) > $O/parity.log
cat $O/parity.log
