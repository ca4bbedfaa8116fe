#!/bin/bash
# round 6, GPU call 7: per-episode K/V indirection in the batched rollout (N1) -- parity + rollout bench at B = 32 / 8; the four-wavefront rule
# of the fused out-projection dgrad -- A/B on config 2.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c7; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_baseline_shapes_gpu.py -q -x -k "rollout" 2>&1 | grep -v amdgpu.ids | tail -8 ) > $O/rollout_tests.log
cat $O/rollout_tests.log
( timeout 600 python tools/rollout_bench.py --B 32 --T 5,15 2>/dev/null | tail -1 ) > $O/rollout_bench_b32.json
( timeout 600 python tools/rollout_bench.py --B 8 --T 5,15 2>/dev/null | tail -1 ) > $O/rollout_bench.json
python - <<PY
import json
for f in ("$O/rollout_bench_b32.json", "$O/rollout_bench.json"):
    try:
        d = json.load(open(f))
        for r in d["rows"]:
            print(f.split("/")[-1], {k: v for k, v in r.items() if k.endswith("_ms") or k in ("T", "B")})
    except Exception as e:
        print(f, "FAILED", e)
PY
B="--steps 100 --warmup 30 --no-cpu-baseline --no-optimizer --no-roofline"
run() { # label, env...
  lbl=$1; shift
  env "$@" timeout 300 python bench.py $B $WL 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$lbl', j['ms_per_step'], j['value'], j['config'].get('env_overrides'))"
}
for i in 1 2 3; do run c2_rule X=1; run c2_off ETP_ATTN_PROJ=0; done > $O/ab_c2.log
cat $O/ab_c2.log
( timeout 900 python -m pytest tests/test_planner_gpu.py tests/test_variants_gpu.py -q -x 2>&1 | grep -v "amdgpu.ids" | tail -6 ) > $O/planner_tests.log
cat $O/planner_tests.log
