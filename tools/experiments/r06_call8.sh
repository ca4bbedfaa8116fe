#!/bin/bash
# round 6, GPU call 8: (a) the extended determinism screen (config 4, fp32 mode, MLM) + the new indirection test; (b) HIP-runtime switches never
# tried on this step (kernel-boundary fences, kernarg placement, dispatch path): same-box A/B on config 2.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c8; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_variants_gpu.py -q -x -s -k "reproduces_every_gradient" 2>&1 | grep -v amdgpu.ids | tail -14 ) > $O/screen_tests.log
cat $O/screen_tests.log | cut -c1-300
( timeout 300 python -m pytest tests/test_baseline_shapes_gpu.py -q -x -k "kv_indirection" 2>&1 | grep -v amdgpu.ids | tail -4 ) > $O/indirection_test.log
cat $O/indirection_test.log
B="--steps 100 --warmup 30 --no-cpu-baseline --no-optimizer --no-roofline"
run() { # label, env...
  lbl=$1; shift
  env "$@" timeout 300 python bench.py $B $WL 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$lbl', j['ms_per_step'], j['value'], j['config'].get('env_overrides'))"
}
for i in 1 2; do
  run base X=1
  run opt_flush0 AMD_OPT_FLUSH=0; run opt_flush1 AMD_OPT_FLUSH=1
  run dev_kernarg0 HIP_FORCE_DEV_KERNARG=0; run dev_kernarg1 HIP_FORCE_DEV_KERNARG=1
  run direct0 AMD_DIRECT_DISPATCH=0
  run sysscope0 ROC_SYSTEM_SCOPE_SIGNAL=0
  run hdpwa0 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0; run hdpwa1 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1
done > $O/ab_runtime.log
cat $O/ab_runtime.log
