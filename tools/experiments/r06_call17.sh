#!/bin/bash
# round 6, GPU call 17: upper bound of hiding the residual read of the N = 768 products behind their reduction -- a timing-only build of
# (the -DETP_EXPT_SKIP_R hook in gemm_shared.h was removed together with the experiment: profiles/r06_ab_runs.json r6c17)
# gemm_mm32.hip that does not read the residual at all (results wrong), same-box A/B on config 2.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c17; mkdir -p $O
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
  run skip_r ETP_LIB=$PWD/etpnav_amd/build/libetp_r6_skipr.so
done > $O/ab_skip_r.log
cat $O/ab_skip_r.log
