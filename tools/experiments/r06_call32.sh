#!/bin/bash
# round 6, GPU call 32: WARM again.  Call 25 measured NOTHING: the experiment's library did not compile (a nontemporal builtin on a HIP vector struct), the build's exit
# status was lost in a pipe, and the runs used the previous library, which ignores ETP_WARM.  This call first checks that the loaded library knows the switch.
# Variants: 1 = stash, nt loads; 5 = stash, plain loads (default cache policy); 7 = stash + weights, plain; 3 = stash + weights, nt.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c32; mkdir -p $O
export TMPDIR=/tmp
python - <<PY | tee $O/switch_check.log
import ctypes
L = ctypes.CDLL("etpnav_amd/libetpnav_hip.so")
L.etp_option_set.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
rc = L.etp_option_set(b"WARM", b"3")
print("etp_option_set(WARM) ->", rc, "(0 = the library knows the switch)")
raise SystemExit(0 if rc == 0 else 1)
PY
[ $? -eq 0 ] || { echo "library without WARM: abort"; exit 1; }
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
  run warm1 ETP_WARM=1
  run warm5 ETP_WARM=5
  run warm7 ETP_WARM=7
  run warm3 ETP_WARM=3
done > $O/ab_warm.log
cat $O/ab_warm.log
( ETP_WARM=7 timeout 300 python tools/chain_waits.py --steps 24 --out $O/chain_waits_warm7.txt > /dev/null 2>&1 ); grep "txt_bwd layer" $O/chain_waits_warm7.txt | cut -c1-125
( ETP_WARM=7 timeout 600 python -m pytest tests/test_planner_gpu.py -q -x -k "golden" 2>&1 | grep -v "amdgpu.ids" | tail -3 ) > $O/parity.log
cat $O/parity.log
