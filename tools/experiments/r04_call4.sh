#!/bin/bash
set -x
O=gpurun_out/r4c4; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python tools/mm32_probe.py > $O/probe_full.json 2> $O/probe_full.err; echo rc $?
ETP_LIB=etpnav_amd/lib_mm32_dmaonly.so timeout 200 python tools/mm32_probe.py > $O/probe_dmaonly.json 2> $O/probe_dma.err; echo rc $?
ETP_LIB=etpnav_amd/lib_mm32_mfmaonly.so timeout 200 python tools/mm32_probe.py > $O/probe_mfmaonly.json 2> $O/probe_mfma.err; echo rc $?
ETP_MM32=0 timeout 200 python tools/mm32_probe.py > $O/probe_old.json 2> $O/probe_old.err; echo rc $?
python - <<'PY'
import json
O='gpurun_out/r4c4/'
f=json.load(open(O+'probe_full.json')); d=json.load(open(O+'probe_dmaonly.json')); m=json.load(open(O+'probe_mfmaonly.json')); o=json.load(open(O+'probe_old.json'))
print(f"{'shape':34s} {'old':>8s} {'full':>8s} {'dma':>8s} {'mfma':>8s}")
for k in f: print(f"{k:34s} {o[k]:8.2f} {f[k]:8.2f} {d[k]:8.2f} {m[k]:8.2f}")
PY
