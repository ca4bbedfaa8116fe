#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5c10; mkdir -p $O
c=c1_single_episode
for v in v1 new; do
  if [ $v == new ]; then L=""; else L=$PWD/etpnav_amd/build/libetp_$v.so; fi
  ETP_LIB=$L python tools/parity_probe.py --case $c --top 2 2>&1 | grep -v amdgpu > $O/probe_$v.txt
  ETP_LIB=$L ETP_MM32=0 python tools/parity_probe.py --case $c --top 2 2>&1 | grep -v amdgpu > $O/probe_${v}_nomm32.txt
  ETP_LIB=$L ETP_GEMM_SMALL=0 python tools/parity_probe.py --case $c --top 2 2>&1 | grep -v amdgpu > $O/probe_${v}_nosmall.txt
done
for f in $O/probe_*.txt; do echo "== $f"; grep "^# output\|^# named.*LayerNorm.bias" $f | cut -c1-120; done
