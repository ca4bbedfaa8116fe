#!/bin/bash
# round 5, GPU call 8: bisect the bf16 error jump (15 % relative L2 on every gradient of the B = 1 fixture, 4.5 % with the base library)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5c8; mkdir -p $O
c=c1_single_episode
for v in v1 v2 base; do ETP_LIB=$PWD/etpnav_amd/build/libetp_$v.so python tools/parity_probe.py --case $c --top 3 2>&1 | grep -v amdgpu > $O/probe_$v.txt; done
python tools/parity_probe.py --case $c --top 3 2>&1 | grep -v amdgpu > $O/probe_new.txt
for v in v1 v2 base new; do echo "== $v"; head -8 $O/probe_$v.txt | cut -c1-120; done
