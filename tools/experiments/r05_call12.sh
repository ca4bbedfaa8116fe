#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5c12; mkdir -p $O
c=c1_single_episode
python tools/parity_probe.py --case $c --top 1 --overlap 0 2>&1 | grep -v amdgpu > $O/new.txt
ETP_LIB=$PWD/etpnav_amd/build/libetp_v1.so python tools/parity_probe.py --case $c --top 1 --overlap 0 2>&1 | grep -v amdgpu > $O/v1.txt
paste <(grep "^# chain" $O/new.txt | awk '{print $4, $7}') <(grep "^# chain" $O/v1.txt | awk '{print $4}')
