#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5c13; mkdir -p $O
python tools/experiments/r05_small_k_wgrad.py 2>&1 | grep -v amdgpu > $O/new.txt
ETP_LIB=$PWD/etpnav_amd/build/libetp_v1.so python tools/experiments/r05_small_k_wgrad.py 2>&1 | grep -v amdgpu > $O/v1.txt
paste $O/new.txt <(awk '{print $NF}' $O/v1.txt) | head -40
