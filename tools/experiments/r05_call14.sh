#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5c14; mkdir -p $O
python tools/experiments/r05_b1_noise.py 2>&1 | grep -v amdgpu > $O/new.txt &
sleep 20
ETP_LIB=$PWD/etpnav_amd/build/libetp_base.so python tools/experiments/r05_b1_noise.py 2>&1 | grep -v amdgpu > $O/base.txt
wait
paste <(awk '{print $1,$2,$(NF-4)}' $O/new.txt) <(awk '{print $(NF-4)}' $O/base.txt)
