#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5c9; mkdir -p $O
python tools/experiments/r05_gelu_epilogue_check.py 2>&1 | grep -v amdgpu > $O/new.txt
ETP_LIB=$PWD/etpnav_amd/build/libetp_v1.so python tools/experiments/r05_gelu_epilogue_check.py 2>&1 | grep -v amdgpu > $O/v1.txt
cat $O/new.txt $O/v1.txt | cut -c1-330
