#!/bin/bash
# round 5, GPU call 29 (the last 36 s): real pano_embed_bwd (12 KB LDS) beside real GEMM kernels, standalone, no Python
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5c29; mkdir -p $O
timeout 25 etpnav_amd/build/r05_pano_bwd_neighbours $PWD/etpnav_amd/build/libetp_panoexpt.so 24 > $O/neighbours.txt 2>&1
echo "rc=$?" >> $O/neighbours.txt
cat $O/neighbours.txt
