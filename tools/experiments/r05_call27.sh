#!/bin/bash
# round 5, GPU call 27: follow-up of call 26 (where do pano_embed_bwd's results differ, and beside which streams)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5c27; mkdir -p $O
ETP_LIB=$PWD/etpnav_amd/build/libetp_panoexpt.so BUDGET_S=60 timeout 90 python tools/experiments/r05_pano_bwd_isolation2.py > $O/isolation2.txt 2> $O/isolation2.err
echo "rc=$?" >> $O/isolation2.txt
cat $O/isolation2.txt; tail -5 $O/isolation2.err
