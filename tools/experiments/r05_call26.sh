#!/bin/bash
# round 5, last GPU call (the remaining 2.9 GPU-minutes): what makes pano_embed_bwd nondeterministic when it shares CUs?
# (tools/experiments/r05_pano_bwd_isolation.py; library from r05_pano_bwd_isolation_build.py, product binary untouched)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5c26; mkdir -p $O
ETP_LIB=$PWD/etpnav_amd/build/libetp_panoexpt.so BUDGET_S=110 timeout 150 python tools/experiments/r05_pano_bwd_isolation.py > $O/isolation.txt 2> $O/isolation.err
echo "rc=$?" >> $O/isolation.txt
cat $O/isolation.txt; tail -5 $O/isolation.err
