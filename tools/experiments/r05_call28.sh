#!/bin/bash
# round 5, GPU call 28: are the FLAT loads of pano_embed_bwd (pointer laundering) what goes wrong beside other streams?
# same screen with a library whose kernels load the same vectors through global_load (r05_pano_bwd_noflat_build.py), 12 KB LDS
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5c28; mkdir -p $O
ETP_LIB=$PWD/etpnav_amd/build/libetp_panoexpt.so SAVE_REF=/tmp/ref_flat WL=c2,c5 timeout 40 python tools/experiments/r05_pano_bwd_isolation2.py > $O/ref_flat.txt 2>&1
ETP_LIB=$PWD/etpnav_amd/build/libetp_panonoflat.so CMP_REF=/tmp/ref_flat WL=c2,c5,sap RUNS=20 MODES=0,2,1 BUDGET_S=45 timeout 70 python tools/experiments/r05_pano_bwd_isolation2.py > $O/noflat.txt 2> $O/noflat.err
echo "rc=$?" >> $O/noflat.txt
cat $O/ref_flat.txt | tail -4; cat $O/noflat.txt; tail -3 $O/noflat.err
