#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5c18; mkdir -p $O
T="timeout 300 python -m pytest tests/test_variants_gpu.py -m gpu -q -x -s -k three_stream_step_reproduces_every_gradient[c5]"
ETP_LIB=$PWD/etpnav_amd/build/libetp_panolds.so ETP_PANO_LDS_ALL=1 $T 2>&1 | grep "tensors above" | cut -c1-300 > $O/lds_all.txt
ETP_LIB=$PWD/etpnav_amd/build/libetp_panolds.so $T 2>&1 | grep "tensors above" | cut -c1-300 > $O/lds_110k.txt
for f in lds_all lds_110k; do echo "== $f"; head -1 $O/$f.txt; done
