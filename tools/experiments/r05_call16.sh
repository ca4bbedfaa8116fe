#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5c16; mkdir -p $O
T="timeout 300 python -m pytest tests/test_variants_gpu.py -m gpu -q -x -s -k three_stream_step_reproduces_every_gradient[c5]"
$T 2>&1 | grep "tensors above" | cut -c1-400 > $O/new.txt
ETP_LIB=$PWD/etpnav_amd/build/libetp_v2.so $T 2>&1 | grep "tensors above" | cut -c1-400 > $O/v2_old_embed.txt
AMD_SERIALIZE_KERNEL=3 $T 2>&1 | grep "tensors above" | cut -c1-400 > $O/new_serialized.txt
for f in new v2_old_embed new_serialized; do echo "== $f"; cat $O/$f.txt; done
