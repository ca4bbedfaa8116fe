#!/bin/bash
# round 5, GPU call 7: which change moved embeddings.LayerNorm.bias of the B = 1 fixture (bf16)?  default / round-4 GELU pair on the
# new library / the base library
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5c7; mkdir -p $O
for c in c1_single_episode; do
python tools/parity_probe.py --case $c > $O/probe_new_$c.txt 2>&1
ETP_LIB=$PWD/etpnav_amd/build/libetp_oldgelu.so python tools/parity_probe.py --case $c > $O/probe_oldgelu_$c.txt 2>&1
ETP_LIB=$PWD/etpnav_amd/build/libetp_base.so python tools/parity_probe.py --case $c > $O/probe_base_$c.txt 2>&1
done
cat $O/probe_*c1*.txt | grep -v amdgpu | cut -c1-150
