#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5c11; mkdir -p $O
c=c1_single_episode
python tools/parity_probe.py --case $c --top 3 --overlap 0 2>&1 | grep -v amdgpu > $O/new_onestream.txt
python tools/parity_probe.py --case $c --top 3 --overlap 1 2>&1 | grep -v amdgpu > $O/new_overlap.txt
AMD_SERIALIZE_KERNEL=3 python tools/parity_probe.py --case $c --top 3 --overlap 1 2>&1 | grep -v amdgpu > $O/new_serialized.txt
ETP_GRAD_OVERWRITE=0 python tools/parity_probe.py --case $c --top 3 --overlap 1 2>&1 | grep -v amdgpu > $O/new_accumulate.txt
ETP_LIB=$PWD/etpnav_amd/build/libetp_v1.so python tools/parity_probe.py --case $c --top 3 --overlap 0 2>&1 | grep -v amdgpu > $O/v1_onestream.txt
for f in $O/*.txt; do echo "== $f"; sed -n 5,12p $f | cut -c1-120; done
