#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5c17; mkdir -p $O
T="timeout 300 python -m pytest tests/test_variants_gpu.py -m gpu -q -x -s -k three_stream_step_reproduces_every_gradient[c5]"
ETP_STREAM_PRIO=0 $T 2>&1 | grep "tensors above" | cut -c1-300 > $O/noprio.txt
ETP_WGRAD_GROUP=0 $T 2>&1 | grep "tensors above" | cut -c1-300 > $O/nogroup.txt
ETP_MM32=0 $T 2>&1 | grep "tensors above" | cut -c1-300 > $O/nomm32.txt
for f in noprio nogroup nomm32; do echo "== $f"; head -1 $O/$f.txt; done
