#!/bin/bash
# round 6, GPU call 3: the full GPU suite with per-test durations on the round-6 library; the extended packed-form reproducer.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c3; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python tools/experiments/r06_pk_opsel_repro.py 2>&1 | grep -v amdgpu.ids ) > $O/pk_opsel_repro_ext.txt
( timeout 1500 python -m pytest tests -m gpu -q -x --durations=70 2>&1 | grep -v "amdgpu.ids\|Gloo\|socket.cpp" | tail -120 ) > $O/gpu_suite.log
tail -90 $O/gpu_suite.log | cut -c1-220
grep -v "^alone\|128x64\|   0 /         0" $O/pk_opsel_repro_ext.txt | cut -c1-220
