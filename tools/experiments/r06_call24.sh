#!/bin/bash
# round 6, GPU call 24: the full GPU suite on the final tree (bench.py refactored since the evidence run of call 15; library sources unchanged).
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c24; mkdir -p $O
export TMPDIR=/tmp
( timeout 1700 python -m pytest tests -m gpu -q -s --durations=45 2>&1 | grep -v "amdgpu.ids\|Gloo\|socket.cpp" ) > $O/gpu_suite.log
tail -52 $O/gpu_suite.log | cut -c1-200
