#!/bin/bash
# round 6, GPU call 43: the full GPU suite in the driver's form on the FINAL library (NN + TN epilogue specialisations).
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c43; mkdir -p $O
export TMPDIR=/tmp
( timeout 780 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "amdgpu.ids\|Gloo\|socket.cpp" ) > $O/gpu_suite.log
tail -6 $O/gpu_suite.log | cut -c1-200
