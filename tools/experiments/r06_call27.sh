#!/bin/bash
# round 6, GPU call 27: the full GPU suite once more, in the driver form (-x -q), on the final tree: a flakiness check, and the first full run with tests/test_zz_two_gpu.py last.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c27; mkdir -p $O
export TMPDIR=/tmp
( timeout 1700 python -m pytest tests -m gpu -x -q --durations=15 2>&1 | grep -v "amdgpu.ids\|Gloo\|socket.cpp" ) > $O/gpu_suite.log
tail -52 $O/gpu_suite.log | cut -c1-200
