#!/bin/bash
# last check of the rebuilt final binary
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_planner_gpu.py tests/test_ops_gpu.py tests/test_mm32_gpu.py -q --tb=short -k "golden or layer_norm or mm32 or gemm" 2>&1 | tail -3
python __graft_entry__.py smoke 2>&1 | grep "^smoke"
