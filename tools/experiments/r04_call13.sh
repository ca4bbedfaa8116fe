#!/bin/bash
O=gpurun_out/r4c13; mkdir -p $O
export TMPDIR=/tmp
T="tests/test_planner_gpu.py"
echo "---- MM32=0 W8=0"
ETP_MM32=0 ETP_GEMM_W8=0 timeout 600 python -m pytest $T -x -q --tb=line -k "golden or fresh or layer_ranges or issue_order" 2>&1 | tail -4
echo "---- MM32=0 only"
ETP_MM32=0 timeout 600 python -m pytest $T -x -q --tb=line -k "layer_ranges or issue_order" 2>&1 | tail -4
echo "---- default, layer_ranges + issue_order"
timeout 600 python -m pytest $T -x -q --tb=line -k "layer_ranges or issue_order" 2>&1 | tail -4
echo "---- default, issue_order only B=32"
timeout 600 python -m pytest $T -x -q --tb=line -k "issue_order and 32" 2>&1 | tail -4
