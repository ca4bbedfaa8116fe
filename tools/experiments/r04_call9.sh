#!/bin/bash
O=gpurun_out/r4c9; mkdir -p $O
export TMPDIR=/tmp
T="tests/test_planner_gpu.py::test_data_parallel_issue_order_gives_the_single_gpu_gradients"
for i in 1 2; do timeout 300 python -m pytest "$T" -q --tb=line 2>&1 | tail -6; done
echo "---- MM32=0"
ETP_MM32=0 timeout 300 python -m pytest "$T" -q --tb=line 2>&1 | tail -6
echo "---- only the failing id"
timeout 300 python -m pytest "$T[32-80-False]" -q --tb=line 2>&1 | tail -6
