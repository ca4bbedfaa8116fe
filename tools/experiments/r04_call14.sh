#!/bin/bash
export TMPDIR=/tmp
T="tests/test_planner_gpu.py"
for i in 1 2 3; do timeout 600 python -m pytest $T -q --tb=short -k "layer_ranges or issue_order" 2>&1 | grep -E "^E |passed|failed" | cut -c1-400; done
