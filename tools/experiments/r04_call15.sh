#!/bin/bash
export TMPDIR=/tmp
cd _r3
for i in 1 2 3; do timeout 600 python -m pytest tests/test_planner_gpu.py -q --tb=short -k "layer_ranges or issue_order" 2>&1 | grep -E "^E |passed|failed" | cut -c1-300; done
