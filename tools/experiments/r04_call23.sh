#!/bin/bash
export TMPDIR=/tmp
S=tools/experiments/r04_gmap_pos_flake2.py
python $S 2>&1 | grep -v amdgpu | cut -c1-170
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q --tb=short -k "gmap or embed" 2>&1 | tail -3
for i in 1 2; do timeout 600 python -m pytest tests/test_planner_gpu.py -q --tb=short -k "layer_ranges or issue_order" 2>&1 | grep -E "^E |passed|failed" | cut -c1-300; done
