#!/bin/bash
# 256x128 grouped weight-gradient class: parity, isolated timing, in-step A/B
export TMPDIR=/tmp
O=gpurun_out/r4c25; mkdir -p $O
timeout 900 python -m pytest tests/test_mm32_gpu.py -x -q --tb=short -k "grouped" 2>&1 | tail -8
python tools/experiments/r04_group_class_probe.py 2>/dev/null | tee $O/group_class.json
for i in 1 2; do
for c in 128 0; do
  ETP_MM32_GROUP=$c python bench.py --no-cpu-baseline --no-roofline --no-optimizer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('group class $c', d['ms_per_step'], d['value'])"
done; done | tee $O/ab.txt
