#!/bin/bash
# 32x64 small-grid tile class: parity, phases, A/B on configs 2 and 5
export TMPDIR=/tmp
O=gpurun_out/r4c29; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_mm32_gpu.py -x -q --tb=short -k "gemm" 2>&1 | tail -3
ETP_GEMM_TILE=32 timeout 600 python -m pytest tests/test_ops_gpu.py -x -q --tb=short -k "gemm and not group and not fp32" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_planner_gpu.py tests/test_baseline_shapes_gpu.py -x -q --tb=short -k "bf16" 2>&1 | tail -3
for i in 1 2; do
for c in 0 1; do
  for wl in c2 c5; do
  ETP_GEMM_SMALL=$c python bench.py --workload $wl --no-cpu-baseline --no-roofline --no-optimizer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('small $c $wl', d['ms_per_step'], d['value'])"
  done
done; done | tee $O/ab.txt
python tools/gemm_phase_probe.py 2>/dev/null | grep "32x64\|grid 96\|grid 192" | cut -c1-200 | tee $O/phases_small.txt
