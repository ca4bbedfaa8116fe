#!/bin/bash
# mm32 128x64 with two slabs per hand-over: parity, phases, A/B
export TMPDIR=/tmp
O=gpurun_out/r4c30; mkdir -p $O
timeout 600 python -m pytest tests/test_mm32_gpu.py -x -q --tb=short 2>&1 | tail -3
timeout 600 python -m pytest tests/test_planner_gpu.py tests/test_baseline_shapes_gpu.py -x -q --tb=short -k "bf16" 2>&1 | tail -3
for i in 1 2; do
for c in 1 2; do
  for wl in c2 c4; do
  ETP_MM32_SPI=$c python bench.py --workload $wl --no-cpu-baseline --no-roofline --no-optimizer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('spi $c $wl', d['ms_per_step'], d['value'])"
  done
done; done | tee $O/ab.txt
python tools/gemm_phase_probe.py 2>/dev/null | grep "128x64\|sum of" | cut -c1-200 | tee $O/phases_spi2.txt
