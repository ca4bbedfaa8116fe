#!/bin/bash
# LayerNorm backward with one row per wavefront (640 blocks at 2560 rows) against two (320 blocks): A/B + parity
export TMPDIR=/tmp
O=gpurun_out/r4c32; mkdir -p $O
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q --tb=short -k "layer_norm" 2>&1 | tail -2
for i in 1 2; do
for c in 512 1024; do
  for wl in c2; do
  ETP_LNBWD_GRID=$c python bench.py --workload $wl --no-cpu-baseline --no-roofline --no-optimizer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('lnbwd cap $c $wl', d['ms_per_step'], d['value'])"
  done
done; done | tee $O/ab.txt
