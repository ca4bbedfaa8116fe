#!/bin/bash
# write-through (sc1) epilogue stores: parity with the switch on, then in-step A/B
export TMPDIR=/tmp
O=gpurun_out/r4c28; mkdir -p $O
ETP_WT_STORES=1 timeout 600 python -m pytest tests/test_mm32_gpu.py tests/test_ops_gpu.py -x -q --tb=short -k "gemm or mm32" 2>&1 | tail -3
ETP_WT_STORES=1 timeout 300 python -m pytest tests/test_planner_gpu.py -x -q --tb=short -k "golden and not bf16" 2>&1 | tail -2
for i in 1 2 3; do
for c in 0 1; do
  ETP_WT_STORES=$c python bench.py --no-cpu-baseline --no-roofline --no-optimizer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('wt $c', d['ms_per_step'], d['value'])"
done; done | tee $O/ab.txt
