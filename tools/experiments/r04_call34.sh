#!/bin/bash
# LayerNorm row reductions through DPP instead of ds_bpermute: parity, then A/B against the bpermute build (ETP_LIB)
export TMPDIR=/tmp
O=gpurun_out/r4c34; mkdir -p $O
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q --tb=short -k "layer_norm" 2>&1 | tail -2
timeout 400 python -m pytest tests/test_planner_gpu.py -x -q --tb=short -k "golden" 2>&1 | tail -2
for i in 1 2; do
for c in etpnav_amd/lib_ln_bpermute.so etpnav_amd/libetpnav_hip.so; do
  ETP_LIB=$PWD/$c python bench.py --no-cpu-baseline --no-roofline --no-optimizer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$c', d['ms_per_step'], d['value'])"
done; done | tee $O/ab.txt
