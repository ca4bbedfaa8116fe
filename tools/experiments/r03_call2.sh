#!/bin/bash
# Round-3 GPU call 2: does the GEMM loop serialise DMA issue and MFMA work inside a wavefront?  (ETP_GEMM_EXPT builds)
# + the parity tests that changed (calibrated bf16 bounds) + the new multi-GPU / accumulation / benchmarked-shape tests.
set -x
O=gpurun_out/c2; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
T="timeout 600"
for e in e1 e2; do
  ETP_LIB=$R/etpnav_amd/libetpnav_hip_$e.so $T python tools/gemm_sweep.py > $O/gemm_sweep_$e.json 2> $O/gemm_sweep_$e.err
done
$T python tools/gemm_sweep.py > $O/gemm_sweep_full.json 2> $O/gemm_sweep_full.err
timeout 1200 python -m pytest tests/test_dp_gpu.py "tests/test_planner_gpu.py" tests/test_baseline_shapes_gpu.py -m gpu -q -rP --tb=short \
   -k "gather_rows or self_launches or accumulation or other_benchmarked or bf16 or mlm or sap or two_rank or world1" > $O/parity.log 2>&1; echo "rc parity $?"
grep -n "passed\|failed" $O/parity.log | tail -3
