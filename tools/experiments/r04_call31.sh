#!/bin/bash
# final: full GPU suite on the final tree, then the evidence script
export TMPDIR=/tmp
O=gpurun_out/r4c31; mkdir -p $O
timeout 1500 python -m pytest tests/ -q -m gpu --tb=short -rs 2>&1 | tee $O/gpu_tests.log | tail -12
bash tools/run_profiles.sh 2>&1 | tail -30
