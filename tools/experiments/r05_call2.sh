#!/bin/bash
# round 5, GPU call 2: the new parity tests (frozen / ablated variants vs the reference fixtures, fp32 1e-3 at B = 32, the determinism
# screen as a test, tightened mm32 bounds, FusedAdamW with frozen parameters)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r5c2
timeout 1500 python -m pytest tests/test_variants_gpu.py tests/test_mm32_gpu.py tests/test_optim_gpu.py -m gpu -q -s -x > gpurun_out/r5c2/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r5c2/tests.log
tail -30 gpurun_out/r5c2/tests.log
