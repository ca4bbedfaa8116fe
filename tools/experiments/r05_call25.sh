#!/bin/bash
# round 5, final GPU call: the round's profile evidence on the final binary, then the whole GPU suite (-s: the bf16 "worst" lines)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r5_final
SKIP_PARITY=1 SKIP_CHAIN=1 bash tools/run_profiles.sh > gpurun_out/r5_final/profiles.log 2>&1
grep "^bench\|^smoke" gpurun_out/r5_final/profiles.log | cut -c1-200
timeout 1000 python -m pytest tests -m gpu -q -s --durations=8 > gpurun_out/r5_final/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r5_final/tests.log
tail -14 gpurun_out/r5_final/tests.log | cut -c1-200
