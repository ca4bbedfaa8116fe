#!/bin/bash
# round 5, GPU call 1: run the LN-prologue prototype (compiled in round 4, never run) against the product's two-launch forms
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r5c1
timeout 600 python tools/experiments/r05_ln_prologue_test.py > gpurun_out/r5c1/ln_prologue.log 2>&1
echo "rc=$?" >> gpurun_out/r5c1/ln_prologue.log
tail -40 gpurun_out/r5c1/ln_prologue.log
