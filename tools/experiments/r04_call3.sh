#!/bin/bash
set -x
O=gpurun_out/r4c3; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/experiments/r04_dual_chain_probe.py > $O/dual_chain.txt 2>&1; echo "rc $?"; cat $O/dual_chain.txt | tail -12
