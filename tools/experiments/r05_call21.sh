#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5c21; mkdir -p $O
timeout 300 python tools/gemm_phase_probe.py > $O/phases_fp16.txt 2> $O/e1
ETP_LIB=$PWD/etpnav_amd/build/libetp_zbf16.so timeout 300 python tools/gemm_phase_probe.py > $O/phases_bf16.txt 2> $O/e2
ETP_LIB=$PWD/etpnav_amd/build/libetp_nomath.so timeout 300 python tools/gemm_phase_probe.py > $O/phases_nomath.txt 2> $O/e3
grep "2560x3072x768" $O/phases_*.txt | cut -c1-200
