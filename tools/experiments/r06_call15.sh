#!/bin/bash
# round 6, GPU call 15: evidence on the final build -- the full GPU suite (-s, durations), then tools/run_profiles.sh under the r06 prefix.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c15; mkdir -p $O
export TMPDIR=/tmp
( timeout 1700 python -m pytest tests -m gpu -q -s --durations=45 2>&1 | grep -v "amdgpu.ids\|Gloo\|socket.cpp" ) > $O/gpu_suite.log
tail -55 $O/gpu_suite.log | cut -c1-200
grep "bf16 worst" $O/gpu_suite.log | cut -c1-260 > $O/parity_bf16_observed.txt
ETP_ROUND=r06 SKIP_PARITY=1 bash tools/run_profiles.sh 2>&1 | tail -30
mkdir -p gpurun_out/profiles_r06; cp profiles/r06_bench_kernel_stats.csv profiles/r06_pmc_traffic.json gpurun_out/profiles_r06/ 2>/dev/null
