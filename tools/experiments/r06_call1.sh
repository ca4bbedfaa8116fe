#!/bin/bash
# round 6, GPU call 1: parity of the split-reduction class, the neighbour matrix + aggressor bisect, isolated timing of the new class,
# same-box A/B of the step with MM32_K2 = 0 / 262 / 264.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
O=gpurun_out/r06c1
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_mm32_gpu.py -x -q 2>&1 | tail -15 ) > $O/mm32_tests.log
( timeout 600 python -m pytest tests/test_neighbours_gpu.py -q -s 2>&1 | tail -40 ) > $O/neighbours.log
for v in default half_epi pad_lds dma_only mfma_only noslp; do
  if [ $v == default ]; then unset ETP_LIB; else export ETP_LIB=$PWD/etpnav_amd/build/libetp_r6_$v.so; fi
  timeout 300 python tools/experiments/r06_neighbour_bisect.py 2>&1 | tail -3
done > $O/bisect.log
unset ETP_LIB
( timeout 600 python tools/experiments/r06_k2_bench.py 2>&1 | tail -15 ) > $O/k2_bench.log
for k2 in 0 262 264 0 264; do
  ETP_MM32_K2=$k2 timeout 300 python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-optimizer --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('MM32_K2=$k2', j['ms_per_step'], j['value'], j['config'].get('env_overrides'))"
done > $O/ab_k2.log
( timeout 300 python bench.py --workload c5 --steps 100 --warmup 30 --no-cpu-baseline --no-optimizer --no-roofline 2>/dev/null | cut -c1-300 ) > $O/c5.log
cat $O/mm32_tests.log $O/neighbours.log $O/bisect.log $O/k2_bench.log $O/ab_k2.log
