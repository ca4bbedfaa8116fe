#!/bin/bash
# round 6, GPU call 2: (a) synthetic packed-multiply reproducer, (b) neighbour matrix on the library whose row kernels carry no packed
# fp32 instructions (default mask and everything shared), (c) parity of the gemm.hip classes after the transposed-operand swizzle,
# (d) SQ counters (LDS conflicts of the small classes) on config 5, (e) A/B: ROW_EXCLUSIVE default / 0, config 5 bench.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c2; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
( timeout 600 python tools/experiments/r06_pk_opsel_repro.py 2>&1 | grep -v amdgpu.ids ) > $O/pk_opsel_repro.txt
( timeout 600 python -m pytest tests/test_neighbours_gpu.py -q -s 2>&1 | grep -v amdgpu.ids | tail -20 ) > $O/neighbours.log
( timeout 300 python tools/experiments/r06_neighbour_bisect.py 2>&1 | tail -2 ) > $O/bisect_nopk.log
( timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_mm32_gpu.py -x -q 2>&1 | tail -6 ) > $O/gemm_tests.log
P="--workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-optimizer --no-roofline"
(cd /tmp && timeout 420 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/ps -o p -- python $R/bench.py $P > /dev/null 2> $R/$O/ps.err)
python tools/pmc_sq.py $O/ps/p_counter_collection.csv --out $O/gemm_counters_c5.json > $O/gemm_counters_c5.txt 2>&1
rm -rf $O/ps
for x in default 0 default 0; do
  if [ $x == default ]; then unset ETP_ROW_EXCLUSIVE; else export ETP_ROW_EXCLUSIVE=$x; fi
  timeout 300 python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-optimizer --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('ROW_EXCLUSIVE=$x', j['ms_per_step'], j['value'], j['config'].get('env_overrides'))"
done > $O/ab_excl.log
unset ETP_ROW_EXCLUSIVE
for i in 1 2; do timeout 300 python bench.py --workload c5 --steps 100 --warmup 30 --no-cpu-baseline --no-optimizer --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('c5', j['ms_per_step'], j['value'])"; done > $O/c5.log
cat $O/pk_opsel_repro.txt $O/neighbours.log $O/bisect_nopk.log $O/gemm_tests.log $O/ab_excl.log $O/c5.log; grep -i "32x64\|64x64" $O/gemm_counters_c5.txt | head -20
