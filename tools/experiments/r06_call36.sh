#!/bin/bash
# round 6, GPU call 36: where the epilogues' time goes -- three TIMING-ONLY variant libraries (results wrong by design; -DETP_EPI_EXPT=1: the tile is computed but not stored,
# 2: the staged tile is not read back from LDS, 3: the accumulators are not staged) under the in-kernel phase probe, against the shipped library.  + the sanity of the shipped library
# after the walker experiment's revert (smoke, ops tests, bench).
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c36; mkdir -p $O
export TMPDIR=/tmp
sel() { grep "span\|NN,128x128\|NT,128x128\|mm32_group<bf16,f32,TN,128x128,s2> 3072\|NT,128x64,s3> 2560x768x3072\|NN,128x64,s3> 2560x768x3072\|NT,128x64,s3> 2560x768x768 " | head -9; }
echo "== shipped"; ( timeout 300 python tools/gemm_phase_probe.py 2>/dev/null | sel ) | tee $O/phases_base.txt
for v in 1 2 3; do
  echo "== ETP_EPI_EXPT=$v"; ( ETP_LIB=$PWD/etpnav_amd/build/libetp_r6_epi$v.so timeout 300 python tools/gemm_phase_probe.py 2>/dev/null | sel ) | tee $O/phases_epi$v.txt
done
python __graft_entry__.py smoke 2>&1 | grep "^smoke" | tee $O/smoke.log
( timeout 900 python -m pytest tests/test_neighbours_gpu.py tests/test_ops_gpu.py tests/test_mm32_gpu.py -q -x 2>&1 | grep -v amdgpu.ids | tail -2 ) | tee $O/ops.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().split("\n")[-1])
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
