#!/bin/bash
# round 6, GPU call 38 (NN products only -- call 37: -1.0 % in the step, the input-gradient epilogues 9.6 -> 4.3 / 4.3 -> 1.4 us, but the NT kernels that carry specialisations start 0.6 - 0.9 us later): call 36 showed that neither the stores (-0.5 .. -0.9 us), nor the read-back of the staged tile, nor the staging itself account for the 4.6 - 9.4 us
# epilogues of the 128x128 classes; their code does -- eight unrolled chunks, each carrying the WHOLE activation / dropout / residual / accumulate decision chain (43 KB
# kernels, most of it jumped over: instruction-cache misses at every taken branch).  Variant library -DETP_EPI_SPECIAL: the step's hot epilogue variants as compile-time
# specialisations (straight-line code).  Parity with the variant, phase probe, six alternating pairs.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c38; mkdir -p $O
export TMPDIR=/tmp
V=$PWD/etpnav_amd/build/libetp_r6_epispec_nn.so
( ETP_LIB=$V timeout 600 python -m pytest tests/test_mm32_gpu.py tests/test_ops_gpu.py -q -x 2>&1 | grep -v "amdgpu.ids" | tail -2 ) | tee $O/parity_ops.log
( ETP_LIB=$V timeout 900 python -m pytest tests/test_planner_gpu.py -q -x -k "golden" 2>&1 | grep -v "amdgpu.ids" | tail -2 ) | tee $O/parity_planner.log
sel() { grep "span\|NN,128x128\|NT,128x128\|NT,128x64,s3> 2560x768x3072\|NN,128x64,s3> 2560x768x3072\|NT,128x64,s3> 2560x768x768 \|NN,128x64,s3> 2560x768x768 \|NN,128x64,s3> 2560x768x2304\|sum of" | head -12; }
echo "== shipped"; ( timeout 300 python tools/gemm_phase_probe.py 2>/dev/null | sel ) | tee $O/phases_base.txt
echo "== ETP_EPI_SPECIAL"; ( ETP_LIB=$V timeout 300 python tools/gemm_phase_probe.py 2>/dev/null | sel ) | tee $O/phases_epispec.txt
B="--steps 100 --warmup 30 --no-cpu-baseline --no-optimizer --no-roofline"
run() { # label, env...
  lbl=$1; shift
  env "$@" timeout 300 python bench.py $B $WL 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$lbl', j['ms_per_step'], j['value'])"
}
for i in 1 2 3 4 5 6; do
  run base X=1
  run epispec ETP_LIB=$V
done > $O/ab_epispec.log
cat $O/ab_epispec.log
