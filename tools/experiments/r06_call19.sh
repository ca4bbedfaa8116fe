#!/bin/bash
# round 6, GPU call 19: the synthetic packed-form reproducer extended by v_pk_mov_b32 (the one packed form with a low-half select that the
# shipped library still contains: 132 x op_sel:[1,0] in the mm32 GEMM epilogues) -- is the register-pair move affected too?
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c19; mkdir -p $O
export TMPDIR=/tmp
( PK_FORMS=1,2,10,11,12 timeout 600 python tools/experiments/r06_pk_opsel_repro.py 2>&1 | grep -v amdgpu.ids ) > $O/pk_mov_repro.txt
cut -c1-200 $O/pk_mov_repro.txt
