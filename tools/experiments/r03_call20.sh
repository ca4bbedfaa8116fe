#!/bin/bash
# round 3, call 20: what the attention kernels cost in the step (launches dropped, timing only), and the GEMM-only floor
set -x
O=gpurun_out/c20; mkdir -p $O
export TMPDIR=/tmp
B="--steps 100 --warmup 20 --no-cpu-baseline --no-optimizer --no-roofline"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B > $O/bench_$name.json 2> $O/bench_$name.err; python -c "import json; d=json.load(open('$O/bench_$name.json')); print('RESULT $name', d['value'], d['ms_per_step'], d['loss'])"; }
run default_a A=1
run no_attn_fwd ETP_SKIP_ATTN=fwd
run no_attn_bwd ETP_SKIP_ATTN=bwd
run no_attn ETP_SKIP_ATTN=fwd,bwd
run no_attn_no_ln ETP_SKIP_ATTN=fwd,bwd ETP_SKIP_LN=fwd,bwd,red
run gemm_chain_only ETP_SKIP_ATTN=fwd,bwd ETP_SKIP_LN=fwd,bwd,red ETP_SKIP_WGRAD=1
run default_b A=1
