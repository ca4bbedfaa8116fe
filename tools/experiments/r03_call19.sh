#!/bin/bash
# round 3, call 19: upper bounds for the LayerNorm fusions -- the step with the LN launches dropped (results wrong, timing only)
set -x
O=gpurun_out/c19; mkdir -p $O
export TMPDIR=/tmp
B="--steps 100 --warmup 20 --no-cpu-baseline --no-optimizer --no-roofline"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B > $O/bench_$name.json 2> $O/bench_$name.err; python -c "import json; d=json.load(open('$O/bench_$name.json')); print('RESULT $name', d['value'], d['ms_per_step'], d['loss'])"; }
run default_a A=1
run no_lnfwd ETP_SKIP_LN=fwd
run no_lnbwd ETP_SKIP_LN=bwd
run no_lnred ETP_SKIP_LN=red
run no_ln_all ETP_SKIP_LN=fwd,bwd,red
run default_b A=1
run no_ln_all_no_wgrad ETP_SKIP_LN=fwd,bwd,red ETP_SKIP_WGRAD=1
