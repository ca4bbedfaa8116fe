#!/bin/bash
O=gpurun_out/r4c10; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q --tb=short -k "attention" > $O/attn_tests.log 2>&1; echo "rc attn $?"; tail -5 $O/attn_tests.log
timeout 900 python -m pytest tests/test_planner_gpu.py -x -q --tb=short -k "train_mode or bf16_step or golden" > $O/planner_tests.log 2>&1; echo "rc planner $?"; tail -5 $O/planner_tests.log
timeout 300 python tools/attn_bench.py > $O/attn_bench_new.json 2> $O/ab.err; ETP_ATTN_BWD2=0 timeout 300 python tools/attn_bench.py > $O/attn_bench_old.json 2>> $O/ab.err
python - <<'PY'
import json
n=json.load(open('gpurun_out/r4c10/attn_bench_new.json')); o=json.load(open('gpurun_out/r4c10/attn_bench_old.json'))
for k in n: print(f"{k:44s} fwd {n[k]['fwd_us']:6.2f}  bwd new {n[k]['bwd_us']:6.2f} old {o[k]['bwd_us']:6.2f}   isolated bwd new {n[k]['bwd_isolated_us']:6.2f} old {o[k]['bwd_isolated_us']:6.2f}")
PY
B="--steps 100 --warmup 20 --no-cpu-baseline --no-optimizer --no-roofline"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B > $O/bench_$name.json 2> $O/bench_$name.err; python -c "import json; d=json.load(open('$O/bench_$name.json')); print('RESULT $name', d['value'], d['ms_per_step'], d['loss'])"; }
run new A=1
run oldattn ETP_ATTN_BWD2=0
run new_b A=1
run oldattn_b ETP_ATTN_BWD2=0
