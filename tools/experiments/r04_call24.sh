#!/bin/bash
# full GPU suite + smoke + default bench on the round-4 tree
export TMPDIR=/tmp
O=gpurun_out/r4c24; mkdir -p $O
timeout 1500 python -m pytest tests/ -q -m gpu --tb=short -rs 2>&1 | tee $O/gpu_tests.log | tail -25
python __graft_entry__.py smoke 2>&1 | grep -v "^/opt/rocm\|amdgpu" | tee $O/smoke.log | tail -4
python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 3000 $O/bench_c2.json; tail -5 $O/bench_c2.err
