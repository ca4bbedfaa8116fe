#!/bin/bash
# K/V cache under the batched rollout call: parity (golden + per-step equality), then the rollout bench
export TMPDIR=/tmp
O=gpurun_out/r4c27; mkdir -p $O
timeout 600 python -m pytest tests/test_baseline_shapes_gpu.py -q --tb=short -k "rollout or long_instruction" 2>&1 | tail -8
timeout 300 python tools/rollout_bench.py --B 8 --T 5,15 > $O/rollout_bench.json 2> $O/rb.err; cat $O/rollout_bench.json; tail -3 $O/rb.err
timeout 300 python tools/rollout_bench.py --B 32 --T 5,15 > $O/rollout_bench_b32.json 2> $O/rb32.err; cat $O/rollout_bench_b32.json; tail -3 $O/rb32.err
