#!/bin/bash
# round 6, GPU call 34: sanity of the final tree after the experiments of calls 32 and 33 were reverted and the library rebuilt: smoke, the option / neighbour / fused-attention tests, one planner golden run, the bench line.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c34; mkdir -p $O
export TMPDIR=/tmp
python __graft_entry__.py smoke 2>&1 | grep "^smoke" | tee $O/smoke.log
( timeout 900 python -m pytest tests/test_neighbours_gpu.py tests/test_ops_gpu.py tests/test_mm32_gpu.py -q -x 2>&1 | grep -v amdgpu.ids | tail -3 ) | tee $O/ops.log
( timeout 900 python -m pytest tests/test_planner_gpu.py -q -x -k "golden or graph_replay or issue_order" 2>&1 | grep -v amdgpu.ids | tail -3 ) | tee $O/planner.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().split("\n")[-1])
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"], d["config"])
PY
