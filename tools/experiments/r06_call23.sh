#!/bin/bash
# round 6, GPU call 23: bench.py after the make_out / LineWatchdog refactor -- (1) the N = 1 line still carries roofline + cpu_baseline + optimizer,
# (2) the two-rank self-launch test of the suite, (3) the watchdog in a real two-rank run: rank 1 never reaches the post-metric legs
# (ETP_BENCH_TEST_HANG=1), the line must appear after ETP_BENCH_LEG_TIMEOUT with comm.watchdog set and the job must end with status 0.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c23; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().split("\n")[-1])
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"], d["optimizer"]["train_iteration"]["ms"], sorted(d))
PY
( timeout 900 python -m pytest tests/test_dp_gpu.py -q -x -k "self_launches" 2>&1 | grep -v amdgpu.ids | tail -3 ) | tee $O/self_launch.log
t0=$(date +%s)
ETP_BENCH_TEST_HANG=1 ETP_BENCH_LEG_TIMEOUT=15 timeout 600 python bench.py --gpus 2 --dist-backend gloo --same-device --steps 2 --warmup 1 --settle 2 > $O/hang.json 2> $O/hang.err
echo "hang run: rc $? after $(( $(date +%s) - t0 )) s" | tee $O/hang.log
python - <<PY | tee -a $O/hang.log
import json
ls = [l for l in open("$O/hang.json") if l.startswith("{")]
print("lines", len(ls))
d = json.loads(ls[-1])
print("n_gpus", d["n_gpus"], "value", d["value"], "comm", d["comm"], "roofline", d["roofline"], "cpu_baseline", d["cpu_baseline"])
PY
tail -5 $O/hang.err
