"""Is the step host-bound?  Times the HOST side of PlannerStep.run_eager (enqueue only, no synchronisation inside the loop)
against the wall time of the same steps.   python tools/host_timing.py [--steps 100]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from etpnav_amd.planner import GlocalTextPathNavCMT, default_config
from etpnav_amd.step import PlannerStep
from etpnav_amd.synthetic import make_batch
from bench import WORKLOADS

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=100); ap.add_argument("--workload", default="c2")
a = ap.parse_args()
w = WORKLOADS[a.workload]
cfg = default_config(w["task"], image_feat_size=w["image_feat_size"])
model = GlocalTextPathNavCMT(cfg, dtype=torch.bfloat16, device="cuda"); model.init_weights(seed=0)
batch = make_batch(cfg.vocab_size, cfg.image_feat_size, cfg.depth_feat_size, w["B"], w["L"], w["V"], w["G"], seed=1234)
step = PlannerStep(model, batch, dropout="config")
for _ in range(20):
    step.run_eager()
torch.cuda.synchronize()
host = 0.0
t0 = time.perf_counter()
for _ in range(a.steps):
    h0 = time.perf_counter()
    step.run_eager()
    host += time.perf_counter() - h0
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
# host cost with an EMPTY queue in front (each step synchronised first): the pure issue cost, no back-pressure from a full queue
pure = 0.0
for _ in range(20):
    torch.cuda.synchronize()
    h0 = time.perf_counter(); step.run_eager(); pure += time.perf_counter() - h0
torch.cuda.synchronize()
print(f"wall {wall / a.steps * 1e3:.3f} ms/step   host issue inside the loop {host / a.steps * 1e3:.3f} ms/step   "
      f"loop returned after {t_enq / a.steps * 1e3:.3f} ms/step   pure host issue (idle GPU) {pure / 20 * 1e3:.3f} ms/step")
step.close()
