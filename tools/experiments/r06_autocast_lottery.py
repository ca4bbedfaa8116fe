"""Round 6: is the per-seed bf16-autocast gap of a single-episode step a stable number?  The c1 fixture's batch (seed 1234) and seed 6
(the HIP path's worst seed in profiles/r05_b1_noise.txt) are re-run with the panorama features scaled by (1 + j * 1e-4), j = 0..7 -- far
below bf16 resolution, the fp32 reference step moves by ~1e-4 -- and the REAL reference module's bf16-autocast gradients are compared
with its own fp32 gradients on the same perturbed input.  If the gap were a property of the seed, the eight numbers would agree.
    python tools/experiments/r06_autocast_lottery.py >> profiles/r06_autocast_gap.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
torch.set_num_threads(int(os.environ.get("GAP_THREADS", "4")))
from oracle import planner_oracle as po
from oracle import ref_harness as rh
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from r06_autocast_gap import gap

cfg = po.PlannerConfig.r2r()
P = po.init_params(cfg, seed=0)
model = rh.build_reference_model(cfg, P)
print("# --- lottery check: the same single-episode batch with rgb_fts scaled by (1 + j * 1e-4); median / max relative L2 of the autocast gradients")
for seed in (1234, 6):
    base = po.make_batch(cfg, seed=seed, B=1, L=20, V=17, G=9, ragged=False)
    meds = []
    for j in range(8):
        b = dict(base); b["rgb_fts"] = base["rgb_fts"] * (1.0 + j * 1e-4)
        r = gap(model, b)
        meds.append(r["median"])
        print(f"seed {seed} j {j}: loss {r['loss16']:.4f} (fp32 {r['loss32']:.4f}); median {r['median']:.4f}, max {r['max']:.4f}, worst sample/abs-max {r['worst_sample']:.4f}", flush=True)
    print(f"# seed {seed}: median gap ranges {min(meds):.4f} .. {max(meds):.4f} over eight inputs that agree to 7e-4")
