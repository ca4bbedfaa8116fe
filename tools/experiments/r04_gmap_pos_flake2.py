"""Sporadic deviation of d(gmap_pos_embeddings.0.weight): which stream schedule shows it?"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import planner_oracle as po
from etpnav_amd.planner import GlocalTextPathNavCMT
from etpnav_amd.step import PlannerStep

cfg = po.PlannerConfig.r2r(vocab_size=4096)
P = po.init_params(cfg, seed=2)
batch = po.make_batch(cfg, B=32, L=80, V=36, G=16, seed=5, ragged=True)
m = GlocalTextPathNavCMT(cfg.to_dict(), dtype=torch.bfloat16, device="cuda")
m.load_state_dict({k: v for k, v in P.items()}, strict=True)
m.eval()
for overlap in (("aux",) if os.environ.get("FLAKE_AUX_ONLY") else (True, False, "s2", "aux")):
    step = PlannerStep(m, batch, overlap=overlap)
    vals = []
    names = ["global_encoder.gmap_pos_embeddings.0.weight", "global_encoder.gmap_pos_embeddings.0.bias",
             "global_encoder.gmap_pos_embeddings.1.weight", "global_encoder.gmap_step_embeddings.weight"]
    prm = dict(m.named_parameters())
    for run in range(60):
        step.run_eager(); torch.cuda.synchronize()
        vals.append({n: prm[n].grad.detach().clone() for n in names})
    # reference = elementwise median over runs
    for n in names:
        stack = torch.stack([v[n] for v in vals])
        med = stack.median(0).values
        dev = (stack - med).abs().flatten(1).max(1).values
        out = [(i, float(d)) for i, d in enumerate(dev) if d > 1e-5]
        print(f"overlap={overlap!s:5s} {n:50s} |med| {float(med.abs().max()):.3e}  outlier runs (>1e-5): {len(out)}/60  {out[:6]}")
    step.close()
