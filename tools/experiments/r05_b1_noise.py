"""round 5: how much does the bf16 gradient error of a B = 1 step move with the inputs?  Five fresh single-episode batches (c1 shape),
per-tensor relative L2 error against the fp32 oracle: median / 90th percentile / max over the tensors with a non-zero gradient."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from oracle import planner_oracle as po
from etpnav_amd.planner import GlocalTextPathNavCMT
from etpnav_amd.step import PlannerStep
cfg = po.PlannerConfig.r2r()
P = po.init_params(cfg, seed=0)
m = GlocalTextPathNavCMT(cfg.to_dict(), dtype=torch.bfloat16, device="cuda"); m.load_state_dict(P, strict=True); m.eval()
print("library", os.environ.get("ETP_LIB", "default"))
B = int(os.environ.get('NOISE_B', '1'))
for seed in ((1234,) + tuple(range(1, 16)) if B == 1 else tuple(range(1, 9))):
    batch = po.make_batch(cfg, seed=seed, B=B, L=20, V=17, G=9, ragged=False)
    outs, ref = po.step_with_grads(P, cfg, batch)
    step = PlannerStep(m, batch, overlap=False); step.run_eager(); torch.cuda.synchronize()
    rel = []
    for k, p in m.named_parameters():
        r = ref[k].double().reshape(-1); nr = float(r.norm())
        if float(r.abs().max()) < 1e-6:
            continue
        rel.append(float((p.grad.detach().double().cpu().reshape(-1) - r).norm()) / nr)
    t = torch.tensor(rel)
    fin = torch.isfinite(outs["global_logits"])
    print(f"seed {seed}: loss {step.loss.item():.4f} (oracle {outs['loss'].item():.4f}), logits err {float((step.logits.float().cpu()[fin] - outs['global_logits'][fin]).abs().max()):.2e}; "
          f"relative L2 over {len(rel)} tensors: median {t.median():.4f}, p90 {t.kthvalue(int(0.9 * len(rel))).values:.4f}, max {t.max():.4f}")
    step.close()
