"""Helpers shared by the golden-vector tests (oracle on CPU, HIP path on GPU)."""
import ast
import os

import numpy as np
import torch

from oracle import planner_oracle as po
from oracle.make_golden import CASES, make_cfg, sample_idx, out_sample_idx

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    z = np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"))
    ckw = ast.literal_eval(str(z["meta.cfg"]))
    bkw = ast.literal_eval(str(z["meta.batch"]))
    assert (ckw, bkw) == CASES[name]
    cfg = make_cfg(**ckw)
    batch = po.make_batch(cfg, seed=1234, **bkw)
    return z, cfg, batch


def compare_outputs(z, outs, atol, rtol=0.0):
    """outs: dict of tensors named like planner_oracle.planner_step's outputs."""
    worst = {}
    pm = torch.from_numpy(z["out.pano_masks"])
    for k in ("txt_embeds", "pano_embeds", "gmap_embeds", "global_logits", "loss"):
        if f"osm.{k}" in z.files:      # large activation stored as strided samples + (sum, abs-max, L2) fingerprint
            got = outs[k].detach().double().cpu().reshape(-1)
            smp, fp = torch.from_numpy(z[f"osm.{k}"]).double(), z[f"ofp.{k}"]
            err = (got[torch.from_numpy(out_sample_idx(got.numel()))] - smp).abs()
            worst[k] = float(err.max())
            assert worst[k] <= atol + rtol * float(smp.abs().max()), f"{k}: sample err {worst[k]:.3e} > {atol}"
            assert abs(float(got.norm()) - fp[2]) <= (atol + rtol) * max(fp[2], 1.0), f"{k}: L2 fingerprint"
            continue
        ref = torch.from_numpy(z[f"out.{k}"])
        got = outs[k].detach().float().cpu().reshape(ref.shape)
        if k == "pano_embeds":      # padded query rows are don't-care
            ref, got = ref[pm], got[pm]
        fin = torch.isfinite(ref)
        assert torch.equal(torch.isfinite(got), fin), f"{k}: -inf pattern differs"
        err = (got[fin] - ref[fin]).abs()
        tol = atol + rtol * ref[fin].abs()
        worst[k] = float(err.max())
        assert bool((err <= tol).all()), f"{k}: max abs err {float(err.max()):.3e} > {atol}"
    return worst


def compare_grads(z, grads, atol, rel=None, rel_sample=None, abs_rel=0.0):
    """grads: name -> tensor.  Checks the 48 strided samples (abs) and the
    fingerprints (relative to the tensor's abs-max) of every parameter."""
    worst = 0.0
    names = [k[4:] for k in z.files if k.startswith("gfp.")]
    for name in names:
        if name.startswith("__input__") and name not in grads:
            continue
        assert name in grads, f"missing gradient {name}"
        g = grads[name].detach().double().cpu().reshape(-1)
        fp, smp = z[f"gfp.{name}"], torch.from_numpy(z[f"gsm.{name}"]).double()
        idx = torch.from_numpy(sample_idx(g.numel()))
        err = float((g[idx] - smp).abs().max())
        worst = max(worst, err)
        scale = max(fp[1], 1e-12)
        assert err <= atol + abs_rel * scale, f"grad {name}: sample err {err:.3e} > {atol} + {abs_rel}*{scale:.3e}"
        if rel_sample is not None:   # per-tensor relative bound on the samples
            assert err <= rel_sample * scale + 2e-6, f"grad {name}: sample err {err:.3e} vs abs-max {scale:.3e}"
        if rel is not None:
            assert abs(float(g.abs().max()) - fp[1]) <= rel * scale + atol, f"grad {name}: abs-max"
            assert abs(float(g.norm()) - fp[2]) <= rel * max(fp[2], 1e-12) + atol, f"grad {name}: L2"
    return worst


def load_rollout():
    import numpy as np
    from oracle.make_golden_rollout import make_case, CASE
    z = np.load(os.path.join(GOLDEN_DIR, "rollout_t3.npz"))
    assert ast.literal_eval(str(z["meta.case"])) == CASE
    cfg, P, ids, masks, steps = make_case()
    return z, cfg, P, ids, masks, steps


def compare_rollout(z, outs, atol):
    """outs: planner_oracle.rollout_step-shaped dict (loss, txt_embeds, steps[t] = {gmap_embeds, global_logits})."""
    assert abs(float(outs["loss"]) - float(z["out.loss"])) <= atol, "loss"
    assert float((outs["txt_embeds"].detach().float().cpu() - torch.from_numpy(z["out.txt_embeds"])).abs().max()) <= atol
    for t, o in enumerate(outs["steps"]):
        ref = torch.from_numpy(z[f"out.logits.{t}"])
        got = o["global_logits"].detach().float().cpu()
        fin = torch.isfinite(ref)
        assert torch.equal(torch.isfinite(got), fin), f"step {t}: -inf pattern"
        assert float((got[fin] - ref[fin]).abs().max()) <= atol, f"step {t}: logits"
        e = float((o["gmap_embeds"].detach().float().cpu() - torch.from_numpy(z[f"out.gmap_embeds.{t}"])).abs().max())
        assert e <= atol, f"step {t}: gmap_embeds {e}"
