"""Helpers shared by the golden-vector tests (oracle on CPU, HIP path on GPU)."""
import ast
import os

import numpy as np
import torch

from oracle import planner_oracle as po
from oracle.make_golden import CASES, make_cfg, sample_idx

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
