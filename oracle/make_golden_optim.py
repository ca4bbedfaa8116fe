"""Generate tests/golden/adamw.npz from the REAL optimizers (build container only; /root/reference needed):
the reference's pre-training AdamW (pretrain_src/pretrain_src/optim/adamw.py, imported by file path) with the
no-decay grouping of optim/misc.py:12-22 and clip_grad_norm_(5.0), and torch.optim.AdamW (the fine-tuning optimizer).

    python oracle/make_golden_optim.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/pretrain_src/pretrain_src/optim/adamw.py"


def trajectories():
    g = torch.Generator().manual_seed(7)
    shapes = {"a.weight": (24, 64), "a.bias": (64,), "b.LayerNorm.weight": (64,), "b.LayerNorm.bias": (64,),
              "c.layer_norm.weight": (64,), "emb.weight": (9, 64)}
    p0 = {k: torch.randn(s, generator=g) * 0.5 for k, s in shapes.items()}
    steps = 4
    grads = [{k: torch.randn(s, generator=g) * (3.0 if t == 1 else 0.3) for k, s in shapes.items()} for t in range(steps)]
    return shapes, p0, grads


def run(opt_ctor, groups_fn, clip):
    shapes, p0, grads = trajectories()
    params = {k: torch.nn.Parameter(v.clone()) for k, v in p0.items()}
    opt = opt_ctor(groups_fn(params))
    out = []
    for gset in grads:
        for k, p in params.items():
            p.grad = gset[k].clone()
        if clip > 0:
            torch.nn.utils.clip_grad_norm_(list(params.values()), clip)
        opt.step()
        out.append({k: p.detach().clone() for k, p in params.items()})
    return out


def main():
    spec = importlib.util.spec_from_file_location("ref_adamw", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]                  # optim/misc.py:13

    def hf_groups(params):
        return [{"params": [p for n, p in params.items() if not any(nd in n for nd in no_decay)], "weight_decay": 0.01},
                {"params": [p for n, p in params.items() if any(nd in n for nd in no_decay)], "weight_decay": 0.0}]

    hf = run(lambda gr: mod.AdamW(gr, lr=5e-3, betas=(0.9, 0.98)), hf_groups, clip=5.0)           # train_r2r defaults' shape
    hf_nobias = run(lambda gr: mod.AdamW(gr, lr=5e-3, betas=(0.9, 0.98), correct_bias=False), hf_groups, clip=0.0)
    th = run(lambda gr: torch.optim.AdamW(gr, lr=2e-3), lambda params: list(params.values()), clip=0.0)
    shapes, p0, grads = trajectories()
    z = {}
    for k in shapes:
        z[f"p0/{k}"] = p0[k].numpy()
        for t, gset in enumerate(grads):
            z[f"g{t}/{k}"] = gset[k].numpy()
        for name, traj in (("hf", hf), ("hf_nobias", hf_nobias), ("torch", th)):
            for t, ps in enumerate(traj):
                z[f"{name}{t}/{k}"] = ps[k].numpy()
    out = os.path.join(ROOT, "tests", "golden", "adamw.npz")
    np.savez_compressed(out, **z)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    sys.exit(main())
