"""Generate tests/golden/pretrain_{sap,mlm}.npz from the REAL pre-training model (GlocalTextPathCMTPreTraining,
pretrain_cmt.py:50-283) through oracle/ref_pretrain_harness.py: per-sample SAP losses / logits and MLM token losses, the
mean loss, and fingerprints + samples of every parameter gradient (same format as oracle/make_golden.py).

    python oracle/make_golden_pretrain.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import planner_oracle as po            # noqa: E402
from oracle import ref_pretrain_harness as rp      # noqa: E402

CASE = dict(cfg=dict(vocab_size=2048, num_l_layers=2, num_pano_layers=1, num_x_layers=2, use_lang2visn_attn=True),
            batch=dict(B=3, L=15, T=3, V=8, n_cand=4, seed=5, ragged=True), param_seed=3)


def make_case():
    from etpnav_amd.synthetic import make_sap_batch
    cfg = po.PlannerConfig.r2r(**CASE["cfg"])
    P = po.init_params(cfg, seed=CASE["param_seed"])
    b = CASE["batch"]
    batch = make_sap_batch(cfg.vocab_size, cfg.image_feat_size, cfg.depth_feat_size, b["B"], b["L"], b["T"], b["V"],
                           n_cand=b["n_cand"], seed=b["seed"], ragged=b["ragged"])
    g = torch.Generator().manual_seed(99)
    lab = torch.full_like(batch["txt_ids"], -1)
    pick = (torch.rand(lab.shape, generator=g) < 0.25) & batch["txt_masks"]
    pick[:, 1] = True                                       # at least one masked token per episode
    lab[pick] = batch["txt_ids"][pick]
    ids = batch["txt_ids"].clone()
    ids[pick] = 103                                         # [MASK]
    batch["txt_ids"], batch["txt_labels"] = ids, lab
    return cfg, P, batch


def fingerprint(z, prefix, grads):
    for k, g in grads.items():
        g = g.detach().float()
        z[f"{prefix}/gsum/{k}"] = np.float64(g.double().sum().item())
        z[f"{prefix}/gabs/{k}"] = np.float32(g.abs().max().item())
        z[f"{prefix}/gl2/{k}"] = np.float64(g.double().pow(2).sum().sqrt().item())
        flat = g.reshape(-1)
        idx = torch.linspace(0, flat.numel() - 1, min(48, flat.numel())).long()
        z[f"{prefix}/gsmp/{k}"] = flat[idx].numpy()


def main():
    cfg, P, batch = make_case()
    model = rp.build_pretrain_model(cfg, P)
    rb = rp.to_ref_batch(batch)
    strip = lambda n: n[5:] if n.startswith("bert.") else n
    z = {}
    for task in ("sap", "mlm"):
        model.zero_grad()
        losses = model(rb, task, True)
        loss = losses.mean()
        loss.backward()
        grads = {strip(n): (p.grad if p.grad is not None else torch.zeros_like(p)) for n, p in model.named_parameters()}
        z[f"{task}/losses"] = losses.detach().numpy()
        z[f"{task}/loss"] = np.float32(loss.item())
        fingerprint(z, task, grads)
        print(task, "loss", float(loss), "n_losses", losses.numel())
    out = os.path.join(ROOT, "tests", "golden", "pretrain_tasks.npz")
    np.savez_compressed(out, **z)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
