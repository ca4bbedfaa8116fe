"""tests/golden/rollout_t3.npz: a T-step rollout through the REAL reference model (run in the build container only).

    python -m oracle.make_golden_rollout

ss_trainer_ETP.py:801-805 computes the instruction embeddings once per episode batch, :878-892 calls forward_navigation
and the cross-entropy at every step on that SAME tensor, and :1055 sums the step losses before the single backward.  This
fixture pins exactly that composition (text K/V of the cross-attention re-used across steps, gradients of all T steps
flowing back through one text encoder) for the oracle (tests/test_oracle_golden.py) and for the HIP path with and without
its text-K/V cache (tests/test_baseline_shapes_gpu.py).  fp32 CPU, eval mode.  TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import planner_oracle as po  # noqa: E402
from oracle.make_golden import fingerprint  # noqa: E402

CASE = dict(B=3, L=22, G=9, T=3, seed=40, param_seed=9)


def make_case():
    cfg = po.PlannerConfig.r2r()
    P = po.init_params(cfg, seed=CASE["param_seed"])
    ids, masks, steps = po.make_rollout(cfg, CASE["B"], CASE["L"], CASE["G"], CASE["T"], CASE["seed"])
    return cfg, P, ids, masks, steps


def main():
    import torch.nn.functional as F
    from oracle import ref_harness as rh
    assert rh.reference_available(), "needs /root/reference"
    cfg, P, ids, masks, steps = make_case()
    model = rh.build_reference_model(cfg, P)
    for p in model.parameters():
        p.grad = None
    txt = model.forward_txt(ids, masks)
    loss, d = 0.0, {}
    for t, st in enumerate(steps):
        o = model.forward_navigation(txt, masks, None, st["gmap_step_ids"], st["gmap_img_fts"], st["gmap_pos_fts"],
                                     st["gmap_masks"], st["gmap_visited_masks"], st["gmap_pair_dists"])
        loss = loss + F.cross_entropy(o["global_logits"], st["labels"], reduction="sum", ignore_index=-100) / ids.shape[0]
        d[f"out.logits.{t}"] = o["global_logits"].detach().numpy()
        d[f"out.gmap_embeds.{t}"] = o["gmap_embeds"].detach().numpy()
    loss.backward()
    d["out.loss"] = loss.detach().numpy()
    d["out.txt_embeds"] = txt.detach().numpy()
    for k, p in model.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        d[f"gfp.{k}"], d[f"gsm.{k}"] = fingerprint(g)
    d["meta.case"] = np.array(repr(CASE))
    path = os.path.join(ROOT, "tests", "golden", "rollout_t3.npz")
    np.savez_compressed(path, **d)
    print("rollout loss", float(loss), os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
