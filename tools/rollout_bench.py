"""Rollout episode through the module API: T navigation steps on one instruction batch, forward + backward.

Three ways of issuing the same work (identical outputs and gradients: tests/test_baseline_shapes_gpu.py):
    per_step        forward_navigation once per step, text K/V re-projected every call  (what the trainer does:
                    ss_trainer_ETP.py:819-822,878 then one backward over the summed losses, :1055)
    per_step_kv     the same with cache_text_kv (one projection of the text K/V serves all T steps)
    batched_reproject  forward_navigation_steps with batch_steps_kv = False: the T steps as one (T * B)-episode call that
                    projects the T-fold stacked text rows to keys/values in every x-layer (round 3's form)
    batched_copy    forward_navigation_steps on the K/V cache: one projection, replicated for the stacked steps by a copy (round 4's form)
    batched         the same with per-episode indirection (round 6): stacked episode e reads instruction e % B inside the
                    cross-attention kernels, no replicated cache (bf16, axes <= 128)

    python tools/rollout_bench.py [--B 8] [--L 80] [--T 5,10,15] [--G 16] [--dtype bf16] [--iters 20]
Prints one JSON object (ms per episode fwd+bwd, episodes/s) -> profiles/r03_rollout_bench.json.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from etpnav_amd.planner import GlocalTextPathNavCMT, default_config  # noqa: E402
from etpnav_amd.synthetic import make_batch  # noqa: E402


def episode(model, ids, masks, steps, how):
    model.zero_grad()
    txt = model.forward_txt(ids, masks)
    if how.startswith("batched"):
        model.batch_steps_kv = how != "batched_reproject"    # "batched_reproject": the T-fold stacked K/V projection (round 3)
        model.kv_indirection = how == "batched"
        outs = model.forward_navigation_steps(txt, masks, steps)
    else:
        outs = [model.forward_navigation(txt, masks, None, st["gmap_step_ids"], st["gmap_img_fts"], st["gmap_pos_fts"],
                                         st["gmap_masks"], st["gmap_visited_masks"], st["gmap_pair_dists"]) for st in steps]
    loss = 0.0
    for o, st in zip(outs, steps):
        loss = loss + F.cross_entropy(o["global_logits"], st["labels"], reduction="sum", ignore_index=-100) / ids.shape[0]
    loss.backward()
    return loss


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--L", type=int, default=80)
    ap.add_argument("--G", type=int, default=16, help="largest node count; step t of T has 4 + (G - 4) * t / (T - 1) nodes")
    ap.add_argument("--T", default="5,10,15")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--mode", default="train", choices=["train", "eval"])
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    cfg = default_config("r2r")
    model = GlocalTextPathNavCMT(cfg, dtype=dt, device="cuda")
    model.train() if a.mode == "train" else model.eval()
    rows = []
    for T in [int(x) for x in a.T.split(",")]:
        base = make_batch(cfg.vocab_size, cfg.image_feat_size, cfg.depth_feat_size, a.B, a.L, 12, a.G, seed=5)
        ids, masks = base["txt_ids"].cuda(), base["txt_masks"].cuda()
        steps = []
        for t in range(T):
            G = 4 + (a.G - 4) * t // max(T - 1, 1)
            b = make_batch(cfg.vocab_size, cfg.image_feat_size, cfg.depth_feat_size, a.B, a.L, 12, G, seed=100 + t)
            gen = torch.Generator().manual_seed(300 + t)
            b["gmap_img_fts"] = torch.randn(a.B, G, cfg.hidden_size, generator=gen) * 0.5
            steps.append({k: v.cuda() for k, v in b.items() if k.startswith("gmap_") or k == "labels"})
        row = {"T": T, "B": a.B, "L": a.L, "G_last": a.G}
        for how in ("per_step", "per_step_kv", "batched_reproject", "batched_copy", "batched"):
            model.cache_text_kv = how == "per_step_kv"
            for _ in range(3):
                episode(model, ids, masks, steps, how)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.iters):
                episode(model, ids, masks, steps, how)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / a.iters * 1e3
            row[how + "_ms"] = round(ms, 3)
        row["batched_speedup_vs_per_step"] = round(row["per_step_ms"] / row["batched_ms"], 2)
        row["batched_speedup_vs_per_step_kv"] = round(row["per_step_kv_ms"] / row["batched_ms"], 2)
        row["batched_speedup_vs_reproject"] = round(row["batched_reproject_ms"] / row["batched_ms"], 2)
        rows.append(row)
    print(json.dumps({"what": "rollout episode fwd+bwd through the module API, wall clock incl. host (one MI355X)",
                      "dtype": a.dtype, "mode": a.mode, "iters": a.iters, "rows": rows}))


if __name__ == "__main__":
    main()
