"""Generate tests/golden/*.npz from the REAL reference (run in the build container only).

    python -m oracle.make_golden            # writes tests/golden/<case>.npz

The reference ships no golden vectors, so these fixtures are
outputs of the reference's own ``GlocalTextPathNavCMT`` (imported from
/root/reference through ``oracle/ref_harness.py``), fp32 CPU, eval mode, on
seeded synthetic inputs (``planner_oracle.make_batch`` seed 1234) with seeded
weights (``planner_oracle.init_params`` seed 0).  Full gradients are 563 MB per
case, so each parameter gradient is stored as a fingerprint (sum, abs-max, L2)
plus 48 values at fixed strided indices; small outputs are stored whole.
TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import planner_oracle as po  # noqa: E402

N_SAMPLE = 48

# name -> (config factory kwargs, batch kwargs)
CASES = {
    # BASELINE.json configs[0]: single episode, 12 pano + 5 cand views, 20 tokens
    "c1_single_episode": (dict(kind="r2r"), dict(B=1, L=20, V=17, G=9, ragged=False)),
    # ragged lengths, ignore_index label, tiny
    "ragged_small": (dict(kind="r2r"), dict(B=3, L=12, V=14, G=7, ragged=True)),
    # per-sample shape of configs[1] (36 views x 768-d, 80 tokens, 16 nodes)
    "c2_shape_b2": (dict(kind="r2r", image_feat_size=768), dict(B=2, L=80, V=36, G=16, ragged=False)),
    # configs[4]: 64 graph nodes (G > V)
    "c5_g64_b2": (dict(kind="r2r"), dict(B=2, L=24, V=36, G=64, ragged=True)),
    # configs[3]: XLM-R vocabulary / eps 1e-5 (short text to keep the fixture cheap)
    "c4_rxr_b1": (dict(kind="rxr"), dict(B=1, L=48, V=14, G=6, ragged=False)),
    # configs[3] at its BASELINE per-sample shape: XLM-R, 512-token instructions (ragged: lengths 256..512), 36 views, 16 nodes
    "c4_rxr_l512_b2": (dict(kind="rxr"), dict(B=2, L=512, V=36, G=16, ragged=True)),
    # configs[4] at its BASELINE per-sample shape: 80 tokens, 36 views, 64 graph nodes
    "c5_g64_l80_b2": (dict(kind="r2r"), dict(B=2, L=80, V=36, G=64, ragged=False)),
    # frozen / ablated model variants the boundary reads (vlnbert_init.py:42-54; vilmodel_cmt.py:422-433,675-682): the frozen
    # parameters get NO gradient in the reference (stored here as zeros), the detached text output stops the backward
    "fix_lang_small": (dict(kind="r2r", fix_lang_embedding=True), dict(B=3, L=12, V=14, G=7, ragged=True)),
    "fix_pano_small": (dict(kind="r2r", fix_pano_embedding=True), dict(B=3, L=12, V=14, G=7, ragged=True)),
    "no_sprels_small": (dict(kind="r2r", graph_sprels=False), dict(B=3, L=12, V=14, G=7, ragged=True)),
    "no_depth_small": (dict(kind="r2r", use_depth_embedding=False), dict(B=3, L=12, V=14, G=7, ragged=True)),
}

# Big activations of the large cases are stored as (fingerprint, strided samples) like the gradients: the full
# [2,512,768] text embeddings would be 3 MB per fixture.
SAMPLED_OUTPUTS = {"c4_rxr_l512_b2": ("txt_embeds",)}
N_OUT_SAMPLE = 4096


def make_cfg(kind="r2r", **kw):
    return po.PlannerConfig.rxr(**kw) if kind == "rxr" else po.PlannerConfig.r2r(**kw)


def sample_idx(n: int) -> np.ndarray:
    if n <= N_SAMPLE:
        return np.arange(n)
    return np.unique(np.linspace(0, n - 1, N_SAMPLE).astype(np.int64))


def fingerprint(t: torch.Tensor):
    f = t.detach().double().reshape(-1)
    fp = np.array([float(f.sum()), float(f.abs().max()), float(f.norm())], dtype=np.float64)
    idx = sample_idx(f.numel())
    return fp, f[torch.from_numpy(idx)].float().numpy()


def out_sample_idx(n: int) -> np.ndarray:
    return np.unique(np.linspace(0, n - 1, N_OUT_SAMPLE).astype(np.int64))


def pack(outs, grads, sampled=()):
    d = {}
    for k in ("txt_embeds", "pano_embeds", "gmap_embeds", "global_logits", "loss", "gmap_img_fts"):
        if k in sampled:
            f = outs[k].float().reshape(-1)
            d[f"osm.{k}"] = f[torch.from_numpy(out_sample_idx(f.numel()))].numpy()
            d[f"ofp.{k}"] = np.array([float(f.double().sum()), float(f.abs().max()), float(f.double().norm())])
            continue
        d[f"out.{k}"] = outs[k].float().numpy()
    d["out.pano_masks"] = outs["pano_masks"].numpy()
    for k, g in grads.items():
        fp, smp = fingerprint(g)
        d[f"gfp.{k}"] = fp
        d[f"gsm.{k}"] = smp
    return d


def main():
    from oracle import ref_harness as rh
    assert rh.reference_available(), "needs /root/reference"
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    only = sys.argv[1:]
    for name, (ckw, bkw) in CASES.items():
        if only and name not in only:
            continue
        cfg = make_cfg(**ckw)
        P = po.init_params(cfg, seed=0)
        model = rh.build_reference_model(cfg, P)
        batch = po.make_batch(cfg, seed=1234, **bkw)
        outs, grads = rh.reference_step(model, batch)
        d = pack(outs, grads, SAMPLED_OUTPUTS.get(name, ()))
        d["meta.cfg"] = np.array(repr(ckw))
        d["meta.batch"] = np.array(repr(bkw))
        path = os.path.join(ROOT, "tests", "golden", f"{name}.npz")
        np.savez_compressed(path, **d)
        print(name, "loss", float(outs["loss"]), os.path.getsize(path) // 1024, "KiB")
        del model


if __name__ == "__main__":
    main()
