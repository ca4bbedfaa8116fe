"""Rollout microbenchmark for the text K/V cache: one forward_txt + T navigation steps on the same
txt_embeds + one backward, through the drop-in Python API, with and without `model.cache_text_kv`.
    python tools/rollout_probe.py [T]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from etpnav_amd.planner import GlocalTextPathNavCMT, default_config
from etpnav_amd.synthetic import make_batch

T = int(sys.argv[1]) if len(sys.argv) > 1 else 15
cfg = default_config("r2r", image_feat_size=768)
model = GlocalTextPathNavCMT(cfg, dtype=torch.bfloat16, device="cuda:0"); model.init_weights(seed=0); model.train()
B, L, V, G = 32, 80, 36, 16
b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in make_batch(cfg.vocab_size, cfg.image_feat_size, cfg.depth_feat_size, B, L, V, G).items()}
img = torch.randn(B, G, cfg.hidden_size, device="cuda")


def rollout():
    model.zero_grad()
    txt = model.forward_txt(b["txt_ids"], b["txt_masks"])
    loss = 0.0
    for _ in range(T):
        o = model.forward_navigation(txt, b["txt_masks"], None, b["gmap_step_ids"], img, b["gmap_pos_fts"], b["gmap_masks"],
                                     b["gmap_visited_masks"], b["gmap_pair_dists"])
        loss = loss + F.cross_entropy(o["global_logits"], b["labels"], reduction="sum") / B
    loss.backward()
    return loss


for cached in (False, True):
    model.cache_text_kv = cached
    for _ in range(2):
        rollout()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        l = rollout()
    torch.cuda.synchronize()
    print(f"T={T} cache_text_kv={cached}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per rollout (fwd+bwd), loss {l.item():.4f}", flush=True)
