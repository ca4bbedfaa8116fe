"""Timing of the two pre-training steps at the pre-training shape: B=32 episodes, T=5 panoramas of 36
views, 80-token instructions, bert-base vocabulary, 15 % masked tokens; bf16, training mode.   python tools/pretrain_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from etpnav_amd.planner import GlocalTextPathNavCMT, default_config
from etpnav_amd.pretrain import MlmStep
from etpnav_amd.step import PlannerStep
from etpnav_amd.synthetic import make_sap_batch

cfg = default_config("r2r", image_feat_size=768, use_lang2visn_attn=True)
model = GlocalTextPathNavCMT(cfg, dtype=torch.bfloat16, device="cuda:0"); model.init_weights(seed=0)
batch = make_sap_batch(cfg.vocab_size, cfg.image_feat_size, cfg.depth_feat_size, 32, 80, 5, 36, seed=1)
g = torch.Generator().manual_seed(2)
lab = torch.full_like(batch["txt_ids"], -1)
pick = torch.rand(lab.shape, generator=g) < 0.15
lab[pick] = batch["txt_ids"][pick]
batch["txt_labels"] = lab


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


mlm = MlmStep(model, batch, dropout="config")
print(f"MLM step (single stream): {timeit(mlm.run_eager):.2f} ms, {int(pick.sum())} masked tokens, loss {mlm.loss.item():.3f}", flush=True)
sap = PlannerStep(model, batch, dropout="config")
print(f"SAP step (three streams, same model): {timeit(sap.run_eager):.2f} ms, loss {sap.loss.item():.3f}", flush=True)
