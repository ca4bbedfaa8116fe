"""Oracle restatement of the pre-training SAP and MLM tasks against the REAL pre-training model
(tests/golden/pretrain_tasks.npz from GlocalTextPathCMTPreTraining; generator oracle/make_golden_pretrain.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import planner_oracle as po
from oracle.make_golden_pretrain import make_case

GOLD = os.path.join(os.path.dirname(__file__), "golden", "pretrain_tasks.npz")


def check_grads(z, task, grads, atol=2e-5, rel=2e-4):
    names = [k.split("/", 2)[2] for k in z.files if k.startswith(f"{task}/gsum/")]
    assert names
    for k in names:
        g = grads[k].detach().float()
        absmax = float(z[f"{task}/gabs/{k}"])
        flat = g.reshape(-1)
        idx = torch.linspace(0, flat.numel() - 1, min(48, flat.numel())).long()
        tol = atol + rel * absmax
        assert np.abs(flat[idx].numpy() - z[f"{task}/gsmp/{k}"]).max() < tol, (task, k)
        assert abs(float(g.abs().max()) - absmax) < tol, (task, k)
        assert abs(float(g.double().pow(2).sum().sqrt()) - float(z[f"{task}/gl2/{k}"])) < tol * max(1.0, g.numel() ** 0.5), (task, k)


def test_oracle_sap_task_matches_real_pretraining_model():
    z = np.load(GOLD)
    cfg, P, batch = make_case()
    outs, grads = po.sap_step_with_grads(P, cfg, batch)
    assert abs(outs["loss"].item() - float(z["sap/loss"])) < 2e-5
    lse = torch.logsumexp(outs["global_logits"], -1)
    per = lse - outs["global_logits"].gather(1, batch["labels"][:, None]).squeeze(1)
    assert np.abs(per.numpy() - z["sap/losses"]).max() < 2e-5
    check_grads(z, "sap", grads)


def test_oracle_mlm_task_matches_real_pretraining_model():
    z = np.load(GOLD)
    cfg, P, batch = make_case()
    outs, grads = po.mlm_step_with_grads(P, cfg, batch)
    assert abs(outs["loss"].item() - float(z["mlm/loss"])) < 5e-5
    sel = batch["txt_labels"] != -1
    lab = batch["txt_labels"][sel]
    per = torch.logsumexp(outs["mlm_logits"], -1) - outs["mlm_logits"].gather(1, lab[:, None]).squeeze(1)
    assert np.abs(per.numpy() - z["mlm/losses"]).max() < 5e-5
    check_grads(z, "mlm", grads)
