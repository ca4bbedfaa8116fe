"""CPU restatement of the two AdamW variants the reference trains the planner with.  TEST INFRASTRUCTURE ONLY (imported
by tests/ and bench.py's cpu legs; the product path is etpnav_amd/csrc/optim.hip and has no CPU fallback).

  torch style  fine-tuning, ``torch.optim.AdamW`` (ss_trainer_ETP.py:213; torch/optim/adamw.py single-tensor path):
                 p *= 1 - lr*wd;  m,v EMAs;  p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
  hf style     pre-training, pretrain_src/pretrain_src/optim/adamw.py:88-110:
                 m,v EMAs;  p -= lr*sqrt(bc2)/bc1 * m / (sqrt(v) + eps);  p -= lr*wd*p      (bc = 1 if not correct_bias)
plus ``clip_grad_norm_`` (coef = max_norm / (norm + 1e-6), clamped to 1) and a GradScaler-style unscale / skip.

Pinned: tests/golden/adamw.npz holds trajectories produced by the REAL classes (the reference's AdamW imported from
/root/reference, and torch.optim.AdamW) — generator oracle/make_golden_optim.py; tests/test_optim_cpu.py checks this
restatement against them.
"""
from __future__ import annotations

import math
from typing import Optional

import torch


def adamw_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float, beta1: float,
               beta2: float, eps: float, weight_decay, hf_style: bool, correct_bias: bool = True, grad_scale: float = 1.0,
               max_norm: float = 0.0, skip: bool = False):
    """In-place on p, m, v (fp32).  weight_decay: float or a per-element tensor (0 where the reference's no-decay
    grouping applies).  Returns the clip coefficient used."""
    coef = 1.0
    if max_norm > 0.0:
        norm = float(torch.linalg.vector_norm(g.double()).item()) * abs(grad_scale)
        coef = min(1.0, max_norm / (norm + 1e-6))
    if skip:
        return coef
    gr = g * (grad_scale * coef)
    m.mul_(beta1).add_(gr, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(gr, gr, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** step if correct_bias else 1.0
    bc2 = 1.0 - beta2 ** step if correct_bias else 1.0
    if hf_style:
        p.addcdiv_(m, v.sqrt().add_(eps), value=-lr * math.sqrt(bc2) / bc1)
        p.sub_(p * (lr * weight_decay))
    else:
        p.mul_(1.0 - lr * weight_decay)
        p.addcdiv_(m, (v.sqrt() / math.sqrt(bc2)).add_(eps), value=-lr / bc1)
    return coef
