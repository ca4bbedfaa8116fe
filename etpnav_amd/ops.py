"""Host helpers mirrored from vlnce_baselines/common/ops.py (mask / pad utilities the trainer calls around the planner)."""
from __future__ import annotations

import torch


def extend_neg_masks(masks, dtype=None):
    """ops.py:25-34: (N, L) -> (N, 1, 1, L) additive -10000 mask.  (The HIP attention takes the bool mask directly;
    kept for callers that build the additive form.)"""
    if dtype is None:
        dtype = torch.float
    return (1.0 - masks.unsqueeze(1).unsqueeze(2).to(dtype=dtype)) * -10000.0


def gen_seq_masks(seq_lens, max_len=None):
    """ops.py:36-44."""
    if max_len is None:
        max_len = int(max(seq_lens))
    return torch.arange(max_len, device=seq_lens.device).unsqueeze(0) < seq_lens.unsqueeze(1)


def pad_tensors_wgrad(tensors, lens=None):
    """ops.py:46-68 (B x [T, ...] -> [B, max T, ...], differentiable) in one pad_sequence call instead of B torch.cat's."""
    return torch.nn.utils.rnn.pad_sequence(list(tensors), batch_first=True)
