"""The bf16 gradient bounds must be ABLE TO FAIL.

The previous bound (8e-2 + 10 % of abs-max) was an absolute floor larger than most gradient tensors of this model: an
all-zero gradient passed.  These CPU tests run the comparators of tests/golden_util.py (the ones every bf16 `-m gpu` test
uses) on the oracle's own gradients of the committed c1 fixture, perturbed the way bf16 arithmetic perturbs them (must
pass), and on mutated gradient sets (must raise):
  * a small-magnitude tensor zeroed (lang_encoder.layer.3.attention.output.LayerNorm.bias: abs-max ~1e-2),
  * a tensor scaled by 0.85, a tensor replaced by noise of the right magnitude, a sign flip.
"""
import numpy as np
import pytest
import torch

from oracle import planner_oracle as po
from tests.golden_util import load_case, compare_grads_bf16, compare_full_bf16

SMALL = "lang_encoder.layer.3.attention.output.LayerNorm.bias"


@pytest.fixture(scope="module")
def case():
    z, cfg, batch = load_case("c1_single_episode")
    P = po.init_params(cfg, seed=0)
    _, grads = po.step_with_grads(P, cfg, batch)
    return z, grads


def bf16_like(grads, rel=4e-3, seed=0):
    """the perturbation class a bf16 step shows: ~0.4 % (2^-8) relative noise per element"""
    g = torch.Generator().manual_seed(seed)
    return {k: v * (1.0 + rel * torch.randn(v.shape, generator=g)) for k, v in grads.items()}


def test_small_tensor_is_below_the_old_absolute_floor(case):
    z, grads = case
    assert float(z[f"gfp.{SMALL}"][1]) < 8e-2          # the old bound 8e-2 + 0.1 * abs-max could not see this tensor at all
    n_small = sum(1 for k in z.files if k.startswith("gfp.") and float(z[k][1]) < 8e-2)
    n_all = sum(1 for k in z.files if k.startswith("gfp."))
    assert n_small > n_all // 2, (n_small, n_all)


def test_bf16_bounds_accept_bf16_sized_noise(case):
    z, grads = case
    noisy = bf16_like(grads)
    compare_grads_bf16(z, noisy)
    compare_full_bf16(noisy, grads)


@pytest.mark.parametrize("mutation", ["zero_small", "scale", "noise", "sign"])
def test_bf16_bounds_reject_mutated_gradients(case, mutation):
    z, grads = case
    bad = bf16_like(grads, seed=1)
    if mutation == "zero_small":
        bad[SMALL] = torch.zeros_like(bad[SMALL])
    elif mutation == "scale":
        k = "lang_encoder.layer.5.output.dense.weight"
        bad[k] = bad[k] * 0.85
    elif mutation == "noise":
        k = "global_encoder.encoder.x_layers.2.visn_inter.dense.weight"
        g = torch.Generator().manual_seed(3)
        bad[k] = torch.randn(bad[k].shape, generator=g) * bad[k].std()
    else:
        k = "img_embeddings.pano_encoder.layers.1.linear1.bias"
        bad[k] = -bad[k]
    with pytest.raises(AssertionError):
        compare_grads_bf16(z, bad)
    with pytest.raises(AssertionError):
        compare_full_bf16(bad, grads)


def test_old_bound_would_have_accepted_the_zeroed_tensor(case):
    z, grads = case
    fp = z[f"gfp.{SMALL}"]
    smp = np.abs(z[f"gsm.{SMALL}"]).max()
    assert smp <= 8e-2 + 0.1 * float(fp[1])             # documents the hole that was closed
