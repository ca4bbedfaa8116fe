"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol the header declares,
the parameter tree equals the reference state-dict contract, and the product path refuses to run without a GPU
(no silent fallback).  No kernel is launched here."""
import ctypes

import pytest
import torch

from etpnav_amd import _lib
from etpnav_amd.planner import GlocalTextPathNavCMT, default_config
from oracle import planner_oracle as po


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    names = _lib.declared_symbols()
    assert len(names) >= 50
    for n in names:
        assert hasattr(L, n), n
    assert L.etp_version().decode().startswith("etpnav_hip")


@pytest.mark.parametrize("kind", ["r2r", "rxr"])
def test_state_dict_contract_matches_reference_names(kind):
    ocfg = po.PlannerConfig.rxr() if kind == "rxr" else po.PlannerConfig.r2r()
    if kind == "rxr":
        ocfg.vocab_size = 1024      # keep the CPU test light; shapes are checked symbolically
    shapes = po.param_shapes(ocfg)
    m = GlocalTextPathNavCMT(ocfg.to_dict(), dtype=torch.float32, device="cpu")
    sd = m.state_dict()
    assert set(sd.keys()) == set(shapes.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    # all parameters are views into one flat arena, 256-byte aligned, non-overlapping
    spans = sorted((p.data_ptr(), p.numel() * 4) for p in m.parameters())
    base = m.flat_params.data_ptr()
    for (a, n), (b, _) in zip(spans, spans[1:]):
        assert a + n <= b
    assert all((a - base) % 256 == 0 for a, _ in spans)
    assert spans[-1][0] + spans[-1][1] <= base + m.flat_params.numel() * 4


def test_fused_qkv_weights_are_adjacent_in_the_arena():
    m = GlocalTextPathNavCMT(po.PlannerConfig.r2r(vocab_size=512).to_dict(), dtype=torch.float32, device="cpu")
    sd = dict(m.named_parameters())
    p = "lang_encoder.layer.0.attention.self."
    q, k, v = sd[p + "query.weight"], sd[p + "key.weight"], sd[p + "value.weight"]
    assert k.data_ptr() == q.data_ptr() + q.numel() * 4 and v.data_ptr() == k.data_ptr() + k.numel() * 4
    x = "global_encoder.encoder.x_layers.1.visual_attention.att."
    k, v = sd[x + "key.weight"], sd[x + "value.weight"]
    assert v.data_ptr() == k.data_ptr() + k.numel() * 4


def test_load_state_dict_roundtrip_and_grad_views():
    cfg = po.PlannerConfig.r2r(vocab_size=512)
    P = po.init_params(cfg, seed=5)
    m = GlocalTextPathNavCMT(cfg.to_dict(), dtype=torch.float32, device="cpu")
    m.load_state_dict(P, strict=True)
    for k, v in m.state_dict().items():
        assert torch.equal(v, P[k]), k
    # grads are views into the flat gradient arena and survive zero_grad(set_to_none=True)-style resets
    n, p = next(iter(m.named_parameters()))
    assert p.grad is not None and p.grad.data_ptr() >= m.flat_grads.data_ptr()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-5)
    opt.zero_grad()            # sets .grad = None
    m._attach_grads()
    assert all(q.grad is not None for q in m.parameters())
    v0 = m.flat_params._version
    with torch.no_grad():
        p.add_(1.0)
    assert m.flat_params._version != v0   # in-place updates of a view bump the arena version (bf16 shadow refresh)


def test_bad_config_raises_like_reference():
    with pytest.raises(ValueError):
        GlocalTextPathNavCMT(default_config(hidden_size=768, num_attention_heads=7), device="cpu")
    with pytest.raises(_lib.EtpError):
        GlocalTextPathNavCMT(default_config(hidden_size=320, num_attention_heads=5), device="cpu")


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only check")
def test_compute_fails_loudly_without_gpu():
    m = GlocalTextPathNavCMT(po.PlannerConfig.r2r(vocab_size=512).to_dict(), dtype=torch.float32, device="cpu")
    with pytest.raises(_lib.EtpError):
        m.forward_txt(torch.zeros(1, 4, dtype=torch.long), torch.ones(1, 4, dtype=torch.bool))


def test_argument_validation_without_launching():
    L = _lib.lib()
    d = _lib.GemmDesc()
    assert L.etp_gemm(ctypes.byref(d), None) == -1
    assert b"null" in L.etp_last_error()
    assert L.etp_ln_fwd(0, None, None, None, None, None, 4, 768, 1e-12, None) == -1
    cfg = _lib.Config()
    assert not L.etp_planner_create(ctypes.byref(cfg))


def test_policy_api_and_checkpoint_remap_cpu(tmp_path):
    """get_vlnbert_models / ETP keep the reference surface (vlnbert_init.py:13-66, Policy_ViewSelection_ETP.py:157-170)."""
    from types import SimpleNamespace
    from etpnav_amd.vlnbert_init import get_vlnbert_models, remap_checkpoint_keys
    from etpnav_amd.ops import gen_seq_masks, extend_neg_masks, pad_tensors_wgrad
    assert remap_checkpoint_keys({"module.bert.embeddings.LayerNorm.weight": 1, "global_sap_head.net.0.bias": 2}) == {
        "embeddings.LayerNorm.weight": 1, "global_sap_head.net.0.bias": 2}
    m = gen_seq_masks(torch.tensor([1, 3]))
    assert m.tolist() == [[True, False, False], [True, True, True]]
    assert extend_neg_masks(m).shape == (2, 1, 1, 3) and float(extend_neg_masks(m)[0, 0, 0, 1]) == -10000.0
    assert pad_tensors_wgrad([torch.ones(2, 4), torch.ones(3, 4)]).shape == (2, 3, 4)
