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


def test_dropout_mask_generator_is_pinned_and_calibrated():
    """The oracle's numpy restatement of the counter-based mask generator equals the library's host-side view of it
    bit for bit (so GPU train-mode parity tests apply identical masks), the keep rate is 1-p, the scale is 1/(1-p),
    sites and seeds give independent streams, and bad rates are refused."""
    import ctypes
    import numpy as np
    L = _lib.lib()
    n = 200000
    seen = []
    for (p, seed, mode, layer, slot) in [(0.1, 12345, 1, 3, 2), (0.4, (7 << 32) | 9, 2, 0, 8), (0.1, 0, 3, 1, 5),
                                         (0.1, 12346, 1, 3, 2), (0.1, 12345, 1, 4, 2)]:
        out = np.empty(n, np.float32)
        assert L.etp_dropout_multipliers(p, seed, mode, layer, slot, n, out.ctypes.data) == 0
        ref = po.DropSpec(seed=seed).mult(p, mode, layer, slot, (n,)).numpy()
        assert np.array_equal(out, ref)
        kept = out != 0
        assert abs(kept.mean() - (1 - p)) < 4e-3
        assert np.allclose(out[kept], np.float32(1) / (np.float32(1) - np.float32(p)))
        seen.append(kept)
    for i in range(len(seen)):
        for j in range(i + 1, len(seen)):
            if i == 1 or j == 1:
                continue        # different p
            agree = (seen[i] == seen[j]).mean()
            assert abs(agree - 0.82) < 0.01, (i, j, agree)      # independent Bernoulli(0.9): 0.81 + 0.01
    assert L.etp_dropout_multipliers(1.0, 0, 1, 0, 0, 1, np.empty(1, np.float32).ctypes.data) != 0
    m = GlocalTextPathNavCMT(po.PlannerConfig.r2r(vocab_size=512).to_dict(), dtype=torch.float32, device="cpu")
    assert L.etp_planner_set_dropout(m._engine.handle, 0.1, 0.1, 0.1, 0.4, 99) == 0
    assert L.etp_planner_set_dropout(m._engine.handle, -0.1, 0.1, 0.1, 0.4, 99) != 0
    assert L.etp_planner_set_dropout(m._engine.handle, 0.1, 1.0, 0.1, 0.4, 99) != 0
    # nn.Module semantics: train() draws a fresh stream per call, eval() turns dropout off
    m.train(); m.seed_dropout(3)
    a, b = m._dropout(), m._dropout()
    assert a[:4] == (0.1, 0.1, 0.1, 0.0) and a[4] == (3 << 32) | 1 and b[4] == (3 << 32) | 2
    m.eval()
    assert m._dropout() is None


def test_dropout_generator_statistics_match_nn_dropout():
    """the train-mode parity tests compare the HIP masks with the oracle's restatement of the SAME hash, which
    says nothing about the hash being a fair Bernoulli(1-p) source.  Here the library's masks (host view, etp_dropout_multipliers)
    are held against torch.nn.Dropout(p) itself: keep rate within 4 binomial sigmas of 1-p (as nn.Dropout's own draw is), the
    multiplier's mean within 4 sigmas of 1 (unbiasedness, what makes dropout a no-op in expectation), lag-1 .. lag-64 serial
    correlation of the keep bits below 4 sigmas (consecutive elements of a row are independent, as with Philox), and per-column
    keep rates of a [rows, 768] activation flat (no column is favoured: element index = row * H + col)."""
    import numpy as np
    L = _lib.lib()
    n = 768 * 512
    for p in (0.1, 0.4):
        out = np.empty(n, np.float32)
        assert L.etp_dropout_multipliers(p, (5 << 32) | 17, 1, 2, 3, n, out.ctypes.data) == 0
        torch.manual_seed(0)
        ref = torch.nn.functional.dropout(torch.ones(n), p, training=True).numpy()
        sig = np.sqrt(p * (1 - p) / n)
        for name, x in (("etp", out), ("nn.Dropout", ref)):
            kept = (x != 0).astype(np.float64)
            assert abs(kept.mean() - (1 - p)) < 4 * sig, (name, p, kept.mean())
            assert abs(x.mean() - 1.0) < 4 * sig / (1 - p), (name, p, x.mean())
        kept = (out != 0).astype(np.float64) - (1 - p)
        var = p * (1 - p)
        for lag in (1, 2, 3, 7, 8, 16, 64, 768):
            r = float((kept[:-lag] * kept[lag:]).mean() / var)
            assert abs(r) < 4 / np.sqrt(n - lag), (p, lag, r)
        cols = (out.reshape(512, 768) != 0).mean(0)
        assert abs(cols - (1 - p)).max() < 5 * np.sqrt(p * (1 - p) / 512), (p, cols.min(), cols.max())
