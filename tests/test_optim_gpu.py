"""Fused AdamW (etpnav_amd/csrc/optim.hip) through the C ABI against the CPU oracle (oracle/optim_oracle.py, itself
pinned to the reference's AdamW and torch.optim.AdamW by tests/test_optim_cpu.py).  fp32 elementwise arithmetic with
a different association order than torch's op-by-op sequence: tolerance 2e-6 abs + 1e-5 rel."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from etpnav_amd import _lib  # noqa: E402
from etpnav_amd._lib import check, ptr  # noqa: E402
from oracle import optim_oracle as oo  # noqa: E402
from oracle import planner_oracle as po  # noqa: E402

DEV = "cuda"


def close(a, b, atol=2e-6, rtol=1e-5):
    a, b = a.float().cpu(), b.float().cpu()
    return bool(((a - b).abs() <= atol + rtol * b.abs()).all())


@pytest.mark.parametrize("hf_style,correct_bias,max_norm", [(0, 1, 0.0), (1, 1, 5.0), (1, 0, 0.0), (0, 1, 1.0)])
def test_adamw_kernel_matches_oracle(hf_style, correct_bias, max_norm):
    torch.manual_seed(3)
    n, n_shadow = 1 << 18, 1 << 17
    p = torch.randn(n) * 0.3; m = torch.zeros(n); v = torch.zeros(n)
    mask = (torch.rand(n // 64) > 0.3).to(torch.uint8)
    wd_elem = mask.float().repeat_interleave(64) * 0.01
    dp, dm, dv = p.to(DEV), m.to(DEV), v.to(DEV)
    shadow = torch.zeros(n_shadow, dtype=torch.bfloat16, device=DEV)
    dmask = mask.to(DEV)
    sumsq = torch.zeros(1, device=DEV); nonfin = torch.zeros(1, dtype=torch.int32, device=DEV)
    L = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    for step in range(1, 4):
        g = torch.randn(n) * (4.0 if step == 2 else 0.2)
        dg = g.to(DEV)
        sumsq.zero_(); nonfin.zero_()
        check(L.etp_grad_sqnorm(ptr(dg), n, ptr(sumsq), ptr(nonfin), s), "sqnorm")
        c = _lib.AdamwCfg(lr=3e-3, beta1=0.9, beta2=0.98, eps=1e-6, weight_decay=0.01, step=step, hf_style=hf_style,
                          correct_bias=correct_bias, grad_scale=0.5, max_norm=max_norm)
        check(L.etp_adamw_step(ptr(dp), ptr(dg), ptr(dm), ptr(dv), ptr(shadow), n_shadow, ptr(dmask), n, ctypes.byref(c),
                               ptr(sumsq), ptr(nonfin), 1, s), "adamw")
        torch.cuda.synchronize()
        oo.adamw_step(p, g, m, v, step, 3e-3, 0.9, 0.98, 1e-6, wd_elem, bool(hf_style), bool(correct_bias), 0.5, max_norm)
        assert abs(sumsq.item() - float((g.double() ** 2).sum())) < 1e-3 * float((g.double() ** 2).sum())
        assert nonfin.item() == 0
        assert close(dp, p) and close(dm, m) and close(dv, v, atol=1e-7), step
        assert torch.equal(shadow.cpu(), dp[:n_shadow].to(torch.bfloat16).cpu())       # shadow = bf16(new masters)
        assert float(dg.abs().max()) == 0.0                                               # gradients zeroed in the same pass
    # GradScaler semantics: a non-finite gradient skips the update (p, m, v, shadow untouched) but still zeroes grads
    g = torch.randn(n); g[12345] = float("inf")
    dg = g.to(DEV)
    before = (dp.clone(), dm.clone(), dv.clone(), shadow.clone())
    sumsq.zero_(); nonfin.zero_()
    check(L.etp_grad_sqnorm(ptr(dg), n, ptr(sumsq), ptr(nonfin), s), "sqnorm")
    c = _lib.AdamwCfg(lr=3e-3, beta1=0.9, beta2=0.98, eps=1e-6, weight_decay=0.01, step=4, hf_style=hf_style,
                      correct_bias=correct_bias, grad_scale=0.5, max_norm=max_norm)
    check(L.etp_adamw_step(ptr(dp), ptr(dg), ptr(dm), ptr(dv), ptr(shadow), n_shadow, ptr(dmask), n, ctypes.byref(c),
                           ptr(sumsq), ptr(nonfin), 1, s), "adamw")
    torch.cuda.synchronize()
    assert nonfin.item() == 1
    assert torch.equal(dp, before[0]) and torch.equal(dm, before[1]) and torch.equal(dv, before[2]) and torch.equal(shadow, before[3])
    assert float(dg.abs().max()) == 0.0


def test_fused_adamw_closes_the_planner_step():
    """Two whole training steps (fwd+bwd through the planner, FusedAdamW in between) in bf16 mode: parameters follow the
    oracle's AdamW applied to the SAME device gradients, the bf16 weight shadow tracks the masters without a separate
    refresh, and gradients are zero when the next step starts (PlannerStep(refresh_weights=False, zero_grads=False))."""
    from etpnav_amd.optim import FusedAdamW
    from etpnav_amd.planner import GlocalTextPathNavCMT
    from etpnav_amd.step import PlannerStep
    cfg = po.PlannerConfig.r2r(vocab_size=1024, num_l_layers=2, num_pano_layers=1, num_x_layers=1)
    P = po.init_params(cfg, seed=1)
    batch = po.make_batch(cfg, B=2, L=12, V=9, G=6, seed=8, ragged=True)
    model = GlocalTextPathNavCMT(cfg.to_dict(), dtype=torch.bfloat16, device="cuda")
    model.load_state_dict(P, strict=True)
    opt = FusedAdamW(model, lr=1e-3, hf_style=True, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01, max_grad_norm=5.0,
                     no_decay=FusedAdamW.reference_no_decay, check_finite=True)
    first = PlannerStep(model, batch)                                        # step 1: plain refresh + memset
    first.run_eager(); torch.cuda.synchronize()
    losses = [first.loss.item()]
    first.close()
    step = PlannerStep(model, batch, refresh_weights=False, zero_grads=False)   # later steps rely on the optimizer
    eng = model._engine
    p_ref = eng.params.detach().cpu().clone()
    m_ref, v_ref = torch.zeros_like(p_ref), torch.zeros_like(p_ref)
    wd = torch.full_like(p_ref, 0.01)
    for name, shape, off in eng.table:
        if FusedAdamW.reference_no_decay(name):
            wd[off:off + int(np.prod(shape))] = 0.0
    for t in (1, 2):
        g = eng.grads.detach().cpu().clone()
        bad = opt.step()
        torch.cuda.synchronize()
        assert bad.item() == 0
        oo.adamw_step(p_ref, g, m_ref, v_ref, t, 1e-3, 0.9, 0.98, 1e-6, wd, True, True, 1.0, 5.0)
        assert close(eng.params, p_ref), t
        assert float(eng.grads.abs().max()) == 0.0
        assert torch.equal(eng.shadow.cpu(), eng.params[:eng.n_matrix].to(torch.bfloat16).cpu())
        step.run_eager(); torch.cuda.synchronize()
        losses.append(step.loss.item())
    assert losses[2] < losses[0]                                              # it trains
    # a skipped step (non-finite gradient) must not advance the step count that drives the bias correction --
    # GradScaler.step() does not call optimizer.step() on overflow (ss_trainer_ETP.py:504-506)
    assert opt.step_count == 2
    good = eng.grads.detach().clone()
    eng.grads[5] = float("inf")
    before = eng.params.detach().clone()
    bad = opt.step()
    torch.cuda.synchronize()
    assert bad.item() >= 1 and opt.step_count == 2 and torch.equal(eng.params, before)
    eng.grads.copy_(good)
    bad = opt.step()
    torch.cuda.synchronize()
    assert bad.item() == 0 and opt.step_count == 3
    oo.adamw_step(p_ref, good.cpu(), m_ref, v_ref, 3, 1e-3, 0.9, 0.98, 1e-6, wd, True, True, 1.0, 5.0)    # t = 3, not 4
    assert close(eng.params, p_ref)
    step.close()
