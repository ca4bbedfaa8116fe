"""Per-kernel parity on the MI355X: every C-ABI operator against plain torch math on the same inputs.

fp32 kernels ("parity mode") are held to 2e-4 abs on O(1) data (fp32 round-off, k-ordered MFMA chains);
bf16 kernels to the bf16 resolution of the result (documented per test).
"""
import ctypes
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from etpnav_amd import _lib  # noqa: E402
from etpnav_amd._lib import GemmDesc, AttnDesc, AttnBwdDesc, check, ptr  # noqa: E402

DEV = "cuda"


def L():
    return _lib.lib()


def stream():
    return torch.cuda.current_stream().cuda_stream


def tdt(dtype):
    return torch.bfloat16 if dtype == _lib.ETP_BF16 else torch.float32


def tol(dtype, scale=1.0):
    return (3e-2 if dtype == _lib.ETP_BF16 else 2e-4) * scale


def gelu(x):
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def gelu_grad(x):
    return 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)


def run_gemm(A, B, C, M, N, K, ta, tb, dtype, c_dtype=None, alpha=1.0, bias=None, R=None, Z=None, act=0, out_mode=0,
             ksplit=1, batch=1, batch_inner=1, strides=(0, 0, 0, 0, 0, 0), lda=None, ldb=None, ldc=None):
    d = GemmDesc()
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), C.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.lda = lda if lda is not None else A.stride(-2)
    d.ldb = ldb if ldb is not None else B.stride(-2)
    d.ldc = ldc if ldc is not None else C.stride(-2)
    d.trans_a, d.trans_b = ta, tb
    d.dtype = dtype
    d.c_dtype = dtype if c_dtype is None else c_dtype
    d.batch, d.batch_inner = batch, batch_inner
    d.sAo, d.sAi, d.sBo, d.sBi, d.sCo, d.sCi = strides
    d.ksplit, d.alpha = ksplit, alpha
    d.bias = bias.data_ptr() if bias is not None else None
    d.R = R.data_ptr() if R is not None else None
    d.ldr = R.stride(-2) if R is not None else 0
    d.Z = Z.data_ptr() if Z is not None else None
    d.ldz = Z.stride(-2) if Z is not None else 0
    d.act, d.out_mode = act, out_mode
    check(L().etp_gemm(ctypes.byref(d), stream()), "etp_gemm")
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", [_lib.ETP_F32, _lib.ETP_BF16])
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(20, 24, 64), (200, 136, 160), (256, 384, 512), (640, 768, 96)])
def test_gemm_layouts(dtype, ta, tb, M, N, K):
    """Asymmetric random operands (catches row/col swaps), ragged tiles, all storage pairings."""
    torch.manual_seed(M * 7 + N * 3 + K + ta * 2 + tb)
    t = tdt(dtype)
    A = torch.randn(M, K, device=DEV).to(t)
    B = torch.randn(N, K, device=DEV).to(t) * 0.5 + 0.1
    ref = A.float() @ B.float().t()
    As = A.t().contiguous() if ta else A
    Bs = B.t().contiguous() if tb else B
    # transposed operands need a leading dim that is a multiple of the 16-byte chunk
    def pad_ld(X):
        ld = (X.shape[1] + 7) // 8 * 8
        buf = torch.zeros(X.shape[0], ld, device=DEV, dtype=t)
        buf[:, :X.shape[1]] = X
        return buf
    As, Bs = pad_ld(As), pad_ld(Bs)
    C = torch.full((M, N), float("nan"), device=DEV, dtype=t)
    run_gemm(As, Bs, C, M, N, K, ta, tb, dtype)
    err = (C.float() - ref).abs().max().item()
    assert err <= tol(dtype, math.sqrt(K)), f"max err {err}"


@pytest.mark.parametrize("dtype", [_lib.ETP_F32, _lib.ETP_BF16])
def test_gemm_epilogues(dtype):
    torch.manual_seed(0)
    t = tdt(dtype)
    M, N, K = 150, 200, 128
    A = torch.randn(M, K, device=DEV).to(t)
    B = (torch.randn(N, K, device=DEV) * 0.1).to(t)
    bias = torch.randn(N, device=DEV)
    R = torch.randn(M, N, device=DEV).to(t)
    v = 0.5 * (A.float() @ B.float().t()) + bias
    # bias + gelu (+ saved pre-activation) + residual
    C = torch.empty(M, N, device=DEV, dtype=t); Z = torch.empty(M, N, device=DEV, dtype=t)
    run_gemm(A, B, C, M, N, K, 0, 0, dtype, alpha=0.5, bias=bias, R=R, Z=Z, act=_lib.ACT_GELU)
    assert (Z.float() - v).abs().max().item() <= tol(dtype, 4)
    assert (C.float() - (gelu(v) + R.float())).abs().max().item() <= tol(dtype, 4)
    # relu
    run_gemm(A, B, C, M, N, K, 0, 0, dtype, alpha=0.5, bias=bias, act=_lib.ACT_RELU)
    assert (C.float() - torch.relu(v)).abs().max().item() <= tol(dtype, 4)
    # gelu backward / relu backward read Z
    Zin = torch.randn(M, N, device=DEV).to(t)
    raw = A.float() @ B.float().t()
    run_gemm(A, B, C, M, N, K, 0, 0, dtype, Z=Zin, act=_lib.ACT_GELU_BWD)
    assert (C.float() - raw * gelu_grad(Zin.float())).abs().max().item() <= tol(dtype, 4)
    run_gemm(A, B, C, M, N, K, 0, 0, dtype, Z=Zin, act=_lib.ACT_RELU_BWD)
    assert (C.float() - raw * (Zin.float() > 0)).abs().max().item() <= tol(dtype, 4)
    # round 5: the pair the planner's FFN blocks use -- the forward saves gelu'(v) instead of v, the backward multiplies by it
    # (bf16 mode: the 2-byte Z buffer of this pair holds IEEE half values)
    tz = torch.float16 if dtype == _lib.ETP_BF16 else torch.float32
    Zg = torch.full((M, N), float("nan"), device=DEV, dtype=tz)
    run_gemm(A, B, C, M, N, K, 0, 0, dtype, alpha=0.5, bias=bias, R=R, Z=Zg, act=_lib.ACT_GELU_SAVEGRAD)
    assert (C.float() - (gelu(v) + R.float())).abs().max().item() <= tol(dtype, 4)
    assert (Zg.float() - gelu_grad(v)).abs().max().item() <= (1e-3 if dtype == _lib.ETP_BF16 else 2e-5)
    Zh = Zin.float().to(tz)
    run_gemm(A, B, C, M, N, K, 0, 0, dtype, Z=Zh, act=_lib.ACT_MUL_Z)
    assert (C.float() - raw * Zh.float()).abs().max().item() <= tol(dtype, 4)
    # accumulate into C
    C0 = torch.randn(M, N, device=DEV).to(t); C = C0.clone()
    run_gemm(A, B, C, M, N, K, 0, 0, dtype, out_mode=1)
    assert (C.float() - (C0.float() + raw)).abs().max().item() <= tol(dtype, 4)


@pytest.mark.parametrize("tile", ["256s2", "256s3", "128s2", "128s3", "ws4", "ws3", "64s3", "64s4"])
@pytest.mark.parametrize("tb", [0, 1])
def test_gemm_tile_classes_with_epilogues(tile, tb, etp_opt):
    """Every LDS-DMA tile class of round 3 -- 256x128 (eight wavefronts), 128x128, 128x64 and 64x64 (four), each with its ring
    depths -- forced through ETP_GEMM_TILE on ragged shapes (partial tiles in both directions, reductions shorter and longer
    than the ring), NT and NN storage, with the epilogues the planner uses:
    bias + fp32 residual into an fp32 stream, bias + GELU with the saved pre-activation, GELU backward, dropout."""
    etp_opt("ETP_GEMM_TILE", tile)
    dtype, t = _lib.ETP_BF16, torch.bfloat16
    for (M, N, K) in [(300, 200, 128), (130, 72, 192), (257, 136, 768), (64, 64, 1024), (520, 392, 256)]:
        torch.manual_seed(M + N + K + tb)
        A = torch.randn(M, K, device=DEV).to(t)
        B = (torch.randn(N, K, device=DEV) * 0.1).to(t)
        Bs = B.t().contiguous() if tb else B                       # NN: B stored [K][N]
        bias = torch.randn(N, device=DEV)
        raw = A.float() @ B.float().t()
        # fp32 stream output with bias + fp32 residual (out-proj / FFN-down / dgrad_s of the planner)
        R = torch.randn(M, N, device=DEV)
        C = torch.full((M, N), float("nan"), device=DEV)
        run_gemm(A, Bs, C, M, N, K, 0, tb, dtype, c_dtype=_lib.ETP_F32, bias=bias, R=R)
        assert (C - (raw + bias + R)).abs().max().item() <= tol(dtype, math.sqrt(K) / 4), (tile, M, N, K, "stream")
        # bf16 output, bias + GELU, pre-activation saved
        Cb = torch.full((M, N), float("nan"), device=DEV, dtype=t); Z = torch.empty(M, N, device=DEV, dtype=t)
        run_gemm(A, Bs, Cb, M, N, K, 0, tb, dtype, bias=bias, Z=Z, act=_lib.ACT_GELU)
        assert (Z.float() - (raw + bias)).abs().max().item() <= tol(dtype, math.sqrt(K) / 4), (tile, M, N, K, "z")
        assert (Cb.float() - gelu(raw + bias)).abs().max().item() <= tol(dtype, math.sqrt(K) / 4), (tile, M, N, K, "gelu")
        # GELU backward reads Z
        Zin = torch.randn(M, N, device=DEV).to(t)
        run_gemm(A, Bs, Cb, M, N, K, 0, tb, dtype, Z=Zin, act=_lib.ACT_GELU_BWD)
        assert (Cb.float() - raw * gelu_grad(Zin.float())).abs().max().item() <= tol(dtype, math.sqrt(K) / 4), (tile, M, N, K, "dgelu")


@pytest.mark.parametrize("tile", ["64s3", "64s4", "ws2", "ws3", "ws4", "128s2", "128s3", "256s2", "256s3"])
def test_gemm_race_screen_under_uneven_load(tile, etp_opt):
    """The round-3 main loop changed the synchronisation structure (one barrier BETWEEN a slab's k-steps, the whole ring in
    flight, counted vmcnt): cdna_hip_programming.md asks for a multi-run race screen of such edits, under UNEVEN load.  Every
    tile class / ring depth runs the same products 12 times while a bandwidth-heavy copy loop on a second stream perturbs the
    DMA timing; a stale or early LDS read would show as a run that differs from the others.  Outputs must be bit-identical
    across runs and match the fp32 reference; reductions of 2, 3, 5 and 24 slabs cover the ring's fill / drain paths."""
    etp_opt("ETP_GEMM_TILE", tile)
    dtype, t = _lib.ETP_BF16, torch.bfloat16
    side = torch.cuda.Stream()
    noise_a = torch.empty(64 << 20, device=DEV, dtype=torch.uint8)
    noise_b = torch.empty_like(noise_a)
    for (M, N, K, tb) in [(512, 384, 128, 0), (384, 256, 192, 1), (640, 768, 320, 0), (768, 512, 1536, 1)]:
        torch.manual_seed(M + K)
        A = torch.randn(M, K, device=DEV).to(t)
        B = (torch.randn(N, K, device=DEV) * 0.1).to(t)
        Bs = B.t().contiguous() if tb else B
        ref = A.float() @ B.float().t()
        first = None
        for it in range(12):
            C = torch.full((M, N), float("nan"), device=DEV)
            if it % 2 == 1:                      # every other run competes with a streaming copy (uneven load)
                with torch.cuda.stream(side):
                    for _ in range(3):
                        noise_b.copy_(noise_a, non_blocking=True)
            run_gemm(A, Bs, C, M, N, K, 0, tb, dtype, c_dtype=_lib.ETP_F32)
            side.synchronize()
            if first is None:
                first = C.clone()
                assert (C - ref).abs().max().item() <= tol(dtype, math.sqrt(K) / 4), (tile, M, N, K)
            else:
                assert torch.equal(C, first), (tile, M, N, K, it, (C - first).abs().max().item())


@pytest.mark.parametrize("dtype", [_lib.ETP_F32, _lib.ETP_BF16])
def test_gemm_wgrad_splitk_fp32_out(dtype):
    """dW[N,K] += dY[M,N]^T X[M,K]: TN storage, fp32 output, RMW and atomic split-K."""
    torch.manual_seed(1)
    t = tdt(dtype)
    M, N, K = 1000, 192, 136
    dY = torch.randn(M, N, device=DEV).to(t)
    X = torch.randn(M, K, device=DEV).to(t)
    ref = dY.float().t() @ X.float()
    W0 = torch.randn(N, K, device=DEV)
    for ks, mode in ((1, 1), (4, 2)):
        W = W0.clone()
        run_gemm(dY, X, W, N, K, M, 1, 1, dtype, c_dtype=_lib.ETP_F32, out_mode=mode, ksplit=ks)
        err = (W - (W0 + ref)).abs().max().item()
        assert err <= tol(dtype, math.sqrt(M) / 4), f"ksplit {ks}: {err}"


@pytest.mark.parametrize("dtype", [_lib.ETP_F32, _lib.ETP_BF16])
@pytest.mark.parametrize("H", [768, 256])
def test_layer_norm(dtype, H):
    torch.manual_seed(2)
    t = tdt(dtype)
    M = 77
    x = (torch.randn(M, H, device=DEV) * 2 + 0.3).to(t)
    g = torch.randn(H, device=DEV); b = torch.randn(H, device=DEV)
    y = torch.empty_like(x); stats = torch.empty(M, 2, device=DEV)
    check(L().etp_ln_fwd(dtype, ptr(x), ptr(g), ptr(b), ptr(y), ptr(stats), M, H, 1e-12, stream()), "ln_fwd")
    xr = x.float().requires_grad_(True); gr = g.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (H,), gr, br, 1e-12)
    assert (y.float() - yr).abs().max().item() <= tol(dtype, 2)
    assert (stats[:, 0] - x.float().mean(-1)).abs().max().item() < 1e-4
    dy = torch.randn(M, H, device=DEV).to(t)
    add = torch.randn(M, H, device=DEV).to(t)
    yr.backward(dy.float())
    dx = torch.empty_like(x); dg = torch.zeros(H, device=DEV); db = torch.zeros(H, device=DEV)
    check(L().etp_ln_bwd(dtype, ptr(dy), ptr(x), ptr(stats), ptr(g), ptr(add), ptr(dx), ptr(dg), ptr(db), M, H, stream()),
          "ln_bwd")
    assert (dx.float() - (xr.grad + add.float())).abs().max().item() <= tol(dtype, 4)
    assert (dg - gr.grad).abs().max().item() <= tol(dtype, 1) + 1e-3
    assert (db - br.grad).abs().max().item() <= 1e-3


@pytest.mark.parametrize("dtype", [_lib.ETP_F32, _lib.ETP_BF16])
@pytest.mark.parametrize("mask_mode", [0, 1])
def test_softmax(dtype, mask_mode):
    torch.manual_seed(3)
    t = tdt(dtype)
    B, nh, Lq, Lk, ldS = 3, 4, 9, 13, 16
    S = torch.full((B, nh, Lq, ldS), float("nan"), device=DEV, dtype=t)
    S[..., :Lk] = torch.randn(B, nh, Lq, Lk, device=DEV).to(t)
    km = torch.rand(B, Lk, device=DEV) > 0.3
    km[:, 0] = True
    dist = torch.rand(B, Lq, Lk, device=DEV)
    w = torch.tensor([0.7], device=DEV); b0 = torch.tensor([-0.2], device=DEV)
    s = S[..., :Lk].float() + (w * dist + b0)[:, None]
    if mask_mode == 0:
        s = s + (1.0 - km.float())[:, None, None, :] * -10000.0
    else:
        s = s.masked_fill(~km[:, None, None, :], float("-inf"))
    sr = s.clone().requires_grad_(True)
    pr = torch.softmax(sr, -1)
    P = S.clone()
    check(L().etp_softmax_fwd(dtype, ptr(P), ptr(km), ptr(dist), ptr(w), ptr(b0), B, nh, Lq, Lk, ldS, mask_mode, stream()),
          "softmax_fwd")
    assert (P[..., :Lk].float() - pr).abs().max().item() <= tol(dtype, 0.5)
    assert (P[..., Lk:].float() == 0).all()
    dP = torch.zeros_like(P)
    dP[..., :Lk] = torch.randn(B, nh, Lq, Lk, device=DEV).to(t)
    pr2 = P[..., :Lk].float().detach().requires_grad_(True)
    # reference backward from the kernel's own (rounded) P so the check isolates the backward math
    dot = (pr2 * dP[..., :Lk].float()).sum(-1, keepdim=True)
    ds_ref = pr2 * (dP[..., :Lk].float() - dot)
    dw = torch.zeros(1, device=DEV); db = torch.zeros(1, device=DEV)
    check(L().etp_softmax_bwd(dtype, ptr(P), ptr(dP), ptr(dist), ptr(dw), ptr(db), B, nh, Lq, Lk, ldS, stream()), "softmax_bwd")
    assert (dP[..., :Lk].float() - ds_ref).abs().max().item() <= tol(dtype, 1)
    assert (dP[..., Lk:].float() == 0).all()
    assert abs(dw.item() - (ds_ref.sum(1) * dist).sum().item()) <= tol(dtype, 2) + 1e-3
    assert abs(db.item() - ds_ref.sum().item()) <= tol(dtype, 2) + 1e-3


@pytest.mark.parametrize("dtype", [_lib.ETP_F32, _lib.ETP_BF16])
@pytest.mark.parametrize("with_dist", [True, False])
@pytest.mark.parametrize("Lq,Lk", [(80, 80), (9, 20), (36, 36), (16, 512), (16, 80), (100, 120), (70, 40), (96, 72), (128, 128),
                                   (37, 37), (23, 80), (80, 17), (1, 1), (113, 16)])
def test_attention_fwd_bwd(dtype, Lq, Lk, with_dist):
    """softmax(QK^T/8 + mask + sprel)V and its backward against torch autograd (head dim 64).  bf16 with both axes <= 128 runs
    the register-resident kernels (attn_rows.hip: one wavefront per 16 queries / 16 keys, every tile count 1..8 on either
    axis is covered by the shape list), with and without the pairwise-distance bias of the graph self-attention."""
    torch.manual_seed(4)
    t = tdt(dtype)
    B, nh, dh = 2, 3, 64
    H = nh * dh
    ldS = (Lk + 7) // 8 * 8
    q = torch.randn(B * Lq, H, device=DEV).to(t)
    kv = torch.randn(B * Lk, 2 * H, device=DEV).to(t)
    km = torch.rand(B, Lk, device=DEV) > 0.2
    km[:, 0] = True
    dist = torch.rand(B, Lq, Lk, device=DEV)
    w = torch.tensor([0.3], device=DEV); b0 = torch.tensor([0.1], device=DEV)
    qr = q.float().requires_grad_(True); kvr = kv.float().requires_grad_(True)
    wr = w.clone().requires_grad_(True); br = b0.clone().requires_grad_(True)
    qh = qr.view(B, Lq, nh, dh).permute(0, 2, 1, 3)
    kh = kvr[:, :H].reshape(B, Lk, nh, dh).permute(0, 2, 1, 3)
    vh = kvr[:, H:].reshape(B, Lk, nh, dh).permute(0, 2, 1, 3)
    s = qh @ kh.transpose(-1, -2) / 8.0 + (1.0 - km.float())[:, None, None, :] * -10000.0
    if with_dist:
        s = s + (wr * dist + br)[:, None]
    ctx_ref = (torch.softmax(s, -1) @ vh).permute(0, 2, 1, 3).reshape(B * Lq, H)
    P = torch.empty(B, nh, Lq, ldS, device=DEV, dtype=t)
    ctx = torch.empty(B * Lq, H, device=DEV, dtype=t)
    d = AttnDesc()
    d.dtype, d.B, d.heads, d.Lq, d.Lk, d.ldS = dtype, B, nh, Lq, Lk, ldS
    d.Q, d.ldq = q.data_ptr(), H
    d.K, d.ldk = kv.data_ptr(), 2 * H
    d.V, d.ldv = kv.data_ptr() + H * q.element_size(), 2 * H
    d.P, d.ctx, d.ldc = P.data_ptr(), ctx.data_ptr(), H
    d.keymask, d.mask_mode = km.data_ptr(), 0
    if with_dist:
        d.dist, d.sp_w, d.sp_b = dist.data_ptr(), w.data_ptr(), b0.data_ptr()
    d.alpha = 0.125
    check(L().etp_attn_fwd(ctypes.byref(d), stream()), "attn_fwd")
    torch.cuda.synchronize()
    assert (ctx.float() - ctx_ref).abs().max().item() <= tol(dtype, 2)
    dctx = torch.randn(B * Lq, H, device=DEV).to(t)
    ctx_ref.backward(dctx.float())
    bd = AttnBwdDesc()
    bd.f = d
    dP = torch.empty_like(P); dq = torch.full_like(q, float("nan")); dkv = torch.full_like(kv, float("nan"))
    dw = torch.zeros(1, device=DEV); db = torch.zeros(1, device=DEV)
    bd.dctx, bd.ldd, bd.dP = dctx.data_ptr(), H, dP.data_ptr()
    bd.dQ, bd.lddq = dq.data_ptr(), H
    bd.dK, bd.lddk = dkv.data_ptr(), 2 * H
    bd.dV, bd.lddv = dkv.data_ptr() + H * q.element_size(), 2 * H
    if with_dist:
        bd.d_sp_w, bd.d_sp_b = dw.data_ptr(), db.data_ptr()
    check(L().etp_attn_bwd(ctypes.byref(bd), stream()), "attn_bwd")
    torch.cuda.synchronize()
    sc = 4 if dtype == _lib.ETP_F32 else 3
    assert (dq.float() - qr.grad).abs().max().item() <= tol(dtype, sc)
    assert (dkv.float() - kvr.grad).abs().max().item() <= tol(dtype, sc)
    if with_dist:
        assert abs(dw.item() - wr.grad.item()) <= tol(dtype, 4) + 2e-3
        assert abs(db.item() - br.grad.item()) <= tol(dtype, 4) + 2e-3


@pytest.mark.parametrize("Lq,Lk", [(512, 512), (200, 200), (16, 512), (130, 300), (64, 129), (300, 70)])
@pytest.mark.parametrize("mask_mode", [0, 1])
def test_streaming_attention_fwd_bwd(Lq, Lk, mask_mode):
    """bf16 streaming ("flash") kernels for Lq or Lk > 128 (RxR's 512-token instructions): online softmax over 128-key tiles,
    probabilities recomputed in the backward -- against torch autograd on the same bf16 operands.  A spiked key forces the
    running-max rescale path in a LATER tile (cdna guide rule 26: the rare branch needs its own input)."""
    torch.manual_seed(Lq * 7 + Lk + mask_mode)
    dtype, t = _lib.ETP_BF16, torch.bfloat16
    B, nh, dh = 2, 3, 64
    H = nh * dh
    ldS = (Lk + 7) // 8 * 8
    q = torch.randn(B * Lq, H, device=DEV).to(t)
    kv = torch.randn(B * Lk, 2 * H, device=DEV).to(t)
    if Lk > 140:
        kv[Lk - 3, :H] = q[0, :H] * 3.0                       # batch 0: the last tile holds the row max of query 0
    km = torch.rand(B, Lk, device=DEV) > 0.2
    km[:, 0] = True
    km[1, Lk // 2:] = False                                  # batch 1: whole trailing tiles masked out
    qr = q.float().requires_grad_(True); kvr = kv.float().requires_grad_(True)
    qh = qr.view(B, Lq, nh, dh).permute(0, 2, 1, 3)
    kh = kvr[:, :H].reshape(B, Lk, nh, dh).permute(0, 2, 1, 3)
    vh = kvr[:, H:].reshape(B, Lk, nh, dh).permute(0, 2, 1, 3)
    add = torch.where(km, 0.0, float("-inf") if mask_mode else -10000.0)[:, None, None, :]
    s = qh @ kh.transpose(-1, -2) / 8.0 + add
    ctx_ref = (torch.softmax(s, -1) @ vh).permute(0, 2, 1, 3).reshape(B * Lq, H)
    P = torch.full((B, nh, Lq, ldS), float("nan"), device=DEV, dtype=t)
    ctx = torch.empty(B * Lq, H, device=DEV, dtype=t)
    d = AttnDesc()
    d.dtype, d.B, d.heads, d.Lq, d.Lk, d.ldS = dtype, B, nh, Lq, Lk, ldS
    d.Q, d.ldq = q.data_ptr(), H
    d.K, d.ldk = kv.data_ptr(), 2 * H
    d.V, d.ldv = kv.data_ptr() + H * q.element_size(), 2 * H
    d.P, d.ctx, d.ldc = P.data_ptr(), ctx.data_ptr(), H
    d.keymask, d.mask_mode = km.data_ptr(), mask_mode
    d.alpha = 0.125
    check(L().etp_attn_fwd(ctypes.byref(d), stream()), "attn_fwd")
    torch.cuda.synchronize()
    assert (ctx.float() - ctx_ref).abs().max().item() <= tol(dtype, 2)
    dctx = torch.randn(B * Lq, H, device=DEV).to(t)
    ctx_ref.backward(dctx.float())
    bd = AttnBwdDesc()
    bd.f = d
    dP = torch.empty_like(P); dq = torch.full_like(q, float("nan")); dkv = torch.full_like(kv, float("nan"))
    bd.dctx, bd.ldd, bd.dP = dctx.data_ptr(), H, dP.data_ptr()
    bd.dQ, bd.lddq = dq.data_ptr(), H
    bd.dK, bd.lddk = dkv.data_ptr(), 2 * H
    bd.dV, bd.lddv = dkv.data_ptr() + H * q.element_size(), 2 * H
    check(L().etp_attn_bwd(ctypes.byref(bd), stream()), "attn_bwd")
    torch.cuda.synchronize()
    assert (dq.float() - qr.grad).abs().max().item() <= tol(dtype, 3) * max(1.0, qr.grad.abs().max().item() / 4)
    assert (dkv.float() - kvr.grad).abs().max().item() <= tol(dtype, 3) * max(1.0, kvr.grad.abs().max().item() / 4)


@pytest.mark.parametrize("B,Lt", [(5, 24), (32, 80), (2, 130), (1, 7)])
def test_text_embedding_fwd_bwd(B, Lt):
    """BertEmbeddings (vilmodel_cmt.py:62-77): LN(word[id] + pos[l] + type[0]) and its backward -- the position-embedding
    gradient is a per-position segment sum over the batch, the word-table gradient a scatter-add with repeated ids and the
    padding row 0 left at zero (:53)."""
    torch.manual_seed(B * 131 + Lt)
    H, vocab = 768, 300
    word = torch.randn(vocab, H, device=DEV); pos = torch.randn(Lt + 3, H, device=DEV); typ = torch.randn(2, H, device=DEV)
    gamma = torch.rand(H, device=DEV) + 0.5; beta = torch.randn(H, device=DEV)
    ids = torch.randint(1, vocab, (B, Lt), device=DEV)
    ids[:, Lt // 2:] = ids[:, :Lt - Lt // 2].clone()            # every id at least twice in a row of the batch
    ids[0, -2:] = 0                                          # padding id
    wr, pr, tr = (t.clone().requires_grad_(True) for t in (word, pos, typ))
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    emb = torch.nn.functional.embedding(ids, wr, padding_idx=0) + pr[:Lt][None] + tr[0]
    ref = torch.nn.functional.layer_norm(emb, (H,), gr, br, 1e-12)
    y = torch.empty(B * Lt, H, device=DEV); stats = torch.empty(B * Lt, 2, device=DEV)
    check(L().etp_text_embed_fwd(_lib.ETP_F32, ptr(ids), ptr(word), ptr(pos), ptr(typ), ptr(gamma), ptr(beta), ptr(y), None,
                                  ptr(stats), B, Lt, H, 1e-12, stream()), "text_embed_fwd")
    assert (y.view(B, Lt, H) - ref).abs().max().item() < 2e-5
    dy = torch.randn(B * Lt, H, device=DEV)
    ref.backward(dy.view(B, Lt, H))
    dword = torch.zeros_like(word); dpos = torch.zeros_like(pos); dtyp = torch.zeros(H, device=DEV)
    dg = torch.zeros(H, device=DEV); db = torch.zeros(H, device=DEV)
    check(L().etp_text_embed_bwd(_lib.ETP_F32, ptr(dy), ptr(ids), ptr(word), ptr(pos), ptr(typ), ptr(gamma), ptr(stats),
                                  ptr(dword), ptr(dpos), ptr(dtyp), ptr(dg), ptr(db), B, Lt, H, stream()), "text_embed_bwd")
    torch.cuda.synchronize()
    tol_ = lambda g: 2e-5 * max(1.0, g.abs().max().item())
    assert (dword - wr.grad).abs().max().item() < tol_(wr.grad) and dword[0].abs().max().item() == 0
    assert (dpos - pr.grad).abs().max().item() < tol_(pr.grad)
    assert (dtyp - tr.grad[0]).abs().max().item() < tol_(tr.grad)
    assert (dg - gr.grad).abs().max().item() < tol_(gr.grad) and (db - br.grad).abs().max().item() < tol_(br.grad)


def test_cross_entropy_and_gather():
    torch.manual_seed(5)
    B, G = 7, 11
    logits = torch.randn(B, G, device=DEV)
    logits[:, 1] = float("-inf")
    labels = torch.randint(2, G, (B,), device=DEV)
    labels[3] = -100
    lr = logits.clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lr, labels, reduction="sum", ignore_index=-100) / B
    ref.backward()
    loss = torch.full((1,), 123.0, device=DEV); dl = torch.empty(B, G, device=DEV)     # the kernel STORES the loss
    check(L().etp_sap_ce(ptr(logits), ptr(labels), ptr(loss), ptr(dl), B, G, 1.0 / B, -100, stream()), "sap_ce")
    assert abs(loss.item() - ref.item()) < 1e-5
    assert (dl - lr.grad).abs().max().item() < 1e-6
    # gather-sum
    for dtype in (_lib.ETP_F32, _lib.ETP_BF16):
        t = tdt(dtype)
        src = torch.randn(10, 768, device=DEV).to(t)
        p = torch.tensor([0, 0, 3, 4], dtype=torch.int32, device=DEV)
        idx = torch.tensor([1, 5, 9, 2], dtype=torch.int32, device=DEV)
        w = torch.tensor([0.5, 0.25, 0.25, 2.0], device=DEV)
        out = torch.full((3, 768), float("nan"), device=DEV, dtype=t)
        check(L().etp_gather_sum(dtype, ptr(src), ptr(p), ptr(idx), ptr(w), ptr(out), 3, 768, 0, stream()), "gather_sum")
        s = src.float()
        ref = torch.stack([torch.zeros(768, device=DEV), 0.5 * s[1] + 0.25 * s[5] + 0.25 * s[9], 2 * s[2]])
        assert (out.float() - ref).abs().max().item() <= tol(dtype, 0.5)


def test_colsum_and_cast():
    torch.manual_seed(6)
    for dtype in (_lib.ETP_F32, _lib.ETP_BF16):
        t = tdt(dtype)
        dy = torch.randn(333, 776, device=DEV).to(t)
        db = torch.ones(776, device=DEV)
        check(L().etp_colsum(dtype, ptr(dy), 776, ptr(db), 333, 776, stream()), "colsum")
        assert (db - (1 + dy.float().sum(0))).abs().max().item() < 2e-3
    x = torch.randn(100003, device=DEV)
    y = torch.empty(100003, device=DEV, dtype=torch.bfloat16)
    check(L().etp_cast_f32_to_bf16(ptr(x), ptr(y), x.numel(), stream()), "cast")
    assert torch.equal(y, x.to(torch.bfloat16))


# ---- grouped GEMM (one grid for the weight gradients of a layer) and two-stage LayerNorm backward -------------------------
def _desc(A, B, C, M, N, K, ta, tb, dtype, c_dtype, out_mode=0):
    d = GemmDesc()
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), C.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = A.stride(-2), B.stride(-2), C.stride(-2)
    d.trans_a, d.trans_b, d.dtype, d.c_dtype = ta, tb, dtype, c_dtype
    d.batch, d.batch_inner, d.ksplit, d.alpha, d.out_mode = 1, 1, 1, 1.0, out_mode
    return d


@pytest.mark.parametrize("dtype", [_lib.ETP_F32, _lib.ETP_BF16])
@pytest.mark.parametrize("tile", ["", "256s2", "256s3", "128s2", "128s3", "64s3", "64s4"])
def test_gemm_group_weight_gradients(dtype, tile, etp_opt):
    """Seven TN products of different shapes / reduction lengths (an x-layer's weight gradients: token counts 512 and 2560,
    ragged 200-wide output) in ONE grid == the same products one by one; store and accumulate modes; every tile class."""
    etp_opt("ETP_GROUP_TILE", tile)
    torch.manual_seed(3)
    t = tdt(dtype)
    shapes = [(768, 768, 512), (1536, 768, 2560), (768, 768, 512), (2304, 768, 512), (200, 136, 512), (3072, 768, 512),
              (768, 3072, 512)]            # (N_out, K_in, tokens)
    descs, refs, outs = (GemmDesc * len(shapes))(), [], []
    keep = []
    for i, (n, k, m) in enumerate(shapes):
        dY = (torch.randn(m, n, device=DEV) * 0.5).to(t)
        X = (torch.randn(m, k, device=DEV) * 0.5 + 0.05).to(t)
        W0 = torch.randn(n, k, device=DEV)
        mode = i % 2                                   # alternate first-touch store / accumulate
        ref = dY.float().t() @ X.float() + (W0 if mode == 1 else 0)
        W = W0.clone()
        descs[i] = _desc(dY, X, W, n, k, m, 1, 1, dtype, _lib.ETP_F32, out_mode=mode)
        keep += [dY, X]
        refs.append(ref); outs.append(W)
    check(L().etp_gemm_group(descs, len(shapes), stream()), "etp_gemm_group")
    torch.cuda.synchronize()
    for (n, k, m), W, ref in zip(shapes, outs, refs):
        err = (W - ref).abs().max().item()
        assert err <= tol(dtype, math.sqrt(m) / 4), f"{(n, k, m)}: {err}"


@pytest.mark.parametrize("tile", ["", "256s2", "256s3", "128s2"])
def test_gemm_group_text_layer_weight_gradients_large_tiles(tile, etp_opt):
    """The four weight gradients of a text layer (the dominant launch of a step) as ONE grouped grid with every tile class
    that fits them -- 256x128 tiles make it 216 workgroups, one per CU -- against fp32 matmuls; tokens = 640 (10 slabs)."""
    etp_opt("ETP_GROUP_TILE", tile)
    torch.manual_seed(5)
    dtype, t = _lib.ETP_BF16, torch.bfloat16
    shapes = [(2304, 768, 640), (768, 768, 640), (3072, 768, 640), (768, 3072, 640)]
    descs, refs, outs, keep = (GemmDesc * len(shapes))(), [], [], []
    for i, (n, k, m) in enumerate(shapes):
        dY = (torch.randn(m, n, device=DEV) * 0.5).to(t)
        X = (torch.randn(m, k, device=DEV) * 0.5 + 0.05).to(t)
        W = torch.full((n, k), float("nan"), device=DEV)
        descs[i] = _desc(dY, X, W, n, k, m, 1, 1, dtype, _lib.ETP_F32, out_mode=0)
        keep += [dY, X]
        refs.append(dY.float().t() @ X.float()); outs.append(W)
    check(L().etp_gemm_group(descs, len(shapes), stream()), "etp_gemm_group")
    torch.cuda.synchronize()
    for (n, k, m), W, ref in zip(shapes, outs, refs):
        err = (W - ref).abs().max().item()
        assert err <= tol(dtype, math.sqrt(m) / 4), f"{tile} {(n, k, m)}: {err}"


@pytest.mark.parametrize("dtype", [_lib.ETP_F32, _lib.ETP_BF16])
def test_gemm_group_forward_products(dtype):
    """NT class with bias epilogue (the text K/V projections of all x-layers in one launch)."""
    torch.manual_seed(4)
    t = tdt(dtype)
    M, N, K = 640, 1536, 768
    X = torch.randn(M, K, device=DEV).to(t)
    descs = (GemmDesc * 4)()
    Ws, Cs, bs = [], [], []
    for i in range(4):
        W = (torch.randn(N, K, device=DEV) * 0.05).to(t); b = torch.randn(N, device=DEV)
        C = torch.empty(M, N, device=DEV, dtype=t)
        d = _desc(X, W, C, M, N, K, 0, 0, dtype, dtype)
        d.bias = b.data_ptr()
        descs[i] = d
        Ws.append(W); Cs.append(C); bs.append(b)
    check(L().etp_gemm_group(descs, 4, stream()), "etp_gemm_group")
    torch.cuda.synchronize()
    for W, C, b in zip(Ws, Cs, bs):
        ref = X.float() @ W.float().t() + b
        assert (C.float() - ref).abs().max().item() <= tol(dtype, 4)


@pytest.mark.parametrize("dtype", [_lib.ETP_F32, _lib.ETP_BF16])
@pytest.mark.parametrize("M", [7, 512, 2560, 5000])
def test_layer_norm_backward_two_stage(dtype, M):
    """ln_bwd_s with per-workgroup slabs + ln_part_reduce == autograd of F.layer_norm (dx, dgamma, dbeta accumulate)."""
    torch.manual_seed(M)
    H = 768
    x = torch.randn(M, H, device=DEV) * 2 + 0.3
    gmm = torch.randn(H, device=DEV); bta = torch.randn(H, device=DEV)
    dy = torch.randn(M, H, device=DEV)
    add = torch.randn(M, H, device=DEV)
    xr = x.clone().requires_grad_(True); gr = gmm.clone().requires_grad_(True); br = bta.clone().requires_grad_(True)
    y = torch.nn.functional.layer_norm(xr, (H,), gr, br, 1e-12)
    y.backward(dy)
    stats = torch.stack([x.mean(1), 1.0 / torch.sqrt(x.var(1, unbiased=False) + 1e-12)], 1).contiguous()
    dx = torch.empty(M, H, device=DEV); dxt = torch.empty(M, H, device=DEV, dtype=tdt(dtype))
    dg0 = torch.randn(H, device=DEV); db0 = torch.randn(H, device=DEV)
    dg, db = dg0.clone(), db0.clone()
    part = torch.empty(int(L().etp_ln_bwd_part_bytes(M, H)), dtype=torch.uint8, device=DEV)
    check(L().etp_ln_stream_bwd_stage1(dtype, ptr(dy), ptr(x), ptr(stats), ptr(gmm), ptr(add), ptr(dx), ptr(dxt), ptr(dg), ptr(db),
                                       ptr(part), M, H, stream()), "ln stage1")
    check(L().etp_ln_part_reduce(ptr(part), M, H, ptr(dg), ptr(db), stream()), "ln stage2")
    torch.cuda.synchronize()
    assert (dx - (xr.grad + add)).abs().max().item() < 2e-4
    assert (dxt.float() - (xr.grad + add)).abs().max().item() <= tol(dtype, 4)
    assert (dg - dg0 - gr.grad).abs().max().item() < 2e-4 * math.sqrt(M)
    assert (db - db0 - br.grad).abs().max().item() < 2e-4 * math.sqrt(M)


@pytest.mark.parametrize("with_dist", [False, True])
@pytest.mark.parametrize("Lq,Lk", [(80, 80), (16, 80), (36, 36), (9, 20), (128, 128), (113, 16), (16, 16), (1, 1)])
def test_attention_bwd_with_fused_out_projection(Lq, Lk, with_dist):
    """etp_attn_bwd_proj (round 6; attn_rows.hip PROJ): the (batch, head) workgroup forms its dctx tile = dY . W_out[:, head] itself.
    Checked against the pair it replaces -- etp_gemm (dgrad of BertSelfOutput.dense, vilmodel_cmt.py:150-154) + etp_attn_bwd -- on
    the same operands: the tile is rounded to bf16 where the GEMM stored it, so the two differ only through the order of the fp32
    sums (an occasional bf16 ulp of dctx), and against torch autograd in fp32."""
    torch.manual_seed(11)
    t = torch.bfloat16
    B, nh, dh = 3, 12, 64
    H = nh * dh
    ldS = (Lk + 7) // 8 * 8
    q = torch.randn(B * Lq, H, device=DEV).to(t)
    kv = torch.randn(B * Lk, 2 * H, device=DEV).to(t)
    km = torch.rand(B, Lk, device=DEV) > 0.2
    km[:, 0] = True
    dist = torch.rand(B, Lq, Lk, device=DEV)
    w = torch.tensor([0.3], device=DEV); b0 = torch.tensor([0.1], device=DEV)
    P = torch.empty(B, nh, Lq, ldS, device=DEV, dtype=t)
    ctx = torch.empty(B * Lq, H, device=DEV, dtype=t)
    d = AttnDesc()
    d.dtype, d.B, d.heads, d.Lq, d.Lk, d.ldS = _lib.ETP_BF16, B, nh, Lq, Lk, ldS
    d.Q, d.ldq = q.data_ptr(), H
    d.K, d.ldk = kv.data_ptr(), 2 * H
    d.V, d.ldv = kv.data_ptr() + H * 2, 2 * H
    d.P, d.ctx, d.ldc = P.data_ptr(), ctx.data_ptr(), H
    d.keymask, d.mask_mode = km.data_ptr(), 0
    if with_dist:
        d.dist, d.sp_w, d.sp_b = dist.data_ptr(), w.data_ptr(), b0.data_ptr()
    d.alpha = 0.125
    check(L().etp_attn_fwd(ctypes.byref(d), stream()), "attn_fwd")
    dy = (torch.randn(B * Lq, H, device=DEV) * 0.5).to(t)
    Wo = (torch.randn(H, H, device=DEV) / math.sqrt(H)).to(t)           # [out][in], as nn.Linear stores it
    dctx = torch.empty(B * Lq, H, device=DEV, dtype=t)
    run_gemm(dy, Wo, dctx, B * Lq, H, H, 0, 1, _lib.ETP_BF16)           # dctx = dy . Wo   (B operand [reduction][N])
    ref_dctx = (dy.float() @ Wo.float())
    assert (dctx.float() - ref_dctx).abs().max().item() <= 2e-2

    def run(fused):
        bd = AttnBwdDesc()
        bd.f = d
        dP = torch.empty_like(P); dq = torch.full_like(q, float("nan")); dkv = torch.full_like(kv, float("nan"))
        dw = torch.zeros(1, device=DEV); db = torch.zeros(1, device=DEV)
        bd.dctx, bd.ldd, bd.dP = (dy if fused else dctx).data_ptr(), H, dP.data_ptr()
        bd.dQ, bd.lddq = dq.data_ptr(), H
        bd.dK, bd.lddk = dkv.data_ptr(), 2 * H
        bd.dV, bd.lddv = dkv.data_ptr() + H * 2, 2 * H
        if with_dist:
            bd.d_sp_w, bd.d_sp_b = dw.data_ptr(), db.data_ptr()
        if fused:
            check(L().etp_attn_bwd_proj(ctypes.byref(bd), Wo.data_ptr(), H, stream()), "attn_bwd_proj")
        else:
            check(L().etp_attn_bwd(ctypes.byref(bd), stream()), "attn_bwd")
        torch.cuda.synchronize()
        return dq.float(), dkv.float(), dw.item(), db.item()

    dq_f, dkv_f, dw_f, db_f = run(True)
    dq_u, dkv_u, dw_u, db_u = run(False)
    assert torch.isfinite(dq_f).all() and torch.isfinite(dkv_f).all()
    # against the unfused pair: a handful of dctx elements may sit one bf16 ulp apart
    for a, b_ in ((dq_f, dq_u), (dkv_f, dkv_u)):
        scale = b_.abs().max().item() + 1e-6
        assert (a - b_).abs().max().item() <= 2e-2 * scale, ((a - b_).abs().max().item(), scale)
        assert (a - b_).norm().item() <= 2e-3 * b_.norm().item() + 1e-6
    if with_dist:
        assert abs(dw_f - dw_u) <= 2e-2 * (abs(dw_u) + 1.0) and abs(db_f - db_u) <= 2e-2 * (abs(db_u) + 1.0)
    # against torch autograd (fp32 math on the bf16 operands, dctx rounded to bf16 as both paths do)
    qr = q.float().requires_grad_(True); kvr = kv.float().requires_grad_(True)
    qh = qr.view(B, Lq, nh, dh).permute(0, 2, 1, 3)
    kh = kvr[:, :H].reshape(B, Lk, nh, dh).permute(0, 2, 1, 3)
    vh = kvr[:, H:].reshape(B, Lk, nh, dh).permute(0, 2, 1, 3)
    s = qh @ kh.transpose(-1, -2) / 8.0 + (1.0 - km.float())[:, None, None, :] * -10000.0
    if with_dist:
        s = s + (w * dist + b0)[:, None]
    ctx_ref = (torch.softmax(s, -1) @ vh).permute(0, 2, 1, 3).reshape(B * Lq, H)
    ctx_ref.backward(ref_dctx.to(t).float())
    assert (dq_f - qr.grad).abs().max().item() <= tol(_lib.ETP_BF16, 3) * max(1.0, qr.grad.abs().max().item())
    assert (dkv_f - kvr.grad).abs().max().item() <= tol(_lib.ETP_BF16, 3) * max(1.0, kvr.grad.abs().max().item())


def test_attention_bwd_proj_refuses_what_the_fused_kernel_does_not_take():
    t = torch.bfloat16
    B, nh, Lq, Lk = 1, 4, 16, 16                                          # heads * 64 = 256: not the fused reduction length
    H = nh * 64
    q = torch.randn(B * Lq, H, device=DEV).to(t); kv = torch.randn(B * Lk, 2 * H, device=DEV).to(t)
    P = torch.empty(B, nh, Lq, 16, device=DEV, dtype=t); ctx = torch.empty(B * Lq, H, device=DEV, dtype=t)
    d = AttnDesc()
    d.dtype, d.B, d.heads, d.Lq, d.Lk, d.ldS = _lib.ETP_BF16, B, nh, Lq, Lk, 16
    d.Q, d.ldq, d.K, d.ldk, d.V, d.ldv = q.data_ptr(), H, kv.data_ptr(), 2 * H, kv.data_ptr() + 2 * H, 2 * H
    d.P, d.ctx, d.ldc, d.alpha = P.data_ptr(), ctx.data_ptr(), H, 0.125
    bd = AttnBwdDesc()
    bd.f = d
    dq = torch.empty_like(q); dkv = torch.empty_like(kv); dP = torch.empty_like(P)
    bd.dctx, bd.ldd, bd.dP = ctx.data_ptr(), H, dP.data_ptr()
    bd.dQ, bd.lddq, bd.dK, bd.lddk, bd.dV, bd.lddv = dq.data_ptr(), H, dkv.data_ptr(), 2 * H, dkv.data_ptr() + 2 * H, 2 * H
    Wo = torch.randn(H, H, device=DEV).to(t)
    assert L().etp_attn_bwd_proj(ctypes.byref(bd), Wo.data_ptr(), H, stream()) != 0


@pytest.mark.parametrize("with_dist", [False, True])
@pytest.mark.parametrize("Lx", [80, 36, 16, 9, 128, 113, 1])
def test_self_attention_fwd_with_fused_qkv_projection(Lx, with_dist):
    """etp_attn_fwd_qkv (round 6; attn_rows.hip QKV): the (batch, head) workgroup projects its own Q / K / V rows from the block's
    input.  Checked against the pair it replaces -- etp_gemm (BertSelfAttention.query / key / value as one [3H, H] product,
    vilmodel_cmt.py:108-110) + etp_attn_fwd -- on the same operands: the stash must hold the GEMM's bf16 values up to the order of
    the fp32 sums, ctx / lse follow; and against torch in fp32."""
    torch.manual_seed(13)
    t = torch.bfloat16
    B, nh, dh = 3, 12, 64
    H = nh * dh
    ldS = (Lx + 7) // 8 * 8
    x = torch.randn(B * Lx, H, device=DEV).to(t)
    W = (torch.randn(3 * H, H, device=DEV) / math.sqrt(H)).to(t)
    bias = torch.randn(3 * H, device=DEV) * 0.1
    km = torch.rand(B, Lx, device=DEV) > 0.2
    km[:, 0] = True
    dist = torch.rand(B, Lx, Lx, device=DEV)
    w = torch.tensor([0.3], device=DEV); b0 = torch.tensor([0.1], device=DEV)

    def run(fused):
        qkv = torch.full((B * Lx, 3 * H), float("nan"), device=DEV, dtype=t)
        if not fused:
            run_gemm(x, W, qkv, B * Lx, 3 * H, H, 0, 0, _lib.ETP_BF16, bias=bias)
        P = torch.zeros(B, nh, Lx, ldS, device=DEV, dtype=t)
        ctx = torch.full((B * Lx, H), float("nan"), device=DEV, dtype=t)
        d = AttnDesc()
        d.dtype, d.B, d.heads, d.Lq, d.Lk, d.ldS = _lib.ETP_BF16, B, nh, Lx, Lx, ldS
        d.Q, d.ldq = qkv.data_ptr(), 3 * H
        d.K, d.ldk = qkv.data_ptr() + 2 * H, 3 * H
        d.V, d.ldv = qkv.data_ptr() + 4 * H, 3 * H
        d.P, d.ctx, d.ldc = P.data_ptr(), ctx.data_ptr(), H
        d.keymask, d.mask_mode = km.data_ptr(), 0
        if with_dist:
            d.dist, d.sp_w, d.sp_b = dist.data_ptr(), w.data_ptr(), b0.data_ptr()
        d.alpha = 0.125
        if fused:
            check(L().etp_attn_fwd_qkv(ctypes.byref(d), x.data_ptr(), H, W.data_ptr(), H, bias.data_ptr(), stream()), "attn_fwd_qkv")
        else:
            check(L().etp_attn_fwd(ctypes.byref(d), stream()), "attn_fwd")
        torch.cuda.synchronize()
        lse = P.view(torch.uint8).view(-1)[: B * nh * Lx * 4].view(torch.float32).clone()
        return qkv.float(), ctx.float(), lse

    qkv_f, ctx_f, lse_f = run(True)
    qkv_u, ctx_u, lse_u = run(False)
    assert torch.isfinite(qkv_f).all() and torch.isfinite(ctx_f).all()
    ref = x.float() @ W.float().t() + bias
    assert (qkv_f - ref).abs().max().item() <= 3e-2
    assert (qkv_f - qkv_u).abs().max().item() <= 2e-2 * (qkv_u.abs().max().item())       # a bf16 ulp where the fp32 sums differ
    assert (qkv_f != qkv_u).float().mean().item() <= 0.02
    assert (ctx_f - ctx_u).abs().max().item() <= 3e-2
    assert (lse_f - lse_u).abs().max().item() <= 3e-2
    qh = ref[:, :H].to(t).float().view(B, Lx, nh, dh).permute(0, 2, 1, 3)
    kh = ref[:, H:2 * H].to(t).float().view(B, Lx, nh, dh).permute(0, 2, 1, 3)
    vh = ref[:, 2 * H:].to(t).float().view(B, Lx, nh, dh).permute(0, 2, 1, 3)
    s = qh @ kh.transpose(-1, -2) / 8.0 + (1.0 - km.float())[:, None, None, :] * -10000.0
    if with_dist:
        s = s + (w * dist + b0)[:, None]
    ctx_ref = (torch.softmax(s, -1) @ vh).permute(0, 2, 1, 3).reshape(B * Lx, H)
    assert (ctx_f - ctx_ref).abs().max().item() <= tol(_lib.ETP_BF16, 2)
