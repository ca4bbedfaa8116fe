"""Round 6: out-projection dgrad + attention backward as two launches (etp_gemm + etp_attn_bwd) against the fused launch
(etp_attn_bwd_proj), chained on one stream over rotating operand sets, for the attention shapes of config 2."""
import ctypes, json, os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from etpnav_amd import _lib
from etpnav_amd._lib import AttnDesc, AttnBwdDesc, GemmDesc, check

L = _lib.lib(); dev = "cuda"; t = torch.bfloat16; NS = 4
st = lambda: torch.cuda.current_stream().cuda_stream


def make(B, nh, Lq, Lk, selfatt):
    H = nh * 64; ldS = (Lk + 7) // 8 * 8; keep = []
    if selfatt:
        qkv = torch.randn(B * Lq, 3 * H, device=dev).to(t); dqkv = torch.empty_like(qkv); keep += [qkv, dqkv]
        Q, K, V, ldq, ldk, ldv = qkv.data_ptr(), qkv.data_ptr() + 2 * H, qkv.data_ptr() + 4 * H, 3 * H, 3 * H, 3 * H
        dQ, dK, dV = dqkv.data_ptr(), dqkv.data_ptr() + 2 * H, dqkv.data_ptr() + 4 * H
    else:
        q = torch.randn(B * Lq, H, device=dev).to(t); kv = torch.randn(B * Lk, 2 * H, device=dev).to(t)
        dq = torch.empty_like(q); dkv = torch.empty_like(kv); keep += [q, kv, dq, dkv]
        Q, K, V, ldq, ldk, ldv = q.data_ptr(), kv.data_ptr(), kv.data_ptr() + 2 * H, H, 2 * H, 2 * H
        dQ, dK, dV = dq.data_ptr(), dkv.data_ptr(), dkv.data_ptr() + 2 * H
    P = torch.empty(B, nh, Lq, ldS, device=dev, dtype=t); ctx = torch.empty(B * Lq, H, device=dev, dtype=t)
    dy = (torch.randn(B * Lq, H, device=dev) * 0.5).to(t); dctx = torch.empty_like(dy); dP = torch.empty_like(P)
    Wo = (torch.randn(H, H, device=dev) / math.sqrt(H)).to(t)
    km = torch.ones(B, Lk, device=dev, dtype=torch.bool); keep += [P, ctx, dy, dctx, dP, Wo, km]
    d = AttnDesc()
    d.dtype, d.B, d.heads, d.Lq, d.Lk, d.ldS = _lib.ETP_BF16, B, nh, Lq, Lk, ldS
    d.Q, d.ldq, d.K, d.ldk, d.V, d.ldv = Q, ldq, K, ldk, V, ldv
    d.P, d.ctx, d.ldc, d.keymask, d.mask_mode, d.alpha = P.data_ptr(), ctx.data_ptr(), H, km.data_ptr(), 0, 0.125
    check(L.etp_attn_fwd(ctypes.byref(d), st()), "fwd")
    bu = AttnBwdDesc(); bu.f = d
    bu.dctx, bu.ldd, bu.dP = dctx.data_ptr(), H, dP.data_ptr()
    bu.dQ, bu.lddq, bu.dK, bu.lddk, bu.dV, bu.lddv = dQ, ldq, dK, ldk, dV, ldv
    bf = AttnBwdDesc(); bf.f = d
    bf.dctx, bf.ldd, bf.dP = dy.data_ptr(), H, dP.data_ptr()
    bf.dQ, bf.lddq, bf.dK, bf.lddk, bf.dV, bf.lddv = dQ, ldq, dK, ldk, dV, ldv
    g = GemmDesc()
    g.A, g.B, g.C, g.M, g.N, g.K, g.lda, g.ldb, g.ldc = dy.data_ptr(), Wo.data_ptr(), dctx.data_ptr(), B * Lq, H, H, H, H, H
    g.trans_a, g.trans_b, g.dtype, g.c_dtype, g.batch, g.batch_inner, g.ksplit, g.alpha = 0, 1, _lib.ETP_BF16, _lib.ETP_BF16, 1, 1, 1, 1.0
    return dict(bu=bu, bf=bf, g=g, Wo=Wo, keep=keep)


def timeit(fn, n=60):
    for _ in range(10): fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


out = {}
for name, (B, nh, Lq, Lk, selfatt) in {"text 80x80": (32, 12, 80, 80, True), "pano 36x36": (32, 12, 36, 36, True),
                                        "graph self 16x16": (32, 12, 16, 16, True), "graph->text 16x80": (32, 12, 16, 80, False), "text 80x80 B=8": (8, 12, 80, 80, True), "text 80x80 B=16": (16, 12, 80, 80, True), "pano 36x36 B=8": (8, 12, 36, 36, True),
                                        "c5 graph self 64x64": (8, 12, 64, 64, True), "c5 graph->text 64x80": (8, 12, 64, 80, False)}.items():
    sets = [make(B, nh, Lq, Lk, selfatt) for _ in range(NS)]
    def unf(i):
        s = sets[i % NS]
        check(L.etp_gemm(ctypes.byref(s["g"]), st()), "gemm"); check(L.etp_attn_bwd(ctypes.byref(s["bu"]), st()), "bwd")
    def gemm_only(i):
        s = sets[i % NS]; check(L.etp_gemm(ctypes.byref(s["g"]), st()), "gemm")
    def bwd_only(i):
        s = sets[i % NS]; check(L.etp_attn_bwd(ctypes.byref(s["bu"]), st()), "bwd")
    def fus(i):
        s = sets[i % NS]; check(L.etp_attn_bwd_proj(ctypes.byref(s["bf"]), s["Wo"].data_ptr(), nh * 64, st()), "proj")
    r = dict(gemm_us=timeit(gemm_only), attn_bwd_us=timeit(bwd_only), pair_us=timeit(unf), fused_us=timeit(fus))
    out[name] = {k: round(v, 2) for k, v in r.items()}
    print(name, out[name], flush=True)
print(json.dumps(out))
