"""Round 6: QKV projection + self-attention forward as two launches (etp_gemm + etp_attn_fwd) against the fused launch
(etp_attn_fwd_qkv), chained on one stream over rotating operand sets."""
import ctypes, json, os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from etpnav_amd import _lib
from etpnav_amd._lib import AttnDesc, GemmDesc, check

L = _lib.lib(); dev = "cuda"; t = torch.bfloat16; NS = 4
st = lambda: torch.cuda.current_stream().cuda_stream


def make(B, nh, Lx):
    H = nh * 64; ldS = (Lx + 7) // 8 * 8
    x = torch.randn(B * Lx, H, device=dev).to(t); W = (torch.randn(3 * H, H, device=dev) / math.sqrt(H)).to(t)
    bias = torch.randn(3 * H, device=dev) * 0.1
    qkv = torch.empty(B * Lx, 3 * H, device=dev, dtype=t)
    P = torch.empty(B, nh, Lx, ldS, device=dev, dtype=t); ctx = torch.empty(B * Lx, H, device=dev, dtype=t)
    km = torch.ones(B, Lx, device=dev, dtype=torch.bool)
    d = AttnDesc()
    d.dtype, d.B, d.heads, d.Lq, d.Lk, d.ldS = _lib.ETP_BF16, B, nh, Lx, Lx, ldS
    d.Q, d.ldq, d.K, d.ldk, d.V, d.ldv = qkv.data_ptr(), 3 * H, qkv.data_ptr() + 2 * H, 3 * H, qkv.data_ptr() + 4 * H, 3 * H
    d.P, d.ctx, d.ldc, d.keymask, d.mask_mode, d.alpha = P.data_ptr(), ctx.data_ptr(), H, km.data_ptr(), 0, 0.125
    g = GemmDesc()
    g.A, g.B, g.C, g.M, g.N, g.K, g.lda, g.ldb, g.ldc = x.data_ptr(), W.data_ptr(), qkv.data_ptr(), B * Lx, 3 * H, H, H, H, 3 * H
    g.trans_a, g.trans_b, g.dtype, g.c_dtype, g.batch, g.batch_inner, g.ksplit, g.alpha = 0, 0, _lib.ETP_BF16, _lib.ETP_BF16, 1, 1, 1, 1.0
    g.bias = bias.data_ptr()
    return dict(d=d, g=g, x=x, W=W, bias=bias, keep=[qkv, P, ctx, km])


def timeit(fn, n=60):
    for _ in range(10): fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


out = {}
for name, (B, nh, Lx) in {"text 80 (B=32)": (32, 12, 80), "pano 36 (B=32)": (32, 12, 36), "graph self 16 (B=32)": (32, 12, 16),
                           "text 80 (B=8)": (8, 12, 80), "pano 36 (B=8)": (8, 12, 36), "graph self 64 (B=8)": (8, 12, 64),
                           "text 80 (B=16)": (16, 12, 80)}.items():
    sets = [make(B, nh, Lx) for _ in range(NS)]
    H = nh * 64
    def pair(i):
        s = sets[i % NS]
        check(L.etp_gemm(ctypes.byref(s["g"]), st()), "gemm"); check(L.etp_attn_fwd(ctypes.byref(s["d"]), st()), "fwd")
    def gemm_only(i):
        s = sets[i % NS]; check(L.etp_gemm(ctypes.byref(s["g"]), st()), "gemm")
    def fwd_only(i):
        s = sets[i % NS]; check(L.etp_attn_fwd(ctypes.byref(s["d"]), st()), "fwd")
    def fus(i):
        s = sets[i % NS]
        check(L.etp_attn_fwd_qkv(ctypes.byref(s["d"]), s["x"].data_ptr(), H, s["W"].data_ptr(), H, s["bias"].data_ptr(), st()), "qkv")
    r = dict(gemm_us=timeit(gemm_only), attn_fwd_us=timeit(fwd_only), pair_us=timeit(pair), fused_us=timeit(fus))
    out[name] = {k: round(v, 2) for k, v in r.items()}
    print(name, out[name], flush=True)
print(json.dumps(out))
