"""Attention kernels of BASELINE config 2 in isolation (run on the GPU box):

    python tools/attn_bench.py                      # register-resident kernels (csrc/attn_rows.hip)
    ETP_ATTN_ROWS=0 python tools/attn_bench.py      # the LDS-tile kernels of csrc/attn.hip

For every attention shape of one planner step (text self-attention 80x80 in the fused-QKV layout, panorama 37x37, graph
self-attention with the pairwise-distance bias, graph->text cross attention) it times forward and backward two ways:
  * chained   N launches back to back on one stream over rotating operand sets (what a dependent chain sees once the
              launch overhead is amortised: per-kernel time including the ramp and drain of every launch);
  * isolated  each launch bracketed by its own events after a device sync (adds the fixed cost of a cold launch).
Prints one JSON object {shape: {fwd_us, bwd_us, fwd_isolated_us, bwd_isolated_us, bytes, GBps_fwd, ...}}.
"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from etpnav_amd import _lib
from etpnav_amd._lib import AttnDesc, AttnBwdDesc, check

L = _lib.lib()
dev = "cuda"
NSETS = 4
t = torch.bfloat16


def make(B, nh, Lq, Lk, fused_qkv, with_dist):
    H = nh * 64
    ldS = (Lk + 7) // 8 * 8
    keep = []
    if fused_qkv:                                   # self-attention: [B*L, 3H] rows, as the planner's QKV GEMM writes them
        qkv = torch.randn(B * Lq, 3 * H, device=dev).to(t); keep.append(qkv)
        Q, K, V, ldq, ldk, ldv = qkv.data_ptr(), qkv.data_ptr() + 2 * H, qkv.data_ptr() + 4 * H, 3 * H, 3 * H, 3 * H
        dqkv = torch.empty_like(qkv); keep.append(dqkv)
        dQ, dK, dV, lddq, lddk, lddv = dqkv.data_ptr(), dqkv.data_ptr() + 2 * H, dqkv.data_ptr() + 4 * H, 3 * H, 3 * H, 3 * H
    else:                                           # cross attention: Q [B*Lq, H], K/V [B*Lk, 2H]
        q = torch.randn(B * Lq, H, device=dev).to(t); kv = torch.randn(B * Lk, 2 * H, device=dev).to(t); keep += [q, kv]
        Q, K, V, ldq, ldk, ldv = q.data_ptr(), kv.data_ptr(), kv.data_ptr() + 2 * H, H, 2 * H, 2 * H
        dq = torch.empty_like(q); dkv = torch.empty_like(kv); keep += [dq, dkv]
        dQ, dK, dV, lddq, lddk, lddv = dq.data_ptr(), dkv.data_ptr(), dkv.data_ptr() + 2 * H, H, 2 * H, 2 * H
    P = torch.empty(B, nh, Lq, ldS, device=dev, dtype=t)
    ctx = torch.empty(B * Lq, H, device=dev, dtype=t)
    dctx = torch.randn(B * Lq, H, device=dev).to(t)
    dP = torch.empty_like(P)
    km = torch.ones(B, Lk, device=dev, dtype=torch.bool)
    keep += [P, ctx, dctx, dP, km]
    d = AttnDesc()
    d.dtype, d.B, d.heads, d.Lq, d.Lk, d.ldS = _lib.ETP_BF16, B, nh, Lq, Lk, ldS
    d.Q, d.ldq, d.K, d.ldk, d.V, d.ldv = Q, ldq, K, ldk, V, ldv
    d.P, d.ctx, d.ldc = P.data_ptr(), ctx.data_ptr(), H
    d.keymask, d.mask_mode, d.alpha = km.data_ptr(), 0, 0.125
    bd = AttnBwdDesc()
    if with_dist:
        dist = torch.rand(B, Lq, Lk, device=dev); w = torch.tensor([0.3], device=dev); b0 = torch.tensor([0.1], device=dev)
        dw = torch.zeros(1, device=dev); db = torch.zeros(1, device=dev)
        d.dist, d.sp_w, d.sp_b = dist.data_ptr(), w.data_ptr(), b0.data_ptr()
        bd.d_sp_w, bd.d_sp_b = dw.data_ptr(), db.data_ptr()
        keep += [dist, w, b0, dw, db]
    bd.f = d
    bd.dctx, bd.ldd, bd.dP = dctx.data_ptr(), H, dP.data_ptr()
    bd.dQ, bd.lddq, bd.dK, bd.lddk, bd.dV, bd.lddv = dQ, lddq, dK, lddk, dV, lddv
    return d, bd, keep


def chained(fn, descs, iters=200):
    s = torch.cuda.current_stream().cuda_stream
    for d in descs:
        fn(d, s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(descs[i % len(descs)], s)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def isolated(fn, descs, iters=40):
    s = torch.cuda.current_stream().cuda_stream
    ts = []
    for i in range(iters):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(descs[i % len(descs)], s); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    fwd = lambda d, s: check(L.etp_attn_fwd(ctypes.byref(d[0]), s), "fwd")
    bwd = lambda d, s: check(L.etp_attn_bwd(ctypes.byref(d[1]), s), "bwd")
    shapes = {"text_self_80x80": (32, 12, 80, 80, True, False), "pano_self_37x37": (32, 12, 37, 37, True, False),
              "graph_self_24x24_dist": (32, 12, 24, 24, True, True), "graph_text_cross_24x80": (32, 12, 24, 80, False, False),
              "text_self_128x128": (32, 12, 128, 128, True, False), "graph_self_64x64_dist": (8, 12, 64, 64, True, True),
              "graph_text_cross_64x80": (8, 12, 64, 80, False, False)}
    out = {"rows_kernels": os.environ.get("ETP_ATTN_ROWS", "1") != "0"}
    for name, (B, nh, Lq, Lk, fq, wd) in shapes.items():
        sets = [make(B, nh, Lq, Lk, fq, wd) for _ in range(NSETS)]
        r = {"fwd_us": round(chained(fwd, sets), 2), "bwd_us": round(chained(bwd, sets), 2),
             "fwd_isolated_us": round(isolated(fwd, sets), 2), "bwd_isolated_us": round(isolated(bwd, sets), 2)}
        io_f = B * nh * ((Lq + 2 * Lk) * 64 * 2 + Lq * 64 * 2)                  # q, k, v in; ctx out
        io_b = B * nh * (2 * (Lq + 2 * Lk) * 64 * 2 + Lq * 64 * 2)              # q, k, v, dO in; dq, dk, dv out
        r["fwd_GBps"] = round(io_f / r["fwd_us"] / 1e3, 1); r["bwd_GBps"] = round(io_b / r["bwd_us"] / 1e3, 1)
        r["fwd_TFLOPs"] = round(4.0 * B * nh * Lq * Lk * 64 / r["fwd_us"] / 1e6, 1)
        r["bwd_TFLOPs"] = round(8.0 * B * nh * Lq * Lk * 64 / r["bwd_us"] / 1e6, 1)
        out[name] = r
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
