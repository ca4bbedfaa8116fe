"""Harness of the round-5 prototype tools/experiments/r05_ln_prologue_gemm.hip (LayerNorm in the consumer GEMM's prologue, node-side
shapes).  NOT part of the product; the prototype has been compiled but never run.  On a GPU box:

    python tools/experiments/r05_ln_prologue_test.py            # builds the prototype, checks it, times it

Check: against the product's own two launches (etp_ln_stream_fwd, then etp_gemm on the bf16 copy) -- y, the bf16 copy and
(mean, rstd) must be bit-identical (same arithmetic in the same order), C within bf16 rounding of the product's (expected
bit-identical: both accumulate k in slab order through v_mfma_f32_16x16x32_bf16).
Timing: ln + gemm as two launches against the fused launch, rotating operand sets (L2-cold, Infinity-Cache-warm), for the node-side
products of one x-layer at M = 512: N = 768 (Q projection), 2304 (QKV), 3072 (FFN-up, GELU + saved pre-activation).
"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from etpnav_amd import _lib
from etpnav_amd._lib import GemmDesc, check

SRC = os.path.join(ROOT, "tools", "experiments", "r05_ln_prologue_gemm.hip")
OUT = os.path.join(ROOT, "etpnav_amd", "build", "libr05_ln_gemm.so")


def build():
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < os.path.getmtime(SRC):
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result",
               "-munsafe-fp-atomics", "-Wno-return-type-c-linkage", "-shared", "-o", OUT, SRC]
        subprocess.check_call(cmd)
    P = ctypes.CDLL(OUT)
    P.r05_ln_gemm.restype = ctypes.c_int
    P.r05_ln_gemm.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_int,
                              ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float,
                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_long,
                              ctypes.c_int, ctypes.c_void_p]
    P.r05_ln_bwd_gemm.restype = ctypes.c_int
    P.r05_ln_bwd_gemm.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p,
                                  ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                  ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p]
    return P


def check_backward(L, P, s):
    """dX = LN_bwd(dY, S) . W (NN storage) against etp_ln_stream_bwd_stage1 + etp_ln_part_reduce + etp_gemm: dx / the bf16 copy
    bit-identical, dgamma / dbeta within fp32 summation-order noise (the slabs group the rows differently), dX within bf16 rounding."""
    H = 768
    for M in (512, 500):
        for N, act in ((768, _lib.ACT_NONE), (3072, _lib.ACT_GELU_BWD)):
            for bm in (32, 64):
                g = torch.Generator(device="cuda").manual_seed(M + N + bm)
                S = torch.randn(M, H, device="cuda", generator=g) * 1.5 + 0.3
                dy = torch.randn(M, H, device="cuda", generator=g) * 0.2
                gamma = 1.0 + 0.1 * torch.randn(H, device="cuda", generator=g)
                W = (torch.randn(H, N, device="cuda", generator=g) * 0.05).to(torch.bfloat16)       # [reduction][N]
                Z = (torch.randn(M, N, device="cuda", generator=g)).to(torch.bfloat16) if act == _lib.ACT_GELU_BWD else None
                mean = S.mean(1); rstd = (S.var(1, unbiased=False) + 1e-12).rsqrt()
                stats = torch.stack([mean, rstd], 1).contiguous()
                dx_r, dx_n = torch.empty(M, H, device="cuda"), torch.empty(M, H, device="cuda")
                dxt_r, dxt_n = torch.empty(M, H, device="cuda", dtype=torch.bfloat16), torch.empty(M, H, device="cuda", dtype=torch.bfloat16)
                dg, db = torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda")
                part = torch.empty(int(L.etp_ln_bwd_part_bytes(M, H)) // 4, device="cuda")
                check(L.etp_ln_stream_bwd_stage1(_lib.ETP_BF16, dy.data_ptr(), S.data_ptr(), stats.data_ptr(), gamma.data_ptr(), None,
                                                 dx_r.data_ptr(), dxt_r.data_ptr(), dg.data_ptr(), db.data_ptr(), part.data_ptr(), M, H, s), "ln_bwd")
                check(L.etp_ln_part_reduce(part.data_ptr(), M, H, dg.data_ptr(), db.data_ptr(), s), "ln_part_reduce")
                C_r = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); C_n = torch.full_like(C_r, float("nan"))
                d = GemmDesc()
                d.A, d.B, d.C = dxt_r.data_ptr(), W.data_ptr(), C_r.data_ptr()
                d.M, d.N, d.K, d.lda, d.ldb, d.ldc = M, N, H, H, N, N
                d.trans_a, d.trans_b, d.dtype, d.c_dtype = 0, 1, _lib.ETP_BF16, _lib.ETP_BF16
                d.batch, d.batch_inner, d.ksplit, d.alpha, d.act = 1, 1, 1, 1.0, act
                if Z is not None:
                    d.Z, d.ldz = Z.data_ptr(), N
                check(L.etp_gemm(ctypes.byref(d), s), "gemm")
                nblk = (M + bm - 1) // bm
                slabs = torch.empty(nblk, 2, H, device="cuda")
                rc = P.r05_ln_bwd_gemm(dy.data_ptr(), S.data_ptr(), stats.data_ptr(), gamma.data_ptr(), None, W.data_ptr(), N, C_n.data_ptr(), N, 0,
                                       None, 0, M, N, dx_n.data_ptr(), dxt_n.data_ptr(), slabs.data_ptr(), act,
                                       Z.data_ptr() if Z is not None else None, N, bm, s)
                assert rc == 0, rc
                torch.cuda.synchronize()
                ok = torch.equal(dx_r, dx_n) and torch.equal(dxt_r, dxt_n)
                eg = (slabs[:, 0].sum(0) - dg).abs().max().item() / max(dg.abs().max().item(), 1e-6)
                eb = (slabs[:, 1].sum(0) - db).abs().max().item() / max(db.abs().max().item(), 1e-6)
                dc = (C_r.float() - C_n.float()).abs().max().item()
                print(f"bwd M={M:4d} N={N:4d} bm={bm}: dx / copy bit-identical {ok}, dgamma rel {eg:.2e}, dbeta rel {eb:.2e}, max |dC| {dc:.3e}")
                assert ok and eg < 1e-4 and eb < 1e-4 and dc <= 6e-2


def make(M, N, act, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    H = 768
    d = dict(S=torch.randn(M, H, device="cuda", generator=g) * 1.5 + 0.3,
             W=(torch.randn(N, H, device="cuda", generator=g) * 0.05).to(torch.bfloat16),
             bias=torch.randn(N, device="cuda", generator=g) * 0.1,
             gamma=1.0 + 0.1 * torch.randn(H, device="cuda", generator=g), beta=0.1 * torch.randn(H, device="cuda", generator=g))
    for tag in ("ref", "new"):
        d["y_" + tag] = torch.full((M, H), float("nan"), device="cuda")
        d["yt_" + tag] = torch.full((M, H), float("nan"), device="cuda", dtype=torch.bfloat16)
        d["st_" + tag] = torch.full((M, 2), float("nan"), device="cuda")
        d["C_" + tag] = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        d["Z_" + tag] = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16) if act == _lib.ACT_GELU else None
    return d


def run_ref(L, d, M, N, act, s):
    check(L.etp_ln_stream_fwd(_lib.ETP_BF16, d["S"].data_ptr(), d["gamma"].data_ptr(), d["beta"].data_ptr(), d["y_ref"].data_ptr(),
                              d["yt_ref"].data_ptr(), d["st_ref"].data_ptr(), M, 768, ctypes.c_float(1e-12), s), "ln")
    g = GemmDesc()
    g.A, g.B, g.C = d["yt_ref"].data_ptr(), d["W"].data_ptr(), d["C_ref"].data_ptr()
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc = M, N, 768, 768, 768, N
    g.trans_a, g.trans_b, g.dtype, g.c_dtype = 0, 0, _lib.ETP_BF16, _lib.ETP_BF16
    g.batch, g.batch_inner, g.ksplit, g.alpha = 1, 1, 1, 1.0
    g.bias, g.act = d["bias"].data_ptr(), act
    if d["Z_ref"] is not None:
        g.Z, g.ldz = d["Z_ref"].data_ptr(), N
    check(L.etp_gemm(ctypes.byref(g), s), "gemm")


def run_new(P, d, M, N, act, bm, s):
    z = d["Z_new"]
    rc = P.r05_ln_gemm(d["S"].data_ptr(), 768, d["W"].data_ptr(), 768, d["C_new"].data_ptr(), N, 0, M, N, d["bias"].data_ptr(),
                       d["gamma"].data_ptr(), d["beta"].data_ptr(), 1e-12, d["y_new"].data_ptr(), d["yt_new"].data_ptr(),
                       d["st_new"].data_ptr(), act, z.data_ptr() if z is not None else None, N, bm, s)
    assert rc == 0, rc


def main():
    L, P = _lib.lib(), build()
    s = torch.cuda.current_stream().cuda_stream
    for M in (512, 500, 64):
        for N, act in ((768, _lib.ACT_NONE), (2304, _lib.ACT_NONE), (3072, _lib.ACT_GELU)):
            for bm in (32, 64):
                d = make(M, N, act, 7 * M + N + bm)
                run_ref(L, d, M, N, act, s); run_new(P, d, M, N, act, bm, s)
                torch.cuda.synchronize()
                ok_y = torch.equal(d["y_ref"], d["y_new"]) and torch.equal(d["yt_ref"], d["yt_new"]) and torch.equal(d["st_ref"], d["st_new"])
                dc = (d["C_ref"].float() - d["C_new"].float()).abs().max().item()
                dz = (d["Z_ref"].float() - d["Z_new"].float()).abs().max().item() if act == _lib.ACT_GELU else 0.0
                print(f"M={M:4d} N={N:4d} bm={bm}: LN outputs bit-identical {ok_y}, max |dC| {dc:.3e}, max |dZ| {dz:.3e}")
                assert ok_y and dc <= 6e-2 and dz <= 6e-2
    check_backward(L, P, s)
    # timing at M = 512
    M, nsets, iters = 512, 6, 60
    for N, act in ((768, _lib.ACT_NONE), (2304, _lib.ACT_NONE), (3072, _lib.ACT_GELU)):
        sets = [make(M, N, act, 100 + i) for i in range(nsets)]
        res = {}
        for name, fn in (("ln + gemm", lambda d: run_ref(L, d, M, N, act, s)), ("fused bm32", lambda d: run_new(P, d, M, N, act, 32, s)),
                         ("fused bm64", lambda d: run_new(P, d, M, N, act, 64, s))):
            for d in sets:
                fn(d)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(iters):
                fn(sets[i % nsets])
            e1.record(); torch.cuda.synchronize()
            res[name] = e0.elapsed_time(e1) / iters * 1e3
        print(f"M=512 N={N}: " + ", ".join(f"{k} {v:.2f} us" for k, v in res.items()))


if __name__ == "__main__":
    main()
