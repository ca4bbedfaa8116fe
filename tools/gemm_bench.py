"""GEMM micro-benchmark on the planner's real shapes (run on the GPU box).  python tools/gemm_bench.py [tile]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from etpnav_amd import _lib
from etpnav_amd._lib import GemmDesc, check

L = _lib.lib()
dev = "cuda"

def run(kind, M, N, K, dtype=_lib.ETP_BF16, iters=30, ksplit=1, batch=1):
    t = torch.bfloat16 if dtype == _lib.ETP_BF16 else torch.float32
    d = GemmDesc()
    if kind == "fwd":      # Y[M,N] = X[M,K] W[N,K]^T + b
        A = torch.randn(M, K, device=dev).to(t); B = torch.randn(N, K, device=dev).to(t); C = torch.empty(M, N, device=dev, dtype=t)
        d.trans_a, d.trans_b, d.c_dtype = 0, 0, dtype; d.lda, d.ldb, d.ldc = K, K, N; d.M, d.N, d.K = M, N, K
        bias = torch.randn(N, device=dev); d.bias = bias.data_ptr()
    elif kind == "fwd_s":  # fp32 stream out + fp32 residual
        A = torch.randn(M, K, device=dev).to(t); B = torch.randn(N, K, device=dev).to(t); C = torch.empty(M, N, device=dev)
        d.trans_a, d.trans_b, d.c_dtype = 0, 0, _lib.ETP_F32; d.lda, d.ldb, d.ldc = K, K, N; d.M, d.N, d.K = M, N, K
        bias = torch.randn(N, device=dev); d.bias = bias.data_ptr(); R = torch.randn(M, N, device=dev); d.R = R.data_ptr(); d.ldr = N
    elif kind == "dgrad_s":
        A = torch.randn(M, N, device=dev).to(t); B = torch.randn(N, K, device=dev).to(t); C = torch.empty(M, K, device=dev)
        d.trans_a, d.trans_b, d.c_dtype = 0, 1, _lib.ETP_F32; d.lda, d.ldb, d.ldc = N, K, K; d.M, d.N, d.K = M, K, N
        R = torch.randn(M, K, device=dev); d.R = R.data_ptr(); d.ldr = K
    elif kind == "dgrad":  # dX[M,K] = dY[M,N] W[N,K]
        A = torch.randn(M, N, device=dev).to(t); B = torch.randn(N, K, device=dev).to(t); C = torch.empty(M, K, device=dev, dtype=t)
        d.trans_a, d.trans_b, d.c_dtype = 0, 1, dtype; d.lda, d.ldb, d.ldc = N, K, K; d.M, d.N, d.K = M, K, N
    else:                  # dW[N,K] += dY[M,N]^T X[M,K]
        A = torch.randn(M, N, device=dev).to(t); B = torch.randn(M, K, device=dev).to(t); C = torch.zeros(N, K, device=dev)
        d.trans_a, d.trans_b, d.c_dtype = 1, 1, _lib.ETP_F32; d.lda, d.ldb, d.ldc = N, K, K; d.M, d.N, d.K = N, K, M
        d.ksplit = ksplit; d.out_mode = 2 if ksplit > 1 else 1
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), C.data_ptr()
    d.dtype = dtype; d.batch, d.batch_inner, d.alpha = 1, 1, 1.0
    if d.ksplit == 0: d.ksplit = 1
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3): check(L.etp_gemm(ctypes.byref(d), s))
    torch.cuda.synchronize()
    if os.environ.get("GEMM_CHECK") and kind in ("fwd", "dgrad"):
        ref = (A.float() @ (B.float().t() if kind == "fwd" else B.float()))
        if kind == "fwd": ref = ref + bias
        err = (C.float() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6)
        assert err < 2e-2, (kind, M, N, K, os.environ.get("ETP_GEMM_TILE"), err)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): check(L.etp_gemm(ctypes.byref(d), s))
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    fl = 2.0 * M * N * K
    return us, fl / us / 1e6

if __name__ == "__main__" and not os.environ.get("GEMM_GROUP_ONLY"):
    shapes = [("fwd_s", 2560, 768, 768), ("fwd_s", 2560, 768, 3072), ("dgrad_s", 2560, 2304, 768), ("dgrad_s", 2560, 3072, 768),
              ("dgrad_s", 512, 3072, 768), ("fwd", 2560, 2304, 768), ("fwd", 2560, 768, 768), ("fwd", 2560, 3072, 768), ("fwd", 2560, 768, 3072),
              ("fwd", 1152, 2304, 768), ("fwd", 512, 768, 768), ("fwd", 512, 3072, 768),
              ("dgrad", 2560, 2304, 768), ("dgrad", 2560, 768, 768), ("dgrad", 2560, 3072, 768), ("dgrad", 2560, 768, 3072),
              ("dgrad", 512, 768, 768),
              ("wgrad", 2560, 2304, 768), ("wgrad", 2560, 768, 768), ("wgrad", 2560, 3072, 768), ("wgrad", 2560, 768, 3072),
              ("wgrad", 512, 768, 768), ("wgrad", 512, 3072, 768)]
    tiles = sys.argv[1:] or ["auto", "128", "64"]
    print(f"{'kind':6} {'M':>5} {'N':>5} {'K':>5} " + " ".join(f"{t+':us':>10} {'TF':>7}" for t in tiles))
    for kind, M, N, K in shapes:
        row = f"{kind:6} {M:5d} {N:5d} {K:5d} "
        for tl in tiles:
            _lib.force_gemm_tile(tl)         # forcing a gemm.hip class also switches the mm32 family off
            if kind == "wgrad":
                best = None
                for ks in (1, 2, 4, 8):
                    us, tf = run(kind, M, N, K, ksplit=ks)
                    if best is None or us < best[0]: best = (us, tf, ks)
                row += f"{best[0]:10.1f} {best[1]:7.1f}(ks{best[2]})"
            else:
                us, tf = run(kind, M, N, K)
                row += f"{us:10.1f} {tf:7.1f} "
        print(row, flush=True)



def run_group(tokens, shapes, tile, iters=20, nsets=4, store=True):
    """One grouped TN launch (etp_gemm_group) over `shapes` = [(N_out, K_in)] with `tokens` rows; nsets rotating operand sets
    keep the operands L2-cold (MALL-warm), like a real backward pass where every dY was just written by another kernel."""
    _lib.set_option("GROUP_TILE", tile)
    t = torch.bfloat16
    sets = []
    for _ in range(nsets):
        descs = (GemmDesc * len(shapes))()
        keep = []
        for i, (n, k) in enumerate(shapes):
            dY = torch.randn(tokens, n, device=dev).to(t); X = torch.randn(tokens, k, device=dev).to(t)
            W = torch.zeros(n, k, device=dev)
            d = GemmDesc()
            d.A, d.B, d.C = dY.data_ptr(), X.data_ptr(), W.data_ptr()
            d.M, d.N, d.K = n, k, tokens
            d.lda, d.ldb, d.ldc = n, k, k
            d.trans_a, d.trans_b, d.dtype, d.c_dtype = 1, 1, _lib.ETP_BF16, _lib.ETP_F32
            d.batch, d.batch_inner, d.ksplit, d.alpha, d.out_mode = 1, 1, 1, 1.0, 0 if store else 1
            descs[i] = d
            keep += [dY, X, W]
        sets.append((descs, keep))
    s = torch.cuda.current_stream().cuda_stream
    for descs, _ in sets: check(L.etp_gemm_group(descs, len(shapes), s))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        descs, _ = sets[i % nsets]
        check(L.etp_gemm_group(descs, len(shapes), s))
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    fl = sum(2.0 * tokens * n * k for n, k in shapes)
    return us, fl / us / 1e6


def group_table():
    layers = {"text layer (2560 tok)": (2560, [(2304, 768), (768, 768), (3072, 768), (768, 3072)]),
              "pano layer (1152 tok)": (1152, [(2304, 768), (768, 768), (3072, 768), (768, 3072)]),
              "x-layer node side (512 tok)": (512, [(768, 768), (768, 768), (2304, 768), (768, 768), (3072, 768), (768, 3072)]),
              "x-layer text K/V (2560 tok)": (2560, [(1536, 768)]),
              "RxR text layer (8192 tok)": (8192, [(2304, 768), (768, 768), (3072, 768), (768, 3072)])}
    tiles = ["256s2", "256s3", "128s2", "128s3", "64s3"]
    print(f"{'grouped weight gradients':34} " + " ".join(f"{t + ':us':>10} {'TF':>7}" for t in tiles))
    for name, (tok, shapes) in layers.items():
        row = f"{name:34} "
        for tl in tiles:
            us, tf = run_group(tok, shapes, tl)
            row += f"{us:10.1f} {tf:7.1f} "
        print(row, flush=True)


if __name__ == "__main__" and os.environ.get("GEMM_GROUP_TABLE"):
    group_table()
