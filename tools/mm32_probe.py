"""Isolated launch times of the text-layer GEMM shapes (cold operands: 6 rotating sets, as tools/gemm_sweep.py) for the
library selected by ETP_LIB -- used to compare the mm32 family's full / DMA-only / MFMA-only measurement builds.

    ETP_LIB=etpnav_amd/lib_mm32_dmaonly.so python tools/mm32_probe.py > out.json
"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from etpnav_amd import _lib
from etpnav_amd._lib import GemmDesc, check
from tools.gemm_sweep import time_variant, L, dev, BF, F32


def group_time(Mt=2560, iters=12, nsets=3):
    H, I = 768, 3072
    specs = [(3 * H, H), (H, H), (I, H), (H, I)]
    sets = []
    for _ in range(nsets):
        keep, descs = [], (GemmDesc * 4)()
        for i, (n, k) in enumerate(specs):
            dY = (torch.randn(Mt, n, device=dev) * 0.5).to(torch.bfloat16)
            X = torch.randn(Mt, k, device=dev).to(torch.bfloat16)
            W = torch.empty(n, k, device=dev)
            db = torch.zeros(n, device=dev)
            d = descs[i]
            d.A, d.B, d.C = dY.data_ptr(), X.data_ptr(), W.data_ptr()
            d.M, d.N, d.K = n, k, Mt
            d.lda, d.ldb, d.ldc = n, k, k
            d.trans_a, d.trans_b, d.dtype, d.c_dtype = 1, 1, BF, F32
            d.batch, d.batch_inner, d.ksplit, d.alpha, d.out_mode = 1, 1, 1, 1.0, 0
            d.a_colsum = db.data_ptr()
            keep += [dY, X, W, db]
        sets.append((descs, keep))
    s = torch.cuda.current_stream().cuda_stream
    for d, _ in sets:
        check(L.etp_gemm_group(d, 4, s))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        check(L.etp_gemm_group(sets[i % nsets][0], 4, s))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    H, I = 768, 3072
    out = {}
    for M in (2560, 8192):
        for kind, N, K in (("fwd", 3 * H, H), ("fwd_g", I, H), ("dg_g", I, H), ("fwd_s", H, H), ("fwd_s", H, I), ("dg_s", H, I),
                           ("dg_s", H, 3 * H), ("dg", H, H)):
            out[f"{kind}:{M}x{N}x{K}"] = round(time_variant(kind, M, N, K, "auto"), 2)
    out["wgrad_group_text_layer"] = round(group_time(), 2)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
