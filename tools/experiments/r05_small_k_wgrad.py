"""round 5: weight-gradient products with a handful of tokens (the c1 fixture: 9 graph nodes, 17 views, 20 words): TN storage, fp32
out, reduction length K = tokens, against fp64 -- per library (ETP_LIB)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from etpnav_amd import _lib
from etpnav_amd._lib import GemmDesc, check
L = _lib.lib(); s = torch.cuda.current_stream().cuda_stream
print("library", os.environ.get("ETP_LIB", "default"))
for tokens in (9, 17, 20, 5, 33, 64):
    for (n_out, k_in) in ((768, 768), (3072, 768), (768, 3072)):
        torch.manual_seed(tokens + n_out)
        dY = (torch.randn(tokens, n_out, device="cuda") * 0.5).to(torch.bfloat16)
        X = torch.randn(tokens, k_in, device="cuda").to(torch.bfloat16)
        for poison in (0, 1):
            dW = torch.full((n_out, k_in), float("nan"), device="cuda")
            db = torch.zeros(n_out, device="cuda")
            if poison:       # fill the allocator's neighbourhood with large values: out-of-range reads show up
                junk = [torch.full((1 << 16,), 3.0e4, device="cuda", dtype=torch.bfloat16) for _ in range(8)]
            d = GemmDesc()
            d.A, d.B, d.C = dY.data_ptr(), X.data_ptr(), dW.data_ptr()
            d.M, d.N, d.K = n_out, k_in, tokens
            d.lda, d.ldb, d.ldc = n_out, k_in, k_in
            d.trans_a, d.trans_b, d.dtype, d.c_dtype = 1, 1, _lib.ETP_BF16, _lib.ETP_F32
            d.batch, d.batch_inner, d.ksplit, d.alpha = 1, 1, 1, 1.0
            check(L.etp_gemm(ctypes.byref(d), s), "gemm"); torch.cuda.synchronize()
            ref = dY.double().t() @ X.double()
            err = (dW.double() - ref).abs().max().item()
            rel = ((dW.double() - ref).norm() / ref.norm()).item()
            print(f"tokens {tokens:3d} dW[{n_out},{k_in}] poison {poison}: max abs err {err:.3e}, relative L2 {rel:.3e}")
