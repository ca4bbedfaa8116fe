"""round 5: the FFN dgrad product (2560 x 3072 x 768, NN, bf16 out) with each epilogue variant, isolated, rotating operand sets:
what does reading the activation operand Z and the activation arithmetic cost this launch?"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from etpnav_amd import _lib
from etpnav_amd._lib import GemmDesc, check
L = _lib.lib(); s = torch.cuda.current_stream().cuda_stream
M, N, K = 2560, 3072, 768
sets = []
for i in range(6):
    torch.manual_seed(i)
    sets.append(dict(A=torch.randn(M, K, device="cuda").to(torch.bfloat16), B=(torch.randn(K, N, device="cuda") * 0.05).to(torch.bfloat16),
                     C=torch.empty(M, N, device="cuda", dtype=torch.bfloat16), Z=torch.randn(M, N, device="cuda").to(torch.bfloat16),
                     Zh=torch.rand(M, N, device="cuda").half()))


def run(d, act):
    g = GemmDesc()
    g.A, g.B, g.C = d["A"].data_ptr(), d["B"].data_ptr(), d["C"].data_ptr()
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc = M, N, K, K, N, N
    g.trans_a, g.trans_b, g.dtype, g.c_dtype = 0, 1, _lib.ETP_BF16, _lib.ETP_BF16
    g.batch, g.batch_inner, g.ksplit, g.alpha, g.act = 1, 1, 1, 1.0, act
    if act:
        g.Z, g.ldz = (d["Zh"] if act == 6 else d["Z"]).data_ptr(), N
    check(L.etp_gemm(ctypes.byref(g), s), "gemm")


print("library", os.environ.get("ETP_LIB", "default"))
for name, act in (("none", 0), ("relu_bwd (reads Z)", 4), ("gelu_bwd (reads Z, erf)", 3), ("mul_z (reads Z)", 6)):
    try:
        for d in sets:
            run(d, act)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(120):
            run(sets[i % 6], act)
        e1.record(); torch.cuda.synchronize()
        print(f"{name:26s} {e0.elapsed_time(e1) / 120 * 1e3:7.2f} us per launch")
    except Exception as ex:
        print(name, "n/a", type(ex).__name__)
