"""round 5: elementwise error of the GELU epilogues (ETP_ACT_GELU / ETP_ACT_GELU_SAVEGRAD / ETP_ACT_GELU_BWD) of every bf16 tile class
against fp64, for the library selected by ETP_LIB -- the unit tests' bf16 tolerance (0.12) cannot see a few-percent error."""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from etpnav_amd import _lib
from etpnav_amd._lib import GemmDesc, check

L = _lib.lib()
s = torch.cuda.current_stream().cuda_stream


def gemm(A, B, C, M, N, K, act, Z=None, bias=None, tb=0):
    d = GemmDesc()
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), C.data_ptr()
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc = M, N, K, K, (N if tb else K), N
    d.trans_a, d.trans_b, d.dtype, d.c_dtype = 0, tb, _lib.ETP_BF16, _lib.ETP_BF16
    d.batch, d.batch_inner, d.ksplit, d.alpha, d.act = 1, 1, 1, 1.0, act
    if Z is not None:
        d.Z, d.ldz = Z.data_ptr(), N
    if bias is not None:
        d.bias = bias.data_ptr()
    check(L.etp_gemm(ctypes.byref(d), s), "gemm")
    torch.cuda.synchronize()


def gelu(x):
    return x * 0.5 * (1 + torch.erf(x / math.sqrt(2)))


def gelu_grad(x):
    return 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)


print("library", os.environ.get("ETP_LIB", "default"))
for (M, N, K) in [(20, 3072, 768), (9, 3072, 768), (17, 3072, 768), (33, 3072, 768), (64, 3072, 768)]:
    torch.manual_seed(M + N)
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    B = (torch.randn(N, K, device="cuda") * 0.06).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda") * 0.3
    v = (A.double() @ B.double().t()) + bias.double()
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); Z = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    gemm(A, B, C, M, N, K, _lib.ACT_GELU, Z, bias)
    e_g = (C.double() - gelu(v)).abs()
    rel = (e_g / (gelu(v).abs() + 1e-2)).max().item()
    Zh = torch.empty(M, N, device="cuda", dtype=torch.float16)
    msg = ""
    if hasattr(_lib, "ACT_GELU_SAVEGRAD") and os.environ.get("NO_SAVEGRAD") is None:
        try:
            gemm(A, B, C, M, N, K, _lib.ACT_GELU_SAVEGRAD, Zh, bias)
            msg = f" | savegrad: gelu max err {(C.double() - gelu(v)).abs().max().item():.3e}, gelu' max err {(Zh.double() - gelu_grad(v)).abs().max().item():.3e}"
        except Exception as ex:
            msg = f" | savegrad n/a ({type(ex).__name__})"
    Zin = (torch.randn(M, N, device="cuda") * 1.2).to(torch.bfloat16)
    raw = A.double() @ B.double().t()
    gemm(A, B, C, M, N, K, _lib.ACT_GELU_BWD, Zin)
    e_b = (C.double() - raw * gelu_grad(Zin.double())).abs()
    # the planner's dgrad form: NN storage (B = W[N_red][K_out]), MUL_Z with the fp16 derivative
    Bn = B.t().contiguous()          # [K, N] -> reduce over K... use as NN: C[M, N] = A[M, K] . Bn[K, N]
    Zh2 = (torch.rand(M, N, device="cuda") * 1.2 - 0.1).half()
    try:
        gemm(A, Bn, C, M, N, K, _lib.ACT_MUL_Z, Zh2, tb=1)
        msg2 = f" | NN mul_z max abs err {(C.double() - raw * Zh2.double()).abs().max().item():.3e}"
    except Exception as ex:
        msg2 = f" | NN mul_z n/a"
    print(f"{M}x{N}x{K}: gelu max abs err {e_g.max().item():.3e} (max rel {rel:.3e}), z err {(Z.double() - v).abs().max().item():.3e}; gelu_bwd max abs err {e_b.max().item():.3e} "
          f"(|ref| max {(raw * gelu_grad(Zin.double())).abs().max().item():.2f}){msg}{msg2}")
