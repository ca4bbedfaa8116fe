"""Cold-operand sweep of the planner's chain GEMM shapes over tile/stage variants (run on the GPU box).

    python tools/gemm_sweep.py > gpurun_out/gemm_sweep.json

Every variant runs the same launch sequence over NSETS rotating operand sets (~80 MB in total: L2-cold, Infinity-Cache
warm -- what a real step sees: activations were just written by another kernel, weights come from HBM/MALL), timed with
HIP events over the whole sequence.  Output: JSON {shape key: {variant: us}} + the best variant per shape, which
csrc/gemm.hip's dispatch table (launch_tiles) is written from.  Also a K sweep at fixed M, N: the intercept at K -> 0 is
the fixed cost of a launch (dispatch + prologue latency + epilogue), the slope the streaming rate.
"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from etpnav_amd import _lib
from etpnav_amd._lib import GemmDesc, check

L = _lib.lib()
dev = "cuda"
BF, F32 = _lib.ETP_BF16, _lib.ETP_F32
NSETS = 6


def make(kind, M, N, K):
    """kind: fwd (NT, bf16 out, bias) | fwd_s (NT, f32 out, bias + residual) | fwd_g (NT, bf16, GELU + saved z) |
    dg (NN, bf16 out) | dg_s (NN, f32 out + residual) | dg_g (NN, bf16 out, GELU backward reads z)"""
    t = torch.bfloat16
    d = GemmDesc()
    keep = []
    if kind.startswith("fwd"):
        A = torch.randn(M, K, device=dev).to(t); B = (torch.randn(N, K, device=dev) * 0.05).to(t)
        d.trans_a, d.trans_b, d.lda, d.ldb = 0, 0, K, K
        bias = torch.randn(N, device=dev); d.bias = bias.data_ptr(); keep.append(bias)
    else:
        A = torch.randn(M, K, device=dev).to(t); B = (torch.randn(K, N, device=dev) * 0.05).to(t)      # W stored [K (reduction)][N]
        d.trans_a, d.trans_b, d.lda, d.ldb = 0, 1, K, N
    f32out = kind.endswith("_s")
    C = torch.empty(M, N, device=dev, dtype=torch.float32 if f32out else t)
    d.c_dtype = F32 if f32out else BF
    if f32out:
        R = torch.randn(M, N, device=dev); d.R = R.data_ptr(); d.ldr = N; keep.append(R)
    if kind.endswith("_g"):
        Z = torch.randn(M, N, device=dev).to(t); d.Z = Z.data_ptr(); d.ldz = N; keep.append(Z)
        d.act = _lib.ACT_GELU if kind == "fwd_g" else _lib.ACT_GELU_BWD
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), C.data_ptr()
    d.M, d.N, d.K, d.ldc = M, N, K, N
    d.dtype, d.batch, d.batch_inner, d.ksplit, d.alpha = BF, 1, 1, 1, 1.0
    keep += [A, B, C]
    return d, keep


def time_variant(kind, M, N, K, variant, iters=24):
    _lib.force_gemm_tile(variant)            # forcing a gemm.hip class also switches the mm32 family off
    sets = [make(kind, M, N, K) for _ in range(NSETS)]
    s = torch.cuda.current_stream().cuda_stream
    for d, _ in sets:
        check(L.etp_gemm(ctypes.byref(d), s))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        check(L.etp_gemm(ctypes.byref(sets[i % NSETS][0]), s))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def ksweep_cold_vs_warm():
    """Fixed cost of a launch with cold (6 rotating operand sets) vs warm (one set, L2/MALL-resident) operands."""
    global NSETS
    out = {}
    for nsets in (6, 1):
        NSETS = nsets
        for M in (2560, 512):
            for K in (128, 768, 3072):
                out[f"nsets{nsets}:fwd_s:{M}x768x{K}"] = round(time_variant("fwd_s", M, 768, K, "64s3"), 2)
                out[f"nsets{nsets}:fwd:{M}x768x{K}"] = round(time_variant("fwd", M, 768, K, "64s3"), 2)
    _lib.force_gemm_tile("")
    print(json.dumps(out, indent=1))


def main():
    if os.environ.get("KSWEEP_ONLY"):
        return ksweep_cold_vs_warm()
    H, I = 768, 3072
    shapes = []
    for M in (2560, 1152, 512):                    # text / panorama / node tokens of config 2
        shapes += [("fwd", M, 3 * H, H), ("fwd_s", M, H, H), ("fwd_g", M, I, H), ("fwd_s", M, H, I),
                   ("dg_g", M, I, H), ("dg_s", M, H, I), ("dg", M, H, H), ("dg_s", M, H, 3 * H)]
    shapes += [("fwd", 2560, 2 * H, H), ("dg_s", 2560, H, 2 * H), ("fwd", 512, H, H), ("dg_s", 512, H, H),
               ("fwd", 8192, 3 * H, H), ("fwd_s", 8192, H, H), ("fwd_g", 8192, I, H), ("fwd_s", 8192, H, I)]
    variants = ["64s3", "64s4", "ws3", "ws4", "128s2", "128s3"]
    if os.environ.get("SWEEP_256"):                # round 3: the eight-wavefront 256x128 class beside the four-wavefront ones
        variants = ["auto", "64s4", "ws4", "128s2", "256s2", "256s3"]
    out = {"variants": variants, "shapes": {}, "ksweep": {}}
    for kind, M, N, K in shapes:
        key = f"{kind}:{M}x{N}x{K}"
        if key in out["shapes"]:
            continue
        row = {}
        for v in variants:
            if (v.startswith("128") and (M < 128 or N < 128)) or (v.startswith("256") and (M < 256 or N < 128)):
                continue
            row[v] = round(time_variant(kind, M, N, K, v), 2)
        best = min(row, key=row.get)
        fl = 2.0 * M * N * K
        out["shapes"][key] = {"us": row, "best": best, "best_tflops": round(fl / row[best] / 1e6, 1)}
        print(key, row, "->", best, file=sys.stderr, flush=True)
    for K in (128, 256, 512, 768, 1536, 3072):
        kv = ("64s3", "ws3", "128s2") + (("256s2",) if os.environ.get("SWEEP_256") else ())
        out["ksweep"][str(K)] = {v: round(time_variant("fwd_s", 2560, 768, K, v), 2) for v in kv}
        print("ksweep", K, out["ksweep"][str(K)], file=sys.stderr, flush=True)
    _lib.force_gemm_tile("")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
