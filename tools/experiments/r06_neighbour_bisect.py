"""Round 6 (VERDICT r5 #1b): ONE bisect on the aggressor of DESIGN.md §3.6.  Victim: the three pano_embed_bwd launches WITHOUT the
CU-exclusive request (ROW_EXCLUSIVE = 0); neighbour stream: the library's bf16 2560 x 3072 x 768 product of the 128x128 class.  Run
once per library build (ETP_LIB = the default library or an experiment build of gemm_mm32.hip / embed.hip, tools/build_variant.sh):
    default                     the shipped kernels
    half_epi  (-DETP_MM32_HALF_EPI)  128x128 epilogue staged in two 64-row halves: 65 536 B of LDS instead of 67 584
    pad_lds   (-DETP_MM32_PAD_LDS)   the 128x128 class asks for 100 KB: ONE aggressor workgroup per CU instead of two
    dma_only  (-DETP_MM32_EXPT=1)    the aggressor's loop issues its LDS-DMA but no fragment reads / MFMAs (wrong product)
    mfma_only (-DETP_MM32_EXPT=2)    fragment reads + MFMAs + barriers, no DMA inside the loop (wrong product)
    noslp     (embed.hip -fno-slp-vectorize)  the VICTIM without packed-fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32:
              the round-5 isolation found the wrong values in elements 0 and 2 of a lane's four -- the low halves of such pairs)
Prints one line: repetitions (of 24) in which da / dd differ bitwise or a gradient sum deviates, per neighbour."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import ctypes
import torch
from etpnav_amd import _lib
import tests.test_neighbours_gpu as nb

victims = [v for v in nb.make_victims() if v.name == "pano_embed_bwd"]
agg = nb.make_aggressors()
keep = {k: v for k, v in agg.items() if k.startswith("mm32 128x128")}
# one more neighbour: this round's eight-wavefront split-reduction 128x64 class on the FFN-down shape
T = torch.bfloat16
A = torch.randn(2560, 3072, device="cuda").to(T); W = (torch.randn(768, 3072, device="cuda") * 0.05).to(T); C = torch.empty(2560, 768, device="cuda", dtype=T)
d = _lib.GemmDesc()
d.A, d.B, d.C = A.data_ptr(), W.data_ptr(), C.data_ptr()
d.M, d.N, d.K, d.lda, d.ldb, d.ldc = 2560, 768, 3072, 3072, 3072, 768
d.trans_a, d.trans_b, d.dtype, d.c_dtype, d.batch, d.batch_inner, d.ksplit, d.alpha = 0, 0, _lib.ETP_BF16, _lib.ETP_BF16, 1, 1, 1, 1.0
keep["mm32 128x64 k2 (8 waves, 2560x768x3072)"] = (lambda s: _lib.check(_lib.lib().etp_gemm(ctypes.byref(d), s), "gemm"), {"MM32": "264"}, (A, W, C, d))
keep["mm32 128x64 (4 waves, 2560x768x3072)"] = (lambda s: _lib.check(_lib.lib().etp_gemm(ctypes.byref(d), s), "gemm"), {"MM32": "64"}, (A, W, C, d))
m = nb.run_matrix(victims, keep, 0)
lib = os.path.basename(os.environ.get("ETP_LIB", "default"))
for v, row in m.items():
    print(f"{lib:28s} " + " | ".join(f"{a}: {r['bad_reps']}/{r['reps']} (rows {r['rows']}, sums {r['worst_sum']:.1e})" for a, r in row.items()), flush=True)
