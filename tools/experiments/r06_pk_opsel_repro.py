"""Driver of tools/experiments/r06_pk_opsel_repro.hip (GPU box): builds it with hipcc, runs every packed-multiply form alone and beside
(a) a synthetic MFMA-dense kernel with 4 / 2 / 1 32x32x16 accumulators per wavefront, (b) the library's 128x128-tile GEMM and its 128x64
class, and prints mismatches of the packed result against two scalar multiplies of the same operands: total, per half (low / high), per
quarter of the wavefront.  No planner, no oracle."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from etpnav_amd import _lib

src = os.path.join(ROOT, "tools", "experiments", "r06_pk_opsel_repro.hip")
so = "/tmp/r06_pkrepro.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-fno-slp-vectorize", "-o", so, src])
L = _lib.lib()
R = ctypes.CDLL(so)
R.pk_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint, ctypes.c_void_p]
R.mfma_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = "cuda"
s_n = torch.cuda.Stream(priority=-1); s_v = torch.cuda.Stream(priority=0)
sink = torch.zeros(4, device=dev)
T = torch.bfloat16
A = torch.randn(2560, 768, device=dev).to(T); W = (torch.randn(3072, 768, device=dev) * 0.05).to(T); C = torch.empty(2560, 3072, device=dev, dtype=T)
A2 = torch.randn(2560, 3072, device=dev).to(T); W2 = (torch.randn(768, 3072, device=dev) * 0.05).to(T); C2 = torch.empty(2560, 768, device=dev, dtype=T)


def desc(a, b, c, M, N, K):
    d = _lib.GemmDesc()
    d.A, d.B, d.C = a.data_ptr(), b.data_ptr(), c.data_ptr()
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc = M, N, K, K, K, N
    d.trans_a, d.trans_b, d.dtype, d.c_dtype, d.batch, d.batch_inner, d.ksplit, d.alpha = 0, 0, _lib.ETP_BF16, _lib.ETP_BF16, 1, 1, 1, 1.0
    return d


d128, d64 = desc(A, W, C, 2560, 3072, 768), desc(A2, W2, C2, 2560, 768, 3072)
NEIGH = {
    "alone": None,
    "synthetic MFMA, 4 accumulators (64 registers)": lambda: R.mfma_run(4, sink.data_ptr(), 4000, 512, s_n.cuda_stream),
    "synthetic MFMA, 2 accumulators": lambda: R.mfma_run(2, sink.data_ptr(), 8000, 512, s_n.cuda_stream),
    "synthetic MFMA, 1 accumulator": lambda: R.mfma_run(1, sink.data_ptr(), 16000, 512, s_n.cuda_stream),
    "library GEMM 128x128 (2560x3072x768)": lambda: [_lib.check(L.etp_gemm(ctypes.byref(d128), s_n.cuda_stream), "gemm") for _ in range(24)],
    "library GEMM 128x64 (2560x768x3072)": lambda: [_lib.check(L.etp_gemm(ctypes.byref(d64), s_n.cuda_stream), "gemm") for _ in range(24)],
}
FORMS = ["v_pk_mul_f32 (plain)", "op_sel:[0,1]", "op_sel:[1,0]", "op_sel_hi:[1,0]", "op_sel_hi:[0,1]", "op_sel:[0,1] op_sel_hi:[0,1]",
         "v_pk_add_f32 op_sel:[0,1]", "v_pk_fma_f32 op_sel:[0,1,0]", "v_pk_fma_f32 op_sel:[0,0,1]", "v_pk_mul_f32 op_sel:[1,1]",
         "v_pk_mov_b32 op_sel:[1,0]", "v_pk_mov_b32 op_sel:[0,1]", "v_pk_mov_b32 op_sel:[1,1]"]
if os.environ.get("PK_FORMS"):
    KEEP = [int(x) for x in os.environ["PK_FORMS"].split(",")]
else:
    KEEP = list(range(len(FORMS)))
ITERS, BLOCKS, REPS = 20000, 2048, 6
print(f"# {BLOCKS} workgroups x 256 threads x {ITERS} packed multiplies per launch, {REPS} launches per cell; cell = mismatching results "
      f"(low half / high half), lanes 0-15 / 16-31 / 32-47 / 48-63")
for nname, neigh in NEIGH.items():
    for v, fname in enumerate(FORMS):
        if v not in KEEP:
            continue
        out = torch.zeros(160, dtype=torch.int32, device=dev)
        for rep in range(REPS):
            torch.cuda.synchronize()
            if neigh is not None:
                neigh()
            rc = R.pk_run(v, out.data_ptr(), ITERS, BLOCKS, 1234 + rep, s_v.cuda_stream)
            assert rc == 0, rc
            if neigh is not None:
                neigh()
        torch.cuda.synchronize()
        o = out.cpu().numpy().astype("int64")
        per = o[:128].reshape(64, 2)
        q = [int(per[16 * i:16 * i + 16].sum()) for i in range(4)]
        first = ""
        if o[128]:
            first = f"; first: lane {o[129]} {'low' if o[132] == 0 else 'high'} got {int(o[130]) & 0xffffffff:#010x} want {int(o[131]) & 0xffffffff:#010x}"
        print(f"{nname:48s} | {fname:30s} | {int(per[:, 0].sum()):9d} / {int(per[:, 1].sum()):9d} | quarters {q}{first}", flush=True)
