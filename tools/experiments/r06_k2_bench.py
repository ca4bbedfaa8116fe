"""Round 6: the split-reduction 128x64 class (mm32::tile KS = 2) against the four-wavefront class it replaces on the
one-round N = 768 grids of config 2, isolated launches (back to back on one stream), us per launch and TFLOP/s.
    python tools/experiments/r06_k2_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["GEMM_GROUP_ONLY"] = "1"
import gemm_bench as gb
from etpnav_amd import _lib

shapes = [("fwd_s", 2560, 768, 768), ("fwd_s", 2560, 768, 3072), ("fwd", 2560, 768, 768), ("fwd", 2560, 768, 3072),
          ("dgrad", 2560, 3072, 768), ("dgrad", 2560, 2304, 768), ("dgrad", 2560, 768, 768), ("dgrad_s", 2560, 3072, 768),
          ("dgrad_s", 2560, 2304, 768)]
# gemm_bench's dgrad(M, N, K) computes dX[M, K] = dY[M, N] W[N, K]: reduction length N, output columns K
modes = [("0", "4 waves, ring 3"), ("262", "8 waves, 2 x ring 2"), ("264", "8 waves, 2 x ring 3")]
print(f"{'kind':8} {'rows':>5} {'cols':>5} {'red.':>5} " + " ".join(f"{m[1] + ' us':>24} {'TF':>6}" for m in modes))
for kind, M, N, K in shapes:
    row = f"{kind:8} {M:5d} {(K if kind.startswith('dgrad') else N):5d} {(N if kind.startswith('dgrad') else K):5d} "
    for val, _ in modes:
        _lib.set_option("MM32_K2", val)
        best = min(gb.run(kind, M, N, K, iters=60)[0] for _ in range(3))
        row += f"{best:24.2f} {2.0 * M * N * K / best / 1e6:6.0f} "
    print(row, flush=True)
_lib.set_option("MM32_K2", None)
