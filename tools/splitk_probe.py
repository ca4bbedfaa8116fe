"""Would split-K on 128x128 tiles beat the 64x64 tiles of the N = 768 chain products?  (run on the GPU box)

The N = 768, K = 2304 / 3072 products (FFN-down forward, the dgrads into the residual stream) are feed-bound on 64x64 tiles
(16 KB of LDS-DMA per 0.5 MFLOP); 128x128 tiles halve the bytes per FLOP but give only 120 tiles, so the reduction is split
2-4 ways with fp32 atomic accumulation (out_mode 2; C pre-zeroed by the caller).  Timed like tools/gemm_sweep.py (rotating
operand sets, HIP events over the sequence).  -> profiles/r03f_splitk_probe.json
"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from etpnav_amd import _lib
from etpnav_amd._lib import check
sys.argv = sys.argv[:1]
import tools.gemm_sweep as gs

L = _lib.lib()


def time_split(kind, M, N, K, tile, ks, iters=24):
    _lib.force_gemm_tile(tile)
    sets = [gs.make(kind, M, N, K) for _ in range(gs.NSETS)]
    for d, _ in sets:
        d.ksplit = ks
        if ks > 1:
            d.out_mode = 2
    s = torch.cuda.current_stream().cuda_stream
    for d, _ in sets:
        check(L.etp_gemm(ctypes.byref(d), s))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        check(L.etp_gemm(ctypes.byref(sets[i % gs.NSETS][0]), s))
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / iters * 1e3, 2)


out = {}
for kind, M, N, K in [("fwd_s", 2560, 768, 3072), ("dg_s", 2560, 768, 3072), ("dg_s", 2560, 768, 2304), ("fwd_s", 1152, 768, 3072),
                      ("fwd_s", 512, 768, 3072), ("fwd_s", 2560, 768, 768)]:
    row = {}
    for tile, ks in [("64s3", 1), ("64s3", 2), ("128s2", 1), ("128s2", 2), ("128s2", 4), ("128s3", 4), ("128s2", 6)]:
        if K // ks % 64 or K // ks < 128:
            continue
        try:
            row[f"{tile},ks{ks}"] = time_split(kind, M, N, K, tile, ks)
        except Exception as e:       # noqa: BLE001
            row[f"{tile},ks{ks}"] = str(e)[:80]
    out[f"{kind}:{M}x{N}x{K}"] = row
    print(kind, M, N, K, row, file=sys.stderr, flush=True)
_lib.force_gemm_tile("")
print(json.dumps(out, indent=1))
