"""Would two half-batch text-encoder chains on two streams beat one full-batch chain?  (round-4 experiment)

Times etp_txt_fwd (+ etp_txt_bwd with the weight gradients inline) for B = 32 on one stream against two B = 16 calls on two
streams issued back to back, same model, bf16, train-mode dropout off (timing only)."""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from etpnav_amd import _lib  # noqa: E402
from etpnav_amd._lib import check, ptr  # noqa: E402
from etpnav_amd.planner import GlocalTextPathNavCMT, default_config  # noqa: E402


def main():
    cfg = default_config("r2r", image_feat_size=768)
    model = GlocalTextPathNavCMT(cfg, dtype=torch.bfloat16, device="cuda:0")
    model.init_weights(seed=0)
    eng = model._engine
    eng.require_gpu()
    eng.refresh_weights(force=True)
    L, h, dev = eng.L, eng.handle, eng.device
    Lt, H = 80, 768
    check(L.etp_planner_set_aux_stream(h, None), "aux")
    check(L.etp_planner_set_aux2_stream(h, None), "aux2")
    check(L.etp_planner_set_lazy_join(h, 0), "lazy")

    def mk(B):
        ids = torch.randint(1000, 20000, (B, Lt), device=dev)
        mask = torch.ones(B, Lt, dtype=torch.bool, device=dev)
        out = torch.empty(B, Lt, H, device=dev)
        dout = torch.randn(B, Lt, H, device=dev) * 1e-3
        st = eng.buf(L.etp_txt_stash_bytes(h, B, Lt))
        ws = eng.buf(L.etp_txt_ws_bytes(h, B, Lt))
        return dict(B=B, ids=ids, mask=mask, out=out, dout=dout, st=st, ws=ws)

    full, ha, hb = mk(32), mk(16), mk(16)
    s1 = torch.cuda.Stream(); s2 = torch.cuda.Stream()

    def fwd(x, s):
        check(L.etp_txt_fwd(h, ptr(x["ids"]), ptr(x["mask"]), x["B"], Lt, ptr(x["out"]), ptr(x["st"]), s.cuda_stream), "txt_fwd")

    def bwd(x, s):
        check(L.etp_txt_bwd(h, ptr(x["dout"]), ptr(x["ids"]), ptr(x["mask"]), x["B"], Lt, ptr(x["st"]), ptr(x["ws"]), s.cuda_stream), "txt_bwd")

    def timeit(fn, n=20):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    res = {}
    res["fwd_full_1stream"] = timeit(lambda: fwd(full, s1))
    res["fwd_half_1stream"] = timeit(lambda: fwd(ha, s1))
    res["fwd_2halves_2streams"] = timeit(lambda: (fwd(ha, s1), fwd(hb, s2)))
    res["fwd_2halves_1stream"] = timeit(lambda: (fwd(ha, s1), fwd(hb, s1)))
    res["fwdbwd_full_1stream"] = timeit(lambda: (fwd(full, s1), bwd(full, s1)))
    res["fwdbwd_half_1stream"] = timeit(lambda: (fwd(ha, s1), bwd(ha, s1)))
    res["fwdbwd_2halves_2streams"] = timeit(lambda: (fwd(ha, s1), bwd(ha, s1), fwd(hb, s2), bwd(hb, s2)))
    for k, v in res.items():
        print(f"{k:28s} {v:8.3f} ms")


if __name__ == "__main__":
    main()
