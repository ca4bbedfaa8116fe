"""How much of a planner step's time is inter-kernel bubbles / underfilled kernels?  Runs two INDEPENDENT half-batch
steps (B=16 each, own planner, own streams) concurrently and compares with one B=32 step.  If 2 x B16 concurrently is
clearly faster than 1 x B32, splitting the batch into independent micro-batch chains pays.   python tools/concurrency_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from etpnav_amd.planner import GlocalTextPathNavCMT, default_config
from etpnav_amd.step import PlannerStep
from etpnav_amd.synthetic import make_batch

cfg = default_config("r2r", image_feat_size=768)


def mk(B, seed):
    m = GlocalTextPathNavCMT(cfg, dtype=torch.bfloat16, device="cuda:0"); m.init_weights(seed=0)
    b = make_batch(cfg.vocab_size, cfg.image_feat_size, cfg.depth_feat_size, B, 80, 36, 16, seed=seed)
    return m, PlannerStep(m, b, dropout="config")


def timeit(steps, streams, iters=40):
    for _ in range(5):
        for st, s in zip(steps, streams):
            with torch.cuda.stream(s):
                st.run_eager()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        for st, s in zip(steps, streams):
            with torch.cuda.stream(s):
                st.run_eager()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


m32, s32 = mk(32, 1)
print("1 x B32            ms/step", round(timeit([s32], [torch.cuda.Stream()]), 3), flush=True)
mA, sA = mk(16, 2)
print("1 x B16            ms/step", round(timeit([sA], [torch.cuda.Stream()]), 3), flush=True)
mB, sB = mk(16, 3)
print("2 x B16 concurrent ms/pair", round(timeit([sA, sB], [torch.cuda.Stream(), torch.cuda.Stream()]), 3), flush=True)
mC, sC = mk(8, 4); mD, sD = mk(8, 5); mE, sE = mk(8, 6); mF, sF = mk(8, 7)
print("1 x B8             ms/step", round(timeit([sC], [torch.cuda.Stream()]), 3), flush=True)
print("4 x B8 concurrent  ms/quad", round(timeit([sC, sD, sE, sF], [torch.cuda.Stream() for _ in range(4)]), 3), flush=True)
