"""Where a GEMM launch's time goes: per-workgroup timestamps taken INSIDE the LDS-DMA kernels (run on the GPU box).

    python tools/gemm_phase_probe.py [--workload c2] [--mode train] > profiles/r03_gemm_phases.txt

With the probe on (include/etpnav_hip.h: etp_gemm_probe_enable), thread 0 of every workgroup of every eager GEMM launch
records s_memrealtime (100 MHz, chip-wide) at entry / exit and s_memtime (shader clock) at entry, first slab visible, end
of the reduction and end of the epilogue, plus HW_ID / XCC_ID.  One single-stream planner step is probed; per launch class
(kernel, M, N, K) the table gives

    span      last exit - first entry of any workgroup (the kernel's device-side duration, no launch gap)
    skew      entry time of the last-dispatched workgroup relative to the first (dispatch spread)
    pro       entry -> first slab visible (argument loads, address setup, first DMA round trip)
    loop      first slab visible -> end of the reduction
    epi       end of the reduction -> last store issued (LDS staging, epilogue operand reads, stores)
    wg/cu     workgroups that shared the busiest CU
    GHz       shader clock (s_memtime ticks per s_memrealtime tick)

The reference's only instrumentation on this path is host time.time() (pretrain_src/pretrain_src/train_r2r.py:227,299-317).
"""
import argparse, ctypes, os, sys
from collections import OrderedDict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
from etpnav_amd import _lib
from etpnav_amd.planner import GlocalTextPathNavCMT, default_config
from etpnav_amd.step import PlannerStep
from etpnav_amd.synthetic import make_batch
from bench import WORKLOADS

WG_MAX = 4096


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--mode", default="train")
    ap.add_argument("--launches", type=int, default=400)
    ap.add_argument("--seq", action="store_true", help="also print every launch in order")
    a = ap.parse_args()
    w = WORKLOADS[a.workload]
    cfg = default_config(w["task"], image_feat_size=w["image_feat_size"])
    model = GlocalTextPathNavCMT(cfg, dtype=torch.bfloat16, device="cuda")
    model.init_weights(seed=0)
    batch = make_batch(cfg.vocab_size, cfg.image_feat_size, cfg.depth_feat_size, w["B"], w["L"], w["V"], w["G"], seed=1234)
    step = PlannerStep(model, batch, overlap=False, dropout="config" if a.mode == "train" else None)
    L = _lib.lib()
    for _ in range(5):
        step.run_eager()
    torch.cuda.synchronize()
    buf = torch.zeros(a.launches * WG_MAX * 8, dtype=torch.int64, device="cuda")
    _lib.check(L.etp_gemm_probe_enable(buf.data_ptr(), a.launches), "probe_enable")
    step.run_eager()
    torch.cuda.synchronize()
    n = int(L.etp_gemm_probe_count())
    metas = []
    for i in range(n):
        nm = ctypes.create_string_buffer(96)
        dims = (ctypes.c_int32 * 4)()
        _lib.check(L.etp_gemm_probe_meta(i, nm, 96, dims), "probe_meta")
        metas.append((nm.value.decode(), tuple(dims)))
    _lib.check(L.etp_gemm_probe_enable(None, 0), "probe_disable")
    rec = buf.cpu().numpy().view(np.uint64).reshape(a.launches, WG_MAX, 8)

    rows = []
    for i, (name, (grid, M, N, K)) in enumerate(metas):
        r = rec[i, :grid].astype(np.int64)
        rt0, rt1, m0, m1, m2, m3, hw = r[:, 0], r[:, 1], r[:, 2], r[:, 3], r[:, 4], r[:, 5], r[:, 6]
        ok = rt1 > 0
        if not ok.all():
            continue
        span = (rt1.max() - rt0.min()) * 0.01                      # 100 MHz ticks -> us
        skew = (rt0.max() - rt0.min()) * 0.01
        ghz = float(np.median((m3 - m0) / np.maximum(rt1 - rt0, 1))) * 0.1
        cyc = 1e-3 / max(ghz, 1e-3)                                 # us per shader cycle
        cu = ((hw >> 32) << 16) | (hw & 0xFF00)                     # XCC_ID | se/sh/cu bits of HW_ID
        _, counts = np.unique(cu, return_counts=True)
        rows.append(dict(name=name, M=M, N=N, K=K, grid=grid, span=span, skew=skew, pro=float((m1 - m0).mean()) * cyc,
                         loop=float((m2 - m1).mean()) * cyc, epi=float((m3 - m2).mean()) * cyc,
                         wg=float((m3 - m0).mean()) * cyc, wgmax=float((m3 - m0).max()) * cyc, ncu=len(counts),
                         percu=int(counts.max()), ghz=ghz, slabs=int(r[0, 7])))
    agg = OrderedDict()
    for r in rows:
        k = (r["name"], r["M"], r["N"], r["K"], r["grid"])
        agg.setdefault(k, []).append(r)
    print(f"# {len(rows)} probed GEMM launches of one single-stream {a.workload} step ({a.mode} mode); times in us, means over "
          f"launches and workgroups")
    print(f"{'n':>3} {'span':>7} {'skew':>6} {'pro':>6} {'loop':>7} {'epi':>6} {'wg':>7} {'wgmax':>7} {'us/slab':>7} {'CUs':>4} "
          f"{'wg/cu':>5} {'GHz':>5}  kernel M N K grid")
    tot = 0.0
    for k, rs in sorted(agg.items(), key=lambda kv: -sum(r["span"] for r in kv[1])):
        m = lambda f: sum(r[f] for r in rs) / len(rs)
        tot += sum(r["span"] for r in rs)
        print(f"{len(rs):3d} {m('span'):7.2f} {m('skew'):6.2f} {m('pro'):6.2f} {m('loop'):7.2f} {m('epi'):6.2f} {m('wg'):7.2f} "
              f"{m('wgmax'):7.2f} {m('loop') / max(rs[0]['slabs'], 1):7.3f} {rs[0]['ncu']:4d} {rs[0]['percu']:5d} {m('ghz'):5.2f}  "
              f"{k[0]} {k[1]}x{k[2]}x{k[3]} grid {k[4]}")
    print(f"# sum of device-side GEMM spans: {tot:.0f} us/step")
    if a.seq:
        print("\n# launch sequence")
        for r in rows:
            print(f"{r['span']:7.2f} {r['pro']:6.2f} {r['loop']:7.2f} {r['epi']:6.2f}  {r['name']} {r['M']}x{r['N']}x{r['K']} grid {r['grid']}")
    step.close()


if __name__ == "__main__":
    main()
