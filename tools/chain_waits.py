"""Where does the dependent chain of a planner step wait?  Un-profiled, from device-side time stamps.

  python tools/chain_waits.py [--steps 20] [--out profiles/r04_chain_waits.txt]

`rocprofv3 --kernel-trace` slows the host (every launch is intercepted) and a step of ~330 launches can become host-bound
under it; gaps read off such a trace may be the profiler's.  This tool runs bench.py's configuration-2 step free-running (no
profiler, no synchronisation between steps) with a stamp sink installed (include/etpnav_hip.h etp_stamp_sink): at the layer
boundaries, forks and joins of the issue order a one-thread kernel stores s_memrealtime (100 MHz, chip-wide) on the stream it
was enqueued on.  Printed: the step time with and without the stamps, the host's enqueue time per step (is the host ahead of
the GPU?), and the mean duration of every interval between consecutive stamps of the dependent chain -- intervals with no
kernel of the chain in them are pure waits.
"""
import argparse
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

TAGS = {1: "step begin", 2: "join panorama fwd (before node assembly)", 3: "loss done", 4: "nav_bwd returned", 5: "node assembly bwd done",
        6: "txt_bwd begin", 7: "step end (panorama joined)", 1099: "txt_fwd end", 1100: "pano_fwd begin", 1199: "pano_fwd end",
        1299: "nav_fwd end", 2000: "nav_bwd begin", 2090: "nav_bwd: before d_txt join", 2091: "nav_bwd: d_txt joined",
        2100: "pano_bwd begin", 2199: "pano_bwd end", 2299: "txt_bwd: embeddings done"}


def tag_name(t):
    if t in TAGS:
        return TAGS[t]
    if 1000 <= t < 1099:
        return f"txt_fwd layer {t - 1000}"
    if 1200 <= t < 1299:
        return f"nav_fwd x-layer {t - 1200}"
    if 2010 <= t < 2090:
        return f"nav_bwd x-layer {(t - 2010) // 10}"
    if 2200 <= t < 2299:
        return f"txt_bwd layer {(t - 2200) // 10}" + (" (ffn done)" if t % 10 == 1 else "")
    return str(t)


def side(t):                       # stamps taken on the panorama stream
    return t in (1100, 1199, 2100, 2199)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from etpnav_amd import _lib
    from etpnav_amd.planner import GlocalTextPathNavCMT, default_config
    from etpnav_amd.step import PlannerStep
    from etpnav_amd.synthetic import make_batch
    import bench
    w = dict(bench.WORKLOADS["c2"])
    cfg = default_config(w["task"], image_feat_size=w["image_feat_size"])
    model = GlocalTextPathNavCMT(cfg, dtype=torch.bfloat16, device="cuda:0")
    model.init_weights(seed=0)
    batch = make_batch(cfg.vocab_size, cfg.image_feat_size, cfg.depth_feat_size, w["B"], w["L"], w["V"], w["G"], seed=1234)
    step = PlannerStep(model, batch, overlap=True, dropout="config", drop_seed=0)
    L = _lib.lib()
    for _ in range(40):
        step.run_eager()
    torch.cuda.synchronize()

    def timed(n):
        host = []
        t0 = time.perf_counter()
        for _ in range(n):
            h0 = time.perf_counter()
            step.run_eager()
            host.append(time.perf_counter() - h0)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, t_issue / n * 1e3, host

    base_ms, base_issue, _ = timed(a.steps)
    cap = a.steps * 96
    buf = torch.zeros(cap, dtype=torch.int64, device="cuda:0")
    _lib.check(L.etp_stamp_sink(ctypes.c_void_p(buf.data_ptr()), cap), "stamp_sink")
    st_ms, st_issue, host = timed(a.steps)
    n = int(L.etp_stamp_count())
    tags = [int(L.etp_stamp_tag(i)) for i in range(n)]
    _lib.check(L.etp_stamp_sink(None, 0), "stamp_sink off")
    t = buf.cpu().numpy()[:n].astype("float64") * 0.01          # us
    lines = []
    p = lines.append
    p(f"# tools/chain_waits.py: {a.steps} free-running config-2 steps (bf16, train mode), no profiler")
    p(f"step time without stamps {base_ms:.3f} ms (host issue {base_issue:.3f} ms/step); with {n // a.steps} stamps per step "
      f"{st_ms:.3f} ms (host issue {st_issue:.3f} ms/step)")
    p(f"host enqueue time per step: min {min(host) * 1e3:.3f} / median {sorted(host)[len(host) // 2] * 1e3:.3f} / max {max(host) * 1e3:.3f} ms"
      f"  -> the host is {'AHEAD of' if st_issue < st_ms * 0.9 else 'NOT ahead of'} the GPU")
    # per step: stamps of the dependent chain in enqueue order
    starts = [i for i, g in enumerate(tags) if g == 1] + [n]
    agg, order = {}, []
    sagg = {}
    for si in range(2, len(starts) - 1):                       # skip the first two steps (queues filling)
        idx = list(range(starts[si], starts[si + 1]))
        chain = [i for i in idx if not side(tags[i])]
        for i0, i1 in zip(chain, chain[1:]):
            key = (tags[i0], tags[i1])
            if key not in agg:
                agg[key] = []
                order.append(key)
            agg[key].append(t[i1] - t[i0])
        nxt = starts[si + 1]
        if nxt < n:
            agg.setdefault((7, 1), []).append(t[nxt] - t[chain[-1]])
        for a0, a1 in ((1100, 1199), (2100, 2199)):
            i0 = [i for i in idx if tags[i] == a0]
            i1 = [i for i in idx if tags[i] == a1]
            if i0 and i1:
                sagg.setdefault((a0, a1), []).append(t[i1[0]] - t[i0[0]])
        for a0, a1 in ((5, 2100), (2199, 7), (1199, 2)):       # fork / join latencies between the chain and the panorama stream
            i0 = [i for i in idx if tags[i] == a0]
            i1 = [i for i in idx if tags[i] == a1]
            if i0 and i1:
                sagg.setdefault((a0, a1), []).append(t[i1[0]] - t[i0[0]])
    if (7, 1) in agg and (7, 1) not in order:
        order.append((7, 1))
    tot = 0.0
    p(f"{'interval on the dependent chain':78s} {'mean us':>9s} {'min':>8s} {'max':>8s}")
    for key in order:
        v = agg[key]
        m = sum(v) / len(v)
        tot += m
        p(f"{tag_name(key[0])[:36]:36s} -> {tag_name(key[1])[:38]:38s} {m:9.1f} {min(v):8.1f} {max(v):8.1f}")
    p(f"{'sum':78s} {tot:9.1f}")
    p("panorama stream and its forks / joins:")
    for key, v in sagg.items():
        p(f"{tag_name(key[0])[:36]:36s} -> {tag_name(key[1])[:38]:38s} {sum(v) / len(v):9.1f} {min(v):8.1f} {max(v):8.1f}")
    p("pure waits (no chain kernel between the two stamps): 'node assembly bwd done -> txt_bwd begin' (only the fork of the panorama "
      "stream is enqueued between them), 'nav_bwd: before d_txt join -> d_txt joined' (the chain waits for the d_txt stream), "
      "'step end -> step begin' (join of the weight-gradient / panorama streams + the next step's prologue).")
    txt = "\n".join(lines)
    print(txt)
    if a.out:
        open(a.out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
