"""Per-kernel budget of one planner step, single stream (each HIP-event pair brackets its kernel alone).

    python tools/chain_budget.py [--workload c2] [--mode train] [--seq]

Prints, per (kernel, grid) class: launches per step, average / total microseconds; with --seq the whole launch sequence of
one step in order.  Every kernel of the library goes through csrc/launch.h, so this covers GEMMs, attention, LayerNorm and
the embedding / head kernels alike (bench.py's roofline leg only times the GEMMs).  The sum over classes is the
single-stream step time minus launch gaps; what the three-stream schedule can hide is the weight-gradient groups only.
"""
import argparse, ctypes, os, re, sys
from collections import OrderedDict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from etpnav_amd import _lib
from etpnav_amd.planner import GlocalTextPathNavCMT, default_config
from etpnav_amd.step import PlannerStep
from etpnav_amd.synthetic import make_batch
from bench import WORKLOADS


def short(name):
    n = re.sub(r"\(anonymous namespace\)::|etp::", "", name)
    n = re.sub(r"^void\s+", "", n)
    n = n.replace("unsigned short", "bf16")
    return n.split("(")[0][:100]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--mode", default="train")
    ap.add_argument("--seq", action="store_true")
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    w = WORKLOADS[a.workload]
    cfg = default_config(w["task"], image_feat_size=w["image_feat_size"])
    model = GlocalTextPathNavCMT(cfg, dtype=torch.bfloat16, device="cuda")
    model.init_weights(seed=0)
    batch = make_batch(cfg.vocab_size, cfg.image_feat_size, cfg.depth_feat_size, w["B"], w["L"], w["V"], w["G"], seed=1234)
    step = PlannerStep(model, batch, overlap=False, dropout="config" if a.mode == "train" else None)
    L = _lib.lib()
    for _ in range(3):
        step.run_eager()
    torch.cuda.synchronize()
    L.etp_ktime_reset(); L.etp_ktime_enable(1)
    for _ in range(a.steps):
        step.run_eager()
    torch.cuda.synchronize()
    L.etp_ktime_enable(0)
    buf = ctypes.create_string_buffer(8 << 20)
    n = L.etp_ktime_report(buf, len(buf))
    lines = buf.raw[:n].decode().strip().split("\n")
    L.etp_ktime_reset()
    recs = [l.split("\t") for l in lines]
    per_step = len(recs) // a.steps
    agg = OrderedDict()
    for us, grid, block, st, name in recs:
        k = (short(name), int(grid))
        e = agg.setdefault(k, [0, 0.0])
        e[0] += 1; e[1] += float(us)
    tot = sum(e[1] for e in agg.values()) / a.steps
    print(f"{per_step} launches per step, sum of kernel spans {tot:.0f} us/step (single stream)")
    print(f"{'us/step':>9} {'n/step':>7} {'avg us':>8} {'grid':>6}  kernel")
    for (name, grid), (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{us / a.steps:9.1f} {cnt / a.steps:7.1f} {us / cnt:8.2f} {grid:6d}  {name}")
    if a.seq:
        print("\n# launch sequence of the last step")
        for us, grid, block, st, name in recs[-per_step:]:
            print(f"{float(us):8.2f} {int(grid):6d}  {short(name)}")
    step.close()


if __name__ == "__main__":
    main()
