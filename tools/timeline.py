"""Timeline analysis of a `rocprofv3 --kernel-trace` CSV of bench.py: where does a planner step's wall time go?

  python tools/timeline.py gpurun_out/prof/r01_kernel_trace.csv [--steps 20]

Steps are delimited by the text-embedding forward kernel (one per step) and the weight-shadow cast in front of it.  For the last `--steps` steps it prints: wall time per step, the union of kernel-busy time (any stream), idle
gaps, per-stream busy time, and per kernel family the summed duration, the "exclusive" duration (time during which it
was the only kernel running) and the launch count.  Exclusive time of small-grid kernels is what the three-stream
schedule failed to overlap: the list to attack first.
"""
import argparse
import csv
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    n = re.sub(r"\(anonymous namespace\)::", "", name)
    n = re.sub(r"^void\s+", "", n)
    m = re.match(r"([\w:]+)<(.*)>\(", n)
    if m:
        args = m.group(2)
        args = re.sub(r"__hip_bfloat16|hip_bfloat16", "bf16", args)
        args = re.sub(r"\s+", "", args)
        return f"{m.group(1).split('::')[-1]}<{args[:48]}>"
    return n.split("(")[0][-60:]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--top", type=int, default=28)
    a = ap.parse_args()
    rows = []
    with open(a.csv, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", "0"),
                         int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]),
                         int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])))
    rows.sort()
    # step starts: one text-embedding forward per step; the step begins with the weight-shadow cast just before it
    marks = [i for i, r in enumerate(rows) if "text_embed_fwd_kernel" in r[2]]
    if not marks:
        sys.exit("no text_embed_fwd kernels found: cannot delimit steps")
    starts = []
    for i in marks:
        j = i
        while j > 0 and i - j < 6 and "cast_f32_bf16_kernel" not in rows[j][2]:
            j -= 1
        starts.append(rows[j][0] if "cast_f32_bf16_kernel" in rows[j][2] else rows[i][0])
    if len(starts) < a.steps + 1:
        a.steps = len(starts) - 1
    t0, t1 = starts[-a.steps - 1], starts[-1]
    win = [r for r in rows if t0 <= r[0] < t1]
    wall = (t1 - t0) / a.steps
    # union busy + exclusive time via sweep
    ev = []
    for i, r in enumerate(win):
        ev.append((r[0], 1, i))
        ev.append((min(r[1], t1), -1, i))
    ev.sort()
    active = set()
    busy = 0
    excl = defaultdict(int)
    conc_hist = defaultdict(int)
    last = t0
    for t, d, i in ev:
        if t > last:
            n = len(active)
            conc_hist[min(n, 4)] += t - last
            if n >= 1:
                busy += t - last
            if n == 1:
                excl[short(win[next(iter(active))][2])] += t - last
            last = t
        if d == 1:
            active.add(i)
        else:
            active.discard(i)
    tot = defaultdict(int)
    cnt = defaultdict(int)
    wgs = defaultdict(int)
    per_stream = defaultdict(int)
    for r in win:
        k = short(r[2])
        tot[k] += r[1] - r[0]
        cnt[k] += 1
        wgs[k] += r[4] // max(r[5], 1)
        per_stream[r[3]] += r[1] - r[0]
    # gaps on the busiest stream (the critical chain): idle time before each kernel, attributed to that kernel's family
    main = max(per_stream, key=lambda k: per_stream[k])
    chain = [r for r in win if r[3] == main]
    gap_by = defaultdict(int)
    gap_n = defaultdict(int)
    gap_tot = 0
    pair_by = defaultdict(int)
    pair_n = defaultdict(int)
    for prev, cur in zip(chain, chain[1:]):
        gp = cur[0] - prev[1]
        if gp > 0:
            gap_tot += gp
            gap_by[short(cur[2])] += gp
            gap_n[short(cur[2])] += 1
            if gp > 20000:          # > 20 us: a dependency wait, not launch latency
                key = (short(prev[2])[:44], short(cur[2])[:44])
                pair_by[key] += gp
                pair_n[key] += 1
    us = lambda x: x / a.steps / 1e3
    print(f"steps analysed: {a.steps}   wall/step {wall / 1e3:.1f} us   kernels/step {len(win) / a.steps:.0f}")
    print(f"busy (>=1 kernel) {us(busy):.1f} us/step   idle {wall / 1e3 - us(busy):.1f} us/step")
    print("concurrency histogram (us/step): " + "  ".join(f"{k if k < 4 else '4+'}:{us(v):.0f}" for k, v in sorted(conc_hist.items())))
    print("per-stream kernel time (us/step): " + "  ".join(f"s{k}:{us(v):.0f}" for k, v in sorted(per_stream.items())))
    print(f"chain stream s{main}: {len(chain) / a.steps:.0f} kernels/step, busy {us(per_stream[main]):.0f} us, gaps {us(gap_tot):.0f} us/step"
          f" (avg {gap_tot / max(len(chain) - 1, 1) / 1e3:.2f} us per kernel); largest gap owners:")
    for k in sorted(gap_by, key=lambda k: -gap_by[k])[:8]:
        print(f"    {k[:72]:72s} {us(gap_by[k]):8.1f} us/step over {gap_n[k] / a.steps:.0f} launches")
    print("    waits > 20 us on the chain stream, by (previous kernel -> waiting kernel):")
    for k in sorted(pair_by, key=lambda k: -pair_by[k])[:10]:
        print(f"      {k[0]:44s} -> {k[1]:44s} {us(pair_by[k]):8.1f} us/step over {pair_n[k] / a.steps:.1f} waits")
    print(f"{'kernel':72s} {'sum us':>8s} {'excl us':>8s} {'n':>5s} {'avg us':>7s} {'avg WGs':>8s}")
    for k in sorted(tot, key=lambda k: -tot[k])[:a.top]:
        print(f"{k[:72]:72s} {us(tot[k]):8.1f} {us(excl[k]):8.1f} {cnt[k] / a.steps:5.0f} {tot[k] / cnt[k] / 1e3:7.1f} {wgs[k] / cnt[k]:8.0f}")


if __name__ == "__main__":
    main()
