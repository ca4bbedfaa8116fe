"""Per-kernel HBM traffic from rocprofv3 PMC passes (one pass per counter: FETCH_SIZE costs 3 of the 4 TCC slots and
WRITE_SIZE 2 — MI355X_MICROARCH.md "rocprofv3 PMC slots").

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch -o p --output-format csv -- python bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write -o p --output-format csv -- python bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch/p_counter_collection.csv gpurun_out/pmc_write/p_counter_collection.csv \
        --out profiles/r01_pmc_traffic.json

Calibration (the guide: on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x and WRITE_SIZE is uncalibrated —
"calibrate on a known byte count in your own access pattern"): the bf16 weight-shadow refresh `cast_f32_bf16_kernel`
streams a known number of bytes (4 B read + 2 B written per element).  Its three large launches per step (text /
panorama / navigation part of the matrix region, all on the capped 4096-block grid) are averaged, so --cast-elems is
the MEAN element count of those launches = etp_planner_matrix_elems / 3 (38 961 152 for the config-2 model).  The
factor known_bytes / counter_value of that kernel is applied to every other kernel of the same pass.  Round-1 result:
2048 B per FETCH_SIZE unit (KB, under-reported 2x on wide reads, as the guide says) and 1024 B per WRITE_SIZE unit.
Since round 3 the text cast is split (layer 0 on the main stream, a smaller grid): the three largest-grid launches cover
n_matrix minus layer 0's elements, so --cast-elems = (116 883 456 - 7 077 888) / 3 = 36 601 856 for the config-2 model.
Check of any calibration: a kernel with a known output (the 2560 x 768 fp32 stream products: 7.86 MB) must read back its size.
"""
import argparse
import csv
import json
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def bench_name(kernel: str):
    """Short kernel class name as bench.py prints it, without the ring-depth suffix (one traffic row per tile class)."""
    try:
        from tools.pmc_sq import short
    except ImportError:                                   # run as a script from tools/
        from pmc_sq import short
    return re.sub(r",s\d+(x2)?>$", ">", short(kernel))


def load(path, counter):
    per = defaultdict(list)
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                per[bench_name(r["Kernel_Name"])].append((float(r["Counter_Value"]), int(r["Grid_Size"])))
    return per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("--cast-elems", type=float, default=None, help="mean elements of the large cast_f32_bf16 launches (matrix elems / 3)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    fetch, write = load(a.fetch_csv, "FETCH_SIZE"), load(a.write_csv, "WRITE_SIZE")
    ck = "cast_f32_bf16_kernel"
    if ck not in fetch or ck not in write:
        sys.exit("no cast_f32_bf16_kernel dispatches: cannot calibrate")
    big = max(g for _, g in fetch[ck])
    fvals = [v for v, g in fetch[ck] if g == big]
    wvals = [v for v, g in write[ck] if g == big]
    out = {"calibration": {"kernel": ck, "largest_grid": big, "fetch_counter_mean": sum(fvals) / len(fvals),
                           "write_counter_mean": sum(wvals) / len(wvals), "launches": len(fvals)}}
    if a.cast_elems:
        out["calibration"]["known_read_bytes"] = a.cast_elems * 4
        out["calibration"]["known_write_bytes"] = a.cast_elems * 2
        kf = a.cast_elems * 4 / (sum(fvals) / len(fvals))
        kw = a.cast_elems * 2 / (sum(wvals) / len(wvals))
    else:
        kf = kw = None
    out["calibration"]["bytes_per_fetch_unit"] = kf
    out["calibration"]["bytes_per_write_unit"] = kw
    ks = {}
    for k in sorted(set(fetch) | set(write)):
        f = [v for v, _ in fetch.get(k, [])]
        w = [v for v, _ in write.get(k, [])]
        ent = {"launches": max(len(f), len(w)), "fetch_counter_mean": sum(f) / len(f) if f else None,
               "write_counter_mean": sum(w) / len(w) if w else None}
        if kf and f:
            ent["fetch_bytes_per_launch"] = kf * ent["fetch_counter_mean"]
        if kw and w:
            ent["write_bytes_per_launch"] = kw * ent["write_counter_mean"]
        if kf and kw and f and w:
            ent["hbm_bytes_per_launch"] = ent["fetch_bytes_per_launch"] + ent["write_bytes_per_launch"]
        ks[k] = ent
    out["kernels"] = ks
    txt = json.dumps(out, indent=1)
    if a.out:
        open(a.out, "w").write(txt + "\n")
    for k, e in sorted(ks.items(), key=lambda kv: -(kv[1].get("hbm_bytes_per_launch") or 0) * kv[1]["launches"])[:16]:
        print(f"{k:44s} n={e['launches']:5d} fetch={e.get('fetch_bytes_per_launch', e['fetch_counter_mean'])!s:>14.14} "
              f"write={e.get('write_bytes_per_launch', e['write_counter_mean'])!s:>14.14}")
    print(json.dumps(out["calibration"]))


if __name__ == "__main__":
    main()
