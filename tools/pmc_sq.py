"""Per-kernel-class SQ / GRBM counters from a rocprofv3 --pmc pass (MFMA utilisation evidence, BASELINE.json north_star).

    cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
        SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
        -d $R/gpurun_out/pmc_sq -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-optimizer
    python tools/pmc_sq.py gpurun_out/pmc_sq/p_counter_collection.csv --out profiles/r03_gemm_counters.json

(8 SQ slots + 2 GRBM slots per pass on gfx950: MI355X_MICROARCH.md "rocprofv3 PMC slots"; never combined with other trace
domains.)  Kernels are grouped by (short name, grid); counters are means per launch.  Derived columns:

    mfma_util   SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)  matrix-pipe busy cycles summed over the chip's
                                                                          1024 SIMDs / cycles the kernel was resident.
                                                                          GRBM_GUI_ACTIVE arrives summed over the 8 XCDs
                                                                          (it reads 8 x duration x clock), hence the / 8;
                                                                          check: the grouped weight-gradient GEMM's 2.2 M
                                                                          MFMAs x ~14.4 busy cycles = 31.9 M = the counter
    mfma_time   SQ_VALU_MFMA_BUSY_CYCLES / (duration * 2.4 GHz * 1024)   the same against the kernel's own duration at the
                                                                          maximum clock: GRBM_GUI_ACTIVE also counts a window
                                                                          around short kernels (it implies > 2.4 GHz for
                                                                          kernels under ~20 us), which deflates mfma_util
    wait_frac   SQ_WAIT_ANY / SQ_WAVE_CYCLES                              wave parked on s_waitcnt / s_barrier
    stall_frac  SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES                         issue stalls
    issue_frac  SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
    lds_conf    SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE                  extra LDS cycles per LDS cycle
    ghz         GRBM_GUI_ACTIVE / 8 / duration                            effective clock

The reference has no counterpart (host time.time() counters only: pretrain_src/pretrain_src/train_r2r.py:227,299-317).
"""
import argparse
import csv
import json
import re
from collections import defaultdict

csv.field_size_limit(1 << 30)
XCDS = 8          # GRBM_GUI_ACTIVE is reported summed over the XCDs on gfx950


def short(kernel: str) -> str:
    m = re.search(r"gemm(_dma|_group)?_kernel<([^>]*)>", kernel)
    if m:
        a = [x.strip() for x in m.group(2).split(",")]
        t = "bf16" if "short" in a[0] else "f32"
        tc = "bf16" if "short" in a[1] else "f32"
        ta, tb = a[2] == "true", a[3] == "true"
        tr = "TN" if (ta and tb) else ("NN" if tb else "NT")
        st = f",s{a[6]}" if len(a) > 6 else ""
        return f"gemm{m.group(1) or ''}<{t},{tc},{tr},{a[4]}x{a[5]}{st}>"
    m = re.search(r"mm32::(group_)?kernel<([^>]*)>", kernel)        # gemm_mm32.hip: <TC, TA, TB, BM, BN, STAGES>, operands always bf16
    if m:
        a = [x.strip() for x in m.group(2).split(",")]
        tc = "f32" if "float" in a[0] else "bf16"
        ta, tb = a[1] == "true", a[2] == "true"
        tr = "TN" if (ta and tb) else ("NN" if tb else "NT")
        spi = "x2" if len(a) > 6 and a[6] == "2" else ""      # slabs per hand-over (absent in builds before it existed)
        return f"mm32{'_group' if m.group(1) else ''}<bf16,{tc},{tr},{a[3]}x{a[4]},s{a[5]}{spi}>"
    m2 = re.search(r"(\w+_kernel)\s*<", kernel) or re.search(r"::(\w+_kernel)", kernel) or re.search(r"(\w+_kernel)", kernel)
    return m2.group(1) if m2 else kernel[:48]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--out", default=None)
    ap.add_argument("--top", type=int, default=24)
    ap.add_argument("--skip-dispatches", type=int, default=0, help="ignore the first N dispatches (warm-up)")
    a = ap.parse_args()
    per = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(dict)
    with open(a.csv, newline="") as f:
        for r in csv.DictReader(f):
            if "etp" not in r["Kernel_Name"]:
                continue
            if int(r["Dispatch_Id"]) <= a.skip_dispatches:
                continue
            k = (short(r["Kernel_Name"]), int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    out = {}
    for k, cs in per.items():
        e = {c: sum(v) / len(v) for c, v in cs.items()}
        n = max(len(v) for v in cs.values())
        d = sum(dur[k].values()) / max(len(dur[k]), 1)
        g = lambda c: e.get(c)
        ent = {"launches": n, "workgroups": k[1], "avg_us_under_pmc": round(d, 2), "counters": {c: round(v, 1) for c, v in e.items()}}
        if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None and g("GRBM_GUI_ACTIVE"):
            ent["mfma_util"] = round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("GRBM_GUI_ACTIVE") / XCDS * 1024.0), 4)
            if d > 0:
                ent["mfma_time"] = round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (d * 2400.0 * 1024.0), 4)
        if g("SQ_WAVE_CYCLES"):
            for name, c in (("wait_frac", "SQ_WAIT_ANY"), ("stall_frac", "SQ_WAIT_INST_ANY"), ("issue_frac", "SQ_ACTIVE_INST_ANY")):
                if g(c) is not None:
                    ent[name] = round(g(c) / g("SQ_WAVE_CYCLES"), 4)
        if g("SQ_LDS_IDX_ACTIVE"):
            ent["lds_conf"] = round((g("SQ_LDS_BANK_CONFLICT") or 0.0) / g("SQ_LDS_IDX_ACTIVE"), 4)
        if g("GRBM_GUI_ACTIVE") and d > 0:
            ent["ghz"] = round(g("GRBM_GUI_ACTIVE") / XCDS / d * 1e-3, 3)
        out[f"{k[0]} grid {k[1]}"] = ent
    rows = sorted(out.items(), key=lambda kv: -kv[1]["avg_us_under_pmc"] * kv[1]["launches"])
    print(f"{'total us':>9} {'n':>4} {'avg us':>8} {'mfma':>6} {'mfma_t':>6} {'wait':>6} {'stall':>6} {'issue':>6} {'ldsconf':>7} {'GHz':>5}  kernel")
    for name, e in rows[:a.top]:
        f = lambda x: f"{e[x]:6.3f}" if x in e else "     -"
        print(f"{e['avg_us_under_pmc'] * e['launches']:9.0f} {e['launches']:4d} {e['avg_us_under_pmc']:8.2f} {f('mfma_util')} {f('mfma_time')} {f('wait_frac')} "
              f"{f('stall_frac')} {f('issue_frac')} {f('lds_conf'):>7} {e.get('ghz', 0):5.2f}  {name}")
    if a.out:
        json.dump({"source": "rocprofv3 --pmc (one pass, 8 SQ + 1 GRBM counters) -- tools/pmc_sq.py", "kernels": dict(rows)},
                  open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
