"""Register / scratch audit of every kernel in libetpnav_hip.so's objects.

    python tools/kernel_resources.py [--write profiles/r05_kernel_resources.txt] [--strict]

Round 4 traced a sporadic wrong gradient to a kernel without any MFMA whose loads the compiler had parked in AGPRs (accumulation
VGPRs) for ~400 instructions: the parked copy went wrong while an MFMA-heavy kernel shared the CU; the mechanism was never found.
Until it is, NO kernel outside the matrix-core families may be given AGPRs, and no kernel at all may spill to scratch.  This tool
reads the code-object metadata (`.agpr_count`, `.private_segment_fixed_size`, `.vgpr_spill_count`, ...) of the gfx950 code object
embedded in each build object (llvm-objcopy --dump-section .hip_fatbin -> clang-offload-bundler --unbundle -> llvm-readelf --notes)
and applies that policy; `etpnav_amd.build.build()` runs it after every build and fails the build on a violation.

The reference has no native code; this guards the MI355X-side replacement of its autograd kernels (vilmodel_cmt.py throughout).
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LLVM = os.environ.get("ETP_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"

# Kernel families that issue MFMAs: accumulators may live in AGPRs there by design.  Everything else must report agpr_count 0.
MFMA_FAMILIES = ("gemm_kernel", "gemm_dma_kernel", "gemm_group_kernel", "mm32::kernel", "mm32::group_kernel", "attn_fwd_kernel",
                 "attn_bwd_kernel", "flash_fwd", "flash_bwd", "rows_fwd_kernel", "rows_bwd_kernel")
FIELDS = ("agpr_count", "vgpr_count", "sgpr_count", "private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count",
          "group_segment_fixed_size", "uses_dynamic_stack")


def code_object_notes(obj: str) -> str:
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", obj, os.path.join(td, "copy.o")])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                               f"--targets={TARGET}", f"--output={co}"])
        return subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)


def parse(notes: str):
    """-> list of dicts, one per kernel, from the amdhsa.kernels YAML in the note (flat `.key: value` lines per `- ` item)."""
    kernels, cur, in_args = [], None, False
    for line in notes.splitlines():
        m = re.match(r"^(\s*)(- )?\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        indent, dash, key, val = len(m.group(1)), bool(m.group(2)), m.group(3), m.group(4).strip()
        if indent == 2 and dash:              # a new item of amdhsa.kernels
            cur = {}
            kernels.append(cur)
            in_args = False
        if cur is None or indent > 4:         # argument lists are nested deeper
            continue
        if key == "args":
            in_args = True
            continue
        if indent == 4 or (indent == 2 and dash):
            in_args = False if indent == 4 and key != "args" else in_args
            if key in FIELDS or key == "name":
                cur[key] = val.strip("'\"")
    return [k for k in kernels if "name" in k and "vgpr_count" in k]


def demangle(names):
    import shutil
    tool = shutil.which("c++filt") or os.path.join(LLVM, "llvm-cxxfilt")
    if not os.path.exists(tool):
        return list(names)
    out = subprocess.run([tool], input="\n".join(names) + "\n", capture_output=True, text=True, check=True).stdout
    return out.strip().splitlines()


def audit(objdir=None):
    from etpnav_amd import build as b
    objdir = objdir or os.path.join(b.HERE, "build")
    rows = []
    for src in b.SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        if not os.path.exists(obj):
            raise SystemExit(f"{obj} is missing: build first (python -m etpnav_amd.build)")
        with open(obj, "rb") as f:
            if b".hip_fatbin" not in f.read():
                continue                      # host-only translation unit (no device code)
        ks = parse(code_object_notes(obj))
        if not ks:
            continue
        names = demangle([k["name"] for k in ks])
        for k, nm in zip(ks, names):
            short = re.sub(r"\(.*$", "", nm.replace("(anonymous namespace)::", "")).replace("void ", "").replace("etp::", "")
            row = dict(src=src, name=short, full=nm, **{f: k.get(f, "0") for f in FIELDS})
            mfma = any(fam in nm for fam in MFMA_FAMILIES)
            # SGPR spills go to VGPR lanes (no memory traffic): listed, not a violation
            scratch = int(row["private_segment_fixed_size"]) > 0 or int(row["vgpr_spill_count"]) > 0
            row["mfma"] = mfma
            row["violation"] = ("AGPRs in a kernel without MFMAs" if (int(row["agpr_count"]) > 0 and not mfma) else
                                "scratch / spills" if scratch else "")
            rows.append(row)
    return rows


def table(rows) -> str:
    lines = ["# tools/kernel_resources.py: code-object metadata of every kernel of libetpnav_hip.so (gfx950).  Policy: AGPRs only in",
             "# the matrix-core families (mm32 / gemm_* / attention), no scratch and no spills anywhere (DESIGN.md §3.6).",
             f"# {len(rows)} kernels, {sum(1 for r in rows if r['violation'])} violations, {sum(1 for r in rows if int(r['agpr_count']) > 0)} with AGPRs",
             f"{'source':14s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'scratch':>8s} {'spill':>6s} {'lds':>7s} mfma  kernel"]
    for r in sorted(rows, key=lambda r: (r["src"], r["name"])):
        lines.append(f"{r['src']:14s} {r['vgpr_count']:>5s} {r['agpr_count']:>5s} {r['sgpr_count']:>5s} {r['private_segment_fixed_size']:>8s} "
                     f"{int(r['vgpr_spill_count']) + int(r['sgpr_spill_count']):>6d} {r['group_segment_fixed_size']:>7s} {'yes ' if r['mfma'] else 'no  '}  "
                     f"{r['name']}{'   <-- ' + r['violation'] if r['violation'] else ''}")
    return "\n".join(lines) + "\n"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", default=None, help="also write the table to this file")
    ap.add_argument("--strict", action="store_true", help="exit 1 on a violation")
    a = ap.parse_args()
    rows = audit()
    t = table(rows)
    if a.write:
        with open(a.write, "w") as f:
            f.write(t)
    bad = [r for r in rows if r["violation"]]
    print(t if not a.write else f"{len(rows)} kernels, {len(bad)} violations -> {a.write}")
    for r in bad:
        print(f"VIOLATION {r['src']}: {r['full']}: {r['violation']} (agpr {r['agpr_count']}, scratch {r['private_segment_fixed_size']})")
    if a.strict and bad:
        sys.exit(1)


if __name__ == "__main__":
    main()
