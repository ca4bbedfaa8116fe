"""Register / scratch audit of every kernel in libetpnav_hip.so's objects.

    python tools/kernel_resources.py [--write profiles/r05_kernel_resources.txt] [--strict]

Round 4 traced a sporadic wrong gradient to a kernel without any MFMA whose loads the compiler had parked in AGPRs (accumulation
VGPRs) for ~400 instructions: the parked copy went wrong while an MFMA-heavy kernel shared the CU; the mechanism was never found.
Until it is, NO kernel outside the matrix-core families may be given AGPRs, and no kernel at all may spill to scratch.  This tool
reads the code-object metadata (`.agpr_count`, `.private_segment_fixed_size`, `.vgpr_spill_count`, ...) of the gfx950 code object
embedded in each build object (llvm-objcopy --dump-section .hip_fatbin -> clang-offload-bundler --unbundle -> llvm-readelf --notes)
and applies that policy; `etpnav_amd.build.build()` runs it after every build and fails the build on a violation.

Second check (round 5): the LDS-DMA statements write M0 from inline asm without being able to declare it; `m0_audit`
disassembles the objects and verifies, per kernel, that every M0 write feeds the DMA right behind it (or is the restore right
behind one) and that the compiler generated no M0 use of its own there.

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


def _with_code_object(obj: str, fn):
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", obj, os.path.join(td, "copy.o")])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                               f"--targets={TARGET}", f"--output={co}"])
        return fn(co)


def code_object_notes(obj: str) -> str:
    return _with_code_object(obj, lambda co: subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True))


def code_object_isa(obj: str) -> str:
    return _with_code_object(obj, lambda co: subprocess.check_output(
        [os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], text=True))


# ---- LDS-DMA / M0 audit (the glds() inline asm writes M0 without telling the compiler) --------------------------------
# global_load_lds takes the LDS byte address of its piece from M0.  hipcc rejects M0 on a clobber list, so the asm statements of
# gemm_mm32.hip (glds) and gemm_tiles.h (glds16) are only correct while (1) every M0 write sits directly in front of the DMA it
# feeds (nothing but s_nop between), or is glds16's restore directly behind one, and (2) no instruction of the compiler's own in a
# kernel with LDS-DMA reads or holds a value in M0 (relative moves, GPR-index mode, GWS, sendmsg, interpolation, buffer ... lds).
# Both are properties of the ISA the build produced, so the build checks them.
M0_IMPLICIT = ("s_movrel", "v_movrel", "s_set_gpr_idx", "ds_gws", "s_sendmsg", "v_interp", "ds_read_addtid", "ds_write_addtid")


def m0_scan(isa: str):
    """-> (stats, violations) of one disassembly: stats = {kernel: number of LDS-DMA instructions}."""
    stats, bad = {}, []
    kern, ins = None, []

    def flush():
        if kern is None or not any(m.startswith("global_load_lds") for m, _ in ins):
            return
        n = 0
        for i, (m, ops) in enumerate(ins):
            toks = re.split(r"[\s,]+", ops)
            if m.startswith("global_load_lds"):
                n += 1
                j = i - 1
                while j >= 0 and ins[j][0] == "s_nop":
                    j -= 1
                if j < 0 or not (ins[j][0] == "s_mov_b32" and ins[j][1].split(",")[0].strip() == "m0"):
                    bad.append((kern, i, f"{m} {ops}: not directly behind its own 's_mov_b32 m0, ...'"))
            elif "m0" in toks:
                dst = ops.split(",")[0].strip()
                if m == "s_mov_b32" and dst == "m0":                       # a write: feeds the next DMA, or restores behind one
                    j = i + 1
                    while j < len(ins) and ins[j][0] == "s_nop":
                        j += 1
                    feeds = j < len(ins) and ins[j][0].startswith("global_load_lds")
                    restores = False                                       # glds16: save, write, (nop), DMA, restore FROM THE SAVE
                    if i > 0 and ins[i - 1][0].startswith("global_load_lds"):
                        j = i - 2
                        while j >= 0 and ins[j][0] == "s_nop":
                            j -= 1                                          # j = the write that fed the DMA
                        src = ops.split(",")[1].strip() if "," in ops else ""
                        restores = (j >= 1 and ins[j - 1][0] == "s_mov_b32" and
                                    [t.strip() for t in ins[j - 1][1].split(",")] == [src, "m0"])
                    if not (feeds or restores):
                        bad.append((kern, i, f"{m} {ops}: an M0 write that is neither consumed by the next LDS-DMA nor a restore behind one"))
                elif m == "s_mov_b32" and dst != "m0":                     # glds16's save
                    nxt = ins[i + 1] if i + 1 < len(ins) else ("", "")
                    if not (nxt[0] == "s_mov_b32" and nxt[1].split(",")[0].strip() == "m0"):
                        bad.append((kern, i, f"{m} {ops}: M0 read outside the save / write / DMA / restore statement"))
                else:
                    bad.append((kern, i, f"{m} {ops}: compiler-generated M0 use in a kernel with LDS-DMA"))
            elif m.startswith(M0_IMPLICIT) or (m.startswith("buffer_load") and " lds" in " " + ops):
                bad.append((kern, i, f"{m} {ops}: implicit M0 use in a kernel with LDS-DMA"))
        stats[kern] = n

    for line in isa.splitlines():
        mk = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if mk:
            flush()
            kern, ins = mk.group(1), []
            continue
        mi = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*(//.*)?$", line)
        if mi and kern is not None:
            ins.append((mi.group(1), mi.group(2)))
    flush()
    return stats, bad


def m0_audit(objdir=None):
    """-> (number of kernels with LDS-DMA, number of LDS-DMA instructions, violations [(src, kernel, text)])."""
    from etpnav_amd import build as b
    objdir = objdir or os.path.join(b.HERE, "build")
    nk = ni = 0
    bad = []
    for src in b.SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        with open(obj, "rb") as f:
            if b".hip_fatbin" not in f.read():
                continue
        isa = code_object_isa(obj)
        if "global_load_lds" not in isa:
            continue
        stats, v = m0_scan(isa)
        nk += len(stats)
        ni += sum(stats.values())
        bad += [(src, k, t) for k, _, t in v]
    return nk, ni, bad


# packed-fp32 VALU instructions that the row-kernel objects may contain: none (etpnav_amd/build.py NO_PACKED_FP32 compiles them with
# -fno-slp-vectorize -fno-vectorize)
PK_ALLOW = 0
PK_RE = re.compile(r"\bv_pk_(?:fma|mul|add)_f32\b")


# The form that misbehaves beside another kernel's MFMAs (profiles/r06_pk_opsel_repro.txt): a packed fp32 instruction whose LOW result
# takes the HIGH register of a source (op_sel:[..1..]).  No object of the library may contain one, GEMM and attention objects included
# (their packed instructions use op_sel_hi only, which the reproducer shows clean).
PK_OPSEL_RE = re.compile(r"\bv_pk_(?:fma|mul|add)_f32\b[^\n/]*\bop_sel:\[[01,]*1[01,]*\]")


def pk_opsel_audit(objdir=None):
    """-> {object file: number of packed fp32 instructions with a low-half operand select} over every object with device code."""
    from etpnav_amd import build as b
    objdir = objdir or os.path.join(b.HERE, "build")
    out = {}
    for src in b.SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        with open(obj, "rb") as f:
            if b".hip_fatbin" not in f.read():
                continue
        out[src] = len(PK_OPSEL_RE.findall(code_object_isa(obj)))
    return out


def pk_audit(objs, objdir=None):
    """-> {object file: number of v_pk_{fma,mul,add}_f32 instructions in its device code}."""
    from etpnav_amd import build as b
    objdir = objdir or os.path.join(b.HERE, "build")
    return {o: len(PK_RE.findall(code_object_isa(os.path.join(objdir, o)))) for o in objs}


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
    nk, ni, m0bad = m0_audit()
    m0line = f"# LDS-DMA M0 audit: {ni} global_load_lds in {nk} kernels, {len(m0bad)} violations (every M0 write sits in front of its DMA or restores behind one; no compiler-generated M0 use)\n"
    lines = t.split("\n")
    lines.insert(3, m0line.rstrip("\n"))
    t = "\n".join(lines)
    if a.write:
        with open(a.write, "w") as f:
            f.write(t)
    bad = [r for r in rows if r["violation"]]
    print(t if not a.write else f"{len(rows)} kernels, {len(bad)} violations; M0 audit {ni} DMA instructions, {len(m0bad)} violations -> {a.write}")
    for src, k, txt in m0bad[:40]:
        print(f"M0 VIOLATION {src}: {k}: {txt}")
    for r in bad:
        print(f"VIOLATION {r['src']}: {r['full']}: {r['violation']} (agpr {r['agpr_count']}, scratch {r['private_segment_fixed_size']})")
    if a.strict and (bad or m0bad):
        sys.exit(1)


if __name__ == "__main__":
    main()
