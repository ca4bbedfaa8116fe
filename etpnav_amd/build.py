"""Build libetpnav_hip.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m etpnav_amd.build [--force]

The shared object is git-ignored but travels with the gpurun snapshot; hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libetpnav_hip.so")
SOURCES = ["gemm.hip", "gemm_mm32.hip", "attn.hip", "attn_rows.hip", "norm.hip", "embed.hip", "optim.hip", "graph.hip", "planner.hip", "capi.hip", "graphrec.hip", "comm.hip"]
HEADERS = ["common.h", "kernels.h", "launch.h", "options.h", "gemm_shared.h", "gemm_tiles.h", os.path.join("..", "..", "include", "etpnav_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -munsafe-fp-atomics: fp32 atomicAdd lowers to the hardware global_atomic_add_f32 / ds_add_f32 instead of a CAS loop
# (all our atomic targets are ordinary coarse-grained device allocations).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result", "-munsafe-fp-atomics",
         "-Wno-return-type-c-linkage"]


# Row kernels (everything that is not a GEMM / attention tile kernel) are compiled WITHOUT the SLP vectorizer, i.e. without packed-fp32
# VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32).  Round 6 (profiles/r06_neighbour_bisect.txt): the
# panorama-embedding backward returned different results in 24 of 24 repetitions whenever wavefronts of a 128x128-tile GEMM shared its CU;
# the same source compiled with -fno-slp-vectorize: 0 of 24, with unchanged register counts.  These kernels are HBM- / latency-bound: the
# packed forms bought nothing measurable.  tools/kernel_resources.py::pk_audit fails the build if one of these objects contains a packed
# fp32 instruction again.
NO_PACKED_FP32 = ["embed.hip", "norm.hip", "optim.hip", "graph.hip"]
PER_SOURCE_FLAGS = {s: ["-fno-slp-vectorize", "-fno-vectorize"] for s in NO_PACKED_FP32}


def _mtime(p):
    return os.path.getmtime(p) if os.path.exists(p) else 0.0


def needs_build() -> bool:
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return _mtime(OUT) < max(_mtime(d) for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return OUT
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdr_t = max(_mtime(os.path.join(CSRC, h)) for h in HEADERS)

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        srcp = os.path.join(CSRC, src)
        if not force and _mtime(obj) >= max(_mtime(srcp), hdr_t):
            return obj
        cmd = [HIPCC, *FLAGS, *PER_SOURCE_FLAGS.get(src, []), "-c", srcp, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    audit_resources(verbose)
    return OUT


def audit_resources(verbose: bool = True):
    """Fail the build when a kernel without MFMAs was given AGPRs or any kernel uses scratch (tools/kernel_resources.py; DESIGN.md
    §3.6: the round-4 gradient corruption sat in AGPR-parked loads of such a kernel).  ETP_BUILD_AUDIT=0 skips it."""
    if os.environ.get("ETP_BUILD_AUDIT", "1") == "0":
        return
    root = os.path.dirname(HERE)
    tool = os.path.join(root, "tools", "kernel_resources.py")
    if not os.path.exists(tool):
        return
    import importlib.util
    spec = importlib.util.spec_from_file_location("kernel_resources", tool)
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    rows = kr.audit()
    bad = [r for r in rows if r["violation"]]
    if verbose:
        print(f"kernel resource audit: {len(rows)} kernels, {sum(1 for r in rows if int(r['agpr_count']) > 0)} with AGPRs (all matrix-core "
              f"families), {len(bad)} violations", flush=True)
    if bad:
        raise RuntimeError("kernel resource policy violated (tools/kernel_resources.py):\n" +
                           "\n".join(f"  {r['src']}: {r['full']}: {r['violation']}" for r in bad))
    npk = kr.pk_audit([s.replace(".hip", ".o") for s in NO_PACKED_FP32])
    if verbose:
        print(f"packed-fp32 audit: {sum(npk.values())} v_pk_*_f32 instructions in {', '.join(NO_PACKED_FP32)} (allowed: {kr.PK_ALLOW} per "
              f"object)", flush=True)
    over = {k: v for k, v in npk.items() if v > kr.PK_ALLOW}
    if over:
        raise RuntimeError(f"packed fp32 instructions in row-kernel objects (build.py NO_PACKED_FP32; DESIGN.md §3.6): {over}")
    sel = {k: v for k, v in kr.pk_opsel_audit().items() if v}
    if verbose:
        print(f"packed-fp32 op_sel audit: {sum(sel.values())} instructions with a low-half operand select in the library "
              f"(v_pk_*_f32 ... op_sel:[.,1]: wrong results beside another kernel's MFMAs, profiles/r06_pk_opsel_repro.txt)", flush=True)
    if sel:
        raise RuntimeError(f"packed fp32 instructions with op_sel low-half selects (DESIGN.md §3.6): {sel}")
    # the LDS-DMA statements write M0 behind the compiler's back: check the ISA it produced around them
    nk, ni, m0bad = kr.m0_audit()
    if verbose:
        print(f"LDS-DMA M0 audit: {ni} global_load_lds in {nk} kernels, {len(m0bad)} violations", flush=True)
    if m0bad:
        raise RuntimeError("M0 discipline around global_load_lds violated (tools/kernel_resources.py::m0_scan):\n" +
                           "\n".join(f"  {src}: {k}: {t}" for src, k, t in m0bad[:20]))


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
