"""ctypes binding of libetpnav_hip.so (the C ABI declared in include/etpnav_hip.h).

The prototypes are parsed from the header itself so the Python side cannot drift from the C side.
There is NO fallback: if the shared object is missing or a symbol cannot be resolved, importing a
compute path raises.  (Build it with ``python -m etpnav_amd.build``.)
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HEADER = os.path.join(ROOT, "include", "etpnav_hip.h")
# ETP_LIB selects another build of the same library for same-box A/B measurements (e.g. the previous round's binary kept
# beside the new one); symbols that build lacks are skipped, everything else is bound as declared.
LIB_PATH = os.environ.get("ETP_LIB") or os.path.join(HERE, "libetpnav_hip.so")

ETP_F32, ETP_BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_RELU, ACT_GELU_BWD, ACT_RELU_BWD, ACT_GELU_SAVEGRAD, ACT_MUL_Z = 0, 1, 2, 3, 4, 5, 6


class EtpError(RuntimeError):
    pass


# ---- structs (must mirror include/etpnav_hip.h) ------------------------------------------------
class GemmDesc(ctypes.Structure):
    _fields_ = [("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("C", ctypes.c_void_p),
                ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32),
                ("lda", ctypes.c_int64), ("ldb", ctypes.c_int64), ("ldc", ctypes.c_int64),
                ("trans_a", ctypes.c_int32), ("trans_b", ctypes.c_int32),
                ("dtype", ctypes.c_int32), ("c_dtype", ctypes.c_int32),
                ("batch", ctypes.c_int32), ("batch_inner", ctypes.c_int32),
                ("sAo", ctypes.c_int64), ("sAi", ctypes.c_int64), ("sBo", ctypes.c_int64),
                ("sBi", ctypes.c_int64), ("sCo", ctypes.c_int64), ("sCi", ctypes.c_int64),
                ("ksplit", ctypes.c_int32), ("alpha", ctypes.c_float),
                ("bias", ctypes.c_void_p), ("R", ctypes.c_void_p), ("ldr", ctypes.c_int64),
                ("Z", ctypes.c_void_p), ("ldz", ctypes.c_int64),
                ("act", ctypes.c_int32), ("out_mode", ctypes.c_int32), ("a_colsum", ctypes.c_void_p)]


class AttnDesc(ctypes.Structure):
    _fields_ = [("dtype", ctypes.c_int32), ("B", ctypes.c_int32), ("heads", ctypes.c_int32),
                ("Lq", ctypes.c_int32), ("Lk", ctypes.c_int32), ("ldS", ctypes.c_int32),
                ("Q", ctypes.c_void_p), ("ldq", ctypes.c_int64),
                ("K", ctypes.c_void_p), ("ldk", ctypes.c_int64),
                ("V", ctypes.c_void_p), ("ldv", ctypes.c_int64),
                ("P", ctypes.c_void_p), ("ctx", ctypes.c_void_p), ("ldc", ctypes.c_int64),
                ("keymask", ctypes.c_void_p), ("mask_mode", ctypes.c_int32),
                ("dist", ctypes.c_void_p), ("sp_w", ctypes.c_void_p), ("sp_b", ctypes.c_void_p),
                ("alpha", ctypes.c_float)]


class AttnBwdDesc(ctypes.Structure):
    _fields_ = [("f", AttnDesc), ("dctx", ctypes.c_void_p), ("ldd", ctypes.c_int64),
                ("dP", ctypes.c_void_p),
                ("dQ", ctypes.c_void_p), ("lddq", ctypes.c_int64),
                ("dK", ctypes.c_void_p), ("lddk", ctypes.c_int64),
                ("dV", ctypes.c_void_p), ("lddv", ctypes.c_int64),
                ("d_sp_w", ctypes.c_void_p), ("d_sp_b", ctypes.c_void_p)]


class Config(ctypes.Structure):
    _fields_ = [("hidden", ctypes.c_int32), ("heads", ctypes.c_int32), ("inter", ctypes.c_int32),
                ("n_l", ctypes.c_int32), ("n_p", ctypes.c_int32), ("n_x", ctypes.c_int32),
                ("vocab", ctypes.c_int32), ("max_pos", ctypes.c_int32), ("type_vocab", ctypes.c_int32),
                ("img_feat", ctypes.c_int32), ("dep_feat", ctypes.c_int32), ("ang_feat", ctypes.c_int32),
                ("max_steps", ctypes.c_int32), ("use_depth", ctypes.c_int32), ("use_sprels", ctypes.c_int32),
                ("ln_eps", ctypes.c_float), ("dtype", ctypes.c_int32), ("use_lang2visn", ctypes.c_int32)]


class ParamInfo(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 128), ("ndim", ctypes.c_int32), ("shape", ctypes.c_int64 * 2),
                ("offset", ctypes.c_int64)]


class AdamwCfg(ctypes.Structure):
    _fields_ = [("lr", ctypes.c_float), ("beta1", ctypes.c_float), ("beta2", ctypes.c_float), ("eps", ctypes.c_float),
                ("weight_decay", ctypes.c_float), ("step", ctypes.c_int32), ("hf_style", ctypes.c_int32),
                ("correct_bias", ctypes.c_int32), ("grad_scale", ctypes.c_float), ("max_norm", ctypes.c_float)]


class ProfEntry(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 96), ("launches", ctypes.c_int64), ("ms", ctypes.c_double),
                ("flops", ctypes.c_double), ("bytes", ctypes.c_double)]


# ---- header parsing --------------------------------------------------------------------------
_SCALARS = {"int": ctypes.c_int, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64,
            "float": ctypes.c_float,
            "etp_stream_t": ctypes.c_void_p}


def _ctype_of(decl: str):
    d = decl.strip()
    if "*" in d:
        if re.match(r"^(const\s+)?char\s*\*$", d.replace(" *", "*").replace("* ", "*")):
            return ctypes.c_char_p
        return ctypes.c_void_p
    d = re.sub(r"\bconst\b", "", d).strip()
    base = d.split()[0]
    if base == "void":
        return None
    if base not in _SCALARS:
        raise EtpError(f"cannot map C type {decl!r}")
    return _SCALARS[base]


def parse_header(path: str = HEADER) -> Dict[str, Tuple[object, List[object]]]:
    """-> {function name: (restype, [argtypes])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", src, flags=re.S)
    src = re.sub(r"^\s*#.*$", " ", src, flags=re.M)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(etp_\w+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef"):
            continue
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                # strip the parameter name (last identifier), keep the type
                a_type = re.sub(r"\b\w+$", "", a).strip() if not a.endswith("*") else a
                argtypes.append(_ctype_of(a_type if a_type else a))
        protos[name] = (_ctype_of(ret), argtypes)
    return protos


_lib = None
_protos = None


def lib() -> ctypes.CDLL:
    """Load the shared object (once) and attach prototypes.  Raises if it is not built."""
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EtpError(f"{LIB_PATH} is missing: the HIP extension is not built "
                       f"(run `python -m etpnav_amd.build`); there is no CPU fallback")
    # torch bundles its own libamdhip64.so.7; load it FIRST so the extension binds to the same HIP runtime that
    # owns torch's device pointers and streams (two runtimes in one process cannot see each other's allocations).
    import torch  # noqa: F401
    if torch.cuda.is_available():
        torch.cuda.init()
    L = ctypes.CDLL(LIB_PATH)
    _protos = parse_header()
    missing = []
    for name, (res, args) in _protos.items():
        if os.environ.get("ETP_LIB") and not hasattr(L, name):
            missing.append(name)
            if res is ctypes.c_int and name.startswith(("etp_stamp", "etp_prof", "etp_gemm_probe")):
                setattr(L, name, lambda *a, **k: 0)      # measurement aids an older A/B build lacks: no-ops
            continue
        fn = getattr(L, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if missing:                        # ETP_LIB builds only (A/B measurement): say what the selected build lacks, once
        import sys
        print(f"[etpnav_amd] ETP_LIB={LIB_PATH}: {len(missing)} declared symbols are absent from this build and stay unbound: "
              f"{', '.join(missing[:6])}{' ...' if len(missing) > 6 else ''}", file=sys.stderr)
    _lib = L
    return L


def declared_symbols() -> List[str]:
    return sorted(parse_header().keys())


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().etp_last_error()
        raise EtpError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def ptr(t) -> int:
    """Device (or host) address of a torch tensor; None -> NULL."""
    if t is None:
        return None
    assert t.is_contiguous(), "C ABI needs contiguous tensors"
    return t.data_ptr()


def set_option(name: str, value=None) -> None:
    """One run-time switch of the library (csrc/options.h; `etp_option_set`).  ``value`` None / "" = unset.  The library reads the
    environment (ETP_<NAME>) only once, at its first lookup: after that a switch changes through this call, not through os.environ."""
    v = None if value is None or value == "" else str(value).encode()
    check(lib().etp_option_set(name.encode(), v), f"etp_option_set({name})")


def get_option(name: str):
    buf = ctypes.create_string_buffer(64)
    n = lib().etp_option_get(name.encode(), buf, 64)
    if n < 0:
        raise EtpError(f"unknown library switch {name!r}")
    return buf.value.decode() if n > 0 else None


def options() -> Dict[str, str]:
    """every switch of the library that is set, however it got there (environment at load, or set_option)"""
    L = lib()
    n = L.etp_option_list(None, 0)
    buf = ctypes.create_string_buffer(n + 1)
    L.etp_option_list(buf, n + 1)
    return dict(line.split("=", 1) for line in buf.value.decode().splitlines() if "=" in line)


class option:
    """``with _lib.option("MM32", 64): ...`` -- set a switch and restore what was there (tests, A/B legs)."""

    def __init__(self, name: str, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = get_option(self.name)
        set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.old)
        return False


def force_gemm_tile(tile: str = "") -> None:
    """Tuning aid: force a tile class of gemm.hip's kernels for the following launches ("" / "auto" = the library's own choice).
    The mm32 family is consulted BEFORE gemm.hip's tile choice (csrc/gemm_mm32.hip::mm32_class), so forcing a gemm.hip class
    also switches mm32 off -- otherwise an eligible bf16 product would silently keep running the mm32 kernel."""
    tile = "" if tile == "auto" else tile
    set_option("GEMM_TILE", tile)
    set_option("MM32", "0" if tile else None)
