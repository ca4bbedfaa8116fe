"""Precomputed view-feature store of the pre-training data pipeline (storage half).

Mirrors ``ReverieTextPathData.get_scanvp_feature`` (pretrain_src/pretrain_src/data/dataset.py:375-388): features live in a
file keyed ``"{scan}_{viewpoint}" -> float array [36, F]`` (one file for the RGB features, one for depth), are read on
demand and optionally cached in memory.

Backends (chosen by file suffix):
  * ``.hdf5`` / ``.h5``  the reference's files: through ``h5py`` where the interpreter has it, else through the built-in
    pure-Python reader of exactly the format subset those files use (``etpnav_amd/hdf5_lite.py``: v0 superblock, symbol-table
    root group, gzip-compressed chunked float32 datasets; validated against a file written by the real h5py);
  * ``.etpf``            a flat pack written by :func:`write_flat_pack` (raw little-endian float32 rows + a JSON index):
    ``np.memmap``-readable without any dependency; :func:`convert_hdf5` turns the reference's files into it where h5py exists.

MI355X-first addition: ``to_device`` places the WHOLE store in HBM as one tensor ``[N, 36, F]`` (R2R: ~10.6 k viewpoints x
36 x 768 x 4 B = 1.2 GB of 288 GB) so that batch assembly is a device-side ``index_select`` instead of per-viewpoint host
reads + H2D copies every step.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

FLAT_MAGIC = b"ETPF0001"


def _key(scan: str, viewpoint: str) -> str:
    return "%s_%s" % (scan, viewpoint)         # dataset.py:376


def write_flat_pack(path: str, items: Iterable[Tuple[str, np.ndarray]]):
    """items: (key, array [V, F] float) pairs, all with the same V and F.  Layout: magic | u64 header length | JSON header
    {"views", "feat", "keys"} | padding to 64 B | float32 rows in key order."""
    keys, rows, shape = [], [], None
    for k, a in items:
        a = np.ascontiguousarray(a, dtype=np.float32)
        if shape is None:
            shape = a.shape
        if a.shape != shape or a.ndim != 2:
            raise ValueError(f"{k}: shape {a.shape} differs from {shape}")
        keys.append(k); rows.append(a)
    if shape is None:
        raise ValueError("empty feature store")
    header = json.dumps({"views": int(shape[0]), "feat": int(shape[1]), "keys": keys}).encode()
    with open(path, "wb") as f:
        f.write(FLAT_MAGIC)
        f.write(np.uint64(len(header)).tobytes())
        f.write(header)
        pad = (-(len(FLAT_MAGIC) + 8 + len(header))) % 64
        f.write(b"\0" * pad)
        for a in rows:
            f.write(a.tobytes())


class _FlatPack:
    def __init__(self, path: str):
        with open(path, "rb") as f:
            if f.read(8) != FLAT_MAGIC:
                raise ValueError(f"{path}: not an ETPF flat pack")
            n = int(np.frombuffer(f.read(8), dtype=np.uint64)[0])
            h = json.loads(f.read(n).decode())
        self.views, self.feat, self.keys = h["views"], h["feat"], h["keys"]
        off = 16 + n
        off += (-off) % 64
        self.data = np.memmap(path, dtype=np.float32, mode="r", offset=off, shape=(len(self.keys), self.views, self.feat))
        self.index = {k: i for i, k in enumerate(self.keys)}

    def read(self, key: str) -> np.ndarray:
        return np.array(self.data[self.index[key]], dtype=np.float32)

    def all(self) -> np.ndarray:
        return self.data


class _Hdf5:
    """The reference's files.  h5py when the interpreter has it, else the built-in reader (etpnav_amd/hdf5_lite.py: the subset of the
    format h5py's default settings write for these files -- validated against a file written by the real h5py,
    tests/golden/feats_small.hdf5); a file outside that subset raises hdf5_lite.Hdf5Unsupported with the reason."""

    def __init__(self, path: str):
        self.path = path
        try:
            import h5py
            self._h5py = h5py
            with h5py.File(path, "r") as f:
                self.keys = list(f.keys())
        except ImportError:
            from . import hdf5_lite
            self._h5py = None
            self._lite = hdf5_lite.File(path)                 # index of the root group; datasets are decoded on demand
            self.keys = self._lite.keys()

    def read(self, key: str) -> np.ndarray:
        if self._h5py is None:
            return self._lite[key].astype(np.float32)
        with self._h5py.File(self.path, "r") as f:     # opened per read, as dataset.py:381-384 does (fork-safe for workers)
            return f[key][...].astype(np.float32)

    def all(self) -> np.ndarray:
        return np.stack([self.read(k) for k in self.keys])


def _open(path: str):
    ext = os.path.splitext(path)[1].lower()
    if ext in (".hdf5", ".h5"):
        return _Hdf5(path)
    if ext == ".etpf":
        return _FlatPack(path)
    raise ValueError(f"{path}: unknown feature-store format (want .hdf5/.h5 or .etpf)")


def convert_hdf5(src: str, dst: str):
    """One-time conversion of a reference HDF5 feature file to the flat pack (memmap-readable, no decompression per read)."""
    h = _Hdf5(src)
    write_flat_pack(dst, ((k, h.read(k)) for k in h.keys))


class FeatureStore:
    """``get_scanvp_feature(scan, viewpoint) -> (view_fts [V, F_img], dep_fts [V, F_dep])`` float32, as dataset.py:375-388."""

    def __init__(self, img_ft_file: str, dep_ft_file: Optional[str] = None, in_memory: bool = True):
        self.img, self.dep = _open(img_ft_file), (_open(dep_ft_file) if dep_ft_file else None)
        self.in_memory = in_memory
        self._feature_store: Dict[str, np.ndarray] = {}
        self._feature_store_depth: Dict[str, np.ndarray] = {}
        self._dev = None

    def get_scanvp_feature(self, scan: str, viewpoint: str):
        key = _key(scan, viewpoint)
        if self.in_memory and key in self._feature_store:
            return self._feature_store[key], self._feature_store_depth.get(key)
        view_fts = self.img.read(key)
        dep_fts = self.dep.read(key) if self.dep is not None else None
        if self.in_memory:
            self._feature_store[key] = view_fts
            if dep_fts is not None:
                self._feature_store_depth[key] = dep_fts
        return view_fts, dep_fts

    # ---- whole store resident in HBM ----
    def to_device(self, device) -> "FeatureStore":
        keys = list(self.img.keys)
        img = torch.from_numpy(np.ascontiguousarray(self.img.all())).to(device)
        dep = None
        if self.dep is not None:
            order = [self.dep.index[k] for k in keys] if hasattr(self.dep, "index") else None
            d = np.ascontiguousarray(self.dep.all())
            if order is not None:
                d = d[order]
            elif list(self.dep.keys) != keys:
                pos = {k: i for i, k in enumerate(self.dep.keys)}
                d = d[[pos[k] for k in keys]]
            dep = torch.from_numpy(d).to(device)
        self._dev = (img, dep, {k: i for i, k in enumerate(keys)})
        return self

    def gather(self, scanvps: Sequence[Tuple[str, str]]):
        """[(scan, viewpoint)] -> (rgb [n, V, F_img], depth [n, V, F_dep] or None) on the device, one index_select each."""
        if self._dev is None:
            raise RuntimeError("call to_device(device) first")
        img, dep, index = self._dev
        rows = torch.tensor([index[_key(s, v)] for s, v in scanvps], dtype=torch.long, device=img.device)
        return img.index_select(0, rows), (dep.index_select(0, rows) if dep is not None else None)
