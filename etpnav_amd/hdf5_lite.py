"""Minimal pure-Python reader for the reference's HDF5 view-feature files (no h5py in this image's interpreter).

What the reference writes (precompute_img_features/extract_rgb_features.py:111-123, extract_depth_features.py:105-117) and reads
(pretrain_src/pretrain_src/data/dataset.py:375-388): one flat root group, one dataset per ``"{scan}_{viewpoint}"`` key, shape
``[36, F]`` float32, ``compression='gzip'`` -- i.e. with h5py's default ``libver='earliest'``:

    superblock version 0 / 1  ->  root symbol-table entry  ->  group B-tree (v1, node type 0) + local heap  ->  symbol nodes
    dataset object header version 1:  dataspace (v1 / v2), datatype (IEEE float 2 / 4 / 8 bytes or integer, little-endian),
    data layout version 3 (contiguous, compact, or chunked through a v1 B-tree of node type 1), filter pipeline (deflate, shuffle,
    fletcher32)

Exactly that subset is implemented, following the HDF5 File Format Specification 2.0 (sections III.A.1 v1 B-trees, III.B symbol
nodes, III.D local heaps, IV.A.1.a version-1 object headers, IV.A.2.b / .d / .i / .l / .r messages); anything else (libver='latest'
files: superblock 2 / 3, "OHDR" object headers, fractal-heap groups) raises ``Hdf5Unsupported`` with the reason.  The reader is
validated against a file written by the real h5py (tests/golden/feats_small.hdf5, generator oracle/make_golden_hdf5.py run with
the image's /opt/conda/bin/python3.9, the one interpreter here that has h5py).  Host-side I/O only: the features go to HBM once
(``FeatureStore.to_device``); nothing of this is on the planner's hot path.
"""
from __future__ import annotations

import mmap
import os
import struct
import zlib
from typing import Dict, List, Tuple

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF



def fletcher32(data: bytes) -> int:
    """H5_checksum_fletcher32 (HDF5 H5checksum.c): two running sums over BIG-endian 16-bit words, each folded with end-around carry
    (x = (x & 0xffff) + (x >> 16)), an odd trailing byte counted as its high half; result (sum2 << 16) | sum1.  Vectorised: with
    the fold the sums are residues modulo 65535 in which a non-zero multiple of 65535 reads 0xffff, so
    sum1 = fold(sum w_i), sum2 = fold(sum (n - i) w_i)."""
    n = len(data)
    w = np.frombuffer(data[:n - (n & 1)], dtype=">u2").astype(np.uint64)
    if n & 1:
        w = np.concatenate([w, np.array([data[-1] << 8], dtype=np.uint64)])
    m = w.size
    if m == 0:
        return 0
    # sum2 = sum (m - i) w_i.  Reducing a weight modulo 65535 does not change the residue; whether the folded value reads 0 or
    # 0xffff depends only on whether the exact sum is zero, i.e. whether any word is non-zero (all terms are non-negative)
    wt = np.arange(m, 0, -1, dtype=np.uint64) % np.uint64(65535)
    s1 = int(w.sum())
    s2 = 0
    for a in range(0, m, 1 << 20):                                    # products < 2^32, 2^20 per block: exact in uint64
        s2 += int((w[a:a + (1 << 20)] * wt[a:a + (1 << 20)]).sum())
    nonzero = s1 != 0

    def fold(x: int) -> int:
        r = x % 65535
        return r if r else (65535 if nonzero else 0)

    return (fold(s2) << 16) | fold(s1)


class Hdf5Unsupported(ValueError):
    pass


class Dataset:
    """One dataset of the root group: shape, numpy dtype and where its bytes are."""

    def __init__(self, f: "File", name: str, header_addr: int):
        self.f, self.name = f, name
        self.shape: Tuple[int, ...] = ()
        self.dtype = None
        self.layout = None           # ("contiguous", addr, size) | ("compact", bytes) | ("chunked", btree_addr, chunk_dims)
        self.filters: List[Tuple[int, Tuple[int, ...]]] = []
        for mtype, data in f._messages(header_addr):
            if mtype == 0x0001:
                self.shape = f._dataspace(data)
            elif mtype == 0x0003:
                self.dtype = f._datatype(data)
            elif mtype == 0x0008:
                self.layout = f._layout(data)
            elif mtype == 0x000B:
                self.filters = f._filters(data)
        if self.dtype is None or self.layout is None:
            raise Hdf5Unsupported(f"{name}: not a dataset this reader understands (no datatype / layout message)")

    def read(self) -> np.ndarray:
        f, n = self.f, int(np.prod(self.shape)) if self.shape else 1
        kind = self.layout[0]
        if kind == "contiguous":
            _, addr, size = self.layout
            if addr == UNDEF:
                return np.zeros(self.shape, dtype=self.dtype)
            return np.frombuffer(f._read(addr, n * self.dtype.itemsize), dtype=self.dtype).reshape(self.shape).copy()
        if kind == "compact":
            return np.frombuffer(self.layout[1][:n * self.dtype.itemsize], dtype=self.dtype).reshape(self.shape).copy()
        _, btree, cdims = self.layout
        out = np.zeros(self.shape, dtype=self.dtype)
        if btree == UNDEF:
            return out
        rank = len(self.shape)
        for size, mask, offs, addr in f._chunks(btree, rank):
            raw = f._read(addr, size)
            for k in range(len(self.filters) - 1, -1, -1):           # the pipeline is undone last filter first
                if mask & (1 << k):
                    continue                                          # this filter was skipped for this chunk
                fid, cd = self.filters[k]
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:                                        # shuffle: byte planes -> elements
                    es = cd[0] if cd else self.dtype.itemsize
                    a = np.frombuffer(raw, dtype=np.uint8)
                    m = a.size // es
                    raw = a[:m * es].reshape(es, m).T.tobytes() + a[m * es:].tobytes()
                elif fid == 3:                                        # fletcher32: checksum in the last four bytes
                    if len(raw) < 4:
                        raise Hdf5Unsupported(f"{self.name}: chunk at {addr} is shorter than its fletcher32 checksum")
                    stored = int.from_bytes(raw[-4:], "little")
                    raw = raw[:-4]
                    want = fletcher32(raw)
                    # H5Zfletcher32.c also accepts the byte-swapped form of each half (files written by libraries <= 1.6.2)
                    swapped = ((want & 0x00ff00ff) << 8) | ((want >> 8) & 0x00ff00ff)
                    if stored not in (want, swapped):
                        raise Hdf5Unsupported(f"{self.name}: fletcher32 mismatch in the chunk at {addr} "
                                              f"(stored {stored:#010x}, computed {want:#010x}): corrupted file")
                else:
                    raise Hdf5Unsupported(f"{self.name}: filter id {fid} is not supported (deflate, shuffle, fletcher32 are)")
            want_bytes = int(np.prod(cdims)) * self.dtype.itemsize
            if len(raw) != want_bytes:                                # a damaged chunk that still inflates
                raise Hdf5Unsupported(f"{self.name}: chunk at {addr} decodes to {len(raw)} bytes, its shape {tuple(cdims)} needs "
                                      f"{want_bytes}: corrupted file")
            chunk = np.frombuffer(raw, dtype=self.dtype).reshape(cdims)
            sel_out = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, self.shape))
            sel_in = tuple(slice(0, so.stop - so.start) for so in sel_out)
            out[sel_out] = chunk[sel_in]
        return out


class File:
    """``File(path).keys()`` / ``File(path)[key] -> np.ndarray`` for the flat root group of a feature file."""

    def __init__(self, path: str):
        self.path = path
        # The real feature files are gigabytes: map the file instead of reading it (pages come in as datasets are decoded, are
        # shared between data-loader workers through the page cache, and nothing is copied at open).
        with open(path, "rb") as fh:
            if os.fstat(fh.fileno()).st_size == 0:
                raise Hdf5Unsupported(f"{path}: empty file")
            self.buf = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
        try:
            self._open()
        except (IndexError, struct.error) as e:                          # a structure that points outside the file
            raise Hdf5Unsupported(f"{path}: damaged or truncated file ({e})") from e

    def _open(self):
        path = self.path
        base = self._find_superblock()
        ver = self.buf[base + 8]
        if ver not in (0, 1):
            raise Hdf5Unsupported(f"{path}: superblock version {ver} (libver='latest' files are not supported: rewrite the file with "
                                  f"h5py's default libver or convert it with etpnav_amd.features.convert_hdf5 where h5py exists)")
        self.O, self.L = self.buf[base + 13], self.buf[base + 14]
        if (self.O, self.L) != (8, 8):
            raise Hdf5Unsupported(f"{path}: {self.O}-byte offsets / {self.L}-byte lengths (only 8 / 8 is supported)")
        p = base + 24 + (4 if ver == 1 else 0)
        self.base_addr = self._u(p, 8)
        p += 4 * 8                                                    # base, free-space info, end of file, driver info
        # root group symbol-table entry: link name offset, object header address, cache type, reserved, scratch pad
        root_header = self._u(p + 8, 8)
        cache_type = self._u(p + 16, 4)
        if cache_type == 1:
            btree, heap = self._u(p + 24, 8), self._u(p + 32, 8)
        else:
            btree = heap = None
            for mtype, data in self._messages(root_header):
                if mtype == 0x0011:
                    btree, heap = struct.unpack_from("<QQ", data, 0)
            if btree is None:
                raise Hdf5Unsupported(f"{path}: the root group has no symbol table (new-style groups are not supported)")
        self._index: Dict[str, int] = {}
        heap_data = self._heap_data(heap)
        for name_off, header in self._group_entries(btree):
            end = self.buf.find(b"\0", heap_data + name_off)
            if end < 0:
                raise Hdf5Unsupported(f"{path}: unterminated link name in the root group's heap")
            self._index[self.buf[heap_data + name_off:end].decode()] = header
        self._cache: Dict[str, Dataset] = {}

    # a File travels to data-loader workers by path (a mapping cannot be pickled); the worker maps the file again
    def __getstate__(self):
        return {"path": self.path}

    def __setstate__(self, state):
        self.__init__(state["path"])

    # ---- public ----
    def keys(self) -> List[str]:
        return list(self._index)

    def __contains__(self, key):
        return key in self._index

    def dataset(self, key: str) -> Dataset:
        if key not in self._cache:
            header = self._index[key]                                    # KeyError for an unknown key, as h5py
            try:
                self._cache[key] = Dataset(self, key, header)
            except (IndexError, struct.error) as e:                      # a structure that points outside the file
                raise Hdf5Unsupported(f"{self.path}: {key}: damaged or truncated file ({e})") from e
        return self._cache[key]

    def __getitem__(self, key: str) -> np.ndarray:
        try:
            return self.dataset(key).read()
        except (IndexError, struct.error) as e:
            raise Hdf5Unsupported(f"{self.path}: {key}: damaged or truncated file ({e})") from e

    # ---- primitives ----
    def _read(self, addr: int, n: int) -> bytes:
        a = self.base_addr + addr
        if a + n > len(self.buf):
            raise Hdf5Unsupported(f"{self.path}: truncated file (wanted {n} bytes at {a})")
        return self.buf[a:a + n]

    def _u(self, pos: int, n: int) -> int:
        if pos < 0 or pos + n > len(self.buf):                      # a slice past EOF would decode fewer bytes into a wrong value
            raise Hdf5Unsupported(f"{self.path}: truncated file (wanted {n} bytes at {pos})")
        return int.from_bytes(self.buf[pos:pos + n], "little")

    def _find_superblock(self) -> int:
        pos = 0
        while pos < len(self.buf):                                    # at 0, 512, 1024, 2048, ... (spec II.A)
            if self.buf[pos:pos + 8] == SIGNATURE:
                return pos
            pos = 512 if pos == 0 else pos * 2
        raise Hdf5Unsupported(f"{self.path}: no HDF5 signature")

    def _messages(self, addr: int):
        """(type, data) of every message of a version-1 object header, continuation blocks included."""
        a = self.base_addr + addr
        if self.buf[a:a + 4] == b"OHDR":
            raise Hdf5Unsupported(f"{self.path}: version-2 object headers (libver='latest') are not supported")
        if self.buf[a] != 1:
            raise Hdf5Unsupported(f"{self.path}: object header version {self.buf[a]} at {addr}")
        nmsg = self._u(a + 2, 2)
        size = self._u(a + 8, 4)
        blocks = [(a + 16, size)]                                     # 12 bytes of prefix + 4 of alignment padding
        seen = 0
        while blocks and seen < nmsg:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and seen < nmsg:
                mtype, msize = self._u(p, 2), self._u(p + 2, 2)
                data = self.buf[p + 8:p + 8 + msize]
                p += 8 + msize
                seen += 1
                if mtype == 0x0010:                                   # continuation: offset, length
                    off, ln = struct.unpack_from("<QQ", data, 0)
                    blocks.append((self.base_addr + off, ln))
                else:
                    yield mtype, data

    def _heap_data(self, heap_addr: int) -> int:
        a = self.base_addr + heap_addr
        if self.buf[a:a + 4] != b"HEAP":
            raise Hdf5Unsupported(f"{self.path}: bad local heap signature")
        return self.base_addr + self._u(a + 24, 8)

    def _group_entries(self, btree_addr: int):
        """(link name offset, object header address) of every entry below a group B-tree node."""
        a = self.base_addr + btree_addr
        if self.buf[a:a + 4] == b"SNOD":
            n = self._u(a + 6, 2)
            for i in range(n):
                e = a + 8 + i * 40
                yield self._u(e, 8), self._u(e + 8, 8)
            return
        if self.buf[a:a + 4] != b"TREE" or self.buf[a + 4] != 0:
            raise Hdf5Unsupported(f"{self.path}: bad group B-tree node at {btree_addr}")
        used = self._u(a + 6, 2)
        p = a + 24                                                    # signature, type, level, entries, left, right
        for i in range(used):                                         # key_i (L), child_i (O), ..., key_used
            child = self._u(p + 8 + i * 16, 8)
            yield from self._group_entries(child)

    def _chunks(self, btree_addr: int, rank: int):
        """(stored size, filter mask, element offsets, address) of every chunk below a chunk B-tree node."""
        a = self.base_addr + btree_addr
        if self.buf[a:a + 4] != b"TREE" or self.buf[a + 4] != 1:
            raise Hdf5Unsupported(f"{self.path}: bad chunk B-tree node at {btree_addr}")
        level, used = self.buf[a + 5], self._u(a + 6, 2)
        ksize = 8 + 8 * (rank + 1)
        p = a + 24
        for i in range(used):
            k = p + i * (ksize + 8)
            size, mask = self._u(k, 4), self._u(k + 4, 4)
            offs = tuple(self._u(k + 8 + 8 * d, 8) for d in range(rank))
            child = self._u(k + ksize, 8)
            if level == 0:
                yield size, mask, offs, child
            else:
                yield from self._chunks(child, rank)

    # ---- messages ----
    def _dataspace(self, d: bytes) -> Tuple[int, ...]:
        ver, rank = d[0], d[1]
        p = 8 if ver == 1 else 4
        return tuple(int.from_bytes(d[p + 8 * i:p + 8 * i + 8], "little") for i in range(rank))

    def _datatype(self, d: bytes) -> np.dtype:
        cls, bits0, size = d[0] & 0x0F, d[1], int.from_bytes(d[4:8], "little")
        if bits0 & 1:
            raise Hdf5Unsupported(f"{self.path}: big-endian data")
        if cls == 1 and size in (2, 4, 8):
            return np.dtype({2: "<f2", 4: "<f4", 8: "<f8"}[size])
        if cls == 0 and size in (1, 2, 4, 8):
            return np.dtype(("<i" if bits0 & 8 else "<u") + str(size))
        raise Hdf5Unsupported(f"{self.path}: datatype class {cls} of {size} bytes")

    def _layout(self, d: bytes):
        ver, cls = d[0], d[1]
        if ver != 3:
            raise Hdf5Unsupported(f"{self.path}: data layout message version {ver}")
        if cls == 1:
            return ("contiguous",) + struct.unpack_from("<QQ", d, 2)
        if cls == 0:
            n = int.from_bytes(d[2:4], "little")
            return ("compact", bytes(d[4:4 + n]))
        if cls == 2:
            ndim = d[2]
            btree = int.from_bytes(d[3:11], "little")
            dims = tuple(int.from_bytes(d[11 + 4 * i:15 + 4 * i], "little") for i in range(ndim))
            return ("chunked", btree, dims[:-1])                      # the last "dimension" is the element size
        raise Hdf5Unsupported(f"{self.path}: layout class {cls}")

    def _filters(self, d: bytes):
        ver, n = d[0], d[1]
        p = 8 if ver == 1 else 2
        out = []
        for _ in range(n):
            fid = int.from_bytes(d[p:p + 2], "little")
            if ver == 1 or fid >= 256:
                nlen = int.from_bytes(d[p + 2:p + 4], "little"); p += 4
            else:
                nlen = 0; p += 2
            ncd = int.from_bytes(d[p + 2:p + 4], "little"); p += 4       # flags (2), number of client data values (2)
            p += (nlen + 7) // 8 * 8 if ver == 1 else nlen
            cd = tuple(int.from_bytes(d[p + 4 * i:p + 4 * i + 4], "little") for i in range(ncd))
            p += 4 * ncd
            if ver == 1 and ncd % 2:
                p += 4
            out.append((fid, cd))
        return out
