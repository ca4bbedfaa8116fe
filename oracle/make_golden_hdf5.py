"""Write tests/golden/feats_small.hdf5 + feats_small_expected.npz with the REAL h5py, in the layout the reference's extractors write
(precompute_img_features/extract_rgb_features.py:111-123: one root-level dataset per "{scan}_{viewpoint}" key, shape [36, F], dtype
float32, compression='gzip', default libver) -- the fixture that pins etpnav_amd/hdf5_lite.py.  TEST INFRASTRUCTURE ONLY.

    /opt/conda/bin/python3.9 oracle/make_golden_hdf5.py        # the one interpreter of this image that has h5py (3.3.0)

The values are coarse (multiples of 1/8 with long runs) so that gzip keeps the file small; 14 keys make the root group's B-tree
span several symbol nodes; the 768-wide key is split into several chunks by h5py's automatic chunking (a chunk B-tree with more
than one entry, edge chunks included); one key uses shuffle + fletcher32 on top of gzip, one is contiguous, one float16; the
feature datasets carry the two string attributes the reference's extractors set (the reader must step over attribute messages).
"""
import os

import h5py
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "feats_small.hdf5")


def coarse(rng, shape):
    a = np.round(rng.standard_normal(shape) * 2) / 8
    a[:, ::3] = a[:, :1]                         # long runs: compressible
    return a.astype(np.float32)


def main():
    rng = np.random.default_rng(7)
    data = {}
    for i in range(10):
        data[f"scan{i % 3}_vp{i:03d}"] = coarse(rng, (36, 16))
    for i in range(1):
        data[f"scanW_vp{i}"] = coarse(rng, (36, 768))
    with h5py.File(OUT, "w") as f:
        for k, v in data.items():
            f.create_dataset(k, v.shape, dtype="float32", compression="gzip")      # extract_rgb_features.py:123
            f[k][...] = v
            scan, vp = k.split("_")
            f[k].attrs["scanId"] = scan                                            # :125-126 (extract_depth_features.py:119-120):
            f[k].attrs["viewpointId"] = vp                                         # attribute messages in every dataset's header
        data["extra_shuffled"] = coarse(rng, (36, 40))
        f.create_dataset("extra_shuffled", data=data["extra_shuffled"], compression="gzip", shuffle=True, fletcher32=True, chunks=(10, 16))
        data["extra_contiguous"] = coarse(rng, (5, 7))
        f.create_dataset("extra_contiguous", data=data["extra_contiguous"])
        data["extra_half"] = coarse(rng, (36, 8)).astype(np.float16)
        f.create_dataset("extra_half", data=data["extra_half"], compression="gzip")
        chunks = {k: f[k].chunks for k in data}
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "feats_small_expected.npz"), **data)
    print(OUT, os.path.getsize(OUT), "bytes; chunks:", {k: c for k, c in chunks.items() if k.startswith(("scanW", "extra"))})


if __name__ == "__main__":
    main()
