"""Generate tests/golden/vp_inputs.npz from the REAL RLTrainer._vp_feature_variable (ss_trainer_ETP.py:308-342), cut out of
the trainer source and run on synthetic observations (oracle/ref_trainer_fns.py; build container only).

    python oracle/make_golden_vp.py
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_obs(seed=5, B=5, F=16, Fd=8):
    rng = np.random.RandomState(seed)
    g = torch.Generator().manual_seed(seed)
    obs = {"cand_img_idxes": [], "cand_rgb": [], "cand_depth": [], "cand_angle_fts": [], "cand_angles": []}
    for i in range(B):
        K = [1, 3, 5, 2, 4][i % 5]
        # candidate image indices (may repeat: two waypoints seen in the same panorama view)
        idx = rng.randint(0, 12, size=K)
        if i == 1:
            idx[1] = idx[0]
        obs["cand_img_idxes"].append(idx.astype(np.int64))
        obs["cand_rgb"].append(torch.randn(K, F, generator=g))
        obs["cand_depth"].append(torch.randn(K, Fd, generator=g))
        obs["cand_angle_fts"].append(torch.randn(K, 4, generator=g))
        obs["cand_angles"].append(list(rng.uniform(0, 6.28, size=K)))
    obs["pano_rgb"] = torch.randn(B, 12, F, generator=g)
    obs["pano_depth"] = torch.randn(B, 12, Fd, generator=g)
    obs["pano_angle_fts"] = torch.randn(12, 4, generator=g)
    return obs


def main():
    from oracle import ref_trainer_fns as rt
    fn = rt.extract(["_vp_feature_variable"])["_vp_feature_variable"]
    obs = make_obs()
    fake = types.SimpleNamespace(envs=types.SimpleNamespace(num_envs=len(obs["cand_rgb"])))
    with rt.shims():
        ref = fn(fake, obs)
    z = {f"out/{k}": v.numpy() for k, v in ref.items()}
    out = os.path.join(ROOT, "tests", "golden", "vp_inputs.npz")
    np.savez_compressed(out, **z)
    print({k: tuple(v.shape) for k, v in ref.items()})
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
