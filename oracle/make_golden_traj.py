"""Generate tests/golden/traj_agg.npz from the REAL GlobalMapEncoder._aggregate_gmap_features
(pretrain_src/pretrain_src/model/vilmodel.py:585-619), cut out of the source with `ast` and run unchanged with the
reference's own gen_seq_masks / pad_tensors_wgrad restated from pretrain_src/model/ops.py (that file's module-level
imports pull in the transformer package; the two helpers are 10-line functions identical to vlnce_baselines/common/ops.py,
which IS imported for them).  Build container only.

    python oracle/make_golden_traj.py
"""
import ast
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC = "/root/reference/pretrain_src/pretrain_src/model/vilmodel.py"


def real_aggregate():
    from oracle import ref_trainer_fns as rt
    ops = rt._load_ops()
    tree = ast.parse(open(SRC).read())
    fn = [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "_aggregate_gmap_features"][0]
    ns = {"torch": torch, "gen_seq_masks": ops.gen_seq_masks, "pad_tensors_wgrad": ops.pad_tensors_wgrad}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), SRC, "exec"), ns)
    return ns["_aggregate_gmap_features"]


def make_case(seed=11, B=4, V=6, H=8):
    """Synthetic trajectories: revisits, nodes seen as candidates several times before being visited, ragged views."""
    rng = np.random.RandomState(seed)
    lens, vpids, cands, gvps, embeds = [], [], [], [], []
    for i in range(B):
        T = [1, 3, 4, 5][i % 4]
        names = [f"n{i}_{k}" for k in range(12)]
        path = [names[0]]
        for t in range(1, T):
            path.append(names[rng.randint(0, 6)] if rng.rand() < 0.3 else names[t])
        ep_lens, ep_c = [], []
        for t in range(T):
            n = rng.randint(3, V + 1)
            ep_lens.append(n)
            k = rng.randint(1, n)                       # the first k views are candidates
            if i == 1 and t == 1:
                k = min(n + 1, V)                       # one candidate slot beyond this step's valid views (a zero row)
            ep_c.append([names[rng.randint(0, 12)] for _ in range(k)])
        seen = []
        for t in range(T):
            for vp in [path[t]] + ep_c[t]:
                if vp not in seen:
                    seen.append(vp)
        lens.append(torch.tensor(ep_lens)); vpids.append(path); cands.append(ep_c); gvps.append([None] + seen)
        embeds.append(torch.from_numpy(rng.standard_normal((T, V, H)).astype(np.float32)))
    return embeds, lens, vpids, cands, gvps


def main():
    fn = real_aggregate()
    embeds, lens, vpids, cands, gvps = make_case()
    out = fn(None, embeds, lens, vpids, cands, gvps)
    path = os.path.join(ROOT, "tests", "golden", "traj_agg.npz")
    np.savez_compressed(path, out=out.numpy())
    print(tuple(out.shape), "wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
