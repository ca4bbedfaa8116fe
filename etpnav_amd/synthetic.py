"""Seeded synthetic planner inputs with the shapes/dtypes of the reference's batches.

There are no R2R-CE features or checkpoints offline, so benchmarks and smoke tests use these: token ids uniform
in [1000, vocab-1), N(0,1) view features, angle features [sin h, cos h, sin e, cos e] on the 12x3 panorama grid
(pretrain_src/.../data/common.py:51-68), candidate views first (ss_trainer_ETP.py:308-342), symmetric pair distances
with a zero [stop] row/column (ss_trainer_ETP.py:371-387).
"""
from __future__ import annotations

import math
from typing import Dict

import torch


def make_batch(vocab_size: int, image_feat_size: int, depth_feat_size: int, B: int, L: int, V: int, G: int,
               seed: int = 1234, ragged: bool = False, n_cand: int = 4) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1000, vocab_size - 1, (B, L), generator=g)
    if ragged:
        tl = torch.randint(max(L // 2, 1), L + 1, (B,), generator=g); tl[0] = L
        vl = torch.randint(max(V - 6, 1), V + 1, (B,), generator=g); vl[0] = V
        gl = torch.randint(max(G // 2, 3), G + 1, (B,), generator=g); gl[0] = G
    else:
        tl, vl, gl = torch.full((B,), L), torch.full((B,), V), torch.full((B,), G)
    ar = lambda n: torch.arange(n)[None, :]
    tmask, vmask, gmask = ar(L) < tl[:, None], ar(V) < vl[:, None], ar(G) < gl[:, None]
    ids = ids * tmask
    rgb = torch.randn(B, V, image_feat_size, generator=g) * vmask[..., None]
    dep = torch.randn(B, V, depth_feat_size, generator=g) * vmask[..., None]
    v = torch.arange(V)
    heading = 2 * math.pi * (v % 12) / 12
    elev = ((v // 12) % 3 - 1) * (math.pi / 6)
    loc = torch.stack([heading.sin(), heading.cos(), elev.sin(), elev.cos()], -1)[None].repeat(B, 1, 1) * vmask[..., None]
    nav = torch.zeros(B, V, dtype=torch.long); nav[:, :n_cand] = 1
    nav = nav * vmask
    step_ids = torch.zeros(B, G, dtype=torch.long); step_ids[:, 1] = 1
    pos = torch.randn(B, G, 7, generator=g) * gmask[..., None]
    d = torch.rand(B, G, G, generator=g)
    d = (d + d.transpose(1, 2)) * 0.5
    d[:, 0, :] = 0; d[:, :, 0] = 0
    d = d * (1 - torch.eye(G))[None] * gmask[:, :, None] * gmask[:, None, :]
    visited = torch.zeros(B, G, dtype=torch.bool); visited[:, 1] = True
    labels = torch.full((B,), 2, dtype=torch.long)
    return {"txt_ids": ids, "txt_masks": tmask, "rgb_fts": rgb, "dep_fts": dep, "loc_fts": loc, "nav_types": nav,
            "view_lens": vl, "gmap_step_ids": step_ids, "gmap_pos_fts": pos, "gmap_masks": gmask,
            "gmap_visited_masks": visited, "gmap_pair_dists": d, "labels": labels}


def make_sap_batch(vocab_size: int, image_feat_size: int, depth_feat_size: int, B: int, L: int, T: int, V: int = 36,
                   n_cand: int = 4, seed: int = 1234, ragged: bool = False) -> dict:
    """One pre-training SAP step (pretrain_cmt.py:223-283): per episode a T-step trajectory with one V-view panorama per
    step (the first n_cand views are candidates: the next node of the path + new ghost nodes), the graph of everything
    seen so far, and the index of a ghost as the action label.  Tensors are flattened over steps as the reference's
    collate does (tasks.py:322-364): rgb_fts [B*T, V, F] etc.; the "traj" entry carries the id lists the aggregation needs."""
    g = torch.Generator().manual_seed(seed)
    base = make_batch(vocab_size, image_feat_size, depth_feat_size, B * T, L, V, 4, seed=seed, ragged=ragged, n_cand=n_cand)
    tl = torch.randint(max(L // 2, 1), L + 1, (B,), generator=g) if ragged else torch.full((B,), L)
    tl[0] = L
    ids = torch.randint(1000, vocab_size - 1, (B, L), generator=g)
    tmask = torch.arange(L)[None, :] < tl[:, None]
    vpids, cands, gvps, lens = [], [], [], []
    for i in range(B):
        path = [f"e{i}_n{t}" for t in range(T)]
        ep_c, seen = [], [path[0]]
        for t in range(T):
            c = []
            for j in range(n_cand):
                if j == 0 and t + 1 < T:
                    c.append(path[t + 1])                       # the next node is first seen as a candidate
                elif j == 1 and t > 0:
                    c.append(path[t - 1])                       # looking back at a visited node
                else:
                    c.append(f"e{i}_g{t}_{j % 3}")              # ghosts; some are re-seen from the next step
            if t > 0 and n_cand > 2:
                c[2] = f"e{i}_g{t - 1}_2"                       # same ghost seen from two steps -> mean of two views
            ep_c.append(c)
            for vp in [path[t]] + c:
                if vp not in seen:
                    seen.append(vp)
        vpids.append(path); cands.append(ep_c); gvps.append([None] + seen)
        lens.append([int(x) for x in base["view_lens"][i * T:(i + 1) * T]])
    G = max(len(x) for x in gvps)
    gl = torch.tensor([len(x) for x in gvps])
    gmask = torch.arange(G)[None, :] < gl[:, None]
    step_ids = torch.zeros(B, G, dtype=torch.long)
    visited = torch.zeros(B, G, dtype=torch.bool)
    labels = torch.zeros(B, dtype=torch.long)
    for i in range(B):
        for gidx, vp in enumerate(gvps[i]):
            if vp in vpids[i]:
                step_ids[i, gidx] = vpids[i].index(vp) + 1
                visited[i, gidx] = True
        ghosts = [k for k in range(1, len(gvps[i])) if not visited[i, k]]
        labels[i] = ghosts[i % len(ghosts)]
    pos = torch.randn(B, G, 7, generator=g) * gmask[..., None]
    d = torch.rand(B, G, G, generator=g)
    d = (d + d.transpose(1, 2)) * 0.5
    d[:, 0, :] = 0; d[:, :, 0] = 0
    d = d * (1 - torch.eye(G))[None] * gmask[:, :, None] * gmask[:, None, :]
    out = {k: base[k] for k in ("rgb_fts", "dep_fts", "loc_fts", "nav_types", "view_lens")}
    out.update({"txt_ids": ids * tmask, "txt_masks": tmask, "gmap_step_ids": step_ids, "gmap_pos_fts": pos, "gmap_masks": gmask,
                "gmap_visited_masks": visited, "gmap_pair_dists": d, "labels": labels,
                "traj": {"traj_step_lens": [T] * B, "traj_vp_lens": lens, "traj_vpids": vpids, "traj_cand_vpids": cands,
                         "gmap_vpids": gvps}})
    return out


# ---- synthetic rollout driver for graph-input assembly (shared by the golden generators, the tests and tools/graph_probe.py)
import numpy as np  # noqa: E402


def simulate_rollout(GraphCls, seed, steps, merge_ghost=True, embed_dim=8, rows_mode=False):
    """Drive a GraphMap-like class the way the rollout does (ss_trainer_ETP.py:842-871,977): identify_node ->
    update_graph -> move to one of the ghosts (delete_ghost) -> repeat.  `cur_ori` is passed as a scalar heading.
    Every node / candidate view gets a random embedding row appended to `store`; the class receives the row tensors
    (reference behaviour) or, with rows_mode, the row indices (GraphMapLite's device-store mode).
    Returns (gmap, cur_vp, cur_pos, cur_heading, store [rows, embed_dim] float32)."""
    rng = np.random.RandomState(seed)
    erng = np.random.RandomState(seed + 1000)
    store = []

    def new_row():
        store.append(erng.standard_normal(embed_dim).astype(np.float32))
        r = len(store) - 1
        return r if rows_mode else torch.from_numpy(store[r])

    gmap = GraphCls(False, 0.5, merge_ghost, 0)             # has_real_pos, loc_noise, merge_ghost, ghost_aug
    pos = np.array([rng.uniform(-2, 2), 0.2, rng.uniform(-2, 2)])
    heading = rng.uniform(0, 2 * np.pi)
    prev_vp = None
    for stepk in range(steps):
        k = rng.randint(2, 6)
        ang = list(rng.uniform(0, 2 * np.pi, size=k))
        dis = list(rng.uniform(0.6, 2.5, size=k))
        cur_vp, cand_vp, cand_pos = gmap.identify_node(pos, heading, ang, dis)
        cur_e = new_row()
        cand_e = [new_row() for _ in range(k)]
        gmap.update_graph(prev_vp, stepk + 1, cur_vp, pos, cur_e, cand_vp, cand_pos, cand_e, None)
        prev_vp = cur_vp
        if stepk == steps - 1:
            break
        ghosts = list(gmap.ghost_pos.keys())
        if not ghosts:
            break
        gvp = ghosts[rng.randint(len(ghosts))]              # move to a ghost; it becomes the next node (:977)
        new_pos = np.array(gmap.ghost_aug_pos[gvp], dtype=np.float64)
        heading = rng.uniform(0, 2 * np.pi)
        gmap.delete_ghost(gvp)
        pos = new_pos + np.array([0.0, rng.uniform(-0.05, 0.05), 0.0])
    return gmap, prev_vp, pos, heading, np.stack(store)
