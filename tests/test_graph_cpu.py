"""Graph-input assembly: the CPU oracle and the host-side GraphMapLite against golden vectors produced
by the reference's REAL GraphMap class (tests/golden/graph_inputs.npz)."""
import numpy as np
import pytest
import torch

from oracle import graph_oracle as go
from tests.graph_util import load_episodes
from etpnav_amd import _lib
from etpnav_amd.graph_inputs import GraphMapLite, pack_episode, pack_batch, assemble_on_device


def test_oracle_matches_real_graphmap_outputs():
    eps, outs = load_episodes()
    assert len(eps) == len(go.GOLDEN_EPISODES)
    for ep, want in zip(eps, outs):
        L = 1 + ep["n_nodes"] + ep["n_ghost"]
        got = go.assemble(ep, G=L + 3)                       # 3 padded entries: must come out as zeros / False
        assert np.array_equal(got["gmap_step_ids"][:L], want["gmap_step_ids"]) and not got["gmap_step_ids"][L:].any()
        assert np.array_equal(got["gmap_visited_masks"][:L], want["gmap_visited_masks"])
        assert got["gmap_masks"][:L].all() and not got["gmap_masks"][L:].any()
        assert np.abs(got["gmap_pos_fts"][:L] - want["gmap_pos_fts"]).max() < 2e-6
        assert not got["gmap_pos_fts"][L:].any()
        assert np.abs(got["gmap_pair_dists"][:L, :L] - want["gmap_pair_dists"]).max() < 2e-6
        assert not got["gmap_pair_dists"][L:].any() and not got["gmap_pair_dists"][:, L:].any()


def test_graphmaplite_replays_the_reference_bookkeeping():
    """Same synthetic episodes, same driver, our numpy-only GraphMapLite instead of the reference class: node / ghost
    sets, positions, edges, ghost fronts and step ids must come out identical (localisation, ghost merging, deletion)."""
    eps, _ = load_episodes()
    for e, (seed, steps) in enumerate(go.GOLDEN_EPISODES):
        gmap, cur_vp, cur_pos, cur_heading, _store = go.simulate(GraphMapLite, seed, steps, merge_ghost=(e % 2 == 0))
        mine = pack_episode(gmap, cur_vp, cur_pos, cur_heading)
        ref = eps[e]
        assert mine["n_nodes"] == ref["n_nodes"] and mine["n_ghost"] == ref["n_ghost"] and mine["cur_node"] == ref["cur_node"]
        for k in ("node_pos", "node_step", "adj", "ghost_pos", "cur_pos"):
            assert np.allclose(mine[k], ref[k], atol=1e-12), (e, k)
        assert [list(f) for f in mine["ghost_fronts"]] == [list(f) for f in ref["ghost_fronts"]]
        assert abs(mine["cur_heading"] - ref["cur_heading"]) < 1e-12


def test_pack_batch_layout_and_limits():
    eps, _ = load_episodes()
    b = pack_batch(eps)
    B, Nmax, Mmax, Fmax = b["_dims"]
    assert B == len(eps) and Nmax == max(e["n_nodes"] for e in eps) and Mmax == max(e["n_ghost"] for e in eps)
    assert b["adj"].shape == (B, Nmax, Nmax) and (b["adj"][0, 1:, :] == -1).all()           # padded rows: no edges
    for i, e in enumerate(eps):
        assert b["front_ptr"][i, e["n_ghost"]] == sum(len(f) for f in e["ghost_fronts"])
        assert (b["front_ptr"][i, e["n_ghost"]:] == b["front_ptr"][i, e["n_ghost"]]).all()
    big = dict(eps[0]); big["n_nodes"] = 65
    with pytest.raises(ValueError):
        pack_batch([big])
    assert _lib.lib().etp_gmap_assemble is not None
    with pytest.raises(_lib.EtpError):
        assemble_on_device(b, "cpu")                          # no CPU fallback


def test_embedding_csr_reproduces_the_reference_node_embeddings():
    """Device-store mode: the CSR packed from GraphMapLite (rows instead of tensors), applied to the store on the CPU,
    equals the gmap_img_fts the REAL _nav_gmap_variable stacked from GraphMap.get_node_embeds (node rows, ghost means,
    zero [stop] row, zero padding); the transposed CSR is its exact adjoint."""
    from etpnav_amd.graph_inputs import pack_img_csr
    eps, outs = load_episodes()
    gmaps, stores, offs = [], [], [0]
    for e, (seed, steps) in enumerate(go.GOLDEN_EPISODES):
        g, _, _, _, store = go.simulate(GraphMapLite, seed, steps, merge_ghost=(e % 2 == 0), rows_mode=True)
        gmaps.append(g); stores.append(store); offs.append(offs[-1] + len(store))
    store = torch.from_numpy(np.concatenate(stores))
    G = max(1 + e["n_nodes"] + e["n_ghost"] for e in eps) + 2
    (pf, xf, wf), (pb, xb, wb) = pack_img_csr(gmaps, offs[:-1], G, store.shape[0])
    out = torch.zeros(len(gmaps) * G, store.shape[1])
    for n in range(len(gmaps) * G):
        for q in range(int(pf[n]), int(pf[n + 1])):
            out[n] += wf[q] * store[xf[q]]
    out = out.view(len(gmaps), G, -1)
    for b, want in enumerate(outs):
        L = want["gmap_img_fts"].shape[0]
        assert np.abs(out[b, :L].numpy() - want["gmap_img_fts"]).max() < 1e-6, b
        assert not out[b, L:].any() and not out[b, 0].any()
    # adjoint: <A x, y> == <x, A^T y>
    y = torch.randn(len(gmaps) * G, store.shape[1])
    aty = torch.zeros_like(store)
    for r in range(store.shape[0]):
        for q in range(int(pb[r]), int(pb[r + 1])):
            aty[r] += wb[q] * y[xb[q]]
    assert abs(float((out.view(-1, store.shape[1]) * y).sum()) - float((store * aty).sum())) < 1e-3


def test_trajectory_aggregation_csr_matches_the_real_pretraining_method():
    """pack_traj_csr applied on the CPU == GlobalMapEncoder._aggregate_gmap_features (pretrain vilmodel.py:585-619) run
    from its own source (tests/golden/traj_agg.npz): revisited nodes, candidates seen before/after being visited,
    ragged view counts, zero [stop] row and padding; the transposed CSR is the exact adjoint."""
    import os
    from oracle.make_golden_traj import make_case
    from etpnav_amd.graph_inputs import pack_traj_csr
    want = np.load(os.path.join(os.path.dirname(__file__), "golden", "traj_agg.npz"))["out"]
    embeds, lens, vpids, cands, gvps = make_case()
    V, H = embeds[0].shape[1], embeds[0].shape[2]
    G = want.shape[1] + 2
    store = torch.cat(embeds, 0).reshape(-1, H)
    (pf, xf, wf), (pb, xb, wb) = pack_traj_csr([l.tolist() for l in lens], vpids, cands, gvps, V, G)
    out = torch.zeros(len(embeds) * G, H)
    for n in range(out.shape[0]):
        for q in range(int(pf[n]), int(pf[n + 1])):
            out[n] += wf[q] * store[xf[q]]
    out = out.view(len(embeds), G, H)
    assert np.abs(out[:, :want.shape[1]].numpy() - want).max() < 1e-6
    assert not out[:, want.shape[1]:].any() and not out[:, 0].any()
    y = torch.randn(len(embeds) * G, H)
    aty = torch.zeros_like(store)
    for r in range(store.shape[0]):
        for q in range(int(pb[r]), int(pb[r + 1])):
            aty[r] += wb[q] * y[xb[q]]
    assert abs(float((out.view(-1, H) * y).sum()) - float((store * aty).sum())) < 1e-3


def test_trajectory_csr_equals_oracle_aggregation_on_random_trajectories():
    """Property check beyond the golden case: for random trajectories (revisits, repeated candidates, candidate slots
    beyond the valid views, ragged view counts) the packed CSR applied to the embeddings equals the oracle's loop
    restatement of _aggregate_gmap_features (itself pinned to the real method)."""
    from oracle import planner_oracle as po
    from etpnav_amd.graph_inputs import pack_traj_csr
    rng = np.random.RandomState(0)
    for trial in range(25):
        B, V, H = rng.randint(1, 5), rng.randint(3, 9), 4
        lens, vpids, cands, gvps, embeds = [], [], [], [], []
        for i in range(B):
            T = rng.randint(1, 6)
            names = [f"n{i}_{k}" for k in range(8)]
            path = [names[rng.randint(0, 4)] for _ in range(T)]
            ep_lens = [int(rng.randint(1, V + 1)) for _ in range(T)]
            mx = max(ep_lens)
            ep_c = [[names[rng.randint(0, 8)] for _ in range(rng.randint(0, mx + 1))] for _ in range(T)]
            seen = []
            for t in range(T):
                for vp in [path[t]] + ep_c[t]:
                    if vp not in seen:
                        seen.append(vp)
            lens.append(ep_lens); vpids.append(path); cands.append(ep_c); gvps.append([None] + seen)
            embeds.append(torch.from_numpy(rng.standard_normal((T, V, H)).astype(np.float32)))
        flat = torch.cat(embeds, 0)
        traj = {"traj_step_lens": [len(e) for e in embeds], "traj_vp_lens": lens, "traj_vpids": vpids,
                "traj_cand_vpids": cands, "gmap_vpids": gvps}
        want = po.aggregate_gmap_features(flat, traj)
        G = want.shape[1]
        (pf, xf, wf), _ = pack_traj_csr(lens, vpids, cands, gvps, V, G)
        store = flat.reshape(-1, H)
        out = torch.zeros(B * G, H)
        for n in range(B * G):
            for q in range(int(pf[n]), int(pf[n + 1])):
                out[n] += wf[q] * store[xf[q]]
        assert (out.view(B, G, H) - want).abs().max().item() < 1e-5, trial
