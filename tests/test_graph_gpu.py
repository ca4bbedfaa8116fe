"""etp_gmap_assemble (csrc/graph.hip) against the outputs of the reference's REAL GraphMap (tests/golden/graph_inputs.npz)
and against the CPU oracle.  fp32 device arithmetic vs the reference's float64-then-cast: 2e-5 abs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import graph_oracle as go  # noqa: E402
from tests.graph_util import load_episodes  # noqa: E402
from etpnav_amd.graph_inputs import GraphMapLite, pack_batch, assemble_on_device, nav_gmap_variable  # noqa: E402


def test_device_assembly_matches_real_graphmap():
    eps, outs = load_episodes()
    got = assemble_on_device(pack_batch(eps), "cuda")
    torch.cuda.synchronize()
    G = got["gmap_step_ids"].shape[1]
    assert G == max(1 + e["n_nodes"] + e["n_ghost"] for e in eps)
    for b, (ep, want) in enumerate(zip(eps, outs)):
        L = 1 + ep["n_nodes"] + ep["n_ghost"]
        assert np.array_equal(got["gmap_step_ids"][b, :L].cpu().numpy(), want["gmap_step_ids"])
        assert not got["gmap_step_ids"][b, L:].any()
        assert np.array_equal(got["gmap_visited_masks"][b, :L].cpu().numpy(), want["gmap_visited_masks"])
        assert not got["gmap_visited_masks"][b, L:].any()
        assert got["gmap_masks"][b, :L].all() and not got["gmap_masks"][b, L:].any()
        pf = got["gmap_pos_fts"][b].cpu().numpy()
        assert np.abs(pf[:L] - want["gmap_pos_fts"]).max() < 2e-5, b
        assert not pf[L:].any()
        pd = got["gmap_pair_dists"][b].cpu().numpy()
        assert np.abs(pd[:L, :L] - want["gmap_pair_dists"]).max() < 2e-5, b
        assert not pd[L:].any() and not pd[:, L:].any()
        assert np.array_equal(pd, pd.T)


def test_device_assembly_padded_and_via_trainer_api():
    """Explicit G larger than needed, and the _nav_gmap_variable-shaped entry point on GraphMapLite episodes."""
    eps, _ = load_episodes()
    a = assemble_on_device(pack_batch(eps[:3]), "cuda", G=40)
    torch.cuda.synchronize()
    for b, ep in enumerate(eps[:3]):
        want = go.assemble(ep, G=40)
        assert np.abs(a["gmap_pos_fts"][b].cpu().numpy() - want["gmap_pos_fts"]).max() < 2e-5
        assert np.abs(a["gmap_pair_dists"][b].cpu().numpy() - want["gmap_pair_dists"]).max() < 2e-5
    gmaps, vps, poss, heads = [], [], [], []
    for e, (seed, steps) in enumerate(go.GOLDEN_EPISODES[:4]):
        g, vp, pos, h = go.simulate(GraphMapLite, seed, steps, merge_ghost=(e % 2 == 0))
        gmaps.append(g); vps.append(vp); poss.append(pos); heads.append(h)
    out = nav_gmap_variable(gmaps, vps, poss, heads, "cuda")
    torch.cuda.synchronize()
    assert out["no_vp_left"] == [False] * 4 and out["gmap_vp_ids"][0][0] is None
    assert len(out["gmap_vp_ids"][3]) == 1 + len(gmaps[3].node_pos) + len(gmaps[3].ghost_pos)
    for b in range(4):
        want = go.assemble(eps[b], G=out["gmap_pos_fts"].shape[1])
        assert np.abs(out["gmap_pos_fts"][b].cpu().numpy() - want["gmap_pos_fts"]).max() < 2e-5
