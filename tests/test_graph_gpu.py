"""etp_gmap_assemble (csrc/graph.hip) against the outputs of the reference's REAL GraphMap (tests/golden/graph_inputs.npz)
and against the CPU oracle.  fp32 device arithmetic vs the reference's float64-then-cast: 2e-5 abs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import graph_oracle as go  # noqa: E402
from tests.graph_util import load_episodes  # noqa: E402
from etpnav_amd.graph_inputs import GraphMapLite, pack_batch, assemble_on_device, nav_gmap_variable, gather_rows  # noqa: E402


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
        g, vp, pos, h, _store = go.simulate(GraphMapLite, seed, steps, merge_ghost=(e % 2 == 0))
        gmaps.append(g); vps.append(vp); poss.append(pos); heads.append(h)
    out = nav_gmap_variable(gmaps, vps, poss, heads, "cuda")
    torch.cuda.synchronize()
    assert out["no_vp_left"] == [False] * 4 and out["gmap_vp_ids"][0][0] is None
    assert len(out["gmap_vp_ids"][3]) == 1 + len(gmaps[3].node_pos) + len(gmaps[3].ghost_pos)
    for b in range(4):
        want = go.assemble(eps[b], G=out["gmap_pos_fts"].shape[1])
        assert np.abs(out["gmap_pos_fts"][b].cpu().numpy() - want["gmap_pos_fts"]).max() < 2e-5


def test_node_embeddings_from_device_store_with_gradient():
    """gather_rows (etp_gather_sum over the packed CSR) == the gmap_img_fts the real trainer method stacked, and its
    autograd backward == torch's on the same linear map."""
    eps, outs = load_episodes()
    gmaps, stores, offs = [], [], [0]
    for e, (seed, steps) in enumerate(go.GOLDEN_EPISODES):
        g, _, _, _, store = go.simulate(GraphMapLite, seed, steps, merge_ghost=(e % 2 == 0), rows_mode=True)
        gmaps.append(g); stores.append(store); offs.append(offs[-1] + len(store))
    H = 256                                                       # etp_gather_sum rows are multiples of 256 columns
    base = torch.from_numpy(np.concatenate(stores))               # [R, 8] golden embeddings
    proj = torch.randn(base.shape[1], H)
    store = (base @ proj).cuda().requires_grad_(True)             # same linear structure at a kernel-sized width
    G = max(1 + e["n_nodes"] + e["n_ghost"] for e in eps)
    out = gather_rows(store, gmaps, offs[:-1], G)
    torch.cuda.synchronize()
    for b, want in enumerate(outs):
        L = want["gmap_img_fts"].shape[0]
        ref = torch.from_numpy(want["gmap_img_fts"]) @ proj
        assert (out[b, :L].detach().cpu() - ref).abs().max().item() < 1e-4, b
        assert not out[b, L:].any()
    wgt = torch.randn_like(out)
    (out * wgt).sum().backward()
    torch.cuda.synchronize()
    # reference gradient: dense matrix of the same CSR on the CPU
    from etpnav_amd.graph_inputs import pack_img_csr
    (pf, xf, wf), _ = pack_img_csr(gmaps, offs[:-1], G, store.shape[0])
    A = torch.zeros(len(gmaps) * G, store.shape[0])
    for n in range(len(gmaps) * G):
        for q in range(int(pf[n]), int(pf[n + 1])):
            A[n, xf[q]] += wf[q]
    want_grad = A.t() @ wgt.detach().cpu().view(-1, H)
    assert (store.grad.cpu() - want_grad).abs().max().item() < 1e-4


def test_vp_feature_variable_matches_the_real_trainer_method():
    """etp_vp_gather x3 == RLTrainer._vp_feature_variable run from the trainer's own source on the same observations
    (tests/golden/vp_inputs.npz, generator oracle/make_golden_vp.py): bit-exact (pure gathers)."""
    import os
    from oracle.make_golden_vp import make_obs
    from etpnav_amd.graph_inputs import vp_feature_variable
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "vp_inputs.npz"))
    got = vp_feature_variable(make_obs(), "cuda")
    torch.cuda.synchronize()
    for k in ("rgb_fts", "dep_fts", "loc_fts", "nav_types", "view_lens"):
        assert np.array_equal(got[k].cpu().numpy(), z[f"out/{k}"]), k
