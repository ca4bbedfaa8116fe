"""Direct oracle-vs-reference check; only runs where /root/reference exists (build container)."""
import pytest
import torch

from oracle import planner_oracle as po
from oracle import ref_harness as rh

pytestmark = pytest.mark.skipif(not rh.reference_available(), reason="reference not present")


def test_oracle_equals_reference_small():
    cfg = po.PlannerConfig.r2r()
    P = po.init_params(cfg, seed=3)
    model = rh.build_reference_model(cfg, P)
    batch = po.make_batch(cfg, B=2, L=10, V=13, G=6, seed=7, ragged=True)
    ro, rg = rh.reference_step(model, batch)
    oo, og = po.step_with_grads(P, cfg, batch)
    assert abs(float(ro["loss"] - oo["loss"])) < 1e-5
    fin = torch.isfinite(ro["global_logits"])
    assert float((ro["global_logits"][fin] - oo["global_logits"][fin]).abs().max()) < 2e-5
    for k in rg:
        assert float((rg[k] - og[k]).abs().max()) < 2e-5, k


def test_saved_pretraining_checkpoint_loads_strictly_into_the_real_model():
    """etpnav_amd.checkpoint.pretrain_state_dict writes exactly the key set of the reference pre-training model
    (utils/save.py:23-46): the REAL GlocalTextPathCMTPreTraining loads it with strict=True and then reproduces the
    planner's own parameters."""
    from oracle import ref_pretrain_harness as rp
    from etpnav_amd import checkpoint as ck
    from etpnav_amd.planner import GlocalTextPathNavCMT
    cfg = po.PlannerConfig.r2r(vocab_size=512, num_l_layers=1, num_pano_layers=1, num_x_layers=1, use_lang2visn_attn=True)
    P = po.init_params(cfg, seed=2)
    m = GlocalTextPathNavCMT(cfg.to_dict(), dtype=torch.float32, device="cpu")
    m.load_state_dict(P, strict=True)
    real = rp.build_pretrain_model(cfg, po.init_params(cfg, seed=5))          # different weights before loading
    missing, unexpected = real.load_state_dict(ck.pretrain_state_dict(m), strict=True)
    assert not missing and not unexpected
    got = {k: v for k, v in real.state_dict().items()}
    for k, v in P.items():
        name = k if k.startswith(("mlm_head.", "global_sap_head.")) else "bert." + k
        assert torch.equal(got[name], v), k
    assert got["mlm_head.predictions.decoder.weight"].data_ptr() == got["bert.embeddings.word_embeddings.weight"].data_ptr()


@pytest.mark.parametrize("seed", list(range(30, 42)))
def test_graph_assembly_randomised_against_the_real_classes(seed):
    """Beyond the committed golden episodes: random rollouts through the REAL GraphMap and through GraphMapLite must pack
    identically, and the oracle's assembly must equal the trainer's REAL _nav_gmap_variable on them."""
    import numpy as np
    from oracle import graph_oracle as go
    from oracle.make_golden_graph import load_reference_graph_utils, reference_batch_outputs
    from etpnav_amd.graph_inputs import GraphMapLite, pack_episode
    gu = load_reference_graph_utils()
    steps = 2 + seed % 9
    merge = bool(seed % 2)
    real = go.simulate(gu.GraphMap, seed, steps, merge_ghost=merge)
    lite = go.simulate(GraphMapLite, seed, steps, merge_ghost=merge)
    a = pack_episode(real[0], real[1], real[2], real[3])
    b = pack_episode(lite[0], lite[1], lite[2], lite[3])
    assert a["n_nodes"] == b["n_nodes"] and a["n_ghost"] == b["n_ghost"] and a["cur_node"] == b["cur_node"]
    for k in ("node_pos", "node_step", "adj", "ghost_pos"):
        assert np.allclose(a[k], b[k], atol=1e-12), k
    assert a["ghost_fronts"] == b["ghost_fronts"]
    ref = reference_batch_outputs(gu, [real[0]], [real[1]], [real[2]], [real[3]])
    L = 1 + a["n_nodes"] + a["n_ghost"]
    got = go.assemble(b, G=L)
    assert np.array_equal(got["gmap_step_ids"], ref["gmap_step_ids"][0].numpy())
    assert np.array_equal(got["gmap_visited_masks"], ref["gmap_visited_masks"][0].numpy())
    assert np.abs(got["gmap_pos_fts"] - ref["gmap_pos_fts"][0].numpy()).max() < 2e-6
    assert np.abs(got["gmap_pair_dists"] - ref["gmap_pair_dists"][0].numpy()).max() < 2e-6


@pytest.mark.parametrize("seed", [3, 4, 5, 6])
def test_graph_map_lite_with_real_positions_and_ghost_jitter_against_the_real_class(seed):
    """The two GraphMap options the rollout driver above leaves off: has_real_pos (training keeps the simulator's positions of the
    candidates per ghost, graph_utils.py:213-214,225-226,236-237) and ghost_aug (position jitter from np.random, :245-252).  Both
    classes are driven with the same candidates and the same numpy seed; every dict of the graph must agree."""
    import numpy as np
    from oracle.make_golden_graph import load_reference_graph_utils
    from etpnav_amd.graph_inputs import GraphMapLite
    gu = load_reference_graph_utils()
    out = []
    for cls in (gu.GraphMap, GraphMapLite):
        rng = np.random.RandomState(seed)
        np.random.seed(1000 + seed)                          # the jitter draws from the global generator in both classes
        g = cls(True, 0.5, bool(seed % 2), 0.3)              # has_real_pos, loc_noise, merge_ghost, ghost_aug
        pos, heading, prev = np.array([0.3, 0.2, -0.4]), 1.0, None
        for k in range(6):
            n = rng.randint(2, 6)
            cur_vp, cand_vp, cand_pos = g.identify_node(pos, heading, list(rng.uniform(0, 6.28, n)), list(rng.uniform(0.3, 2.0, n)))
            real = [p + rng.normal(0, 0.05, 3) for p in cand_pos]
            g.update_graph(prev, k + 1, cur_vp, pos, float(k), cand_vp, cand_pos, [float(10 * k + i) for i in range(n)], real)
            prev = cur_vp
            if k == 5:
                break        # compare right after an update (the reference leaves a deleted ghost in ghost_aug_pos until the next one)
            ghosts = list(g.ghost_pos.keys())
            gv = ghosts[rng.randint(len(ghosts))]
            pos, heading = np.array(g.ghost_real_pos[gv][0], dtype=np.float64), rng.uniform(0, 6.28)
            g.delete_ghost(gv)
        out.append(g)
    a, b = out
    edges_a = {tuple(sorted((u, v))): w for u, v, w in a.graph_nx.edges(data="weight")}
    assert edges_a.keys() == b.edges.keys() and all(abs(edges_a[k] - b.edges[k]) < 1e-12 for k in edges_a)
    assert list(a.node_pos) == list(b.node_pos) and a.node_stepId == b.node_stepId and a.ghost_cnt == b.ghost_cnt
    for name in ("ghost_mean_pos", "ghost_aug_pos"):
        da, db = getattr(a, name), getattr(b, name)
        assert list(da) == list(db), name
        assert all(np.allclose(da[k], db[k], atol=1e-12) for k in da), name
    for name in ("ghost_pos", "ghost_real_pos"):
        da, db = getattr(a, name), getattr(b, name)
        assert list(da) == list(db) and all(np.allclose(np.asarray(da[k]), np.asarray(db[k]), atol=1e-12) for k in da), name
    assert a.ghost_fronts == b.ghost_fronts
    assert {k: (float(v[0]), v[1]) for k, v in a.ghost_embeds.items()} == {k: (float(v[0]), v[1]) for k, v in b.ghost_embeds.items()}
