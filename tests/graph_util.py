"""Loader for tests/golden/graph_inputs.npz: outputs of the REAL RLTrainer._nav_gmap_variable on REAL GraphMap objects
(generator oracle/make_golden_graph.py)."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(__file__), "golden", "graph_inputs.npz")


def load_episodes():
    z = np.load(GOLD)
    eps, outs = [], []
    for e in range(int(z["n_episodes"])):
        g = lambda k: z[f"ep{e}/{k}"]
        ptr, idx = g("ghost_front_ptr"), g("ghost_front_idx")
        m = int(g("n_ghost"))
        eps.append({"n_nodes": int(g("n_nodes")), "n_ghost": m, "node_pos": g("node_pos"), "node_step": g("node_step"),
                    "adj": g("adj"), "ghost_pos": g("ghost_pos"),
                    "ghost_fronts": [list(idx[ptr[i]:ptr[i + 1]]) for i in range(m)],
                    "cur_node": int(g("cur_node")), "cur_pos": g("cur_pos"), "cur_heading": float(g("cur_heading"))})
        outs.append({"gmap_step_ids": g("out_step_ids"), "gmap_visited_masks": g("out_visited"),
                     "gmap_pos_fts": g("out_pos_fts"), "gmap_pair_dists": g("out_pair_dists"),
                     "gmap_img_fts": g("out_img_fts")})
    return eps, outs
