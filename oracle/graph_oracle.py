"""CPU restatement (numpy, float64) of the reference's graph-input assembly for forward_navigation.  TEST INFRASTRUCTURE
ONLY (tests/, bench legs); the product path is etpnav_amd/csrc/graph.hip through etp_gmap_assemble.

Follows, from compact per-episode arrays instead of Python dicts:
  * shortest paths over the visited-node graph: GraphMap.update_graph's nx.all_pairs_dijkstra_path(_length)
    (vlnce_baselines/models/graph_utils.py:256-257) -> Floyd-Warshall with node counts of the shortest path;
  * GraphMap.front_to_ghost_dist (:259-270): nearest front node of each ghost (first minimum in list order);
  * GraphMap.get_pos_fts (:278-322) with calculate_vp_rel_pos_fts (:21-44) and get_angle_fts (:46-52):
    [sin h, cos h, sin e, cos e, line_dist/30, shortest_dist/30, shortest_step/10] per node;
  * RLTrainer._nav_gmap_variable (ss_trainer_ETP.py:344-417): [stop] + visited nodes + ghosts ordering, step ids,
    visited / validity masks, the pairwise distance matrix (:371-387) and zero padding to the batch maximum.
Pinned by tests/golden/graph_inputs.npz (outputs of the REAL GraphMap class; generator oracle/make_golden_graph.py).
"""
from __future__ import annotations

import numpy as np

MAX_DIST = 30.0     # graph_utils.py:9
MAX_STEP = 10.0     # graph_utils.py:10


def rel_pos_fts(a, b, base_heading):
    """calculate_vp_rel_pos_fts(a, b, base_heading, 0, to_clock=True) graph_utils.py:21-44."""
    dx, dy, dz = b[0] - a[0], b[1] - a[1], b[2] - a[2]
    xz = max(np.sqrt(dx * dx + dz * dz), 1e-8)
    xyz = max(np.sqrt(dx * dx + dy * dy + dz * dz), 1e-8)
    heading = np.arcsin(-dx / xz)
    if b[2] > a[2]:
        heading = np.pi - heading
    heading -= base_heading
    heading = 2 * np.pi - heading
    elevation = np.arcsin(dz / xyz)
    return heading, elevation, xyz


def shortest_paths(adj, n):
    """All-pairs shortest distances and the number of nodes on the shortest path (len(nx path)); adj < 0 = no edge."""
    D = np.full((n, n), np.inf)
    C = np.zeros((n, n), dtype=np.int64)
    for i in range(n):
        for j in range(n):
            if i == j:
                D[i, j], C[i, j] = 0.0, 1
            elif adj[i, j] >= 0:
                D[i, j], C[i, j] = adj[i, j], 2
    for k in range(n):
        for i in range(n):
            for j in range(n):
                if D[i, k] + D[k, j] < D[i, j]:
                    D[i, j] = D[i, k] + D[k, j]
                    C[i, j] = C[i, k] + C[k, j] - 1
    return D, C


def assemble(ep, G):
    """ep: dict of compact arrays of ONE episode (see etpnav_amd/graph_inputs.py: pack_graphs); returns the padded rows."""
    n, m = int(ep["n_nodes"]), int(ep["n_ghost"])
    D, C = shortest_paths(ep["adj"], n)
    cur = int(ep["cur_node"])
    fd, fv = np.zeros(m), np.zeros(m, dtype=np.int64)
    for g in range(m):
        best, bv = 10000.0, -1
        for f in ep["ghost_fronts"][g]:
            d = np.sqrt(((ep["node_pos"][f] - ep["ghost_pos"][g]) ** 2).sum())
            if d < best:
                best, bv = d, f
        fd[g], fv[g] = best, bv
    L = 1 + n + m
    step_ids = np.zeros(G, dtype=np.int64)
    visited = np.zeros(G, dtype=bool)
    mask = np.zeros(G, dtype=bool)
    pos = np.zeros((G, 7), dtype=np.float32)
    pair = np.zeros((G, G), dtype=np.float32)
    mask[:L] = True
    step_ids[1:1 + n] = ep["node_step"][:n]
    visited[1:1 + n] = True
    pos[0] = [0, 1, 0, 1, 0, 0, 0]                                     # vp None: angles (0,0) -> sin 0, cos 0
    for t in range(1, L):
        if t <= n:
            v = t - 1
            h, e, dist = rel_pos_fts(ep["cur_pos"], ep["node_pos"][v], ep["cur_heading"])
            sd, ss = D[cur, v], C[cur, v]
        else:
            g = t - 1 - n
            h, e, dist = rel_pos_fts(ep["cur_pos"], ep["ghost_pos"][g], ep["cur_heading"])
            sd, ss = D[cur, fv[g]] + fd[g], C[cur, fv[g]] + 1
        pos[t] = [np.sin(h), np.cos(h), np.sin(e), np.cos(e), dist / MAX_DIST, sd / MAX_DIST, ss / MAX_STEP]

    def sp(t):           # (anchor node, extra distance) of gmap entry t >= 1
        return (t - 1, 0.0) if t <= n else (fv[t - 1 - n], fd[t - 1 - n])
    for j in range(1, L):
        for k in range(j + 1, L):
            (a, da), (b, db) = sp(j), sp(k)
            pair[j, k] = pair[k, j] = (da + D[a, b] + db) / MAX_DIST
    return {"gmap_step_ids": step_ids, "gmap_visited_masks": visited, "gmap_masks": mask, "gmap_pos_fts": pos,
            "gmap_pair_dists": pair}


# ---- synthetic episode driver shared by the golden generator (real GraphMap) and the tests (GraphMapLite) ------------
GOLDEN_EPISODES = [(1, 1), (2, 3), (3, 6), (4, 9), (5, 14), (6, 20)]     # (seed, steps): one node ... 20 visited nodes


from etpnav_amd.synthetic import simulate_rollout as simulate          # noqa: E402,F401  (synthetic data generator)
