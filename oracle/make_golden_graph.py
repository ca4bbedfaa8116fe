"""Generate tests/golden/graph_inputs.npz from the REAL GraphMap (vlnce_baselines/models/graph_utils.py, imported from
/root/reference; build container only).  habitat is not installed: its two geometry helpers are only used by
heading_from_quaternion, which is replaced by the identity on a scalar heading (quaternion -> heading conversion belongs
to the simulator side and is out of scope); everything else is the reference's own code: identify_node, update_graph
(localisation, ghost merging, networkx all-pairs Dijkstra), delete_ghost, front_to_ghost_dist, get_pos_fts, and the
pair-distance loop of ss_trainer_ETP.py:371-387 restated over the real object's shortest_dist.

    python oracle/make_golden_graph.py
"""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load_reference_graph_utils():
    for name in ("habitat", "habitat.tasks", "habitat.tasks.utils", "habitat.utils", "habitat.utils.geometry_utils"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["habitat.tasks.utils"].cartesian_to_polar = lambda x, y: (np.hypot(x, y), np.arctan2(y, x))
    sys.modules["habitat.utils.geometry_utils"].quaternion_rotate_vector = None
    sys.modules["habitat.utils.geometry_utils"].quaternion_from_coeff = None
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_graph_utils", "/root/reference/vlnce_baselines/models/graph_utils.py")
    gu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gu)
    gu.heading_from_quaternion = lambda q: float(q) % (2 * np.pi)      # callers pass the heading itself
    return gu


def reference_outputs(gu, gmap, cur_vp, cur_pos, cur_heading):
    """ss_trainer_ETP.py:350-394 for one episode, on the real object."""
    node_vp_ids = list(gmap.node_pos.keys())
    ghost_vp_ids = list(gmap.ghost_pos.keys())
    gmap_vp_ids = [None] + node_vp_ids + ghost_vp_ids
    step_ids = [0] + [gmap.node_stepId[vp] for vp in node_vp_ids] + [0] * len(ghost_vp_ids)
    visited = [0] + [1] * len(node_vp_ids) + [0] * len(ghost_vp_ids)
    pos_fts = gmap.get_pos_fts(cur_vp, cur_pos, cur_heading, gmap_vp_ids)
    n = len(gmap_vp_ids)
    pair = np.zeros((n, n), dtype=np.float32)
    for j in range(1, n):
        for k in range(j + 1, n):
            vp1, vp2 = gmap_vp_ids[j], gmap_vp_ids[k]
            if not vp1.startswith('g') and not vp2.startswith('g'):
                dist = gmap.shortest_dist[vp1][vp2]
            elif not vp1.startswith('g') and vp2.startswith('g'):
                fd2, fv2 = gmap.front_to_ghost_dist(vp2)
                dist = gmap.shortest_dist[vp1][fv2] + fd2
            else:
                fd1, fv1 = gmap.front_to_ghost_dist(vp1)
                fd2, fv2 = gmap.front_to_ghost_dist(vp2)
                dist = fd1 + gmap.shortest_dist[fv1][fv2] + fd2
            pair[j, k] = pair[k, j] = dist / gu.MAX_DIST
    return np.array(step_ids), np.array(visited, dtype=bool), pos_fts.astype(np.float32), pair


def main():
    gu = load_reference_graph_utils()
    from etpnav_amd.graph_inputs import pack_episode
    from oracle.graph_oracle import simulate, GOLDEN_EPISODES
    z = {}
    specs = GOLDEN_EPISODES
    for e, (seed, steps) in enumerate(specs):
        gmap, cur_vp, cur_pos, cur_heading = simulate(gu.GraphMap, seed, steps, merge_ghost=(e % 2 == 0))
        step_ids, visited, pos_fts, pair = reference_outputs(gu, gmap, cur_vp, cur_pos, cur_heading)
        ep = pack_episode(gmap, cur_vp, cur_pos, cur_heading)
        for k, v in ep.items():
            if k == "ghost_fronts":
                z[f"ep{e}/ghost_front_ptr"] = np.cumsum([0] + [len(f) for f in v]).astype(np.int32)
                z[f"ep{e}/ghost_front_idx"] = np.array([x for f in v for x in f], dtype=np.int32)
            else:
                z[f"ep{e}/{k}"] = np.asarray(v)
        z[f"ep{e}/out_step_ids"], z[f"ep{e}/out_visited"] = step_ids, visited
        z[f"ep{e}/out_pos_fts"], z[f"ep{e}/out_pair_dists"] = pos_fts, pair
        print(f"episode {e}: {len(gmap.node_pos)} nodes, {len(gmap.ghost_pos)} ghosts")
    z["n_episodes"] = np.array(len(specs))
    out = os.path.join(ROOT, "tests", "golden", "graph_inputs.npz")
    np.savez_compressed(out, **z)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
