"""Graph inputs of forward_navigation assembled on the device.

The reference rebuilds, at every rollout step and on the host, everything forward_navigation needs besides the node
embeddings: ``RLTrainer._nav_gmap_variable`` (ss_trainer_ETP.py:344-417) walks Python dicts of every episode's
``GraphMap``, which itself re-runs networkx all-pairs Dijkstra after every update (graph_utils.py:256-257), fills the
pairwise distance matrix in an O(G^2) Python loop and copies six tensors to the GPU.  Here the host only keeps COMPACT
arrays per episode (positions, edge lengths, ghost fronts) and one kernel launch (``etp_gmap_assemble``) produces the padded
device tensors.

* ``pack_episode(gmap, cur_vp, cur_pos, cur_heading)`` reads any object with the reference GraphMap's attributes
  (``node_pos, node_stepId, ghost_aug_pos, ghost_fronts`` and either ``graph_nx`` or ``edges``) — so the reference's own
  GraphMap can be used unchanged;
* ``GraphMapLite`` is a numpy-only GraphMap with the same update rules (graph_utils.py:118-257) that simply skips the
  per-step Dijkstra (the device does the shortest paths);
* ``nav_gmap_variable(gmaps, cur_vp, cur_pos, cur_heading, device)`` returns the same dict as the reference method
  (without ``gmap_img_fts``; see the next item);
* ``gmap_img_fts``: the reference stacks per-node tensors in Python (``get_node_embeds``, ss_trainer_ETP.py:360-365).  In
  **device-store mode** GraphMapLite is given ROW INDICES into one embedding store tensor instead of tensors
  (``update_graph(..., cur_embeds=row, cand_embeds=[rows])``); ``pack_img_csr`` turns the graphs into a CSR and
  ``gather_rows`` (autograd wrapper of ``etp_gather_sum``) produces the padded ``[B,G,H]`` tensor in one launch, with
  the gradient flowing back into the store through the transposed CSR.

``cur_heading`` is the scalar heading (radians) that the reference obtains with ``heading_from_quaternion(cur_ori)``
(graph_utils.py:54-59); quaternion handling belongs to the simulator side and is out of scope.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr

MAX_NODES, MAX_GHOSTS = 64, 192      # limits of etp_gmap_assemble (csrc/graph.hip)


def _is_row(x) -> bool:
    return isinstance(x, (int, np.integer))


def _dist(a, b) -> float:
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.sqrt(((b - a) ** 2).sum()))          # calc_position_distance graph_utils.py:13-19


class GraphMapLite:
    """GraphMap (graph_utils.py:118-257) without networkx: same node / ghost bookkeeping, edges kept as a dict; shortest
    paths are left to the device.  Embeddings are stored as given (tensors), exactly like the reference."""

    def __init__(self, has_real_pos: bool, loc_noise: float, merge_ghost: bool, ghost_aug: float, rng=None):
        self.node_pos: Dict[str, np.ndarray] = {}
        self.node_embeds: Dict[str, object] = {}
        self.node_stepId: Dict[str, int] = {}
        self.edges: Dict[tuple, float] = {}
        self.ghost_cnt = 0
        self.ghost_pos: Dict[str, list] = {}
        self.ghost_mean_pos: Dict[str, np.ndarray] = {}
        self.ghost_aug_pos: Dict[str, np.ndarray] = {}
        self.ghost_embeds: Dict[str, list] = {}
        self.ghost_fronts: Dict[str, list] = {}
        self.ghost_real_pos: Dict[str, list] = {}
        self.has_real_pos, self.merge_ghost, self.ghost_aug, self.loc_noise = has_real_pos, merge_ghost, ghost_aug, loc_noise
        self.rng = rng if rng is not None else np.random

    def _nearest(self, queries: np.ndarray, keys: Dict[str, np.ndarray]) -> List[Optional[str]]:
        """For every query position the key within ``loc_noise`` of it that is nearest (first in insertion order among equals), or
        None -- the rule of GraphMap._localize (graph_utils.py:146-158) for ALL queries against all keys in one distance matrix
        instead of a Python loop per pair."""
        if not len(keys) or not len(queries):
            return [None] * len(queries)
        names = list(keys.keys())
        K = np.asarray([keys[k] for k in names], dtype=np.float64).reshape(len(names), 3)
        Q = np.asarray(queries, dtype=np.float64).reshape(-1, 3)
        d2 = ((Q[:, None, :] - K[None, :, :]) ** 2).sum(-1)
        j = d2.argmin(1)                                     # argmin returns the FIRST minimum, as the reference's strict `<` does
        best = d2[np.arange(len(Q)), j] ** 0.5
        return [names[jj] if dd <= self.loc_noise and dd < 10000 else None for jj, dd in zip(j.tolist(), best.tolist())]

    def _localize(self, qpos, kpos_dict):                   # single-query form (kept: the trainer side calls it, graph_utils.py:146)
        return self._nearest(np.asarray(qpos, dtype=np.float64).reshape(1, 3), kpos_dict)[0]

    def identify_node(self, cur_pos, cur_heading, cand_ang, cand_dis):      # :160-166 + estimate_cand_pos :61-71
        cur_vp = str(len(self.node_pos))
        cand_vp = [f"{cur_vp}_{i}" for i in range(len(cand_ang))]
        ang = (float(cur_heading) + np.asarray(cand_ang, dtype=np.float64)) % (2 * np.pi)
        dis = np.asarray(cand_dis, dtype=np.float64)
        cand_pos = np.zeros((len(cand_vp), 3))
        cand_pos[:, 0] = cur_pos[0] - dis * np.sin(ang)
        cand_pos[:, 1] = cur_pos[1]
        cand_pos[:, 2] = cur_pos[2] - dis * np.cos(ang)
        return cur_vp, cand_vp, [p for p in cand_pos]

    def _add_edge(self, u, v, w):
        self.edges[(u, v) if u <= v else (v, u)] = float(w)

    def delete_ghost(self, vp):                              # :168-175
        for d in (self.ghost_pos, self.ghost_mean_pos, self.ghost_embeds, self.ghost_fronts):
            d.pop(vp)
        self.ghost_aug_pos.pop(vp, None)
        if self.has_real_pos:
            self.ghost_real_pos.pop(vp)

    def update_graph(self, prev_vp, step_id, cur_vp, cur_pos, cur_embeds, cand_vp, cand_pos, cand_embeds, cand_real_pos):
        """The bookkeeping of GraphMap.update_graph (graph_utils.py:177-257) without its two networkx all-pairs calls (the device
        computes the shortest paths, csrc/graph.hip), organised around what depends on what:
          1. the visited node and its edge to the previous one;
          2. every candidate against the VISITED nodes at once (their positions do not change during the call): a match is an edge;
          3. the remaining candidates, in order, against the ghosts (sequential by nature: each may create or move a ghost);
          4. the position jitter of the ghosts."""
        cur_pos = np.asarray(cur_pos, dtype=np.float64)
        if prev_vp is not None:
            self._add_edge(prev_vp, cur_vp, _dist(self.node_pos[prev_vp], cur_pos))
        self.node_pos[cur_vp], self.node_embeds[cur_vp], self.node_stepId[cur_vp] = cur_pos, cur_embeds, step_id
        n = min(len(cand_vp), len(cand_pos), len(cand_embeds))
        cand_xyz = np.asarray([np.asarray(p, dtype=np.float64) for p in cand_pos[:n]], dtype=np.float64).reshape(n, 3)
        on_node = self._nearest(cand_xyz, self.node_pos)
        for i in range(n):
            if on_node[i] is not None:
                self._add_edge(cur_vp, on_node[i], _dist(cur_pos, self.node_pos[on_node[i]]))
            else:
                self._absorb(cur_vp, cand_xyz[i], cand_embeds[i], cand_real_pos[i] if self.has_real_pos else None)
        self._jitter_ghosts()

    def _absorb(self, front_vp, pos, embeds, real_pos):
        """One candidate that is no visited node: it joins the ghost it localises to (merge_ghost) or becomes a new ghost."""
        gvp = self._localize(pos, self.ghost_mean_pos) if self.merge_ghost else None
        if gvp is None:
            gvp = f"g{self.ghost_cnt}"
            self.ghost_cnt += 1
            self.ghost_pos[gvp], self.ghost_mean_pos[gvp], self.ghost_fronts[gvp] = [pos], pos, [front_vp]
            # device-store mode keeps the ROWS of the embedding store (summed on the device later); tensor mode the running sum
            self.ghost_embeds[gvp] = [[int(embeds)] if _is_row(embeds) else embeds, 1]
            if self.has_real_pos:
                self.ghost_real_pos[gvp] = [real_pos]
            return
        self.ghost_pos[gvp].append(pos)
        self.ghost_mean_pos[gvp] = np.mean(self.ghost_pos[gvp], axis=0)
        acc = self.ghost_embeds[gvp]
        if _is_row(embeds):
            acc[0].append(int(embeds))
        else:
            acc[0] = acc[0] + embeds
        acc[1] += 1
        self.ghost_fronts[gvp].append(front_vp)
        if self.has_real_pos:
            self.ghost_real_pos[gvp].append(real_pos)

    def _jitter_ghosts(self):                                # graph_utils.py:245-252
        self.ghost_aug_pos = {k: np.array(v, dtype=np.float64) for k, v in self.ghost_mean_pos.items()}
        if self.ghost_aug != 0:
            for gvp, gpos in self.ghost_aug_pos.items():
                noise = self.rng.normal(loc=(0, 0, 0), scale=(self.ghost_aug, 0, self.ghost_aug), size=(3,))
                self.ghost_aug_pos[gvp] = gpos + np.clip(noise, -self.ghost_aug, self.ghost_aug)

    def get_node_embeds(self, vp):                           # :272-276 (tensor mode only)
        if not vp.startswith("g"):
            return self.node_embeds[vp]
        return self.ghost_embeds[vp][0] / self.ghost_embeds[vp][1]

    def embed_rows(self, vp):
        """Device-store mode: (rows of the embedding store, weight) whose weighted sum is get_node_embeds(vp)."""
        if not vp.startswith("g"):
            return [int(self.node_embeds[vp])], 1.0
        rows, cnt = self.ghost_embeds[vp]
        return list(rows), 1.0 / cnt


def pack_episode(gmap, cur_vp: str, cur_pos, cur_heading: float) -> dict:
    """Compact arrays of one episode from a GraphMap-like object (reference GraphMap or GraphMapLite)."""
    nodes = list(gmap.node_pos.keys())
    ghosts = list(gmap.ghost_pos.keys())
    idx = {vp: i for i, vp in enumerate(nodes)}
    n, m = len(nodes), len(ghosts)
    adj = np.full((n, n), -1.0, dtype=np.float64)
    if hasattr(gmap, "graph_nx"):
        edge_iter = ((u, v, w) for u, v, w in gmap.graph_nx.edges(data="weight"))
    else:
        edge_iter = ((u, v, w) for (u, v), w in gmap.edges.items())
    for u, v, w in edge_iter:
        adj[idx[u], idx[v]] = adj[idx[v], idx[u]] = w
    return {
        "n_nodes": n, "n_ghost": m,
        "node_pos": np.array([gmap.node_pos[vp] for vp in nodes], dtype=np.float64).reshape(n, 3),
        "node_step": np.array([gmap.node_stepId[vp] for vp in nodes], dtype=np.int64),
        "adj": adj,
        "ghost_pos": np.array([gmap.ghost_aug_pos[vp] for vp in ghosts], dtype=np.float64).reshape(m, 3),
        "ghost_fronts": [[idx[f] for f in gmap.ghost_fronts[vp]] for vp in ghosts],
        "cur_node": idx[cur_vp], "cur_pos": np.asarray(cur_pos, dtype=np.float64), "cur_heading": float(cur_heading),
    }


def pack_batch(episodes: Sequence[dict]) -> Dict[str, np.ndarray]:
    """Pad the per-episode arrays to batch maxima (host side, O(total nodes + edges))."""
    B = len(episodes)
    Nmax = max(1, max(e["n_nodes"] for e in episodes))
    Mmax = max(e["n_ghost"] for e in episodes)
    Fmax = max(1, max(sum(len(f) for f in e["ghost_fronts"]) for e in episodes))
    if Nmax > MAX_NODES or Mmax > MAX_GHOSTS:
        raise ValueError(f"etp_gmap_assemble handles <= {MAX_NODES} visited nodes and <= {MAX_GHOSTS} ghosts per episode")
    out = {
        "node_pos": np.zeros((B, Nmax, 3), np.float32), "node_step": np.zeros((B, Nmax), np.int32),
        "n_nodes": np.zeros(B, np.int32), "adj": np.full((B, Nmax, Nmax), -1.0, np.float32),
        "ghost_pos": np.zeros((B, max(Mmax, 1), 3), np.float32), "n_ghost": np.zeros(B, np.int32),
        "front_ptr": np.zeros((B, Mmax + 1), np.int32), "front_idx": np.zeros((B, Fmax), np.int32),
        "cur_node": np.zeros(B, np.int32), "cur_pos": np.zeros((B, 3), np.float32), "cur_heading": np.zeros(B, np.float32),
    }
    for b, e in enumerate(episodes):
        n, m = e["n_nodes"], e["n_ghost"]
        out["n_nodes"][b], out["n_ghost"][b] = n, m
        out["node_pos"][b, :n] = e["node_pos"]
        out["node_step"][b, :n] = e["node_step"]
        out["adj"][b, :n, :n] = e["adj"]
        if m:
            out["ghost_pos"][b, :m] = e["ghost_pos"]
        q = 0
        for g, fr in enumerate(e["ghost_fronts"]):
            out["front_idx"][b, q:q + len(fr)] = fr
            q += len(fr)
            out["front_ptr"][b, g + 1] = q
        out["front_ptr"][b, m + 1:] = q
        out["cur_node"][b], out["cur_pos"][b], out["cur_heading"][b] = e["cur_node"], e["cur_pos"], e["cur_heading"]
    out["_dims"] = (B, Nmax, Mmax, Fmax)
    return out


def assemble_on_device(batch: Dict[str, np.ndarray], device, G: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """One H2D copy per compact array + one kernel.  Raises without the HIP library / a GPU (no CPU fallback)."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.EtpError("etp_gmap_assemble needs an MI355X (cuda/hip device); no CPU fallback exists")
    L = _lib.lib()
    B, Nmax, Mmax, Fmax = batch["_dims"]
    need = int((1 + batch["n_nodes"] + batch["n_ghost"]).max())
    G = need if G is None else G
    if G < need:
        raise ValueError(f"G={G} < 1 + nodes + ghosts = {need}")
    t = {k: torch.from_numpy(v).to(dev) for k, v in batch.items() if k != "_dims"}
    out = {
        "gmap_step_ids": torch.empty(B, G, dtype=torch.int64, device=dev),
        "gmap_masks": torch.empty(B, G, dtype=torch.bool, device=dev),
        "gmap_visited_masks": torch.empty(B, G, dtype=torch.bool, device=dev),
        "gmap_pos_fts": torch.empty(B, G, 7, dtype=torch.float32, device=dev),
        "gmap_pair_dists": torch.empty(B, G, G, dtype=torch.float32, device=dev),
    }
    check(L.etp_gmap_assemble(ptr(t["node_pos"]), ptr(t["node_step"]), ptr(t["n_nodes"]), ptr(t["adj"]), ptr(t["ghost_pos"]),
                              ptr(t["n_ghost"]), ptr(t["front_ptr"]), ptr(t["front_idx"]), ptr(t["cur_node"]), ptr(t["cur_pos"]),
                              ptr(t["cur_heading"]), B, Nmax, Mmax, Fmax, G, ptr(out["gmap_step_ids"]), ptr(out["gmap_masks"]),
                              ptr(out["gmap_visited_masks"]), ptr(out["gmap_pos_fts"]), ptr(out["gmap_pair_dists"]),
                              torch.cuda.current_stream(dev).cuda_stream), "etp_gmap_assemble")
    return out


def nav_gmap_variable(gmaps: Sequence, cur_vp: Sequence[str], cur_pos, cur_heading: Sequence[float], device) -> dict:
    """Drop-in for RLTrainer._nav_gmap_variable (ss_trainer_ETP.py:344-417) minus ``gmap_img_fts`` (see module docstring);
    ``cur_heading[i]`` replaces ``cur_ori[i]`` (= heading_from_quaternion(cur_ori[i]))."""
    eps = [pack_episode(g, cur_vp[i], cur_pos[i], cur_heading[i]) for i, g in enumerate(gmaps)]
    out = assemble_on_device(pack_batch(eps), device)
    out["gmap_vp_ids"] = [[None] + list(g.node_pos.keys()) + list(g.ghost_pos.keys()) for g in gmaps]
    out["no_vp_left"] = [len(g.ghost_pos) == 0 for g in gmaps]
    return out


# ---- node embeddings from a device-resident store (device-store mode) ---------------------------------------------------
def pack_img_csr(gmaps: Sequence, row_offsets: Sequence[int], G: int, n_store_rows: int):
    """CSR over the embedding store for the padded [B*G] node list ([stop] and padding: empty rows -> zeros), plus its
    transpose for the backward.  row_offsets[b] shifts episode b's row ids into the concatenated store."""
    ptr_f, idx_f, w_f = [0], [], []
    rev: List[list] = [[] for _ in range(n_store_rows)]
    for b, g in enumerate(gmaps):
        vps = [None] + list(g.node_pos.keys()) + list(g.ghost_pos.keys())
        if len(vps) > G:
            raise ValueError(f"G={G} < {len(vps)} graph entries")
        for t in range(G):
            if 1 <= t < len(vps):
                rows, w = g.embed_rows(vps[t])
                for r in rows:
                    idx_f.append(row_offsets[b] + r); w_f.append(w); rev[row_offsets[b] + r].append((b * G + t, w))
            ptr_f.append(len(idx_f))
    ptr_b, idx_b, w_b = [0], [], []
    for r in rev:
        for n, w in r:
            idx_b.append(n); w_b.append(w)
        ptr_b.append(len(idx_b))
    i32 = lambda x: torch.tensor(x, dtype=torch.int32)
    f32 = lambda x: torch.tensor(x, dtype=torch.float32)
    return (i32(ptr_f), i32(idx_f), f32(w_f)), (i32(ptr_b), i32(idx_b), f32(w_b))


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, store, n_out, fwd, bwd):
        L = _lib.lib()
        R, H = store.shape
        out = torch.empty(n_out, H, dtype=torch.float32, device=store.device)
        s = torch.cuda.current_stream(store.device).cuda_stream
        check(L.etp_gather_sum(_lib.ETP_F32, ptr(store), ptr(fwd[0]), ptr(fwd[1]), ptr(fwd[2]), ptr(out), n_out, H, 0, s),
              "etp_gather_sum")
        ctx.bwd, ctx.R = bwd, R
        return out

    @staticmethod
    def backward(ctx, d_out):
        L = _lib.lib()
        d_out = d_out.float().contiguous()
        H = d_out.shape[1]
        d_store = torch.empty(ctx.R, H, dtype=torch.float32, device=d_out.device)
        s = torch.cuda.current_stream(d_out.device).cuda_stream
        b = ctx.bwd
        check(L.etp_gather_sum(_lib.ETP_F32, ptr(d_out), ptr(b[0]), ptr(b[1]), ptr(b[2]), ptr(d_store), ctx.R, H, 0, s),
              "etp_gather_sum (transposed)")
        return d_store, None, None, None


def gather_rows(store: torch.Tensor, gmaps: Sequence, row_offsets: Sequence[int], G: int) -> torch.Tensor:
    """gmap_img_fts [B,G,H] (fp32) from the embedding store [R,H] on the device; differentiable w.r.t. the store."""
    if store.device.type != "cuda":
        raise _lib.EtpError("etp_gather_sum needs an MI355X (cuda/hip device); no CPU fallback exists")
    fwd, bwd = pack_img_csr(gmaps, row_offsets, G, store.shape[0])
    dev = store.device
    fwd = tuple(x.to(dev) for x in fwd)
    bwd = tuple(x.to(dev) for x in bwd)
    out = _GatherRows.apply(store.float().contiguous(), len(gmaps) * G, fwd, bwd)
    return out.view(len(gmaps), G, store.shape[1])


# ---- panorama inputs: candidate views first, then the remaining panorama views (row a13, first half) -------------------
def vp_feature_variable(obs: dict, device) -> dict:
    """Drop-in for RLTrainer._vp_feature_variable (ss_trainer_ETP.py:308-342) on the device: three gathers
    (etp_vp_gather) instead of per-episode torch.cat / pad loops.  `obs` keys as in the reference: cand_img_idxes,
    cand_rgb, cand_depth, cand_angle_fts (lists of per-episode tensors), pano_rgb [B,12,F], pano_depth [B,12,Fd],
    pano_angle_fts [12,4].  (Forward only: these are detached perception features in the reference as well.)"""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.EtpError("etp_vp_gather needs an MI355X (cuda/hip device); no CPU fallback exists")
    L = _lib.lib()
    B = len(obs["cand_rgb"])
    P = obs["pano_rgb"].shape[1]
    ks = [int(x.shape[0]) for x in obs["cand_rgb"]]
    cand_ptr = torch.tensor(np.concatenate([[0], np.cumsum(ks)]), dtype=torch.int32, device=dev)
    mask = torch.zeros(B, P, dtype=torch.uint8)
    for i in range(B):
        mask[i, torch.as_tensor(np.asarray(obs["cand_img_idxes"][i], dtype=np.int64))] = 1
    V = max(k + P - int(mask[i].sum()) for i, k in enumerate(ks))
    mask = mask.to(dev)
    s = torch.cuda.current_stream(dev).cuda_stream
    nav_types = torch.empty(B, V, dtype=torch.int64, device=dev)
    view_lens = torch.empty(B, dtype=torch.int64, device=dev)
    out = {}
    for name, cand_key, pano_key in (("rgb_fts", "cand_rgb", "pano_rgb"), ("dep_fts", "cand_depth", "pano_depth"),
                                     ("loc_fts", "cand_angle_fts", "pano_angle_fts")):
        cand = torch.cat([torch.as_tensor(x, dtype=torch.float32) for x in obs[cand_key]], 0).to(dev).contiguous()
        pano = torch.as_tensor(obs[pano_key], dtype=torch.float32).to(dev).contiguous()
        F = pano.shape[-1]
        stride = 0 if pano.dim() == 2 else P * F
        o = torch.empty(B, V, F, dtype=torch.float32, device=dev)
        first = name == "rgb_fts"
        check(L.etp_vp_gather(ptr(cand), ptr(cand_ptr), ptr(pano), stride, ptr(mask), B, P, F, V, ptr(o),
                              ptr(nav_types) if first else None, ptr(view_lens) if first else None, s), "etp_vp_gather")
        out[name] = o
    out["nav_types"], out["view_lens"] = nav_types, view_lens
    return out


# ---- pre-training: GlobalMapEncoder._aggregate_gmap_features (pretrain_src/pretrain_src/model/vilmodel.py:585-619) ------
def pack_traj_csr(traj_vp_lens: Sequence[Sequence[int]], traj_vpids: Sequence[Sequence[str]],
                  traj_cand_vpids: Sequence[Sequence[Sequence[str]]], gmap_vpids: Sequence[Sequence], V: int, G: int):
    """CSR (+ transpose) that turns the flat panorama embeddings [sum_i T_i, V, H] of a batch of trajectories into the
    padded node features [B, G, H]: a visited node = mean of the valid views of the step that visited it (a later visit
    overwrites an earlier one); an unvisited node = mean of the candidate-view embeddings (view j of step t) over the
    steps at which it was seen while not yet visited; entry 0 ([stop]) and padding are zero rows."""
    ptr_f, idx_f, w_f = [0], [], []
    n_rows = sum(len(x) for x in traj_vp_lens) * V
    rev: List[list] = [[] for _ in range(n_rows)]
    base = 0
    for i in range(len(traj_vp_lens)):
        visited, unvisited = {}, {}
        for t, vp in enumerate(traj_vpids[i]):
            n = int(traj_vp_lens[i][t])
            visited[vp] = ([(base + t) * V + j for j in range(n)], 1.0 / n)
            for j, cvp in enumerate(traj_cand_vpids[i][t]):
                if cvp not in visited:
                    # a candidate slot beyond the step's valid views is a zero row in the reference (embeds * vp_masks,
                    # :599-600) that still counts in the mean: keep the slot, drop the row
                    unvisited.setdefault(cvp, []).append((base + t) * V + j if j < n else None)
        if len(gmap_vpids[i]) > G:
            raise ValueError(f"G={G} < {len(gmap_vpids[i])} graph entries")
        for g in range(G):
            if 1 <= g < len(gmap_vpids[i]):
                vp = gmap_vpids[i][g]
                rows, w = visited[vp] if vp in visited else (unvisited[vp], 1.0 / len(unvisited[vp]))
                for r in rows:
                    if r is None:
                        continue
                    idx_f.append(r); w_f.append(w); rev[r].append((i * G + g, w))
            ptr_f.append(len(idx_f))
        base += len(traj_vp_lens[i])
    ptr_b, idx_b, w_b = [0], [], []
    for r in rev:
        for n, w in r:
            idx_b.append(n); w_b.append(w)
        ptr_b.append(len(idx_b))
    i32 = lambda x: torch.tensor(x, dtype=torch.int32)
    f32 = lambda x: torch.tensor(x, dtype=torch.float32)
    return (i32(ptr_f), i32(idx_f), f32(w_f)), (i32(ptr_b), i32(idx_b), f32(w_b))


def aggregate_gmap_features(traj_embeds: torch.Tensor, traj_vp_lens, traj_vpids, traj_cand_vpids, gmap_vpids, G: int):
    """Device version of GlobalMapEncoder._aggregate_gmap_features: traj_embeds [sum T, V, H] (output of the panorama
    encoder for every trajectory step) -> gmap_img_fts [B, G, H] incl. the zero [stop] row; differentiable."""
    if traj_embeds.device.type != "cuda":
        raise _lib.EtpError("etp_gather_sum needs an MI355X (cuda/hip device); no CPU fallback exists")
    R, V, H = traj_embeds.shape
    fwd, bwd = pack_traj_csr(traj_vp_lens, traj_vpids, traj_cand_vpids, gmap_vpids, V, G)
    dev = traj_embeds.device
    fwd, bwd = tuple(x.to(dev) for x in fwd), tuple(x.to(dev) for x in bwd)
    out = _GatherRows.apply(traj_embeds.float().contiguous().view(R * V, H), len(traj_vp_lens) * G, fwd, bwd)
    return out.view(len(traj_vp_lens), G, H)
