"""Cost of one rollout step's graph-input assembly for B=32 episodes after 15 steps:
host packing of the compact arrays + etp_gmap_assemble (H2D of the compact arrays + one kernel).  The reference's own
host path for the same episodes (real GraphMap + networkx, measured in the build container) is ~92 ms
per step plus ~20 ms of all-pairs Dijkstra per graph update.    python tools/graph_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from etpnav_amd.synthetic import simulate_rollout as simulate
from etpnav_amd.graph_inputs import GraphMapLite, pack_episode, pack_batch, assemble_on_device

B = 32
eps = [simulate(GraphMapLite, 100 + i, 15, True)[:4] for i in range(B)]
t0 = time.perf_counter()
for _ in range(20):
    batch = pack_batch([pack_episode(g, vp, pos, h) for g, vp, pos, h in eps])
t_pack = (time.perf_counter() - t0) / 20
for _ in range(3):
    out = assemble_on_device(batch, "cuda")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    out = assemble_on_device(batch, "cuda")
torch.cuda.synchronize()
t_dev = (time.perf_counter() - t0) / 50
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
import ctypes
from etpnav_amd import _lib
print(f"B={B}, nodes/ghosts of episode 0: {len(eps[0][0].node_pos)}/{len(eps[0][0].ghost_pos)}, G={out['gmap_pos_fts'].shape[1]}")
print(f"host pack {t_pack * 1e3:.2f} ms; H2D + kernel + output allocation {t_dev * 1e3:.3f} ms per step")
