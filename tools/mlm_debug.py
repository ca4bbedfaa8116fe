import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from oracle import planner_oracle as po
from oracle import make_golden_pretrain as mg
from etpnav_amd.planner import GlocalTextPathNavCMT
from etpnav_amd.pretrain import MlmStep

def run(nx, ragged, tag):
    mg.CASE["cfg"]["num_x_layers"] = nx
    mg.CASE["batch"]["ragged"] = ragged
    cfg, P, batch = mg.make_case()
    outs, grads = po.mlm_step_with_grads(P, cfg, batch)
    m = GlocalTextPathNavCMT(cfg.to_dict(), dtype=torch.float32, device="cuda"); m.load_state_dict(P, strict=True); m.eval()
    st = MlmStep(m, batch); st.run_eager(); torch.cuda.synchronize()
    gi = po.aggregate_gmap_features(po.forward_panorama(P, cfg, batch["rgb_fts"], batch["dep_fts"], batch["loc_fts"], batch["nav_types"], batch["view_lens"])[0], batch["traj"])
    G = batch["gmap_step_ids"].shape[1]
    print(tag, "nx", nx, "ragged", ragged, "G", G, "gimg shape", tuple(gi.shape), "loss", st.loss.item(), outs["loss"].item(),
          "gimg err", (st.gimg.cpu()[:, :gi.shape[1]] - gi).abs().max().item(), "gmask lens", batch["gmap_masks"].sum(1).tolist(),
          "view_lens", batch["view_lens"].tolist())

run(0, True, "A"); run(1, True, "B"); run(1, False, "C"); run(2, False, "D")
