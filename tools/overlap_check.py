import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from etpnav_amd.planner import GlocalTextPathNavCMT, default_config
from etpnav_amd.step import PlannerStep
from etpnav_amd.synthetic import make_batch
cfg = default_config("r2r", image_feat_size=768)
model = GlocalTextPathNavCMT(cfg, dtype=torch.bfloat16, device="cuda:0"); model.init_weights(seed=0)
batch = make_batch(cfg.vocab_size, cfg.image_feat_size, cfg.depth_feat_size, 32, 80, 36, 16)
ref = None
for mode in (False, False, "aux", "s2", True, True):
    step = PlannerStep(model, batch, overlap=mode)
    for it in range(3):
        step.run_eager(); torch.cuda.synchronize()
        g = model.flat_grads.clone(); outs = [step.txt.float().clone(), step.pano.float().clone(), step.gemb.float().clone(), step.logits.clone()]
        if ref is None:
            ref = (g, outs)
        dg = (g - ref[0]).abs().max().item()
        do = [float((a - b).abs().max()) for a, b in zip(outs, ref[1])]
        print(mode, it, "loss", step.loss.item(), "max|dgrad|", dg, "outs", do, "gmax", g.abs().max().item(), flush=True)
    step.close()
