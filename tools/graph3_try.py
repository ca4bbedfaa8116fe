import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, faulthandler
faulthandler.enable()
from etpnav_amd.planner import GlocalTextPathNavCMT, default_config
from etpnav_amd.step import PlannerStep
from etpnav_amd.synthetic import make_batch
cfg = default_config("r2r", image_feat_size=768)
model = GlocalTextPathNavCMT(cfg, dtype=torch.bfloat16, device="cuda:0"); model.init_weights(seed=0)
batch = make_batch(cfg.vocab_size, cfg.image_feat_size, cfg.depth_feat_size, 32, 80, 36, 16)
step = PlannerStep(model, batch, overlap=True, dropout="config")
step.run_eager(); torch.cuda.synchronize(); print("eager loss", step.loss.item(), flush=True)
step.capture(); print("captured", flush=True)
for _ in range(3): step.replay()
step.sync()
t0 = time.perf_counter()
for _ in range(50): step.replay()
step.sync(); print("graph(3 streams) ms/step", (time.perf_counter() - t0) * 20, "loss", step.loss.item(), flush=True)
