import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, faulthandler
faulthandler.enable()
from etpnav_amd.planner import GlocalTextPathNavCMT, default_config
from etpnav_amd.step import PlannerStep
from etpnav_amd.synthetic import make_batch
mode = sys.argv[1]
mode = {"none": False, "both": True}.get(mode, mode)
cfg = default_config("r2r", image_feat_size=768)
model = GlocalTextPathNavCMT(cfg, dtype=torch.bfloat16, device="cuda:0"); model.init_weights(seed=0)
batch = make_batch(cfg.vocab_size, cfg.image_feat_size, cfg.depth_feat_size, 32, 80, 36, 16)
step = PlannerStep(model, batch, overlap=mode)
step.run_eager(); torch.cuda.synchronize(); print(mode, "eager loss", step.loss.item(), flush=True)
t0 = time.perf_counter()
for _ in range(10): step.run_eager()
torch.cuda.synchronize(); print(mode, "eager ms/step", (time.perf_counter() - t0) * 100, flush=True)
step.capture(); print(mode, "captured", flush=True)
for _ in range(3): step.replay()
step.sync()
t0 = time.perf_counter()
for _ in range(20): step.replay()
step.sync(); print(mode, "graph ms/step", (time.perf_counter() - t0) * 50, "loss", step.loss.item(), flush=True)
