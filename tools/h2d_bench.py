"""PCIe-inclusive rate of the planner step: a fresh batch is uploaded from pinned host memory before EVERY step.

bench.py's `value` starts with the inputs resident in HBM (the reference's trainer also holds its batch on the device when
policy.net runs: ss_trainer_ETP.py:801-805 `.cuda()` in the collate helpers).  This tool measures the same step when the
boundary is handed host buffers instead: PlannerStep.load_batch copies the 13 input tensors (non-blocking, from pinned memory)
and rebuilds the small node-aggregation index arrays, then the step runs.

    python tools/h2d_bench.py [--workload c2] [--steps 100] > profiles/r03_h2d_bench.json
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS  # noqa: E402
from etpnav_amd.planner import GlocalTextPathNavCMT, default_config  # noqa: E402
from etpnav_amd.step import PlannerStep  # noqa: E402
from etpnav_amd.synthetic import make_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=30)
    a = ap.parse_args()
    w = WORKLOADS[a.workload]
    cfg = default_config(w["task"], image_feat_size=w["image_feat_size"])
    model = GlocalTextPathNavCMT(cfg, dtype=torch.bfloat16, device="cuda")
    model.init_weights(seed=0)
    mk = lambda seed: make_batch(cfg.vocab_size, cfg.image_feat_size, cfg.depth_feat_size, w["B"], w["L"], w["V"], w["G"], seed=seed)
    host = [{k: v.pin_memory() for k, v in mk(1234 + i).items()} for i in range(4)]
    nbytes = sum(v.numel() * v.element_size() for v in host[0].values())
    step = PlannerStep(model, mk(1), overlap=True, dropout="config", drop_seed=0)

    def run(n, upload):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            if upload:
                step.load_batch(host[i % len(host)])
            step.run_eager()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    run(a.warmup, True)
    res = {"workload": a.workload, "batch_bytes_host": nbytes, "steps": a.steps,
           "ms_per_step_resident": round(run(a.steps, False), 4), "ms_per_step_with_upload": round(run(a.steps, True), 4)}
    res["steps_per_s_resident"] = round(1e3 / res["ms_per_step_resident"], 2)
    res["steps_per_s_with_upload"] = round(1e3 / res["ms_per_step_with_upload"], 2)
    res["note"] = ("with_upload: PlannerStep.load_batch from pinned host tensors before every step (13 non-blocking copies + the "
                   "node-aggregation index rebuild on the host), then the step; resident: the same loop without the upload")
    print(json.dumps(res))
    step.close()


if __name__ == "__main__":
    main()
