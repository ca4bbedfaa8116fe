"""Time the REAL reference planner (vlnce_baselines/models/etp/vilmodel_cmt.py, imported through oracle/ref_harness.py) on the
host cores of the BUILD container, BASELINE.json configs[1] (B=32, L=80, V=36x768, G=16), fp32, fwd+bwd.

    python tools/cpu_reference_timing.py > profiles/r03_cpu_reference.json

/root/reference does not exist on the GPU box, so bench.py's `cpu_baseline` there times the oracle port; this file is the
`kind: "reference"` counterpart measured where the reference can be imported (all physical cores, stated).
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from oracle import planner_oracle as po
from oracle import ref_harness as rh


def main():
    assert rh.reference_available(), "needs /root/reference"
    cores = len(os.sched_getaffinity(0))
    torch.set_num_threads(cores)
    out = {"workload": "BASELINE.json configs[1]: B=32, L=80, V=36x768, G=16, bert-base planner 9/2/4 layers, fp32, synthetic",
           "cores": cores, "kind": "reference", "unit": "steps/s", "runs": {}}
    cfg = po.PlannerConfig.r2r(image_feat_size=768)
    P = po.init_params(cfg, seed=0)
    model = rh.build_reference_model(cfg, P)
    batch = po.make_batch(cfg, B=32, L=80, V=36, G=16, seed=1234)
    for mode in ("eval", "train"):
        model.train(mode == "train")
        rh.reference_step(model, batch)                       # warm-up
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); rh.reference_step(model, batch); ts.append(time.perf_counter() - t0)
        ts.sort()
        out["runs"][mode] = {"median_s_per_step": round(ts[len(ts) // 2], 4), "steps_per_s": round(1.0 / ts[len(ts) // 2], 4),
                             "all_s": [round(t, 4) for t in ts]}
    # the oracle port on the same cores, for the port-vs-reference ratio
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    def one():
        for v in Pg.values(): v.grad = None
        po.planner_step(Pg, cfg, batch)["loss"].backward()
    one(); ts = []
    for _ in range(5):
        t0 = time.perf_counter(); one(); ts.append(time.perf_counter() - t0)
    ts.sort()
    out["oracle_port_eval"] = {"median_s_per_step": round(ts[2], 4), "steps_per_s": round(1.0 / ts[2], 4)}
    out["torch"] = torch.__version__
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
