"""Race screen of the whole planner step under its real three-stream schedule: run the same step N times (eval mode: identical
inputs and masks) and compare EVERY parameter gradient and output of every run with the element-wise median over the runs.
Atomically reduced tensors differ by summation order only (<= ~1e-6 of the tensor's magnitude); anything larger is a race or a
stale read.  Found the sporadic d(gmap_pos_embeddings.0.weight) corruption of rounds 3-4 (profiles/r04_gmap_pos_race.txt).

    python tools/determinism_screen.py [--runs 60] [--workload c2]
"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=60)
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--rel", type=float, default=2e-5)
    a = ap.parse_args()
    import bench
    from etpnav_amd.planner import GlocalTextPathNavCMT, default_config
    from etpnav_amd.step import PlannerStep
    from etpnav_amd.synthetic import make_batch
    w = dict(bench.WORKLOADS[a.workload])
    cfg = default_config(w["task"], image_feat_size=w["image_feat_size"])
    model = GlocalTextPathNavCMT(cfg, dtype=torch.bfloat16, device="cuda:0")
    model.init_weights(seed=0)
    batch = make_batch(cfg.vocab_size, cfg.image_feat_size, cfg.depth_feat_size, w["B"], w["L"], w["V"], w["G"], seed=1234)
    step = PlannerStep(model, batch, overlap=True, dropout=None)
    prm = dict(model.named_parameters())
    outs = ("txt", "pano", "gemb", "logits", "d_txt", "d_gimg", "d_pano")
    runs = []
    for _ in range(a.runs):
        step.run_eager(); torch.cuda.synchronize()
        snap = {n: p.grad.detach().clone() for n, p in prm.items() if p.numel() <= (1 << 22)}       # vectors, small matrices
        snap.update({"[sum] " + n: p.grad.detach().float().sum().reshape(1) for n, p in prm.items() if p.numel() > (1 << 22)})
        snap.update({"[out] " + n: getattr(step, n).detach().clone().float().nan_to_num(0.0, 0.0, 0.0) for n in outs})
        runs.append(snap)
    bad = []
    for n in runs[0]:
        stack = torch.stack([r[n].float().reshape(-1) for r in runs])
        med = stack.median(0).values
        scale = max(float(med.abs().max()), 1e-6)
        dev = (stack - med).abs().max(1).values / scale
        k = int((dev > a.rel).sum())
        if k:
            bad.append((float(dev.max()), n, k))
    bad.sort(reverse=True)
    print(f"# tools/determinism_screen.py: workload {a.workload}, {a.runs} runs of the three-stream step, eval mode; tensors whose deviation from the "
          f"per-element median exceeds {a.rel:g} of the tensor's abs-max in at least one run")
    for d, n, k in bad:
        print(f"  {n:78s} worst {d:.3e}   in {k}/{a.runs} runs")
    print(f"{len(bad)} of {len(runs[0])} tensors deviate" if bad else f"all {len(runs[0])} tensors reproduce within {a.rel:g}")


if __name__ == "__main__":
    main()
