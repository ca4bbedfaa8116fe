"""Round 6 (SURVEY.md §7 hard part (i)): the yardstick for the bf16 bounds of the small-batch fixtures.

The REAL reference planner (vilmodel_cmt.py GlocalTextPathNavCMT, imported through oracle/ref_harness.py; build container only) runs
the same step twice on the CPU -- fp32, and under torch.autocast("cpu", torch.bfloat16) (the bf16 analogue of the fp16 autocast the
reference trains under, ss_trainer_ETP.py:502-504: Linear / matmul in bf16, LayerNorm / softmax / cross-entropy in fp32) -- and the
per-tensor relative L2 error of the autocast gradients against the fp32 ones is recorded exactly as tools/experiments/r05_b1_noise.py
records the HIP path's: median / p90 / max over the tensors with a non-zero gradient.  Same seeds as profiles/r05_b1_noise.txt
(16 single-episode batches, 8 three-episode batches) plus every committed fixture that has a bf16 GPU test.

    python tools/experiments/r06_autocast_gap.py > profiles/r06_autocast_gap.txt      # ~10 min, CPU only
Also writes tests/golden/bf16_autocast_gap.json (fixture -> median / p90 / max / worst sample ratio), which tests/golden_util.py reads.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

torch.set_num_threads(int(os.environ.get("GAP_THREADS", "4")))
from oracle import planner_oracle as po
from oracle import ref_harness as rh
from oracle.make_golden import CASES, make_cfg, sample_idx


def gap(model, batch):
    o32, g32 = rh.reference_step(model, batch)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        o16, g16 = rh.reference_step(model, batch)
    rel, worst_sample = [], (0.0, "")
    for k, r in g32.items():
        r = r.double().reshape(-1)
        if float(r.abs().max()) < 1e-6:
            continue
        g = g16[k].double().reshape(-1)
        rel.append((float((g - r).norm()) / float(r.norm()), k))
        idx = torch.from_numpy(sample_idx(r.numel()))                 # what compare_grads_bf16 looks at: 48 strided samples / abs-max
        s = float((g[idx] - r[idx]).abs().max()) / float(r.abs().max())
        if s > worst_sample[0]:
            worst_sample = (s, k)
    t = torch.tensor([x for x, _ in rel])
    fin = torch.isfinite(o32["global_logits"])
    cos = min(float(torch.nn.functional.cosine_similarity(g16[k].double().reshape(-1), g32[k].double().reshape(-1), dim=0))
              for _, k in rel)
    return dict(loss32=float(o32["loss"]), loss16=float(o16["loss"]),
                logits_err=float((o16["global_logits"].float()[fin] - o32["global_logits"][fin]).abs().max()),
                n=len(rel), median=float(t.median()), p90=float(t.kthvalue(max(1, int(0.9 * len(rel)))).values), max=float(t.max()),
                max_name=max(rel)[1], worst_sample=worst_sample[0], worst_sample_name=worst_sample[1], min_cos=cos)


def main():
    t0 = time.time()
    cfg = po.PlannerConfig.r2r()
    P = po.init_params(cfg, seed=0)
    model = rh.build_reference_model(cfg, P)
    print("# reference GlocalTextPathNavCMT (real module), CPU: bf16 autocast vs its own fp32 run; per-tensor relative L2 of the gradients")
    print("# --- B = 1 (c1 shape L=20 V=17 G=9), the 16 seeds of profiles/r05_b1_noise.txt")
    table = {}
    for B, seeds in ((1, (1234,) + tuple(range(1, 16))), (3, tuple(range(1, 9)))):
        if B == 3:
            print("# --- B = 3 (rollout fixture shape), the 8 seeds of the second table of profiles/r05_b1_noise.txt")
        for seed in seeds:
            batch = po.make_batch(cfg, seed=seed, B=B, L=20, V=17, G=9, ragged=False)
            r = gap(model, batch)
            table[f"B{B}_seed{seed}"] = r
            print(f"B {B} seed {seed:5d}: loss {r['loss16']:.4f} (fp32 {r['loss32']:.4f}), logits err {r['logits_err']:.2e}; relative L2 over "
                  f"{r['n']} tensors: median {r['median']:.4f}, p90 {r['p90']:.4f}, max {r['max']:.4f} ({r['max_name']}); worst sample/abs-max "
                  f"{r['worst_sample']:.4f}; min cosine {r['min_cos']:.4f}", flush=True)
    del model
    print("# --- committed fixtures with a bf16 GPU test (oracle/make_golden.py CASES: same seeds, same weights)")
    fixtures = {}
    only = os.environ.get("GAP_CASES")
    for name, (ckw, bkw) in CASES.items():
        if only and name not in only.split(","):
            continue
        if name == "c4_rxr_l512_b2" and os.environ.get("GAP_BIG", "1") == "0":
            continue
        c = make_cfg(**ckw)
        Pc = po.init_params(c, seed=0)
        m = rh.build_reference_model(c, Pc)
        batch = po.make_batch(c, seed=1234, **bkw)
        r = gap(m, batch)
        fixtures[name] = {k: (round(v, 5) if isinstance(v, float) else v) for k, v in r.items()}
        print(f"fixture {name:20s} B {bkw['B']}: median {r['median']:.4f}, p90 {r['p90']:.4f}, max {r['max']:.4f} ({r['max_name']}); worst "
              f"sample/abs-max {r['worst_sample']:.4f} ({r['worst_sample_name']}); min cosine {r['min_cos']:.4f}", flush=True)
        del m, Pc
    out = {"note": "bf16-autocast vs fp32 gap of the REAL reference module on the CPU (tools/experiments/r06_autocast_gap.py); "
                   "median / p90 / max = per-tensor relative L2 of the gradients, worst_sample = worst |sample error| / abs-max over the "
                   "48 strided samples compare_grads_bf16 checks, min_cos = worst per-tensor cosine",
           "seeds": {k: {kk: (round(vv, 5) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in table.items()},
           "fixtures": fixtures}
    dst = os.path.join(ROOT, "tests", "golden", "bf16_autocast_gap.json")
    if not only:
        json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
        print(f"# wrote {dst}; {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
