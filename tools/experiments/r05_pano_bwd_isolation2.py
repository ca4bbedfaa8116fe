"""round 5, follow-up of r05_pano_bwd_isolation.py (same experiment library): WHERE do the results of pano_embed_bwd differ when it
shares the chip with the step's other streams?  Reference = the launches run alone (drain before and after); then ten repetitions
each of: all three streams; only the panorama stream beside the chain (no weight-gradient stream); only the weight-gradient stream
(panorama branch on the chain's stream); all three streams without stream priorities.  Reported per mode:
  * da / dd (the per-row outputs, plain stores): rows that differ from the reference, elements per row, size of the difference
  * every gradient the three launches accumulate: repetitions affected, relative error, how many of the 768 columns are off
"""
import os
import sys
import time

t_start = time.time()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from etpnav_amd.planner import GlocalTextPathNavCMT, default_config  # noqa: E402
from etpnav_amd.step import PlannerStep  # noqa: E402
from etpnav_amd.synthetic import make_batch, make_sap_batch  # noqa: E402

RUNS = int(os.environ.get("RUNS", "10"))
WLS = os.environ.get("WL", "c2,c5").split(",")
w = dict(bench.WORKLOADS[WLS[0]])
cfg = default_config(w["task"], image_feat_size=w["image_feat_size"])
model = GlocalTextPathNavCMT(cfg, dtype=torch.bfloat16, device="cuda")
model.init_weights(seed=0)
prm = dict(model.named_parameters())
EMB = [n for n in prm if n.startswith("img_embeddings.") and "pano_encoder" not in n] + ["embeddings.token_type_embeddings.weight"]
H, I = 768, 3072


def run(overlap, lds, drain, reps, prio="1"):
    os.environ["ETP_PANO_BWD_LDS"] = str(lds)
    os.environ["ETP_PANO_BWD_DRAIN"] = str(drain)
    os.environ["ETP_STREAM_PRIO"] = prio
    step = PlannerStep(model, batch, overlap=overlap, dropout=None, drop_seed=9)
    M = step.Bp * step.V
    plan = (step.ws_pano.numel() - 256) // 4
    o_dd = 3 * plan + M * H * 16
    o_da = o_dd + M * I * 2
    out = []
    for _ in range(reps):
        step.step_no = 0
        step.run_eager()
        torch.cuda.synchronize()
        ws = step.ws_pano
        dd = ws[o_dd:o_dd + M * H * 2].view(torch.bfloat16).view(M, H).clone()
        da = ws[o_da:o_da + M * H * 2].view(torch.bfloat16).view(M, H).clone()
        spare = ws[o_dd + M * H * 2:o_da].clone()                 # the rest of dI: nothing of the embedding backward writes there
        out.append((da, dd, spare, {n: prm[n].grad.detach().double().clone() for n in EMB}))
    step.close()
    return out, M


def rows_report(tag, x, ref):
    bad = (x != ref).any(1).nonzero().reshape(-1)
    if not bad.numel():
        return f"{tag} 0 rows"
    nd = (x[bad] != ref[bad]).sum(1)
    rel = ((x[bad].float() - ref[bad].float()).abs().max(1).values / ref[bad].float().abs().max(1).values.clamp_min(1e-20))
    ex = ", ".join(f"row {int(r)} (blk-slot {int(r) // 4}, wave {int(r) % 4}): {int(n)} el, {float(e):.1e}" for r, n, e in
                   list(zip(bad.tolist(), nd.tolist(), rel.tolist()))[:4])
    return (f"{tag} {bad.numel()} rows; elements/row min {int(nd.min())} med {int(nd.median())} max {int(nd.max())}; "
            f"rel diff min {float(rel.min()):.1e} max {float(rel.max()):.1e}; e.g. {ex}")


for key in WLS:
    w = dict(bench.WORKLOADS[key])
    if key == "sap":
        batch = make_sap_batch(cfg.vocab_size, cfg.image_feat_size, cfg.depth_feat_size, 8, w["L"], w["T"], w["V"], seed=1234)
    else:
        batch = make_batch(cfg.vocab_size, cfg.image_feat_size, cfg.depth_feat_size, w["B"], w["L"], w["V"], w["G"], seed=1234)
    (r0, r1), M = run(True, 12288, 2, 2)
    if os.environ.get("SAVE_REF"):              # reference of THIS library for a later process running another library
        torch.save({"da": r0[0].cpu(), "dd": r0[1].cpu(), "g": {n: v.cpu() for n, v in r0[3].items()}}, os.environ["SAVE_REF"] + "." + key)
        print(f"== {key}: reference saved"); continue
    if os.environ.get("CMP_REF") and os.path.exists(os.environ["CMP_REF"] + "." + key):
        z = torch.load(os.environ["CMP_REF"] + "." + key)
        ge = max(float((r0[3][n].cpu() - z["g"][n]).abs().max() / z["g"][n].abs().max().clamp_min(1e-20)) for n in EMB)
        print(f"== {key}: against the other library's reference (both alone): da bit-identical {bool((r0[0].cpu() == z['da']).all())}, "
              f"dd bit-identical {bool((r0[1].cpu() == z['dd']).all())}, gradients within {ge:.1e}")
    same = bool((r0[0] == r1[0]).all() and (r0[1] == r1[1]).all())
    gref = r0[3]
    gdev = max(float((r1[3][n] - gref[n]).abs().max() / gref[n].abs().max().clamp_min(1e-20)) for n in EMB)
    print(f"== {key}: M = {M} rows; reference (alone) twice: da/dd bit-identical {same}, gradients within {gdev:.1e}  (t = {time.time() - t_start:.1f} s)", flush=True)
    modes = [("three streams", True, "1"), ("panorama stream only (overlap='s2')", "s2", "1"),
             ("weight-gradient stream only (overlap='aux')", "aux", "1"), ("three streams, no priorities", True, "0")]
    if os.environ.get("MODES"):
        modes = [modes[int(i)] for i in os.environ["MODES"].split(",")]
    for tag, overlap, prio in modes:
        if time.time() - t_start > float(os.environ.get("BUDGET_S", "100")):
            print(tag, "skipped (time budget)")
            continue
        reps, _ = run(overlap, 12288, 0, RUNS, prio)
        nbad_da = [int((r[0] != r0[0]).any(1).sum()) for r in reps]
        nbad_dd = [int((r[1] != r0[1]).any(1).sum()) for r in reps]
        spare_changed = any(not bool((r[2] == reps[0][2]).all()) for r in reps[1:])
        print(f"-- {tag}: rows of da differing per repetition {nbad_da}; dd {nbad_dd}; unused part of the dI buffer changed: {spare_changed}")
        worst = max(range(len(reps)), key=lambda i: nbad_da[i] + nbad_dd[i])
        print("   " + rows_report(f"da, repetition {worst}:", reps[worst][0], r0[0]))
        print("   " + rows_report(f"dd, repetition {worst}:", reps[worst][1], r0[1]))
        for n in EMB:
            ref = gref[n].reshape(-1)
            scale = float(ref.abs().max().clamp_min(1e-20))
            errs = [float((r[3][n].reshape(-1) - ref).abs().max()) / scale for r in reps]
            hit = [i for i, e in enumerate(errs) if e > 2e-5]
            if hit:
                i = max(hit, key=lambda j: errs[j])
                d = (reps[i][3][n].reshape(-1) - ref).abs()
                cols = (d > 2e-5 * scale).nonzero().reshape(-1)
                print(f"   {n}: {len(hit)}/{len(reps)} repetitions off, worst {errs[i]:.1e} (repetition {i}): {cols.numel()} of {ref.numel()} "
                      f"elements off, first {cols[:6].tolist()} last {cols[-3:].tolist()}")
        sys.stdout.flush()
print(f"total {time.time() - t_start:.1f} s")
