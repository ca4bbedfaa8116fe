"""Helpers shared by the golden-vector tests (oracle on CPU, HIP path on GPU)."""
import ast
import os

import numpy as np
import torch

from oracle import planner_oracle as po
from oracle.make_golden import CASES, make_cfg, sample_idx, out_sample_idx

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    z = np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"))
    ckw = ast.literal_eval(str(z["meta.cfg"]))
    bkw = ast.literal_eval(str(z["meta.batch"]))
    assert (ckw, bkw) == CASES[name]
    cfg = make_cfg(**ckw)
    batch = po.make_batch(cfg, seed=1234, **bkw)
    return z, cfg, batch


def compare_outputs(z, outs, atol, rtol=0.0):
    """outs: dict of tensors named like planner_oracle.planner_step's outputs."""
    worst = {}
    pm = torch.from_numpy(z["out.pano_masks"])
    for k in ("txt_embeds", "pano_embeds", "gmap_embeds", "global_logits", "loss"):
        if f"osm.{k}" in z.files:      # large activation stored as strided samples + (sum, abs-max, L2) fingerprint
            got = outs[k].detach().double().cpu().reshape(-1)
            smp, fp = torch.from_numpy(z[f"osm.{k}"]).double(), z[f"ofp.{k}"]
            err = (got[torch.from_numpy(out_sample_idx(got.numel()))] - smp).abs()
            worst[k] = float(err.max())
            assert worst[k] <= atol + rtol * float(smp.abs().max()), f"{k}: sample err {worst[k]:.3e} > {atol}"
            assert abs(float(got.norm()) - fp[2]) <= (atol + rtol) * max(fp[2], 1.0), f"{k}: L2 fingerprint"
            continue
        ref = torch.from_numpy(z[f"out.{k}"])
        got = outs[k].detach().float().cpu().reshape(ref.shape)
        if k == "pano_embeds":      # padded query rows are don't-care
            ref, got = ref[pm], got[pm]
        fin = torch.isfinite(ref)
        assert torch.equal(torch.isfinite(got), fin), f"{k}: -inf pattern differs"
        err = (got[fin] - ref[fin]).abs()
        tol = atol + rtol * ref[fin].abs()
        worst[k] = float(err.max())
        assert bool((err <= tol).all()), f"{k}: max abs err {float(err.max()):.3e} > {atol}"
    return worst


def compare_grads(z, grads, atol, rel=None, rel_sample=None, abs_rel=0.0):
    """grads: name -> tensor.  Checks the 48 strided samples (abs) and the
    fingerprints (relative to the tensor's abs-max) of every parameter."""
    worst = 0.0
    names = [k[4:] for k in z.files if k.startswith("gfp.")]
    for name in names:
        if name.startswith("__input__") and name not in grads:
            continue
        assert name in grads, f"missing gradient {name}"
        g = grads[name].detach().double().cpu().reshape(-1)
        fp, smp = z[f"gfp.{name}"], torch.from_numpy(z[f"gsm.{name}"]).double()
        idx = torch.from_numpy(sample_idx(g.numel()))
        err = float((g[idx] - smp).abs().max())
        worst = max(worst, err)
        scale = max(fp[1], 1e-12)
        assert err <= atol + abs_rel * scale, f"grad {name}: sample err {err:.3e} > {atol} + {abs_rel}*{scale:.3e}"
        if rel_sample is not None:   # per-tensor relative bound on the samples
            assert err <= rel_sample * scale + 2e-6, f"grad {name}: sample err {err:.3e} vs abs-max {scale:.3e}"
        if rel is not None:
            assert abs(float(g.abs().max()) - fp[1]) <= rel * scale + atol, f"grad {name}: abs-max"
            assert abs(float(g.norm()) - fp[2]) <= rel * max(fp[2], 1e-12) + atol, f"grad {name}: L2"
    return worst


# ---- bf16 (performance mode) gradient bounds --------------------------------------------------------------------------
# Round-3 change: the bf16 bound used to be 8e-2 + 10 % of the tensor's abs-max.  Most gradient tensors
# of this model have abs-max < 8e-2 (278 of 307 at the c2 fixture), so an all-zero gradient passed.  Now every bound is
# RELATIVE TO THE TENSOR ITSELF (the reference's own bf16-autocast-vs-fp32 gap is ~7 % of abs-max per tensor):
#   fixtures (48 strided samples + fingerprints):  sample err <= 1e-1 * absmax_ref + 1e-4,  |L2 - L2_ref| <= 5e-2 * L2_ref + 1e-4
#   full tensors (fresh-input oracle tests):       ||g - g_ref|| <= 1.2e-1 * ||g_ref|| + 1e-4  and  cosine >= 0.99
# The 1e-4 floors only matter for tensors whose reference gradient is numerically zero.
# Calibration (GPU call r3-c1, profiles/r03_parity_bf16_observed.txt): the worst tensors of the bf16 step sit at 7.2 % of
# abs-max on samples (lang_encoder.layer.0.attention.self.value.bias) and 9.0 % relative L2 (word embeddings at B = 2: their
# gradient has passed all nine layers' bf16 products), everything else below 5 %; VERDICT r2 proposed 6 % / 5 %, which the
# reference's own bf16-vs-fp32 gap (~7 %) would not meet either.  tests/test_bf16_bounds_cpu.py: zeroed / 0.85-scaled /
# noise / sign-flipped tensors are all rejected at these bounds.
BF16_SAMPLE_REL, BF16_L2_REL, BF16_FULL_REL, BF16_ABS_FLOOR, BF16_COS_MIN = 1e-1, 5e-2, 1.2e-1, 1e-4, 0.99
# Round 5: SINGLE-EPISODE fixtures (B = 1: nine graph-node rows reach the SAP head) get their own bounds.  There the bf16 error is one
# COMMON factor of every gradient tensor -- it enters at the head (ReLU-mask flips and the LayerNorm backward over nine rows: d net.4 /
# d net.2 sit at 0.65 %, d net.0.weight already carries all of it) -- and it moves between 1 % and 26 % (median over the tensors; worst
# single tensor 33 %) with the INPUTS, for the round-4 library as much as for this round's: profiles/r05_b1_noise.txt (16 seeds, both
# libraries, paired).  The c1 fixture's seed sat at 3.8 % in round 4 by luck of the draw; any one-ulp change of a forward value moves it.
BF16_B1_SAMPLE_REL, BF16_B1_L2_REL = 3.5e-1, 2e-1


_GAP = None


def autocast_gap():
    """tests/golden/bf16_autocast_gap.json: the bf16-autocast-vs-fp32 gap of the REAL reference module on the CPU (generator:
    tools/experiments/r06_autocast_gap.py; table and the same-seed lottery check: profiles/r06_autocast_gap.txt)."""
    global _GAP
    if _GAP is None:
        import json
        _GAP = json.load(open(os.path.join(GOLDEN_DIR, "bf16_autocast_gap.json")))
    return _GAP


def _tier_quantile(prefix: str, key: str, q: float) -> float:
    vals = sorted(v[key] for k, v in autocast_gap()["seeds"].items() if k.startswith(prefix))
    return vals[min(len(vals) - 1, max(0, int(round(q * (len(vals) - 1)))))]


YARDSTICK_K = 2.0


def bf16_bounds(B: int, fixture: str = None) -> dict:
    """bf16 gradient bounds per batch-size tier, held against the YARDSTICK SURVEY.md §7 (i) names: the reference's own
    bf16-autocast-vs-fp32 gap, measured on the real module for the 16 single-episode and 8 three-episode seeds of
    profiles/r05_b1_noise.txt and for every fixture (profiles/r06_autocast_gap.txt).  What that table says:
      * B = 1: the reference's gap is 2 .. 15 % (median over the tensors), up to 18 % on a tensor and 23 % on a sample, seed to seed --
        and the SAME seed moves between 3 % and 20 % (28 % on a tensor) when the input changes by 1e-4 (lottery check, same file): the
        per-seed value is not a property of the seed, so a fixture is bounded by the tier's distribution, not by its own draw;
      * B = 3: 6 .. 11 % median, up to 19 %.
    A bound = min(round-5 tier, YARDSTICK_K x max(the fixture's own gap, the tier's upper-quartile gap)), YARDSTICK_K = 2; the cosine
    bound uses 1 - K^2 (1 - cos) (a relative error e costs e^2 / 2 of cosine).  For B = 1 this gives 34 % / 34 % / 0.95 against the
    blanket 35 % / 35 % / 0.93 of round 5; from B = 2 on the round-5 tiers are already tighter than 2 x the yardstick and stay.
    The HIP path's distribution over seeds is held to the yardstick's by tests/test_planner_gpu.py::
    test_small_batch_bf16_error_distribution_matches_the_autocast_yardstick.
    Keys: sample_rel / l2_rel for compare_grads_bf16, rel / cos_min for compare_full_bf16."""
    if B <= 1:
        base, prefix = dict(sample_rel=BF16_B1_SAMPLE_REL, l2_rel=BF16_B1_L2_REL, rel=0.35, cos_min=0.93), "B1_"
    elif B <= 4:
        base, prefix = dict(sample_rel=0.15, l2_rel=0.08, rel=0.18, cos_min=0.98), "B3_"
    else:
        return dict(sample_rel=BF16_SAMPLE_REL, l2_rel=BF16_L2_REL, rel=BF16_FULL_REL, cos_min=BF16_COS_MIN)
    fx = autocast_gap()["fixtures"].get(fixture, {}) if fixture else {}
    k = YARDSTICK_K
    rel = k * max(fx.get("max", 0.0), _tier_quantile(prefix, "max", 0.75))
    smp = k * max(fx.get("worst_sample", 0.0), _tier_quantile(prefix, "worst_sample", 0.75))
    cos = 1.0 - k * k * (1.0 - min(fx.get("min_cos", 1.0), _tier_quantile(prefix, "min_cos", 0.25)))
    return dict(sample_rel=min(base["sample_rel"], smp), l2_rel=base["l2_rel"], rel=min(base["rel"], rel),
                cos_min=max(base["cos_min"], cos))


def fixture_bounds(B: int, fixture: str = None) -> dict:
    b = bf16_bounds(B, fixture)
    return dict(sample_rel=b["sample_rel"], l2_rel=b["l2_rel"])


def full_bounds(B: int, fixture: str = None) -> dict:
    b = bf16_bounds(B, fixture)
    return dict(rel=b["rel"], cos_min=b["cos_min"])


def compare_grads_bf16(z, grads, sample_rel=BF16_SAMPLE_REL, l2_rel=BF16_L2_REL):
    """Golden-fixture check of a bf16-mode gradient set; returns (worst sample ratio, worst L2 ratio, their tensor names)."""
    worst_s, worst_l = (0.0, ""), (0.0, "")
    names = [k[4:] for k in z.files if k.startswith("gfp.")]
    for name in names:
        if name.startswith("__input__") and name not in grads:
            continue
        assert name in grads, f"missing gradient {name}"
        g = grads[name].detach().double().cpu().reshape(-1)
        fp, smp = z[f"gfp.{name}"], torch.from_numpy(z[f"gsm.{name}"]).double()
        idx = torch.from_numpy(sample_idx(g.numel()))
        err = float((g[idx] - smp).abs().max())
        amax, l2 = float(fp[1]), float(fp[2])
        assert err <= sample_rel * amax + BF16_ABS_FLOOR, \
            f"grad {name}: sample err {err:.3e} > {sample_rel} * abs-max {amax:.3e} + {BF16_ABS_FLOOR}"
        dl = abs(float(g.norm()) - l2)
        assert dl <= l2_rel * l2 + BF16_ABS_FLOOR, f"grad {name}: |L2 - L2_ref| {dl:.3e} > {l2_rel} * {l2:.3e}"
        if amax > 1e-6:
            worst_s = max(worst_s, (err / amax, name))
        if l2 > 1e-6:
            worst_l = max(worst_l, (dl / l2, name))
    return worst_s, worst_l


# Absolute floors of single tensors whose gradient is ONE scalar summed with cancellation over every (episode, head, query, key)
# score gradient: global_encoder.sprel_linear is Linear(1, 1) (vilmodel_cmt.py:619), |ref| = 7e-4 at B = 32 against score
# gradients that sum to ~1e-1 in absolute value; the bf16 error of that sum was 1.8e-4 (round 4) / 1.7e-4 (round 3).
BF16_NAMED_FLOOR = {"global_encoder.sprel_linear.weight": 3e-4, "global_encoder.sprel_linear.bias": 3e-4}


def compare_full_bf16(mine, ref, skip_prefix="__input__", rel=None, cos_min=None):
    """Full-tensor check of a bf16-mode gradient set against oracle gradients: relative L2 error and cosine per tensor.
    rel / cos_min default to the B <= 2 fixture bounds (12 % / 0.99); the benchmarked shapes pass their own, tighter ones
    (tests/test_baseline_shapes_gpu.py: set from the observed worst tensors, profiles/r04_parity_bf16_observed.txt)."""
    BF16_FULL_REL_, BF16_COS_MIN_ = (BF16_FULL_REL if rel is None else rel), (BF16_COS_MIN if cos_min is None else cos_min)
    worst_r, worst_c = (0.0, ""), (1.0, "")
    via_floor = []
    for k, g in ref.items():
        if k.startswith(skip_prefix):
            continue
        a, b = mine[k].detach().double().cpu().reshape(-1), g.detach().double().cpu().reshape(-1)
        nb, d = float(b.norm()), float((a - b).norm())
        floor = BF16_NAMED_FLOOR.get(k, BF16_ABS_FLOOR)
        assert d <= BF16_FULL_REL_ * nb + floor, f"grad {k}: ||err|| {d:.3e} > {BF16_FULL_REL_} * ||ref|| {nb:.3e} + {floor}"
        if d > BF16_FULL_REL_ * nb:        # within the bound only thanks to the absolute floor: say so
            via_floor.append(f"{k} (||err|| {d:.2e} = {d / max(nb, 1e-30):.1%} of ||ref|| {nb:.2e}, floor {floor:g})")
        if nb > 1e-3:
            cos = float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-30))
            assert cos >= BF16_COS_MIN_, f"grad {k}: cosine {cos:.5f} < {BF16_COS_MIN_}"
            worst_c = min(worst_c, (cos, k))
            worst_r = max(worst_r, (d / nb, k))
    if via_floor:
        print("bf16: tensors inside the bound only through the absolute floor:", "; ".join(via_floor))
    return worst_r, worst_c


def load_rollout():
    import numpy as np
    from oracle.make_golden_rollout import make_case, CASE
    z = np.load(os.path.join(GOLDEN_DIR, "rollout_t3.npz"))
    assert ast.literal_eval(str(z["meta.case"])) == CASE
    cfg, P, ids, masks, steps = make_case()
    return z, cfg, P, ids, masks, steps


def compare_rollout(z, outs, atol):
    """outs: planner_oracle.rollout_step-shaped dict (loss, txt_embeds, steps[t] = {gmap_embeds, global_logits})."""
    assert abs(float(outs["loss"]) - float(z["out.loss"])) <= atol, "loss"
    assert float((outs["txt_embeds"].detach().float().cpu() - torch.from_numpy(z["out.txt_embeds"])).abs().max()) <= atol
    for t, o in enumerate(outs["steps"]):
        ref = torch.from_numpy(z[f"out.logits.{t}"])
        got = o["global_logits"].detach().float().cpu()
        fin = torch.isfinite(ref)
        assert torch.equal(torch.isfinite(got), fin), f"step {t}: -inf pattern"
        assert float((got[fin] - ref[fin]).abs().max()) <= atol, f"step {t}: logits"
        e = float((o["gmap_embeds"].detach().float().cpu() - torch.from_numpy(z[f"out.gmap_embeds.{t}"])).abs().max())
        assert e <= atol, f"step {t}: gmap_embeds {e}"


def compare_full_fp32(got, model_grads, outs, grads, tol=1e-3):
    """BASELINE.json north_star's tolerance, full tensors: embeddings / logits / loss within `tol` absolute of the oracle, the -inf
    pattern of the logits identical, every parameter gradient within `tol` absolute AND (1e-3 relative L2 or a 2e-6 absolute floor
    for tensors that are zero in exact arithmetic).  `got`: txt_embeds / pano_embeds / gmap_embeds / global_logits / loss of the HIP
    step, `outs` / `grads`: the oracle's.  Returns (worst outputs dict, (worst abs err, name), (worst relative L2, name))."""
    worst = {}
    pm = outs["pano_masks"]
    for k in ("txt_embeds", "pano_embeds", "gmap_embeds"):
        a, b = got[k].float().cpu(), outs[k]
        if k == "pano_embeds":
            a, b = a[pm], b[pm]                             # padded query rows are don't-care
        worst[k] = float((a - b).abs().max())
        assert worst[k] <= tol, (k, worst[k])
    fin = torch.isfinite(outs["global_logits"])
    assert torch.equal(torch.isfinite(got["global_logits"].cpu()), fin)
    worst["logits"] = float((got["global_logits"].cpu()[fin] - outs["global_logits"][fin]).abs().max())
    worst["loss"] = abs(got["loss"].item() - outs["loss"].item())
    assert worst["logits"] <= tol and worst["loss"] <= tol, worst
    wg, wr = (0.0, ""), (0.0, "")
    for k, g in grads.items():
        if k.startswith("__input__"):
            continue
        err = float((model_grads[k] - g).abs().max())
        assert err <= tol, (k, err)
        nr = float(g.norm())
        rel = float((model_grads[k] - g).norm()) / nr if nr > 1e-6 else 0.0
        assert rel <= 1e-3 or err <= 2e-6, (k, rel, err)
        wg, wr = max(wg, (err, k)), max(wr, (rel, k))
    return worst, wg, wr
