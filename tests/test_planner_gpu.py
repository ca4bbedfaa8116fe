"""Planner parity on the MI355X: the HIP path (through the C ABI) against
  (a) the golden fixtures generated from the REAL reference (tests/golden/*.npz), and
  (b) the CPU oracle on fresh seeded inputs,
for outputs AND every parameter gradient.

Tolerances: fp32 parity mode must meet BASELINE.json's "within 1e-3 fp32" — we hold it to 2e-4 abs on
outputs/gradients (observed ~1e-5) plus a 2e-3 per-tensor relative bound on gradient samples.
bf16 performance mode (bf16 GEMM/attention operands, fp32 residual stream — autocast's policy) is compared with the
bound the reference itself shows between its bf16-autocast and fp32 runs (8e-3 logits, 2.3e-2 embeds,
5.7e-2 abs / ~7 % of abs-max on gradients): 5e-2 logits/embeds; gradients PER TENSOR, relative to that tensor: 10 % of its abs-max
on the fixture samples, 5 % on its L2 norm, and on full tensors relative L2 error <= 12 % with cosine >= 0.99
(tests/golden_util.py: compare_grads_bf16 / compare_full_bf16; tests/test_bf16_bounds_cpu.py shows the bounds can fail).
"""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import planner_oracle as po  # noqa: E402  (checker only)
from tests.golden_util import load_case, compare_outputs, compare_grads, compare_grads_bf16, compare_full_bf16  # noqa: E402
from etpnav_amd.planner import GlocalTextPathNavCMT  # noqa: E402
from etpnav_amd.step import PlannerStep  # noqa: E402

CASES = ["c1_single_episode", "ragged_small", "c2_shape_b2", "c5_g64_b2", "c4_rxr_b1"]


def build_model(cfg, P, dtype):
    m = GlocalTextPathNavCMT(cfg.to_dict(), dtype=dtype, device="cuda")
    m.load_state_dict({k: v for k, v in P.items()}, strict=True)
    return m.eval()      # parity fixtures are eval-mode (the reference's dropout RNG cannot be reproduced); see the train-mode tests


def grads_of(model):
    return {k: p.grad.detach().float().cpu() for k, p in model.named_parameters()}


def schedule_mismatches(model, ref_flat, rel, floor=1.0, k=6):
    """Parameters whose gradient in model.flat_grads deviates from ref_flat by more than rel * max(floor, |ref| of THAT
    parameter).  The two sides are the same computation under different stream schedules: everything on the dependent chain
    is bit-identical, leaf reductions that end in fp32 atomics (bias / LayerNorm / embedding-table gradients, e.g.
    gmap_pos_embeddings with its metre-scale position features) differ by summation order only -- hence a bound relative to
    the parameter's own magnitude, not to the largest gradient of the model."""
    rows = []
    for name, p in model.named_parameters():
        off = (p.grad.data_ptr() - model.flat_grads.data_ptr()) // 4
        r = ref_flat[off:off + p.numel()]
        d = (p.grad.detach().reshape(-1) - r).abs()
        m = d.max().item()
        if m > rel * max(floor, r.abs().max().item()):
            rows.append((m, name, int((d > 0).sum().item()), p.numel(), r.abs().max().item()))
    rows.sort(reverse=True)
    return "; ".join(f"{n}: max diff {m:.3e} of |ref| {a:.3e} ({c}/{t} differ)" for m, n, c, t, a in rows[:k])


def step_outputs(step):
    torch.cuda.synchronize()
    return {"txt_embeds": step.txt, "pano_embeds": step.pano, "gmap_embeds": step.gemb, "global_logits": step.logits,
            "loss": step.loss.reshape(())}


@pytest.mark.parametrize("name", CASES)
def test_fp32_step_matches_reference_golden(name):
    z, cfg, batch = load_case(name)
    P = po.init_params(cfg, seed=0)
    model = build_model(cfg, P, torch.float32)
    step = PlannerStep(model, batch)
    step.run_eager()
    worst = compare_outputs(z, step_outputs(step), atol=2e-4)
    g = compare_grads(z, grads_of(model), atol=2e-4, rel=2e-3, rel_sample=2e-3)
    print(name, "worst output err", worst, "worst grad err", g)


@pytest.mark.parametrize("name", ["ragged_small", "c2_shape_b2"])
def test_fp32_autograd_boundary_matches_golden(name):
    """Same check through the drop-in Python API (forward_txt / forward_panorama / forward_navigation + torch
    autograd + F.cross_entropy as in ss_trainer_ETP.py:801-892), including the gradient w.r.t. rgb_fts."""
    z, cfg, batch = load_case(name)
    P = po.init_params(cfg, seed=0)
    model = build_model(cfg, P, torch.float32)
    b = {k: v.cuda() for k, v in batch.items()}
    rgb = b["rgb_fts"].clone().requires_grad_(True)
    model.zero_grad()
    txt = model.forward_txt(b["txt_ids"], b["txt_masks"])
    pano, pmask = model.forward_panorama(rgb, b["dep_fts"], b["loc_fts"], b["nav_types"], b["view_lens"])
    G = b["gmap_step_ids"].shape[1]
    m = pmask.to(pano.dtype)
    avg = (pano * m[..., None]).sum(1) / m.sum(1, keepdim=True)
    idx = torch.arange(G - 2, device="cuda")[None, :] % b["view_lens"][:, None]
    views = torch.gather(pano, 1, idx[..., None].expand(-1, -1, pano.shape[-1]))
    gimg = torch.cat([torch.zeros_like(avg[:, None]), avg[:, None], views], 1)
    outs = model.forward_navigation(txt, b["txt_masks"], None, b["gmap_step_ids"], gimg, b["gmap_pos_fts"], b["gmap_masks"],
                                    b["gmap_visited_masks"], b["gmap_pair_dists"])
    loss = F.cross_entropy(outs["global_logits"], b["labels"], reduction="sum", ignore_index=-100) / b["txt_ids"].shape[0]
    loss.backward()
    torch.cuda.synchronize()
    compare_outputs(z, {"txt_embeds": txt, "pano_embeds": pano, "gmap_embeds": outs["gmap_embeds"],
                        "global_logits": outs["global_logits"], "loss": loss}, atol=2e-4)
    grads = grads_of(model)
    grads["__input__.rgb_fts"] = rgb.grad.float().cpu()
    compare_grads(z, grads, atol=2e-4, rel=2e-3, rel_sample=2e-3)


@pytest.mark.parametrize("name", ["c1_single_episode", "c2_shape_b2"])
def test_bf16_step_close_to_reference_golden(name):
    z, cfg, batch = load_case(name)
    P = po.init_params(cfg, seed=0)
    model = build_model(cfg, P, torch.bfloat16)
    step = PlannerStep(model, batch)
    step.run_eager()
    compare_outputs(z, step_outputs(step), atol=5e-2)
    from tests.golden_util import fixture_bounds      # per-batch-size tiers held against the reference's own autocast gap (golden_util.bf16_bounds)
    ws, wl = compare_grads_bf16(z, grads_of(model), **fixture_bounds(batch["txt_ids"].shape[0], name))
    print(name, "bf16 worst sample err / abs-max", ws, "worst |dL2| / L2", wl)


@pytest.mark.parametrize("B,seeds", [(1, (1, 2, 3, 4, 5, 6, 7, 8)), (3, (1, 2, 3, 4))])
def test_small_batch_bf16_error_distribution_matches_the_autocast_yardstick(B, seeds):
    """VERDICT r5 #5 / SURVEY.md §7 (i): at B <= 4 the bf16 gradient error is a lottery over the inputs -- for the reference's own
    bf16 autocast as much as for this path (profiles/r06_autocast_gap.txt: the same seed moves between 3 % and 20 % when the input
    changes by 1e-4) -- so a single fixture cannot tell a 10 % regression from an unlucky draw.  The DISTRIBUTION can: over the first
    seeds of that study (the c1 / rollout per-sample shape L = 20, V = 17, G = 9), the median over the seeds of the per-tensor median
    relative L2 error against the fp32 oracle must stay within 1.75 x the yardstick's median over the same seeds, and the worst seed
    within 2 x the yardstick's worst seed.  (Round-5 library, same seeds: 1.35 x the median, 1.8 x the worst.)"""
    from tests.golden_util import autocast_gap
    ref = autocast_gap()["seeds"]
    cfg = po.PlannerConfig.r2r()
    P = po.init_params(cfg, seed=0)
    model = build_model(cfg, P, torch.bfloat16)
    med, mx = [], []
    for seed in seeds:
        batch = po.make_batch(cfg, seed=seed, B=B, L=20, V=17, G=9, ragged=False)
        _, grads = po.step_with_grads(P, cfg, batch)
        step = PlannerStep(model, batch, overlap=False)
        step.run_eager()
        torch.cuda.synchronize()
        rel = []
        for k, p in model.named_parameters():
            r = grads[k].double().reshape(-1)
            if float(r.abs().max()) < 1e-6:
                continue
            rel.append(float((p.grad.detach().double().cpu().reshape(-1) - r).norm()) / float(r.norm()))
        t = torch.tensor(rel)
        med.append(float(t.median()))
        mx.append(float(t.quantile(0.9)))
        step.close()
    y_med = [ref[f"B{B}_seed{s}"]["median"] for s in seeds]
    y_p90 = [ref[f"B{B}_seed{s}"]["p90"] for s in seeds]
    mm, ym = float(torch.tensor(med).median()), float(torch.tensor(y_med).median())
    print(f"B={B}: per-seed median rel-L2 {[round(x, 4) for x in med]} (yardstick {[round(x, 4) for x in y_med]}); median over seeds "
          f"{mm:.4f} vs {ym:.4f} = {mm / ym:.2f} x; worst seed p90 {max(mx):.4f} vs {max(y_p90):.4f} = {max(mx) / max(y_p90):.2f} x")
    assert mm <= 1.75 * ym, (mm, ym)
    assert max(mx) <= 2.0 * max(y_p90), (max(mx), max(y_p90))


def test_fp32_step_vs_oracle_fresh_inputs_and_graph_replay():
    """Fresh seed (not in the fixtures), ragged lengths; also checks that hipGraph replay reproduces the eager step."""
    cfg = po.PlannerConfig.r2r()
    P = po.init_params(cfg, seed=11)
    batch = po.make_batch(cfg, B=4, L=33, V=19, G=10, seed=99, ragged=True)
    outs, grads = po.step_with_grads(P, cfg, batch)
    model = build_model(cfg, P, torch.float32)
    step = PlannerStep(model, batch)
    step.run_eager()
    got = step_outputs(step)
    fin = torch.isfinite(outs["global_logits"])
    assert torch.equal(torch.isfinite(got["global_logits"].cpu()), fin)
    assert (got["global_logits"].cpu()[fin] - outs["global_logits"][fin]).abs().max().item() < 2e-4
    assert abs(got["loss"].item() - outs["loss"].item()) < 2e-4
    mine = grads_of(model)
    for k, g in grads.items():
        if k.startswith("__input__"):
            continue
        err = (mine[k] - g).abs().max().item()
        assert err < 2e-4 + 2e-3 * g.abs().max().item(), f"{k}: {err}"
    eager_loss = got["loss"].item()
    eager_grad = model.flat_grads.clone()
    step.close()
    step = PlannerStep(model, batch, overlap="s2")   # graph replay: one side stream (see PlannerStep.capture)
    step.capture()
    step.replay(); step.replay()
    step.sync()
    assert abs(step.loss.item() - eager_loss) < 2e-6
    # atomically-accumulated sums may differ in the last bits between runs
    assert (model.flat_grads - eager_grad).abs().max().item() < 1e-4
    step.close()


def test_text_backward_in_layer_ranges_equals_single_call():
    """etp_txt_bwd_range over [6,9),[3,6),[0,3) (the data-parallel overlap schedule of bench.py) == one etp_txt_bwd."""
    cfg = po.PlannerConfig.r2r(vocab_size=4096)
    P = po.init_params(cfg, seed=2)
    batch = po.make_batch(cfg, B=3, L=24, V=14, G=8, seed=5, ragged=True)
    model = build_model(cfg, P, torch.bfloat16)
    step = PlannerStep(model, batch)
    step.run_eager(); torch.cuda.synchronize()
    ref = model.flat_grads.clone()
    s = model._engine.stream()
    step.enqueue_main(s, True, join_pano=True)
    for lo, hi in ((6, 9), (3, 6), (0, 3)):
        step.enqueue_txt_bwd(s, lo, hi)
    torch.cuda.synchronize()
    bad = schedule_mismatches(model, ref, rel=2e-5)
    assert not bad, bad


@pytest.mark.parametrize("overlapped", [True, False])
@pytest.mark.parametrize("B,L", [(3, 24), (32, 80)])
def test_data_parallel_issue_order_gives_the_single_gpu_gradients(overlapped, B, L):
    """PlannerStep.run_data_parallel (what bench.py's multi-rank path issues): text backward in layer groups with the bucket
    callbacks in between, free-running (side streams handed to the reducer) or joined -- after the last group everything is
    joined, and every gradient equals the plain step's.  The callbacks see bucket 0 (non-text) and one bucket per group, and in
    the overlapped form the side streams that hold their producers; a consumer that waits for exactly those streams (as
    dp.NativeComm.after does for the communication stream) must see complete gradients for the announced bucket."""
    from etpnav_amd import dp, _lib
    cfg = po.PlannerConfig.r2r(vocab_size=4096)
    P = po.init_params(cfg, seed=2)
    batch = po.make_batch(cfg, B=B, L=L, V=14 if B == 3 else 36, G=8 if B == 3 else 16, seed=5, ragged=True)
    model = build_model(cfg, P, torch.bfloat16)
    step = PlannerStep(model, batch)
    step.run_eager(); torch.cuda.synchronize()
    ref = model.flat_grads.clone()
    ranges, sparse, groups = dp.planner_buckets_layered(model, text_groups=3)
    Lb = _lib.lib()
    watcher = ctypes.c_void_p()
    _lib.check(Lb.etp_stream_create(ctypes.byref(watcher)), "stream_create")
    s = model._engine.stream()
    seen, snaps = [], []

    def bucket_ready(i, side):
        seen.append((i, tuple(side)))
        # a consumer stream ordered after the main stream and the announced side streams copies the bucket: what a
        # communication stream would read
        for st in (s,) + tuple(side):
            _lib.check(Lb.etp_stream_after(st, watcher.value), "stream_after")
        lo, hi = ranges[i]
        snap = torch.empty(hi - lo, device="cuda")
        with torch.cuda.stream(torch.cuda.ExternalStream(watcher.value)):
            snap.copy_(model.flat_grads[lo:hi], non_blocking=True)
        snaps.append((i, snap))

    for rep in range(2):
        seen.clear(); snaps.clear()
        step.run_data_parallel(groups, bucket_ready, stream=s, overlapped=overlapped)
        torch.cuda.synchronize()
        assert [i for i, _ in seen] == [0, 1, 2, 3]
        assert all((len(side) > 0) == overlapped for _, side in seen)
        bad = schedule_mismatches(model, ref, rel=2e-5)
        assert not bad, bad
        for i, snap in snaps:                       # the bucket was complete when its consumer read it
            lo, hi = ranges[i]
            d = (snap - ref[lo:hi]).abs().max().item()
            assert d <= 2e-5 * max(1.0, ref[lo:hi].abs().max().item()), (rep, i, d)
    _lib.check(Lb.etp_stream_destroy(watcher), "stream_destroy")
    step.close()


@pytest.mark.parametrize("dtype,rel", [(torch.float32, 2e-5), (torch.bfloat16, 2e-4)])
def test_micro_batched_step_equals_full_batch_step(dtype, rel):
    """MicroBatchedStep: two half-batch chains on independent stream sets, the second accumulating onto the first one's
    weight-gradient stores, must leave the gradient of the full-batch mean loss (ss_trainer_ETP.py:892) -- and re-store it
    on the next step instead of accumulating across steps."""
    from etpnav_amd.step import MicroBatchedStep
    cfg = po.PlannerConfig.r2r(vocab_size=4096)
    P = po.init_params(cfg, seed=2)
    batch = po.make_batch(cfg, B=4, L=24, V=14, G=8, seed=5, ragged=True)
    model = build_model(cfg, P, dtype)
    full = PlannerStep(model, batch)
    full.run_eager(); torch.cuda.synchronize()
    ref, loss_ref = model.flat_grads.clone(), full.loss.item()
    full.close()
    model.flat_grads.fill_(float("nan"))                 # the micro-batched step owns the zeroing / first-touch stores
    mb = MicroBatchedStep(model, batch, n_micro=2)
    for _ in range(2):
        mb.run_eager(); torch.cuda.synchronize()
        assert abs(mb.loss.item() - loss_ref) < (1e-5 if dtype == torch.float32 else 2e-3)
        bad = schedule_mismatches(model, ref, rel=rel)
        assert not bad, bad
    mb.close()


# ---- training mode: dropout at the reference's sites, masks from the documented counter-based generator ---------------
RATES = (0.1, 0.1, 0.1, 0.4)     # hidden, attention-probs, SAP head (vlnbert_init.py:58), drop_env (Policy_ViewSelection_ETP.py:102)


def _assert_step_matches(outs, grads, got, mine, atol=2e-4, rel=2e-3, bf16=False, B=8):
    """bf16=True: outputs to `atol`, gradients by the per-tensor relative bf16 bounds (rel is ignored)."""
    for k in ("txt_embeds", "pano_embeds", "gmap_embeds"):
        assert (got[k].float().cpu() - outs[k]).abs().max().item() < atol, k
    fin = torch.isfinite(outs["global_logits"])
    assert torch.equal(torch.isfinite(got["global_logits"].cpu()), fin)
    assert (got["global_logits"].cpu()[fin] - outs["global_logits"][fin]).abs().max().item() < atol
    assert abs(got["loss"].item() - outs["loss"].item()) < atol
    if bf16:
        from tests.golden_util import full_bounds
        wr, wc = compare_full_bf16(mine, grads, **full_bounds(B))
        print("bf16 worst rel-L2", wr, "worst cosine", wc)
        return
    for k, g in grads.items():
        if k.startswith("__input__"):
            continue
        err = (mine[k] - g).abs().max().item()
        assert err < atol + rel * g.abs().max().item(), f"{k}: {err}"


@pytest.mark.parametrize("L", [21, 70])      # 70 > 64: fp32 mode leaves the fused attention kernels (batched-GEMM path + drop_rows)
def test_fp32_train_mode_step_matches_oracle_with_same_masks(L):
    """policy.train() (ss_trainer_ETP.py:483): every dropout of the path active.  The oracle applies the same masks at
    the reference's dropout sites, so outputs and all gradients must agree to fp32 accuracy; two steps draw different
    masks; rates 0 reproduce eval."""
    cfg = po.PlannerConfig.r2r(vocab_size=2048)
    P = po.init_params(cfg, seed=4)
    batch = po.make_batch(cfg, B=3, L=L, V=12, G=9, seed=17, ragged=True)
    model = build_model(cfg, P, torch.float32)
    step = PlannerStep(model, batch, dropout=RATES, drop_seed=77)
    for k in (1, 2):
        step.run_eager()
        got = step_outputs(step)
        drop = po.DropSpec(*RATES, seed=(77 << 32) | k)
        outs, grads = po.step_with_grads(P, cfg, batch, drop=drop)
        _assert_step_matches(outs, grads, got, grads_of(model))
        if k == 1:
            first = got["loss"].item()
    assert abs(first - got["loss"].item()) > 1e-4          # fresh masks every step
    eval_outs, _ = po.step_with_grads(P, cfg, batch)
    assert abs(eval_outs["loss"].item() - got["loss"].item()) > 1e-4
    step.close()
    step = PlannerStep(model, batch, dropout=(0.0, 0.0, 0.0, 0.0))
    step.run_eager()
    assert abs(step_outputs(step)["loss"].item() - eval_outs["loss"].item()) < 2e-4
    step.close()


def test_fp32_train_mode_module_api_matches_oracle():
    """model.train() through the drop-in API + torch autograd: each entry-point call draws its own mask stream
    (seed = (seed_dropout << 32) | call counter) and its backward recomputes the same masks."""
    cfg = po.PlannerConfig.r2r(vocab_size=2048)
    P = po.init_params(cfg, seed=5)
    batch = po.make_batch(cfg, B=2, L=17, V=10, G=7, seed=23, ragged=True)
    model = build_model(cfg, P, torch.float32).train()
    model.seed_dropout(1234)
    b = {k: v.cuda() for k, v in batch.items()}
    rgb = b["rgb_fts"].clone().requires_grad_(True)
    model.zero_grad()
    txt = model.forward_txt(b["txt_ids"], b["txt_masks"])
    pano, pmask = model.forward_panorama(rgb, b["dep_fts"], b["loc_fts"], b["nav_types"], b["view_lens"])
    G = b["gmap_step_ids"].shape[1]
    m = pmask.to(pano.dtype)
    avg = (pano * m[..., None]).sum(1) / m.sum(1, keepdim=True)
    idx = torch.arange(G - 2, device="cuda")[None, :] % b["view_lens"][:, None]
    views = torch.gather(pano, 1, idx[..., None].expand(-1, -1, pano.shape[-1]))
    gimg = torch.cat([torch.zeros_like(avg[:, None]), avg[:, None], views], 1)
    outs = model.forward_navigation(txt, b["txt_masks"], None, b["gmap_step_ids"], gimg, b["gmap_pos_fts"], b["gmap_masks"],
                                    b["gmap_visited_masks"], b["gmap_pair_dists"])
    loss = F.cross_entropy(outs["global_logits"], b["labels"], reduction="sum", ignore_index=-100) / b["txt_ids"].shape[0]
    loss.backward()
    torch.cuda.synchronize()
    # oracle: the three calls used call counters 1, 2, 3
    r = (0.1, 0.1, 0.1, 0.0)          # default_config rates: hidden, attention_probs, pred_head; drop_env stays with the caller
    Pg = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    ob = dict(batch)
    ob["rgb_fts"] = batch["rgb_fts"].clone().requires_grad_(True)
    o_txt = po.forward_txt(Pg, cfg, ob["txt_ids"], ob["txt_masks"], po.DropSpec(*r, seed=(1234 << 32) | 1))
    o_pano, o_mask = po.forward_panorama(Pg, cfg, ob["rgb_fts"], ob["dep_fts"], ob["loc_fts"], ob["nav_types"], ob["view_lens"],
                                         po.DropSpec(*r, seed=(1234 << 32) | 2))
    o_gimg = po.assemble_gmap_img_fts(o_pano, o_mask, ob["view_lens"], G)
    o = po.forward_navigation(Pg, cfg, o_txt, ob["txt_masks"], ob["gmap_step_ids"], o_gimg, ob["gmap_pos_fts"], ob["gmap_masks"],
                              ob["gmap_visited_masks"], ob["gmap_pair_dists"], po.DropSpec(*r, seed=(1234 << 32) | 3))
    o_loss = po.cross_entropy_sum(o["global_logits"], ob["labels"]) / ob["txt_ids"].shape[0]
    o_loss.backward()
    ref_outs = {"txt_embeds": o_txt.detach(), "pano_embeds": o_pano.detach(), "gmap_embeds": o["gmap_embeds"].detach(),
                "global_logits": o["global_logits"].detach(), "loss": o_loss.detach()}
    ref_grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in Pg.items()}
    _assert_step_matches(ref_outs, ref_grads, {"txt_embeds": txt.detach(), "pano_embeds": pano.detach(),
                                               "gmap_embeds": outs["gmap_embeds"].detach(),
                                               "global_logits": outs["global_logits"].detach(), "loss": loss.detach()},
                         grads_of(model))
    assert (rgb.grad.cpu() - ob["rgb_fts"].grad).abs().max().item() < 2e-4
    # eval() switches every site off again
    model.eval()
    e_txt = model.forward_txt(b["txt_ids"], b["txt_masks"])
    assert (e_txt.detach().cpu() - po.forward_txt(P, cfg, batch["txt_ids"], batch["txt_masks"])).abs().max().item() < 2e-4


def test_bf16_train_mode_step_close_to_oracle_with_same_masks():
    cfg = po.PlannerConfig.r2r(vocab_size=2048)
    P = po.init_params(cfg, seed=6)
    batch = po.make_batch(cfg, B=4, L=40, V=20, G=12, seed=31, ragged=True)
    model = build_model(cfg, P, torch.bfloat16)
    step = PlannerStep(model, batch, dropout=RATES, drop_seed=5)
    step.run_eager()
    got = step_outputs(step)
    outs, grads = po.step_with_grads(P, cfg, batch, drop=po.DropSpec(*RATES, seed=(5 << 32) | 1))
    _assert_step_matches(outs, grads, got, grads_of(model), atol=8e-2, bf16=True, B=4)
    # ranged text backward (DP overlap schedule) recomputes the same masks
    ref = model.flat_grads.clone()
    s = model._engine.stream()
    step.step_no -= 1                      # same masks as the step above
    step.enqueue_main(s, True, join_pano=True)
    for lo, hi in ((6, 9), (3, 6), (0, 3)):
        step.enqueue_txt_bwd(s, lo, hi)
    torch.cuda.synchronize()
    bad = schedule_mismatches(model, ref, rel=2e-3, floor=5e-2)
    assert not bad, bad
    step.close()


# ---- text K/V cache across rollout steps -----------------------------------------------------------
def _rollout(model, batches, txt_ids, txt_masks):
    """ss_trainer_ETP.py:801-892 in miniature: one forward_txt, then T navigation steps on the SAME txt_embeds, losses
    summed, one backward."""
    model.zero_grad()
    txt = model.forward_txt(txt_ids, txt_masks)
    outs, loss = [], 0.0
    for b in batches:
        o = model.forward_navigation(txt, txt_masks, None, b["gmap_step_ids"], b["gmap_img_fts"], b["gmap_pos_fts"],
                                     b["gmap_masks"], b["gmap_visited_masks"], b["gmap_pair_dists"])
        outs.append(o)
        loss = loss + F.cross_entropy(o["global_logits"], b["labels"], reduction="sum", ignore_index=-100) / txt_ids.shape[0]
    loss.backward()
    torch.cuda.synchronize()
    return loss.detach(), [{k: v.detach().clone() for k, v in o.items()} for o in outs], grads_of(model)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 3e-2)])
def test_text_kv_cache_rollout_equals_per_step_projection(dtype, tol):
    cfg = po.PlannerConfig.r2r(vocab_size=2048)
    P = po.init_params(cfg, seed=9)
    B, L, G, T = 3, 22, 9, 3
    base = po.make_batch(cfg, B=B, L=L, V=8, G=G, seed=40, ragged=True)
    batches = []
    for t in range(T):
        bt = po.make_batch(cfg, B=B, L=L, V=8, G=G, seed=41 + t, ragged=True)
        gen = torch.Generator().manual_seed(100 + t)
        bt["gmap_img_fts"] = torch.randn(B, G, cfg.hidden_size, generator=gen) * 0.5
        batches.append({k: v.cuda() for k, v in bt.items()})
    model = build_model(cfg, P, dtype)
    ids, masks = base["txt_ids"].cuda(), base["txt_masks"].cuda()
    model.cache_text_kv = False
    loss0, outs0, g0 = _rollout(model, batches, ids, masks)
    model.cache_text_kv = True
    loss1, outs1, g1 = _rollout(model, batches, ids, masks)
    kv_first = model._kv_cache[2]
    assert abs(loss0.item() - loss1.item()) < tol
    for a, b in zip(outs0, outs1):
        fin = torch.isfinite(a["global_logits"])
        assert torch.equal(fin, torch.isfinite(b["global_logits"]))
        assert (a["global_logits"][fin] - b["global_logits"][fin]).abs().max().item() < tol
        assert (a["gmap_embeds"] - b["gmap_embeds"]).abs().max().item() < tol
    for k in g0:
        err = (g0[k] - g1[k]).abs().max().item()
        assert err < tol * (1.0 + 10.0 * g0[k].abs().max().item()), f"{k}: {err}"
    # one projection served all T steps (same tensor object); a new forward_txt result misses and re-projects
    txt2 = model.forward_txt(ids, masks)
    model.forward_navigation(txt2, masks, None, batches[0]["gmap_step_ids"], batches[0]["gmap_img_fts"], batches[0]["gmap_pos_fts"],
                             batches[0]["gmap_masks"], batches[0]["gmap_visited_masks"], batches[0]["gmap_pair_dists"])
    assert model._kv_cache[2] is not kv_first


# ---- the pre-training SAP unit (SURVEY.md §8d second unit; pretrain_cmt.py:223-283) -----------------------------------
@pytest.mark.parametrize("dtype,atol,rel", [(torch.float32, 2e-4, 2e-3), (torch.bfloat16, 8e-2, None)])
def test_sap_pretraining_step_matches_oracle(dtype, atol, rel):
    """T-step trajectories: panorama encoder over every step, node features aggregated over the steps (visited = pano
    mean, unvisited = mean of the candidate views that saw it), global encoder, SAP head, mean CE -- outputs and all
    gradients vs the oracle composition (each part pinned to the reference: forward_* by the golden fixtures,
    _aggregate_gmap_features by tests/golden/traj_agg.npz)."""
    from etpnav_amd.synthetic import make_sap_batch
    cfg = po.PlannerConfig.r2r(vocab_size=2048)
    P = po.init_params(cfg, seed=12)
    batch = make_sap_batch(cfg.vocab_size, cfg.image_feat_size, cfg.depth_feat_size, B=3, L=19, T=3, V=9, n_cand=4, seed=77,
                           ragged=True)
    outs, grads = po.sap_step_with_grads(P, cfg, batch)
    model = build_model(cfg, P, dtype)
    step = PlannerStep(model, batch)
    step.run_eager()
    got = step_outputs(step)
    _assert_step_matches(outs, grads, got, grads_of(model), atol=atol, rel=rel, bf16=dtype == torch.bfloat16)
    assert (step.gimg.cpu() - outs["gmap_img_fts"]).abs().max().item() < atol
    step.close()


# ---- the pre-training MLM task (pretrain_cmt.py:141-163) --------------------------------------------
@pytest.mark.parametrize("dtype,atol,rel", [(torch.float32, 3e-4, 2e-3), (torch.bfloat16, 1e-1, None)])
def test_mlm_pretraining_step_matches_oracle(dtype, atol, rel):
    """Same case as tests/golden/pretrain_tasks.npz (where the oracle is pinned to the REAL pre-training model): text ->
    forward_lang2visn through every x-layer -> tied MLM head on the masked tokens -> mean CE; the loss and the gradient of
    every parameter (incl. the tied decoder's contribution to the word embeddings and the language-side x-layer weights)."""
    from oracle.make_golden_pretrain import make_case
    from etpnav_amd.pretrain import MlmStep
    cfg, P, batch = make_case()
    outs, grads = po.mlm_step_with_grads(P, cfg, batch)
    model = build_model(cfg, P, dtype)
    step = MlmStep(model, batch)
    step.run_eager()
    torch.cuda.synchronize()
    assert abs(step.loss.item() - outs["loss"].item()) < atol * 3
    mine = grads_of(model)
    if dtype == torch.bfloat16:
        print("mlm bf16", compare_full_bf16(mine, grads))
    else:
        for k, g in grads.items():
            err = (mine[k] - g).abs().max().item()
            assert err < atol + rel * g.abs().max().item(), f"{k}: {err}"
    # train mode: dropout through the language-side blocks with the oracle's masks
    if dtype == torch.float32:
        step = MlmStep(model, batch, dropout=(0.1, 0.1, 0.1, 0.0), drop_seed=9)
        step.run_eager(); torch.cuda.synchronize()
        outs, grads = po.mlm_step_with_grads(P, cfg, batch, drop=po.DropSpec(0.1, 0.1, 0.1, 0.0, seed=(9 << 32) | 1))
        assert abs(step.loss.item() - outs["loss"].item()) < atol * 3
        mine = grads_of(model)
        for k, g in grads.items():
            err = (mine[k] - g).abs().max().item()
            assert err < atol + rel * g.abs().max().item(), f"train {k}: {err}"


def test_pretrain_driver_mixes_sap_and_mlm_steps_and_trains():
    """SURVEY.md §8f N3 remainder: MetaLoader task mixing + per-task steps + warm-up-linear LR + fused AdamW
    (train_r2r.py:229-300).  The first step of each task reproduces the oracle's loss for that batch (eval-mode dropout so the
    numbers are comparable), the schedule drives the optimizer's rate, same-shape SAP batches reuse one preallocated step, and
    the losses on the (two-batch) synthetic streams go down."""
    from oracle.make_golden_pretrain import make_case
    from etpnav_amd.pretrain import PretrainDriver
    from etpnav_amd.synthetic import make_sap_batch
    from etpnav_amd.optim import get_lr_sched
    cfg, P, mlm_batch = make_case()
    sap_batches = [make_sap_batch(cfg.vocab_size, cfg.image_feat_size, cfg.depth_feat_size, B=3, L=19, T=3, V=9, n_cand=4,
                                  seed=70 + i, ragged=False) for i in range(2)]
    model = build_model(cfg, P, torch.float32)
    g = torch.Generator().manual_seed(1)
    drv = PretrainDriver(model, {"mlm": ([mlm_batch], 1, lambda e: None), "sap": (sap_batches, 1, lambda e: None)},
                         learning_rate=2e-4, warmup_steps=4, num_train_steps=40, dropout=None, generator=g)
    ref = {"sap": po.sap_step_with_grads(P, cfg, sap_batches[0])[0]["loss"].item(),
           "mlm": po.mlm_step_with_grads(P, cfg, mlm_batch)[0]["loss"].item()}
    seen, first = {"sap": [], "mlm": []}, None
    for k, (name, loss) in enumerate(drv.run(24)):
        seen[name].append(loss)
        if k == 0:
            first = name
            assert abs(loss.item() - ref[name]) < 3e-4, (name, loss.item(), ref[name])     # untouched weights: oracle parity
        assert drv.lr_history[k] == pytest.approx(get_lr_sched(k + 1, 2e-4, 4, 40))     # rate used by optimizer step k+1
    torch.cuda.synchronize()
    assert len(seen["sap"]) >= 4 and len(seen["mlm"]) >= 4
    assert len(drv._sap) == 1                                           # both SAP batches share one preallocated step
    for name, ls in seen.items():
        v = [x.item() for x in ls]
        assert all(torch.isfinite(torch.tensor(v))), (name, v)
        assert sum(v[-2:]) / 2 < sum(v[:2]) / 2, (name, v)               # it trains
    drv.close()


def test_pretrain_driver_gradient_accumulation_matches_oracle():
    """gradient_accumulation_steps = 2 (train_r2r.py:231-300): two SAP micro-steps, each adding the gradient of
    mean-loss / 2, then ONE optimizer step.  The arena handed to the optimizer must be the mean of the two batches' oracle
    gradients; a second window must start from zero again."""
    from oracle.make_golden_pretrain import make_case
    from etpnav_amd.pretrain import PretrainDriver
    from etpnav_amd.synthetic import make_sap_batch
    cfg, P, _ = make_case()
    batches = [make_sap_batch(cfg.vocab_size, cfg.image_feat_size, cfg.depth_feat_size, B=3, L=19, T=3, V=9, n_cand=4,
                              seed=90 + i, ragged=False) for i in range(2)]
    model = build_model(cfg, P, torch.float32)
    drv = PretrainDriver(model, {"sap": (batches, 1, lambda e: None)}, learning_rate=1e-4, warmup_steps=2, num_train_steps=10,
                         dropout=None, accum_steps=2)
    seen = []
    real_step = drv.opt.step
    drv.opt.step = lambda *a, **k: (seen.append(model.flat_grads.clone()), real_step(*a, **k))[1]
    ref = {}
    for b in batches:
        _, g = po.sap_step_with_grads(P, cfg, b)
        for k, v in g.items():
            ref[k] = ref.get(k, 0) + v / 2
    res = drv.run(1)
    torch.cuda.synchronize()
    assert len(res) == 2 and len(seen) == 1 and drv.global_step == 1
    off = {n: (o, s) for n, s, o in model._engine.table}
    for k, v in ref.items():
        if k.startswith("__input__") or k not in off:
            continue
        o, shp = off[k]
        mine = seen[0][o:o + v.numel()].cpu().reshape(v.shape)
        err = (mine - v).abs().max().item()
        assert err < 2e-4 + 2e-3 * v.abs().max().item(), f"{k}: {err}"
    l0 = po.sap_step_with_grads(P, cfg, batches[0])[0]["loss"].item()
    assert abs(res[0][1].item() - l0 / 2) < 3e-4                 # the logged micro-step loss is loss / accumulation steps
    drv.close()
