"""Round-5 parity additions on the MI355X (VERDICT r4 "Next round" item 1), all through the C ABI of libetpnav_hip.so:

  * the frozen / ablated model variants the boundary reads (vlnbert_init.py:42-54 -> vilmodel_cmt.py:422-433,675-682):
    fix_lang_embedding, fix_pano_embedding, use_sprels = False, use_depth_embedding = False -- each against a fixture produced by
    the REAL reference (tests/golden/{fix_lang,fix_pano,no_sprels,no_depth}_small.npz, generator oracle/make_golden.py), through
    PlannerStep and through the module API + torch autograd; frozen parameters get no gradient, leave the data-parallel buckets
    and are left alone by FusedAdamW;
  * BASELINE.json's tolerance at BASELINE.json's headline shape: configs[1] (B = 32, L = 80, V = 36 x 768, G = 16) in the fp32
    parity mode, eval and train mode (same dropout masks), outputs and ALL 307 parameter gradients full-tensor against the oracle
    at north_star's 1e-3 (until round 4 the fp32 whole-step checks ran at B <= 3 only: M = 2560 rows takes other tile classes,
    one-row-per-wavefront LayerNorm grids and 640-block reductions);
  * the race screen of DESIGN.md §3.6 (tools/determinism_screen.py) as a test: the three-stream step repeated on identical inputs
    and masks must reproduce every gradient (config 2 in train mode, config 5, the SAP pre-training unit).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import planner_oracle as po  # noqa: E402  (checker only)
from oracle import optim_oracle as oo  # noqa: E402
from tests.golden_util import load_case, compare_outputs, compare_grads  # noqa: E402
from tests import golden_util as gu  # noqa: E402
from etpnav_amd.planner import GlocalTextPathNavCMT  # noqa: E402
from etpnav_amd.step import PlannerStep  # noqa: E402

VARIANTS = ["fix_lang_small", "fix_pano_small", "no_sprels_small", "no_depth_small"]


def build_model(cfg, P, dtype):
    m = GlocalTextPathNavCMT(cfg.to_dict(), dtype=dtype, device="cuda")
    m.load_state_dict({k: v for k, v in P.items()}, strict=True)
    return m.eval()


def module_grads(model):
    """.grad of every parameter; frozen parameters have .grad = None (as in the reference, whose fixture stores zeros for them)"""
    out = {}
    for k, p in model.named_parameters():
        assert (p.grad is None) == (not p.requires_grad), k
        out[k] = p.grad.detach().float().cpu() if p.grad is not None else torch.zeros(p.shape)
    return out


def step_outputs(step):
    torch.cuda.synchronize()
    return {"txt_embeds": step.txt, "pano_embeds": step.pano, "gmap_embeds": step.gemb, "global_logits": step.logits,
            "loss": step.loss.reshape(())}


@pytest.mark.parametrize("name", VARIANTS)
def test_variant_step_matches_reference_golden(name):
    z, cfg, batch = load_case(name)
    P = po.init_params(cfg, seed=0)
    assert set(P) == {n for n, _, _ in GlocalTextPathNavCMT(cfg.to_dict(), dtype=torch.float32, device="cpu")._engine.table}
    model = build_model(cfg, P, torch.float32)
    step = PlannerStep(model, batch)
    step.run_eager()
    worst = compare_outputs(z, step_outputs(step), atol=2e-4)
    g = compare_grads(z, module_grads(model), atol=2e-4, rel=2e-3, rel_sample=2e-3)
    frozen = [k for k, p in model.named_parameters() if not p.requires_grad]
    assert sorted(frozen) == sorted(k for k in P if po.is_frozen(cfg, k))
    assert step.train_txt == (not cfg.fix_lang_embedding) and step.train_pano     # fix_pano alone: token_type(1) still trains
    print(name, "worst output err", worst, "worst grad err", g, "frozen", len(frozen))
    # a second step into the same arena (overwrite mode for the touched matrices, zeroed tail): still the same gradients
    step.run_eager()
    torch.cuda.synchronize()
    compare_grads(z, module_grads(model), atol=2e-4, rel=2e-3, rel_sample=2e-3)
    step.close()


def test_both_sides_frozen_skip_the_text_and_panorama_backward():
    """fix_lang_embedding AND fix_pano_embedding: nothing behind the text encoder or the panorama branch requires a gradient, so the
    step runs neither backward (autograd would not either); the trainable half (global_encoder.*, global_sap_head.*) must equal
    the oracle's gradients and the frozen slots of the arena stay untouched (zero)."""
    cfg = po.PlannerConfig.r2r(fix_lang_embedding=True, fix_pano_embedding=True)
    P = po.init_params(cfg, seed=0)
    batch = po.make_batch(cfg, B=3, L=12, V=14, G=7, seed=77, ragged=True)
    outs, grads = po.step_with_grads(P, cfg, batch)
    for dtype, tol in ((torch.float32, 2e-4), (torch.bfloat16, None)):
        model = build_model(cfg, P, dtype)
        step = PlannerStep(model, batch)
        assert not step.train_txt and not step.train_pano
        step.run_eager()
        torch.cuda.synchronize()
        eng = model._engine
        for k, shape, off in eng.table:
            sl = eng.grads[off:off + int(np.prod(shape))]
            if po.is_frozen(cfg, k):
                assert float(sl.abs().max()) == 0.0, k
            elif tol is not None:
                err = float((sl.view(shape).float().cpu() - grads[k]).abs().max())
                assert err <= tol + 2e-3 * float(grads[k].abs().max()), (k, err)
            else:
                r, nr = grads[k], float(grads[k].norm())
                # B = 3: the small-batch tier of golden_util.bf16_bounds (the reference's own autocast gap x 2, at most 18 %), not the
                # 12 % of the B >= 8 shapes -- round 6 call 4 saw gmap_pos_embeddings.0.bias at 13.3 % after a rebuild that moved one ulp
                assert float((sl.view(shape).float().cpu() - r).norm()) <= gu.full_bounds(3)["rel"] * nr + 3e-4, k
        assert abs(step.loss.item() - outs["loss"].item()) < (2e-4 if tol else 5e-2)
        step.close()


@pytest.mark.parametrize("name", VARIANTS)
def test_variant_autograd_boundary_matches_reference_golden(name):
    """The same fixtures through forward_txt / forward_panorama / forward_navigation + torch autograd (ss_trainer_ETP.py:801-892):
    with fix_pano_embedding the reference still differentiates through the panorama branch when the features require a gradient
    (d rgb_fts is in the fixture), and leaves the frozen parameters without a .grad."""
    z, cfg, batch = load_case(name)
    P = po.init_params(cfg, seed=0)
    model = build_model(cfg, P, torch.float32)
    b = {k: v.cuda() for k, v in batch.items()}
    rgb = b["rgb_fts"].clone().requires_grad_(True)
    model.zero_grad()
    txt = model.forward_txt(b["txt_ids"], b["txt_masks"])
    assert txt.requires_grad == (not cfg.fix_lang_embedding)                      # LanguageEncoder.forward :431-432
    pano, pmask = model.forward_panorama(rgb, b["dep_fts"], b["loc_fts"], b["nav_types"], b["view_lens"])
    G = b["gmap_step_ids"].shape[1]
    m = pmask.to(pano.dtype)
    avg = (pano * m[..., None]).sum(1) / m.sum(1, keepdim=True)
    idx = torch.arange(G - 2, device="cuda")[None, :] % b["view_lens"][:, None]
    views = torch.gather(pano, 1, idx[..., None].expand(-1, -1, pano.shape[-1]))
    gimg = torch.cat([torch.zeros_like(avg[:, None]), avg[:, None], views], 1)
    outs = model.forward_navigation(txt, b["txt_masks"], None, b["gmap_step_ids"], gimg, b["gmap_pos_fts"], b["gmap_masks"],
                                    b["gmap_visited_masks"], b["gmap_pair_dists"] if cfg.graph_sprels else None)
    loss = F.cross_entropy(outs["global_logits"], b["labels"], reduction="sum", ignore_index=-100) / b["txt_ids"].shape[0]
    loss.backward()
    torch.cuda.synchronize()
    compare_outputs(z, {"txt_embeds": txt, "pano_embeds": pano, "gmap_embeds": outs["gmap_embeds"],
                        "global_logits": outs["global_logits"], "loss": loss}, atol=2e-4)
    grads = {}
    for k, p in model.named_parameters():
        if p.requires_grad:
            grads[k] = p.grad.detach().float().cpu()
        else:
            assert p.grad is None, k
            grads[k] = torch.zeros(p.shape)
    grads["__input__.rgb_fts"] = rgb.grad.float().cpu()
    compare_grads(z, grads, atol=2e-4, rel=2e-3, rel_sample=2e-3)


@pytest.mark.parametrize("name", ["fix_lang_small", "fix_pano_small"])
def test_fused_adamw_leaves_frozen_parameters_alone(name):
    """FusedAdamW on a model with frozen parameters (VERDICT r4 missing #4: it used to raise): the trainable parameters follow the
    oracle's AdamW on the device gradients, the frozen ones -- and their moments and bf16 shadow -- stay bit-identical even when
    their gradient slots hold garbage, and the clipping norm leaves those slots out (clip_grad_norm_ only sees parameters with
    a .grad)."""
    from etpnav_amd.optim import FusedAdamW
    z, cfg, batch = load_case(name)
    P = po.init_params(cfg, seed=0)
    model = build_model(cfg, P, torch.bfloat16)
    eng = model._engine
    opt = FusedAdamW(model, lr=1e-3, hf_style=True, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01, max_grad_norm=5.0,
                     no_decay=FusedAdamW.reference_no_decay, check_finite=True)
    assert opt.frozen_names and all(po.is_frozen(cfg, k) for k in opt.frozen_names)
    step = PlannerStep(model, batch)
    step.run_eager(); torch.cuda.synchronize()
    frozen_mask = torch.zeros(eng.total, dtype=torch.bool)
    wd = torch.full((eng.total,), 0.01)
    for nm, shape, off in eng.table:
        n = int(np.prod(shape))
        if po.is_frozen(cfg, nm):
            frozen_mask[off:off + (n + 63) // 64 * 64] = True
        if FusedAdamW.reference_no_decay(nm):
            wd[off:off + n] = 0.0
    eng.grads[frozen_mask.cuda()] = 7.0                     # garbage where the reference has no .grad at all
    g = eng.grads.detach().cpu().clone()
    g[frozen_mask] = 0.0
    p_ref = eng.params.detach().cpu().clone()
    p0, sh0 = eng.params.detach().clone(), eng.shadow.detach().clone()
    m_ref, v_ref = torch.zeros_like(p_ref), torch.zeros_like(p_ref)
    norm = float(opt.grad_norm().item())
    assert abs(norm - float(g.double().norm())) <= 1e-4 * float(g.double().norm())
    bad = opt.step()
    torch.cuda.synchronize()
    assert bad.item() == 0
    oo.adamw_step(p_ref, g, m_ref, v_ref, 1, 1e-3, 0.9, 0.98, 1e-6, wd, True, True, 1.0, 5.0)
    live = ~frozen_mask
    got = eng.params.detach().cpu()
    assert bool(((got[live] - p_ref[live]).abs() <= 2e-6 + 1e-5 * p_ref[live].abs()).all())
    fm = frozen_mask.cuda()
    assert torch.equal(eng.params[fm], p0[fm])
    assert float(opt.exp_avg[fm].abs().max()) == 0.0 and float(opt.exp_avg_sq[fm].abs().max()) == 0.0
    fsh = fm[:eng.n_matrix]
    assert torch.equal(eng.shadow[fsh], sh0[fsh])
    assert torch.equal(eng.shadow[~fsh].cpu(), eng.params[:eng.n_matrix][~fsh].to(torch.bfloat16).cpu())
    assert float(eng.grads.abs().max()) == 0.0             # zeroed everywhere, frozen slots included
    step.close()


# ---- north_star's fp32 tolerance at the headline shape ------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_benchmarked_shape_b32_fp32_matches_oracle(mode):
    """configs[1] exactly as `bench.py --dtype fp32` runs it (B = 32, L = 80, V = 36 x 768, G = 16): loss, logits, embeddings and
    every parameter gradient, full tensors, within 1e-3 absolute of the CPU oracle (BASELINE.json north_star: 'logits/gradients
    matching the reference PyTorch path within 1e-3 fp32'); the oracle is pinned to the reference at 2e-5
    (tests/test_oracle_golden.py).  Train mode: every dropout site on, the oracle applies the same masks."""
    cfg = po.PlannerConfig.r2r(image_feat_size=768)
    P = po.init_params(cfg, seed=0)
    batch = po.make_batch(cfg, B=32, L=80, V=36, G=16, seed=1234, ragged=False)
    rates = (0.1, 0.1, 0.1, 0.4)
    drop = po.DropSpec(*rates, seed=(5 << 32) | 1) if mode == "train" else None
    outs, grads = po.step_with_grads(P, cfg, batch, drop=drop)
    model = build_model(cfg, P, torch.float32)
    step = PlannerStep(model, batch, dropout=rates if mode == "train" else None, drop_seed=5)
    step.run_eager()
    got = step_outputs(step)
    mine = {k: p.grad.detach().float().cpu() for k, p in model.named_parameters()}
    assert len(mine) == 307
    from tests.golden_util import compare_full_fp32
    worst, wg, wr = compare_full_fp32(got, mine, outs, grads, tol=1e-3)
    print(f"fp32 B=32 {mode}: outputs {worst}; worst gradient abs err {wg}, worst per-tensor relative L2 {wr}")
    step.close()


# ---- determinism / race screen as a test ----------------------------------------------------------------------
def _screen(step, model, runs, rel):
    prm = dict(model.named_parameters())
    outs = ("txt", "pano", "gemb", "logits", "d_txt", "d_gimg", "d_pano")
    snaps = []
    for _ in range(runs):
        step.step_no = 0                                    # same dropout masks every repetition (the counter is part of the seed)
        step.run_eager(); torch.cuda.synchronize()
        snap = {n: p.grad.detach().clone() for n, p in prm.items() if p.numel() <= (1 << 22)}
        snap.update({"[sum] " + n: p.grad.detach().double().sum().reshape(1) for n, p in prm.items() if p.numel() > (1 << 22)})
        snap.update({"[out] " + n: getattr(step, n).detach().clone().float().nan_to_num(0.0, 0.0, 0.0) for n in outs
                     if isinstance(getattr(step, n, None), torch.Tensor)})
        snaps.append(snap)
    bad = []
    for n in snaps[0]:
        stack = torch.stack([r[n].double().reshape(-1) for r in snaps])
        med = stack.median(0).values
        scale = max(float(med.abs().max()), 1e-6)
        adev = float((stack - med).abs().max())
        if adev / scale > rel and adev > _ABS_FLOOR.get(n, 0.0):
            bad.append((adev / scale, n))
    return sorted(bad, reverse=True)


# cancelling sums whose value is ~1e-3 of their summands (or exactly zero in exact arithmetic: d(net.4.bias) = sum of the CE
# gradient over the nodes of every episode = sum(softmax) - 1 = 0): fp32 atomic order shows at 1e-4 .. 1e-2 of the tiny result
# (first GPU run of this test: net.4.bias 3.2e-2 of a 1e-6 floor, [sum] word_embeddings 7e-4)
_NEAR_ZERO_SUMS = ("global_encoder.sprel_linear.", "[sum] ")
# d(global_sap_head.net.4.bias) = sum over all nodes of the CE gradient = B^-1 * sum_b (sum_g softmax - 1) = 0 in exact arithmetic:
# what is left is the fp32 atomic-order noise of ~500 summands of magnitude 1/B (observed 3e-8 .. 9e-8 absolute)
# (d net.2.bias = w4 * sum of the CE gradient over the nodes: the same exact zero in eval mode, i.e. without the head's dropout mask)
_ABS_FLOOR = {"global_sap_head.net.4.bias": 1e-6, "global_sap_head.net.2.bias": 1e-6}


@pytest.mark.parametrize("workload", ["c2_train", "c5", "sap", "c4", "c2_fp32"])
def test_three_stream_step_reproduces_every_gradient(workload):
    """tools/determinism_screen.py as a gate: 10 repetitions of the free-running three-stream step (weight gradients and the
    panorama branch on side streams) on identical inputs; every gradient and output must stay within 2e-5 of its abs-max of the
    per-element median.  This is the screen that exposed the sporadic d(gmap_pos_embeddings.0.weight) corruption of rounds 3-4
    (profiles/r04_gmap_pos_race.txt); it now also runs in train mode and on the SAP unit, which the tool never did.
    Round 6 (VERDICT r5 #1c): + config 4 (RxR rows: the text backward's fork order differs there) and the fp32 parity
    mode of config 2; the MLM step has its own case below."""
    import bench
    from etpnav_amd.planner import default_config
    from etpnav_amd.synthetic import make_batch, make_sap_batch
    key = workload.split("_")[0]
    w = dict(bench.WORKLOADS[key])
    cfg = default_config(w["task"], image_feat_size=w["image_feat_size"])
    model = GlocalTextPathNavCMT(cfg, dtype=torch.float32 if workload.endswith("fp32") else torch.bfloat16, device="cuda")
    model.init_weights(seed=0)
    if key == "sap":
        batch = make_sap_batch(cfg.vocab_size, cfg.image_feat_size, cfg.depth_feat_size, 8, w["L"], w["T"], w["V"], seed=1234)
    else:
        batch = make_batch(cfg.vocab_size, cfg.image_feat_size, cfg.depth_feat_size, w["B"], w["L"], w["V"], w["G"], seed=1234)
    step = PlannerStep(model, batch, overlap=True, dropout=(0.1, 0.1, 0.1, 0.4) if workload.endswith("train") else None, drop_seed=9)
    bad = _screen(step, model, runs=6 if workload in ("c4", "c2_fp32") else 10, rel=2e-5)
    hard = [(d, n) for d, n in bad if not (n.startswith(_NEAR_ZERO_SUMS) and d < 2e-3)]
    print(workload, "tensors above 2e-5:", bad[:8])
    assert not hard, hard[:8]
    step.close()


def test_three_stream_mlm_step_reproduces_every_gradient():
    """The same screen on the pre-training MLM step (MlmStep: text chain + panorama branch + weight-gradient stream; the language-side
    x-layers and the tied decoder's word-embedding gradient), bf16, dropout on."""
    from oracle.make_golden_pretrain import make_case
    from etpnav_amd.pretrain import MlmStep
    cfg, P, batch = make_case()
    model = build_model(cfg, P, torch.bfloat16)
    step = MlmStep(model, batch, dropout=(0.1, 0.1, 0.1, 0.0), drop_seed=9)
    bad = _screen(step, model, runs=10, rel=2e-5)
    hard = [(d, n) for d, n in bad if not (n.startswith(_NEAR_ZERO_SUMS) and d < 2e-3)]
    print("mlm tensors above 2e-5:", bad[:8])
    assert not hard, hard[:8]
