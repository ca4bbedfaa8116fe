"""Checkpoint formats (utils/save.py:23-46, ss_trainer_ETP.py:74-83,229-236): key layouts and optimizer-state exchange
with torch.optim.AdamW, all on the CPU (arenas are plain tensors)."""
import io

import pytest
import torch

from oracle import planner_oracle as po
from etpnav_amd import checkpoint as ck
from etpnav_amd.optim import FusedAdamW
from etpnav_amd.planner import GlocalTextPathNavCMT


def small(pretrain=False):
    cfg = po.PlannerConfig.r2r(vocab_size=512, num_l_layers=1, num_pano_layers=1, num_x_layers=1, use_lang2visn_attn=pretrain)
    m = GlocalTextPathNavCMT(cfg.to_dict(), dtype=torch.float32, device="cpu")
    m.load_state_dict(po.init_params(cfg, seed=1), strict=True)
    return cfg, m


def test_pretrain_model_step_roundtrip_and_key_layout():
    cfg, m = small(pretrain=True)
    sd = ck.pretrain_state_dict(m)
    assert "bert.embeddings.word_embeddings.weight" in sd and "mlm_head.predictions.bias" in sd
    assert "global_sap_head.net.0.weight" in sd and "bert.global_sap_head.net.0.weight" not in sd
    assert sd["mlm_head.predictions.decoder.weight"] is sd["bert.embeddings.word_embeddings.weight"]      # tied
    assert "bert.global_encoder.encoder.x_layers.0.lang_self_att.self.query.weight" in sd
    buf = io.BytesIO(); torch.save(sd, buf); buf.seek(0)
    loaded = torch.load(buf)
    # into the pre-training variant: everything; into the fine-tuning variant: the shared part (vlnbert_init.py:61-64)
    _, m2 = small(pretrain=True)
    with torch.no_grad():
        m2.flat_params.zero_()
    ck.load_pretrain_state_dict(m2, {("module." + k): v for k, v in loaded.items()}, strict=True)
    assert torch.equal(m2.flat_params, m.flat_params)
    _, m3 = small(pretrain=False)
    r = ck.load_pretrain_state_dict(m3, loaded)
    assert not r.missing_keys
    for k, v in m3.state_dict().items():
        assert torch.equal(v, m.state_dict()[k])


def _stepped_torch_adamw(m, extra=()):
    """A torch AdamW over (extra +) m.parameters() with the reference's decay / no-decay grouping (optim/misc.py:12-22)
    that has taken two real steps on random gradients.  Returns (optimizer, [names in optimizer order])."""
    named = [(f"net.other.{i}", p) for i, p in enumerate(extra)] + [("net.vln_bert." + n, p) for n, p in m.named_parameters()]
    nd = lambda n: any(s in n for s in ("bias", "LayerNorm.bias", "LayerNorm.weight"))
    groups = [{"params": [p for n, p in named if not nd(n)], "weight_decay": 0.01},
              {"params": [p for n, p in named if nd(n)], "weight_decay": 0.0}]
    names = [n for n, _ in named if not nd(n)] + [n for n, _ in named if nd(n)]
    opt = torch.optim.AdamW(groups, lr=3e-4, betas=(0.9, 0.98), eps=1e-6)
    g = torch.Generator().manual_seed(5)
    for _ in range(2):
        for _, p in named:
            p.grad = torch.randn(p.shape, generator=g) * 1e-2
        opt.step()
    return opt, names


def test_optimizer_state_maps_every_parameter_by_name_in_both_directions():
    """torch numbers optimizer state in model.parameters() (module-tree) order, the arena table is ordered
    q.w,k.w,v.w,q.b,...; every parameter's moments must land on the right slice, also when the optimizer covers more than
    the planner (policy.parameters(), ss_trainer_ETP.py:213) and with decay / no-decay groups."""
    cfg, m = small()
    extra = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7))]
    topt, names = _stepped_torch_adamw(m, extra)
    sd = topt.state_dict()
    by_param = {n: topt.state[p] for n, p in [(f"net.other.{i}", p) for i, p in enumerate(extra)] +
                [("net.vln_bert." + n, p) for n, p in m.named_parameters()]}
    opt = FusedAdamW(m)
    with pytest.raises(ValueError):
        ck.adamw_state_from_torch(opt, sd)                       # covers more parameters than the planner: needs the names
    ck.adamw_state_from_torch(opt, sd, param_names=names, prefix="net.vln_bert.")
    slots = {name: (off, n, shape) for name, off, n, shape in ck._param_slots(m)}
    assert [n for n, *_ in ck._param_slots(m)] == [n for n, _ in m.named_parameters()]
    for name, (off, n, shape) in slots.items():
        st = by_param["net.vln_bert." + name]
        assert torch.equal(opt.exp_avg[off:off + n].view(shape), st["exp_avg"]), name
        assert torch.equal(opt.exp_avg_sq[off:off + n].view(shape), st["exp_avg_sq"]), name
    assert opt.step_count == 2 and opt.lr == pytest.approx(3e-4) and opt.betas == (0.9, 0.98) and opt.eps == pytest.approx(1e-6)
    assert opt.weight_decay == pytest.approx(0.01)
    assert sorted(opt.no_decay_names) == sorted(n for n, _ in m.named_parameters() if FusedAdamW.reference_no_decay(n))
    assert opt.decay_mask is not None and int(opt.decay_mask.sum()) < opt.decay_mask.numel()
    # export: what torch.optim.AdamW(model.parameters()) would have saved -- load it into one and compare BY PARAMETER
    out = ck.adamw_state_to_torch(opt)
    plist = list(m.parameters())
    t2 = torch.optim.AdamW([torch.nn.Parameter(torch.zeros_like(p)) for p in plist], lr=1.0)
    t2.load_state_dict(out)
    for (name, _), q in zip(m.named_parameters(), t2.param_groups[0]["params"]):
        st = by_param["net.vln_bert." + name]
        assert torch.equal(t2.state[q]["exp_avg"], st["exp_avg"]) and t2.state[q]["exp_avg"].shape == st["exp_avg"].shape, name
        assert torch.equal(t2.state[q]["exp_avg_sq"], st["exp_avg_sq"]), name
    # FusedAdamW's own state dict carries the no-decay set and check_finite
    opt3 = FusedAdamW(small()[1])
    opt3.load_state_dict(opt.state_dict())
    assert sorted(opt3.no_decay_names) == sorted(opt.no_decay_names) and opt3.weight_decay == pytest.approx(0.01)


def test_finetune_checkpoint_and_optimizer_state_exchange_with_torch_adamw():
    cfg, m = small()
    opt = FusedAdamW(m, lr=2e-4)
    g = torch.Generator().manual_seed(0)
    opt.exp_avg.copy_(torch.randn(opt.exp_avg.shape, generator=g) * 1e-2)
    opt.exp_avg_sq.copy_(torch.rand(opt.exp_avg_sq.shape, generator=g) * 1e-4)
    opt.step_count = 7
    ckpt = ck.finetune_checkpoint(m, opt, iteration=1200, extra_policy_state={"net.rgb_encoder.x": torch.ones(1)})
    assert ckpt["iteration"] == 1200 and "net.vln_bert.global_sap_head.net.4.bias" in ckpt["state_dict"]
    # the optimizer state is a valid torch.optim.AdamW state dict for the same parameter list
    ref = torch.optim.AdamW([torch.nn.Parameter(torch.zeros_like(p)) for p in m.parameters()], lr=1e-3)
    ref.load_state_dict(ckpt["optim_state"])
    assert ref.param_groups[0]["lr"] == pytest.approx(2e-4)
    first = ref.state[ref.param_groups[0]["params"][0]]
    _, off, n, shape = ck._param_slots(m)[0]
    assert torch.equal(first["exp_avg"], opt.exp_avg[off:off + n].view(shape)) and float(first["step"]) == 7
    # and back: a torch AdamW state (two param groups, as a decay / no-decay split) into a fresh fused optimizer
    sd = ref.state_dict()
    k = len(sd["param_groups"][0]["params"]) // 2
    g0 = dict(sd["param_groups"][0])
    sd["param_groups"] = [dict(g0, params=g0["params"][:k]), dict(g0, params=g0["params"][k:], weight_decay=0.0)]
    _, m2 = small()
    opt2 = FusedAdamW(m2)
    r = ck.load_finetune_checkpoint(m2, dict(ckpt, optim_state=sd), opt2)
    assert not r.missing_keys
    assert torch.equal(m2.flat_params, m.flat_params)
    assert opt2.step_count == 7 and opt2.lr == pytest.approx(2e-4)
    assert len(opt2.no_decay_names) == len(sd["param_groups"][1]["params"])        # the decay mask follows the checkpoint's groups
    for _, off, n, _ in m._views:       # padding between parameters is not part of any state dict
        assert torch.equal(opt2.exp_avg[off:off + n], opt.exp_avg[off:off + n])
        assert torch.equal(opt2.exp_avg_sq[off:off + n], opt.exp_avg_sq[off:off + n])
    bad = ref.state_dict()
    bad["state"][0]["step"] = torch.tensor(9.0)
    with pytest.raises(ValueError):
        ck.adamw_state_from_torch(opt2, bad)
