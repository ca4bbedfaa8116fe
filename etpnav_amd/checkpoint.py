"""Checkpoint formats of the reference, for the arena-backed planner (storage half: checkpoints).

* pre-training  ``model_step_N.pt``   flat ``state_dict`` of GlocalTextPathCMTPreTraining (utils/save.py:23-46): keys
                 ``bert.<planner name>``, ``mlm_head.*``, ``global_sap_head.*`` (+ the tied
                 ``mlm_head.predictions.decoder.weight``), optimizer in ``train_state_N.pt``;
* fine-tuning   ``ckpt.iterN.pth``    ``{"state_dict": policy.state_dict(), "config", "optim_state", "iteration"}``
                 (ss_trainer_ETP.py:74-83), planner keys under ``net.vln_bert.`` (ILPolicy.net -> ETP.vln_bert).

Model weights go through the module's own ``state_dict()`` / ``load_state_dict()`` (names equal the reference's).  The
fused optimizer keeps its moments as two flat arenas; ``adamw_state_to_torch`` / ``adamw_state_from_torch`` convert to and
from the per-parameter layout of ``torch.optim.AdamW.state_dict()`` (same parameter order as ``model.parameters()``), so
``optim_state`` entries written by either side load on the other.
"""
from __future__ import annotations

from typing import Dict

import torch

HEAD_PREFIXES = ("mlm_head.", "global_sap_head.")
FINETUNE_PREFIX = "net.vln_bert."


def pretrain_state_dict(model) -> Dict[str, torch.Tensor]:
    """What utils/save.py:ModelSaver.save would write for the reference pre-training model holding these weights."""
    out = {}
    for k, v in model.state_dict().items():
        out[k if k.startswith(HEAD_PREFIXES) else "bert." + k] = v.detach().cpu().clone()
    if any(k.startswith("mlm_head.") for k in out):
        out["mlm_head.predictions.decoder.weight"] = out["bert.embeddings.word_embeddings.weight"]     # tied (pretrain_cmt.py:79-82)
    return out


def load_pretrain_state_dict(model, state: Dict[str, torch.Tensor], strict: bool = False):
    """model_step_N.pt -> planner (fine-tuning or pre-training variant): strips 'module.' / 'bert.' as vlnbert_init.py:22-30,
    drops the tied decoder copy and, unless strict, whatever the variant does not have (heads of other tasks)."""
    from .vlnbert_init import remap_checkpoint_keys
    state = {k: v for k, v in remap_checkpoint_keys(state).items() if k != "mlm_head.predictions.decoder.weight"}
    own = set(model.state_dict().keys())
    if strict:
        return model.load_state_dict(state, strict=True)
    return model.load_state_dict({k: v for k, v in state.items() if k in own}, strict=False)


def finetune_checkpoint(model, optimizer=None, iteration: int = 0, config=None, extra_policy_state=None) -> dict:
    """The dict ss_trainer_ETP.py:74-83 saves; `extra_policy_state` = the policy's non-planner entries (rgb/depth encoders...)."""
    sd = dict(extra_policy_state or {})
    sd.update({FINETUNE_PREFIX + k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    ck = {"state_dict": sd, "config": config, "iteration": iteration}
    if optimizer is not None:
        ck["optim_state"] = adamw_state_to_torch(optimizer)
    return ck


def load_finetune_checkpoint(model, ckpt: dict, optimizer=None):
    sd = {k[len(FINETUNE_PREFIX):]: v for k, v in ckpt["state_dict"].items() if k.startswith(FINETUNE_PREFIX)}
    r = model.load_state_dict(sd, strict=False)         # ss_trainer_ETP.py:229 loads with strict=False as well
    if optimizer is not None and ckpt.get("optim_state") is not None:
        adamw_state_from_torch(optimizer, ckpt["optim_state"])
    return r


def _param_slots(model):
    """[(name, offset, numel, shape)] in ``model.parameters()`` order -- the order torch.optim numbers its parameters in
    (module-tree traversal: query.weight, query.bias, key.weight, ...), which differs from the arena's table order
    (q.w, k.w, v.w, q.b, ...)."""
    by_id = {id(p): (off, n, shape) for p, off, n, shape in model._views}
    return [(name,) + by_id[id(p)] for name, p in model.named_parameters()]


def adamw_state_to_torch(opt) -> dict:
    """FusedAdamW -> torch.optim.AdamW.state_dict() layout (one param group; parameter index = position in
    model.parameters(), exactly what ``torch.optim.AdamW(model.parameters())`` would save)."""
    slots = _param_slots(opt.model)
    state = {}
    for i, (_, off, n, shape) in enumerate(slots):
        state[i] = {"step": torch.tensor(float(opt.step_count)),
                    "exp_avg": opt.exp_avg[off:off + n].view(shape).detach().cpu().clone(),
                    "exp_avg_sq": opt.exp_avg_sq[off:off + n].view(shape).detach().cpu().clone()}
    group = {"lr": opt.lr, "betas": tuple(opt.betas), "eps": opt.eps, "weight_decay": opt.weight_decay, "amsgrad": False,
             "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
             "params": list(range(len(slots)))}
    return {"state": state, "param_groups": [group]}


def adamw_state_from_torch(opt, sd: dict, param_names=None, prefix: str = ""):
    """torch.optim.AdamW.state_dict() -> FusedAdamW.

    Without `param_names` the state must cover exactly ``model.parameters()`` in order (an optimizer built over the planner
    alone).  The reference optimizer covers ``policy.parameters()`` (ss_trainer_ETP.py:213: planner + perception encoders):
    pass `param_names` = the names of ``policy.named_parameters()`` in order and `prefix` = "net.vln_bert." to select the
    planner's entries BY NAME.  Per-group weight decay is validated against the optimizer's decay mask."""
    slots = _param_slots(opt.model)
    order = [i for g in sd["param_groups"] for i in g["params"]]
    wd_of = {i: float(g.get("weight_decay", 0.0)) for g in sd["param_groups"] for i in g["params"]}
    if param_names is None:
        if len(order) != len(slots):
            raise ValueError(f"optimizer state covers {len(order)} parameters, the planner has {len(slots)}; pass "
                             f"param_names/prefix to select the planner's entries of a larger optimizer by name")
        pairs = list(zip(order, slots))
    else:
        if len(param_names) != len(order):
            raise ValueError(f"param_names lists {len(param_names)} parameters, the optimizer state {len(order)}")
        by_name = {prefix + name: (name, off, n, shape) for name, off, n, shape in slots}
        pairs = [(idx, by_name[pn]) for idx, pn in zip(order, param_names) if pn in by_name]
        if len(pairs) != len(slots):
            missing = sorted(set(by_name) - set(param_names))
            raise ValueError(f"optimizer state lacks {len(missing)} planner parameters, e.g. {missing[:3]}")
    steps, decays = set(), {}
    with torch.no_grad():
        for idx, (name, off, n, shape) in pairs:
            st = sd["state"].get(idx)
            decays[name] = wd_of.get(idx, 0.0)
            if st is None:                                       # parameter never stepped: zero moments
                opt.exp_avg[off:off + n].zero_(); opt.exp_avg_sq[off:off + n].zero_()
                continue
            if tuple(st["exp_avg"].shape) != tuple(shape):
                raise ValueError(f"{name}: moment shape {tuple(st['exp_avg'].shape)} != {tuple(shape)}")
            opt.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1).to(opt.exp_avg.device))
            opt.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1).to(opt.exp_avg_sq.device))
            steps.add(int(float(st["step"])))
    if len(steps) > 1:
        raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): one fused step count cannot represent them")
    opt.step_count = steps.pop() if steps else 0
    g0 = sd["param_groups"][0]
    opt.lr, opt.betas, opt.eps = float(g0["lr"]), tuple(g0["betas"]), float(g0["eps"])
    # weight decay: one non-zero value for the decayed parameters, 0 for the no-decay set; the fused optimizer's decay mask
    # is REBUILT from the checkpoint's grouping (optim/misc.py:12-37 style decay / no-decay groups)
    nz = sorted({w for w in decays.values() if w != 0.0})
    if len(nz) > 1:
        raise ValueError(f"parameter groups use several non-zero weight decays {nz}: not representable")
    if nz:
        opt.weight_decay = nz[0]
        opt.set_no_decay_names([n for n, w in decays.items() if w == 0.0])
    else:
        opt.weight_decay = 0.0
        opt.set_no_decay_names([])
