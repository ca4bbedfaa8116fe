"""Fused AdamW for the MI355X planner: one HIP kernel over the model's flat fp32 arenas.

Drop-in for the two optimizers the reference uses on the planner:

  * fine-tuning  ``torch.optim.AdamW(self.policy.parameters(), lr=...)``            ss_trainer_ETP.py:213
                 + ``GradScaler`` unscale / skip-on-inf / ``zero_grad()``            :463,499-506
  * pre-training ``AdamW`` of pretrain_src/pretrain_src/optim/adamw.py:53-112 with the no-decay grouping of
                 optim/misc.py:12-22 and ``clip_grad_norm_``                          (hf_style=True)

``step()`` = unscale -> (optional) global-norm clip -> Adam moments -> decoupled weight decay -> parameter update ->
bf16 GEMM-weight shadow -> gradient zeroing, all in a single pass over HBM (etp_adamw_step), preceded by one
reduction pass (etp_grad_sqnorm) only when clipping or the non-finite check is requested.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes
from typing import Callable, Iterable, Optional

import torch

from . import _lib
from ._lib import check, ptr

NO_DECAY_SUBSTRINGS = ("bias", "LayerNorm.bias", "LayerNorm.weight")      # optim/misc.py:13


class FusedAdamW:
    def __init__(self, model, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 hf_style: bool = False, correct_bias: bool = True, max_grad_norm: float = 0.0,
                 no_decay: Optional[Callable[[str], bool]] = None, check_finite: bool = False):
        """Defaults = torch.optim.AdamW's (the fine-tuning optimizer).  For the pre-training optimizer use
        ``hf_style=True, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01, max_grad_norm=5.0,
        no_decay=FusedAdamW.reference_no_decay``.  `no_decay(name) -> bool` selects parameters WITHOUT weight decay."""
        self.model = model
        eng = self.eng = model._engine
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), tuple(betas), float(eps), float(weight_decay)
        self.hf_style, self.correct_bias = bool(hf_style), bool(correct_bias)
        self.max_grad_norm, self.check_finite = float(max_grad_norm), bool(check_finite)
        self._step_host = 0
        dev = eng.device
        # with the non-finite check the number of APPLIED steps is only known on the device (a skipped step must not advance
        # it: GradScaler.step() does not call optimizer.step() on overflow)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev) if self.check_finite else None
        self.exp_avg = torch.zeros_like(eng.params)
        self.exp_avg_sq = torch.zeros_like(eng.params)
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.nonfinite = torch.zeros(1, dtype=torch.int32, device=dev)
        self.decay_mask = None
        self.no_decay_names = []        # parameters excluded from weight decay (checkpoint validation)
        # frozen parameters (requires_grad = False: fix_lang_embedding / fix_pano_embedding, vilmodel_cmt.py:675-682; the
        # reference hands torch.optim.AdamW parameters without a .grad, which it skips): bit 1 of the per-block mask byte --
        # the kernel leaves p / m / v / shadow of those blocks alone, the norm / non-finite scan leaves them out
        self.frozen_names = [name for name, p in model.named_parameters() if not p.requires_grad]
        self._build_mask()
        if no_decay is not None:
            self.set_no_decay_names([name for name, _, _ in eng.table if no_decay(name)])

    def set_no_decay_names(self, names):
        """Exclude exactly these parameters from weight decay (one mask byte per 64 elements; every parameter starts on a
        64-element boundary of the arena).  An empty list = every parameter decays (no mask)."""
        names = set(names)
        self.no_decay_names = [name for name, _, _ in self.eng.table if name in names]
        self._build_mask()

    def _build_mask(self):
        """One byte per 64 arena elements (include/etpnav_hip.h, etp_adamw_step): bit 0 = weight decay applies, bit 1 = frozen.
        None when every parameter decays and none is frozen."""
        eng = self.eng
        nd, fz = set(self.no_decay_names), set(self.frozen_names)
        if not nd and not fz:
            self.decay_mask = None
            return
        mask = torch.ones((eng.total + 63) // 64, dtype=torch.uint8)
        for name, shape, off in eng.table:
            if name in nd or name in fz:
                n = 1
                for s in shape:
                    n *= s
                mask[off // 64:(off + n + 63) // 64] = (0 if name in nd else 1) | (2 if name in fz else 0)
        self.decay_mask = mask.to(eng.device)

    def _sqnorm(self, s):
        eng = self.eng
        self.sumsq.zero_(); self.nonfinite.zero_()
        if self.frozen_names:
            check(eng.L.etp_grad_sqnorm_masked(ptr(eng.grads), eng.total, ptr(self.decay_mask), ptr(self.sumsq), ptr(self.nonfinite), s),
                  "grad_sqnorm_masked")
        else:
            check(eng.L.etp_grad_sqnorm(ptr(eng.grads), eng.total, ptr(self.sumsq), ptr(self.nonfinite), s), "grad_sqnorm")

    @property
    def step_count(self) -> int:
        """Number of applied updates (reads the device counter -- one host sync -- when check_finite is on)."""
        if self.step_dev is not None:
            return int(self.step_dev.item())
        return self._step_host

    @step_count.setter
    def step_count(self, v: int):
        self._step_host = int(v)
        if self.step_dev is not None:
            self.step_dev.fill_(int(v))

    @staticmethod
    def reference_no_decay(name: str) -> bool:
        """optim/misc.py:12-22: substring match on the parameter name (note: 'layer_norm'/'norm1' do NOT match)."""
        return any(nd in name for nd in NO_DECAY_SUBSTRINGS)

    def zero_grad(self, set_to_none: bool = False):
        self.model.zero_grad()

    def grad_norm(self) -> torch.Tensor:
        """Global L2 norm of the (still scaled) gradients, as a device scalar."""
        self._sqnorm(self.eng.stream())
        return self.sumsq.sqrt()

    def step(self, grad_scale: float = 1.0, zero_grads: bool = True):
        """grad_scale multiplies the raw gradients (1/loss_scale for a GradScaler, 1/world_size after a summed
        all-reduce).  Returns the device int32 tensor counting non-finite gradient values (0 -> the step was applied)
        when check_finite, else None; reading it is the only host synchronisation and is left to the caller."""
        eng = self.eng
        eng.require_gpu()
        s = eng.stream()
        need_scan = self.max_grad_norm > 0.0 or self.check_finite
        if need_scan:
            self._sqnorm(s)
        self._step_host += 1
        c = _lib.AdamwCfg(lr=self.lr, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, weight_decay=self.weight_decay,
                          step=self._step_host, hf_style=int(self.hf_style), correct_bias=int(self.correct_bias),
                          grad_scale=float(grad_scale), max_norm=self.max_grad_norm)
        args = (ptr(eng.params), ptr(eng.grads), ptr(self.exp_avg), ptr(self.exp_avg_sq), ptr(eng.shadow),
                eng.n_matrix if eng.shadow is not None else 0, ptr(self.decay_mask), eng.total, ctypes.byref(c),
                ptr(self.sumsq) if self.max_grad_norm > 0.0 else None, ptr(self.nonfinite) if self.check_finite else None,
                int(zero_grads))
        if self.step_dev is not None:
            check(eng.L.etp_adamw_step_counted(*args, ptr(self.step_dev), s), "adamw_step_counted")
        else:
            check(eng.L.etp_adamw_step(*args, s), "adamw_step")
        eng.mark_shadow_current()      # the kernel rewrote the masters AND their bf16 shadow
        return self.nonfinite if self.check_finite else None

    # ---- torch.optim-like state handling (checkpoint format of the trainer: optimizer.state_dict()) ----
    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "hyper": dict(lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=self.weight_decay,
                              hf_style=self.hf_style, correct_bias=self.correct_bias, max_grad_norm=self.max_grad_norm,
                              check_finite=self.check_finite),
                "no_decay": list(self.no_decay_names)}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        for k, v in sd.get("hyper", {}).items():
            setattr(self, k, tuple(v) if k == "betas" else v)
        if "no_decay" in sd:
            self.set_no_decay_names(sd["no_decay"])


# ---- learning-rate schedule of the pre-training loop (pretrain_src/pretrain_src/optim/sched.py:17-30) -------------------
def warmup_linear(step: int, warmup_step: int, tot_step: int) -> float:
    """BERT schedule (sched.py:17-21): linear warm-up to 1 over `warmup_step`, then linear decay to 0 at `tot_step`."""
    if step < warmup_step:
        return step / warmup_step
    return max(0, (tot_step - step) / (tot_step - warmup_step))


def get_lr_sched(global_step: int, learning_rate: float, warmup_steps: int, num_train_steps: int) -> float:
    """sched.py:24-30 with the option fields spelled out: never returns a non-positive rate (floor 1e-8)."""
    lr = learning_rate * warmup_linear(global_step, warmup_steps, num_train_steps)
    return 1e-8 if lr <= 0 else lr


class WarmupLinearLR:
    """Drives ``FusedAdamW.lr`` the way train_r2r.py sets ``param_group['lr']`` before every optimizer step
    (pretrain_src/pretrain_src/train_r2r.py: lr_this_step = get_lr_sched(global_step, opts)).  The rate is a kernel argument
    of the fused step (no device state), so changing it costs nothing."""

    def __init__(self, optimizer, learning_rate: float, warmup_steps: int, num_train_steps: int):
        self.opt, self.base, self.warmup, self.total = optimizer, float(learning_rate), int(warmup_steps), int(num_train_steps)

    def step(self, global_step: int) -> float:
        lr = get_lr_sched(global_step, self.base, self.warmup, self.total)
        self.opt.lr = lr
        return lr
