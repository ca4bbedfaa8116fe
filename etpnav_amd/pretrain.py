"""The pre-training MLM task as one device-side step, straight through the C ABI.

Mirrors ``GlocalTextPathCMTPreTraining.forward(batch, 'mlm')`` + ``.mean()`` + ``backward()``
(pretrain_src/pretrain_src/model/pretrain_cmt.py:141-163, train_r2r.py:240-250): text encoder, panorama encoder over every
trajectory step, node aggregation, ``forward_lang2visn`` through every x-layer, tied MLM head on the masked tokens, mean
token cross-entropy, gradients of every parameter.  The model must be built with ``use_lang2visn_attn=True`` (the
pre-training config, run_pt/r2r_model_config_dep.json).  The SAP task of the same model is ``PlannerStep`` with
``batch['traj']`` (etpnav_amd/step.py).
"""
from __future__ import annotations

from typing import List, Dict

import torch

from . import _lib
from ._lib import check, ptr
from .graph_inputs import pack_traj_csr
from .planner import GlocalTextPathNavCMT


class MlmStep:
    def __init__(self, model: GlocalTextPathNavCMT, batch: Dict[str, torch.Tensor], dropout=None, drop_seed: int = 0,
                 overlap: bool = True):
        """batch: etpnav_amd.synthetic.make_sap_batch layout + ``txt_labels`` [B,L] (-1 = not masked, else the token id).
        overlap: the three-stream schedule of PlannerStep -- weight gradients on an `aux` stream, the panorama branch
        (forward and backward) on `s2` beside the text branch; False = everything on the caller's stream."""
        eng = self.eng = model._engine
        eng.require_gpu()
        if not eng.cconf.use_lang2visn:
            raise _lib.EtpError("the MLM task needs a model built with use_lang2visn_attn=True (pre-training config)")
        self.model, self.L = model, eng.L
        dev = eng.device
        if dropout == "config":
            c = model.config
            g = lambda k: float(c[k] if isinstance(c, dict) and k in c else getattr(c, k, 0.1))
            dropout = (g("hidden_dropout_prob"), g("attention_probs_dropout_prob"), g("pred_head_dropout_prob"), 0.0)
        self.dropout, self.drop_seed, self.step_no = dropout, int(drop_seed) & 0xFFFFFFFF, 0
        # gradient accumulation (PretrainDriver): later micro-steps keep the arena (zero_grads False) and every micro-step's mean
        # loss is scaled by 1 / accumulation steps (train_r2r.py:250-252)
        self.zero_grads, self.loss_scale = True, 1.0
        B, Lt = batch["txt_ids"].shape
        Bp, V = batch["rgb_fts"].shape[:2]
        G = batch["gmap_step_ids"].shape[1]
        H = eng.cconf.hidden
        self.dims = (B, Lt, Bp, V, G, H)
        mv = lambda x, dt=None: (x.to(dt) if dt is not None else x).contiguous().to(dev)
        self.inp = {
            "txt_ids": mv(batch["txt_ids"], torch.int64), "txt_masks": mv(batch["txt_masks"], torch.bool),
            "rgb": mv(batch["rgb_fts"], torch.float32), "dep": mv(batch["dep_fts"], torch.float32),
            "loc": mv(batch["loc_fts"], torch.float32), "nav": mv(batch["nav_types"], torch.int64),
            "view_lens": mv(batch["view_lens"], torch.int64), "step_ids": mv(batch["gmap_step_ids"], torch.int64),
            "pos": mv(batch["gmap_pos_fts"], torch.float32), "gmask": mv(batch["gmap_masks"], torch.bool),
        }
        tr = batch["traj"]
        fwd, bwd = pack_traj_csr(tr["traj_vp_lens"], tr["traj_vpids"], tr["traj_cand_vpids"], tr["gmap_vpids"], V, G)
        self.csr_f, self.csr_b = tuple(x.to(dev) for x in fwd), tuple(x.to(dev) for x in bwd)
        # masked positions: row gather of the text (and its transpose for the backward)
        sel = (batch["txt_labels"] != -1).reshape(-1)
        rows = torch.nonzero(sel).reshape(-1).to(torch.int32)
        Nm = self.Nm = int(rows.numel())
        if Nm == 0:
            raise ValueError("no masked tokens in the batch")
        i32 = lambda x: torch.as_tensor(x, dtype=torch.int32).to(dev)
        self.sel = (i32(torch.arange(Nm + 1)), rows.to(dev), torch.ones(Nm, dtype=torch.float32, device=dev))
        ptr_t = torch.zeros(B * Lt + 1, dtype=torch.int32)
        ptr_t[1:] = torch.cumsum(sel.to(torch.int32), 0)
        self.selT = (ptr_t.to(dev), i32(torch.arange(Nm)), torch.ones(Nm, dtype=torch.float32, device=dev))
        self.labels = mv(batch["txt_labels"].reshape(-1)[sel], torch.int64)
        e = lambda *s, dt=torch.float32: torch.empty(*s, dtype=dt, device=dev)
        self.txt, self.pano, self.pmask = e(B, Lt, H), e(Bp, V, H), e(Bp, V, dt=torch.bool)
        self.gimg, self.loss = e(B, G, H), torch.zeros(1, dtype=torch.float32, device=dev)
        self.d_txt, self.d_gimg, self.d_pano = e(B, Lt, H), e(B, G, H), e(Bp, V, H)
        h = eng.handle
        self.st_txt = eng.buf(self.L.etp_txt_stash_bytes(h, B, Lt)); self.ws_txt = eng.buf(self.L.etp_txt_ws_bytes(h, B, Lt))
        self.st_pano = eng.buf(self.L.etp_pano_stash_bytes(h, Bp, V)); self.ws_pano = eng.buf(self.L.etp_pano_ws_bytes(h, Bp, V))
        self.st_mlm = eng.buf(self.L.etp_mlm_stash_bytes(h, B, Lt, G, Nm)); self.ws_mlm = eng.buf(self.L.etp_mlm_ws_bytes(h, B, Lt, G, Nm))
        self.aux = self.s2 = None
        if overlap:
            import ctypes
            a, b2 = ctypes.c_void_p(), ctypes.c_void_p()
            check(self.L.etp_stream_create(ctypes.byref(a)), "stream_create")
            check(self.L.etp_stream_create(ctypes.byref(b2)), "stream_create")
            self.aux, self.s2 = a.value, b2.value

    @staticmethod
    def batch_shape_key(batch):
        B, Lt = batch["txt_ids"].shape
        Bp, V = batch["rgb_fts"].shape[:2]
        return (B, Lt, Bp, V, batch["gmap_step_ids"].shape[1], int((batch["txt_labels"] != -1).sum()))

    def shape_key(self):
        B, Lt, Bp, V, G, _ = self.dims
        return (B, Lt, Bp, V, G, self.Nm)

    def close(self):
        self.L.etp_planner_set_aux_stream(self.eng.handle, None)
        for st in (self.aux, self.s2):
            if st is not None:
                self.L.etp_stream_destroy(st)
        self.aux = self.s2 = None

    def run_eager(self, backward: bool = True):
        L, eng, i = self.L, self.eng, self.inp
        h, s = eng.handle, eng.stream()
        s2 = self.s2 if self.s2 is not None else s
        B, Lt, Bp, V, G, H = self.dims
        self.step_no += 1
        eng.set_dropout(None if self.dropout is None else
                        tuple(self.dropout) + ((self.drop_seed << 32) | (self.step_no & 0xFFFFFFFF),))
        check(L.etp_planner_set_aux_stream(h, self.aux), "set_aux_stream")
        check(L.etp_planner_set_aux2_stream(h, None), "set_aux2_stream")
        check(L.etp_planner_set_lazy_join(h, 0), "set_lazy_join")      # (PlannerStep re-installs its own streams at every enqueue)
        # text weights first on the main stream; everything else (panorama / x-layer casts, the gradient memset) rides on the
        # panorama stream, whose join precedes the MLM forward and every backward kernel
        check(L.etp_planner_refresh_part(h, 0, s), "refresh text weights")
        check(L.etp_memset_async(ptr(self.loss), 0, 4, s), "memset loss")
        check(L.etp_stream_after(s, s2), "fork")
        check(L.etp_planner_refresh_part(h, 1, s2), "refresh panorama weights")
        check(L.etp_planner_refresh_part(h, 2, s2), "refresh navigation weights")
        if backward and self.zero_grads:
            check(L.etp_memset_async(ptr(eng.grads), 0, eng.grads.numel() * 4, s2), "memset grads")
        check(L.etp_txt_fwd(h, ptr(i["txt_ids"]), ptr(i["txt_masks"]), B, Lt, ptr(self.txt), ptr(self.st_txt), s), "txt_fwd")
        check(L.etp_pano_fwd(h, ptr(i["rgb"]), ptr(i["dep"]), ptr(i["loc"]), ptr(i["nav"]), ptr(i["view_lens"]), Bp, V,
                             ptr(self.pano), ptr(self.pmask), ptr(self.st_pano), s2), "pano_fwd")
        check(L.etp_stream_after(s2, s), "join")
        pf, xf, wf = self.csr_f
        check(L.etp_gather_sum(_lib.ETP_F32, ptr(self.pano), ptr(pf), ptr(xf), ptr(wf), ptr(self.gimg), B * G, H, 0, s), "aggregate")
        check(L.etp_mlm_fwd(h, ptr(self.txt), ptr(i["txt_masks"]), ptr(i["step_ids"]), ptr(self.gimg), ptr(i["pos"]),
                            ptr(i["gmask"]), ptr(self.sel[0]), ptr(self.sel[1]), ptr(self.sel[2]), ptr(self.labels), B, Lt, G,
                            self.Nm, self.loss_scale / self.Nm, ptr(self.loss), ptr(self.st_mlm), s), "mlm_fwd")
        if not backward:
            return
        check(L.etp_mlm_bwd(h, ptr(self.txt), ptr(i["txt_masks"]), ptr(i["step_ids"]), ptr(i["pos"]), ptr(i["gmask"]),
                            ptr(self.selT[0]), ptr(self.selT[1]), ptr(self.selT[2]), B, Lt, G, self.Nm, ptr(self.d_txt),
                            ptr(self.d_gimg), ptr(self.st_mlm), ptr(self.ws_mlm), s), "mlm_bwd")
        pb, xb, wb = self.csr_b
        check(L.etp_gather_sum(_lib.ETP_F32, ptr(self.d_gimg), ptr(pb), ptr(xb), ptr(wb), ptr(self.d_pano), Bp * V, H, 0, s),
              "aggregate bwd")
        check(L.etp_stream_after(s, s2), "fork")                  # panorama backward beside the text backward
        check(L.etp_pano_bwd(h, ptr(self.d_pano), ptr(i["rgb"]), ptr(i["dep"]), ptr(i["loc"]), ptr(i["nav"]), Bp, V, None,
                             ptr(self.st_pano), ptr(self.ws_pano), s2), "pano_bwd")
        check(L.etp_txt_bwd(h, ptr(self.d_txt), ptr(i["txt_ids"]), ptr(i["txt_masks"]), B, Lt, ptr(self.st_txt), ptr(self.ws_txt),
                            s), "txt_bwd")
        check(L.etp_stream_after(s2, s), "join")
        check(L.etp_planner_set_aux_stream(h, None), "set_aux_stream")


# ---- multi-task driver pieces (what pretrain_src/pretrain_src/data/loader.py:18-75 and train_r2r.py:229-300 provide) -------------
class _TaskSource:
    """One task's stream of batches: a re-iterable (DataLoader, list of batches, ...) that is restarted when it runs dry, after
    telling the owner which epoch begins (DistributedSampler.set_epoch in the reference, loader.py:63-71)."""

    def __init__(self, name: str, batches, weight: float = 1.0, on_epoch=None):
        self.name, self.batches, self.weight, self.on_epoch = name, batches, float(weight), on_epoch
        self.epoch, self._it = 0, iter(batches)

    def take(self):
        try:
            return next(self._it)
        except StopIteration:
            self.epoch += 1
            if self.on_epoch is not None:
                self.on_epoch(self.epoch)
            self._it = iter(self.batches)
            return next(self._it)


class MetaLoader:
    """Task mixing for multi-task pre-training: yields ``(task_name, batch)`` forever; the task is held for ``accum_steps``
    consecutive draws (one gradient-accumulation window trains one task) and windows pick their task with probability
    proportional to the mixing weights -- the contract of the reference's loader (loader.py:54-63).

    Unlike the reference, which draws and (in distributed runs) broadcasts one task id per window from the training loop, the
    schedule here is PLANNED: ``horizon`` windows are drawn in one ``torch.multinomial`` call and rank 0's plan is broadcast
    once, so all ranks walk the same task sequence (the per-task gradient bucket sets of etpnav_amd.dp.task_grad_ranges rely
    on that) without a collective -- and a device round trip on the RCCL backend -- in front of every optimizer step.

    loaders: {name: batches  |  (batches, weight, on_epoch)} with ``batches`` anything ``iter()`` accepts repeatedly."""

    def __init__(self, loaders: Dict, accum_steps: int = 1, distributed: bool = False, device=None, generator=None,
                 horizon: int = 1024):
        if not isinstance(loaders, dict) or not loaders:
            raise ValueError("loaders: a non-empty {task: batches | (batches, weight, on_epoch)} mapping")
        self.sources: List[_TaskSource] = []
        for name, spec in loaders.items():
            batches, weight, on_epoch = spec if isinstance(spec, tuple) else (spec, 1.0, None)
            self.sources.append(_TaskSource(name, batches, weight, on_epoch))
        self.accum_steps, self.distributed, self.device = max(1, int(accum_steps)), bool(distributed), device
        self.generator, self.horizon = generator, int(horizon)
        self._plan: List[int] = []
        self._held, self._left = 0, 0          # task of the running window, draws left in it

    @property
    def names(self) -> List[str]:
        return [s.name for s in self.sources]

    def _extend_plan(self):
        w = torch.tensor([s.weight for s in self.sources], dtype=torch.float32)
        plan = torch.multinomial(w, self.horizon, replacement=True, generator=self.generator)
        if self.distributed:
            import torch.distributed as dist
            t = plan.to(self.device) if self.device is not None else plan
            dist.broadcast(t, 0)
            plan = t.cpu()
        self._plan = plan.tolist()[::-1]       # popped from the end

    def __iter__(self):
        while True:
            if self._left == 0:
                if not self._plan:
                    self._extend_plan()
                self._held, self._left = self._plan.pop(), self.accum_steps
            self._left -= 1
            src = self.sources[self._held]
            yield src.name, src.take()


class PretrainDriver:
    """The multi-task pre-training loop of train_r2r.py:229-300 on the MI355X planner: MetaLoader task mixing -> one
    device-side step of the drawn task (SAP = PlannerStep over trajectories, MLM = MlmStep) -> (multi-GPU) mean of exactly
    the gradient ranges that task writes (etpnav_amd.dp.task_grad_ranges: static per-task bucket sets instead of DDP's
    find_unused_parameters=True, utils/misc.py:58) -> warm-up-linear learning rate (optim/sched.py) -> fused AdamW with the
    reference's decay grouping and gradient clipping (optim/misc.py, optim/adamw.py, train_r2r.py:283-300).

    Step objects (preallocated stash / workspace / streams) are cached per batch shape and refilled in place for SAP; the MLM
    step is rebuilt per batch (its masked-token count changes the buffer plan)."""

    def __init__(self, model: GlocalTextPathNavCMT, loaders: Dict, learning_rate: float = 5e-5, warmup_steps: int = 10000,
                 num_train_steps: int = 100000, grad_norm: float = 5.0, accum_steps: int = 1, dropout="config", seed: int = 0,
                 distributed: bool = False, max_cached_steps: int = 4, generator=None, max_txt_len: int = 100):
        """accum_steps = gradient_accumulation_steps of train_r2r.py:231-300: that many consecutive (task, batch) draws (the
        MetaLoader keeps the task fixed over them, loader.py:61-63) each add the gradient of mean-loss / accum_steps to the
        arena, then ONE reduction / clipping / optimizer step follows.  max_txt_len = the dataset's truncation length
        (run_pt/r2r_pretrain_habitat.json: 100): with several ranks the row-sparse word-embedding exchange uses the
        rank-independent capacity B * max_txt_len (the collate pads to the per-batch maximum, so L differs across ranks)."""
        from .optim import FusedAdamW, WarmupLinearLR
        from . import dp
        if accum_steps < 1:
            raise ValueError("accum_steps must be >= 1")
        self.accum_steps, self.max_txt_len = int(accum_steps), int(max_txt_len)
        self._micro = 0                     # micro-steps since the last optimizer step
        self._touched = []                  # word rows touched by the SAP micro-steps of the current accumulation window
        self.model, self.dropout, self.seed = model, dropout, int(seed)
        self.meta = MetaLoader(loaders, accum_steps=accum_steps, distributed=distributed, device=model._engine.device,
                               generator=generator)
        self.opt = FusedAdamW(model, lr=learning_rate, hf_style=True, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01,
                              max_grad_norm=grad_norm, no_decay=FusedAdamW.reference_no_decay)
        self.sched = WarmupLinearLR(self.opt, learning_rate, warmup_steps, num_train_steps)
        self.global_step = 0
        self.distributed = distributed
        self._sap = {}                      # shape key -> PlannerStep
        self._max = max_cached_steps
        self.reducers = {}
        if distributed:
            for task in ("sap", "mlm"):
                ranges, sparse = dp.task_grad_ranges(model, task)
                self.reducers[task] = dp.GradReducer(model.flat_grads, ranges, sparse_rows=sparse)
        self.task_losses = {}
        self.lr_history = []

    def _sap_step(self, batch):
        from .step import PlannerStep
        key = PlannerStep.batch_shape_key(batch)
        st = self._sap.get(key)
        if st is None:
            if len(self._sap) >= self._max:
                self._sap.pop(next(iter(self._sap))).close()
            # accumulate-mode gradients + full zeroing: the pre-training variant carries weights this task does not touch
            st = PlannerStep(self.model, batch, dropout=self.dropout, drop_seed=self.seed, zero_grads=True, grad_overwrite=False)
            self._sap[key] = st
        else:
            st.load_batch(batch)
        return st

    def train_step(self, name: str, batch) -> torch.Tensor:
        """One micro-step on `batch` of task `name` ('sap...' / 'mlm...': the part before '_' selects the task, as
        train_r2r.py:235); every accum_steps-th call closes the window with the gradient reduction and the optimizer step.
        Returns the device loss tensor of this micro-step, already scaled by 1 / accum_steps as the reference logs it
        (no host sync)."""
        task = name.split("_")[0]
        first = self._micro == 0
        A = self.accum_steps
        if task == "sap":
            st = self._sap_step(batch)
            st.step_no = self.global_step * A + self._micro
            st.zero_grads = first
            st.loss_scale = 1.0 / (st.B * A)
            st.run_eager()
            if st.Lt > self.max_txt_len and self.distributed:
                raise ValueError(f"instruction length {st.Lt} exceeds max_txt_len={self.max_txt_len} (the exchange capacity)")
            self._touched.append((st.inp["txt_ids"].reshape(-1).clone() if A > 1 else st.inp["txt_ids"].reshape(-1), st.B))
        elif task == "mlm":
            st = MlmStep(self.model, batch, dropout=self.dropout, drop_seed=self.seed)
            st.step_no = self.global_step * A + self._micro
            st.zero_grads = first
            st.loss_scale = 1.0 / A
            st.run_eager()
        else:
            raise ValueError(f"unknown task {task!r}: this fork pre-trains with 'mlm' and 'sap' (pretrain_cmt.py:141-163,223-283)")
        loss = st.loss.clone()
        self._micro += 1
        self.task_losses.setdefault(name, []).append(loss)
        if self._micro < A:
            if task == "mlm":
                torch.cuda.current_stream().synchronize()
                st.close()
            return loss
        if self.distributed:
            red = self.reducers[task]
            for i in range(len(red.ranges)):
                red.reduce_bucket(i)
            if task == "sap":
                ids = torch.cat([t for t, _ in self._touched]) if len(self._touched) > 1 else self._touched[0][0]
                red.reduce_sparse_rows(ids, capacity=sum(b for _, b in self._touched) * self.max_txt_len)
            red.finish()
        self._micro = 0
        self._touched = []
        self.global_step += 1
        self.lr_history.append(self.sched.step(self.global_step))
        self.opt.step()
        if task == "mlm":
            torch.cuda.current_stream().synchronize()      # the per-batch MLM step object is released: its buffers must be idle
            st.close()
        return loss

    def run(self, num_steps: int):
        it = iter(self.meta)
        out = []
        for _ in range(num_steps * self.accum_steps):       # num_steps OPTIMIZER steps
            name, batch = next(it)
            out.append((name, self.train_step(name, batch)))
        return out

    def close(self):
        for st in self._sap.values():
            st.close()
        self._sap.clear()
