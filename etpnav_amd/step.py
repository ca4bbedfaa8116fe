"""One whole planner training step (the unit of BASELINE.json's metric) driven straight through the C ABI.

``PlannerStep`` preallocates every buffer for a fixed (B, L, V, G) shape and runs

    weight refresh (bf16) -> zero grads -> forward_txt -> forward_panorama -> node assembly (gather-mean)
    -> forward_navigation -> cross-entropy(sum)/B -> backward of all of it into the flat gradient arena

mirroring one rollout step of ss_trainer_ETP.py:801-892 plus the backward of :504.  Nothing is
allocated or synchronised inside ``run_eager`` so the step can be captured into one hipGraph
(``capture()`` / ``replay()``) — launch-bound inner loops replay from the graph instead of being re-issued.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional

import torch

from . import _lib
from ._lib import check, ptr
from .planner import GlocalTextPathNavCMT


def build_node_csr(view_lens: torch.Tensor, V: int, G: int):
    """CSR (and its transpose) of the benchmark node assembly: node 0 = [stop] (empty), node 1 = mean of the valid
    views (ss_trainer_ETP.py:838-839), node g>=2 = view (g-2) mod view_len (one-occurrence ghost nodes,
    graph_utils.py:224-234).  Host-side index bookkeeping only (the reference does this in Python too)."""
    B = view_lens.numel()
    vl = [int(x) for x in view_lens.tolist()]
    ptr_f, idx_f, w_f = [0], [], []
    rev = [[] for _ in range(B * V)]
    for b in range(B):
        for g in range(G):
            n = b * G + g
            if g == 1:
                for v in range(vl[b]):
                    idx_f.append(b * V + v); w_f.append(1.0 / vl[b]); rev[b * V + v].append((n, 1.0 / vl[b]))
            elif g >= 2:
                v = (g - 2) % vl[b]
                idx_f.append(b * V + v); w_f.append(1.0); rev[b * V + v].append((n, 1.0))
            ptr_f.append(len(idx_f))
    ptr_b, idx_b, w_b = [0], [], []
    for r in rev:
        for n, w in r:
            idx_b.append(n); w_b.append(w)
        ptr_b.append(len(idx_b))
    i32 = lambda x: torch.tensor(x, dtype=torch.int32)
    f32 = lambda x: torch.tensor(x, dtype=torch.float32)
    return (i32(ptr_f), i32(idx_f), f32(w_f)), (i32(ptr_b), i32(idx_b), f32(w_b))


def _side_stream(L, level: int):
    """leaf stream (weight gradients / the panorama branch) at the lowest priority.  (Confining these streams to a slice of the
    chip with hipExtStreamCreateWithCUMask was measured 1.8x SLOWER -- 7.8 ms per step whatever the share,
    profiles/r03_ab_runs.json group c4 -- and is not offered.)"""
    h = ctypes.c_void_p()
    check(L.etp_stream_create_prio(ctypes.byref(h), level), "stream_create")
    return h.value


class PlannerStep:
    def __init__(self, model: GlocalTextPathNavCMT, batch: Dict[str, torch.Tensor], overlap: bool = True,
                 dropout=None, drop_seed: int = 0, refresh_weights: bool = True, zero_grads: bool = True,
                 grad_overwrite: Optional[bool] = None, share_side_streams: Optional["PlannerStep"] = None):
        """dropout: None (eval-mode step), "config" (the model config's rates, the reference's policy.train()), or a
        tuple (p_hidden, p_attn, p_head, p_env).  refresh_weights / zero_grads = False when etpnav_amd.optim.FusedAdamW
        closes the step: its kernel already wrote the bf16 weight shadow and zeroed the gradient arena.
        grad_overwrite: weight-gradient GEMMs store instead of accumulating (etp_planner_set_grad_overwrite) and only the
        vector/table tail of the gradient arena is zeroed per step.  Default: on when this step owns the zeroing (fresh
        gradients every step) and every weight matrix is touched exactly once per step (not the pre-training variant, whose
        language-side x-layer weights this step does not touch)."""
        self.model = model
        self.refresh_weights, self.zero_grads = refresh_weights, zero_grads
        # frozen sub-models (vlnbert_init.py:51-54 -> vilmodel_cmt.py:422-433,675-682): with fix_lang_embedding the text encoder's
        # output is detached (LanguageEncoder.forward :431-432) -- no text backward runs and no gradient reaches `embeddings.*` /
        # `lang_encoder.*`.  fix_pano_embedding alone does NOT stop the panorama backward: forward_panorama adds
        # embeddings.token_type_embeddings(1) (vilmodel_cmt.py:706-708), which stays trainable, so the gradient still crosses the
        # (frozen) panorama encoder to reach it; only with BOTH flags nothing behind the panorama branch requires a gradient
        # (the features are inputs) and autograd never enters it -- neither does this step.  Gradient slots of frozen parameters
        # in the flat arena are unspecified (their .grad is None, FusedAdamW and the data-parallel buckets leave them out).
        from .planner import _cfg_get
        self.train_txt = not bool(_cfg_get(model.config, "fix_lang_embedding", False))
        self.train_pano = self.train_txt or not bool(_cfg_get(model.config, "fix_pano_embedding", False))
        if grad_overwrite is None:
            grad_overwrite = bool(zero_grads) and not model._engine.cconf.use_lang2visn and \
                os.environ.get("ETP_GRAD_OVERWRITE", "1") != "0"
        self.grad_overwrite = bool(grad_overwrite)
        if dropout == "config":
            c = model.config
            dropout = (float(getattr(c, "hidden_dropout_prob", 0.1)), float(getattr(c, "attention_probs_dropout_prob", 0.1)),
                       float(getattr(c, "pred_head_dropout_prob", 0.1)), float(model.drop_env_prob))
        self.dropout = dropout
        self.drop_seed = int(drop_seed) & 0xFFFFFFFF
        self.step_no = 0
        self.loss_scale = 1.0 / batch["txt_ids"].shape[0]      # cross_entropy(sum) / batch_size (ss_trainer_ETP.py:892)
        eng = self.eng = model._engine
        eng.require_gpu()
        dev = eng.device
        self.L = eng.L
        self.B, self.Lt = batch["txt_ids"].shape
        self.V = batch["rgb_fts"].shape[1]
        self.G = batch["gmap_step_ids"].shape[1]
        B, Lt, V, G, H = self.B, self.Lt, self.V, self.G, eng.cconf.hidden
        t = eng.tdtype
        mv = lambda x, dt=None: (x.to(dt) if dt is not None else x).contiguous().to(dev)
        self.inp = {
            "txt_ids": mv(batch["txt_ids"], torch.int64), "txt_masks": mv(batch["txt_masks"], torch.bool),
            "rgb": mv(batch["rgb_fts"], torch.float32), "dep": mv(batch["dep_fts"], torch.float32),
            "loc": mv(batch["loc_fts"], torch.float32), "nav": mv(batch["nav_types"], torch.int64),
            "view_lens": mv(batch["view_lens"], torch.int64), "step_ids": mv(batch["gmap_step_ids"], torch.int64),
            "pos": mv(batch["gmap_pos_fts"], torch.float32), "gmask": mv(batch["gmap_masks"], torch.bool),
            "visited": mv(batch["gmap_visited_masks"], torch.bool), "dists": mv(batch["gmap_pair_dists"], torch.float32),
            "labels": mv(batch["labels"], torch.int64),
        }
        # Panorama batch: B (one panorama per episode, the fine-tuning rollout step) or sum of trajectory steps (the
        # pre-training SAP task, pretrain_cmt.py:223-283: one panorama per step of every episode); the node features are a
        # CSR gather over all panorama embeddings either way.
        Bp = self.Bp = batch["rgb_fts"].shape[0]
        if "traj" in batch:
            from .graph_inputs import pack_traj_csr
            tr = batch["traj"]
            (pf, xf, wf), (pb, xb, wb) = pack_traj_csr(tr["traj_vp_lens"], tr["traj_vpids"], tr["traj_cand_vpids"],
                                                       tr["gmap_vpids"], V, G)
        else:
            if Bp != B:
                raise ValueError("rgb_fts batch differs from txt_ids batch: pass the trajectory lists as batch['traj']")
            (pf, xf, wf), (pb, xb, wb) = build_node_csr(batch["view_lens"].cpu(), V, G)
        self.csr_f = tuple(x.to(dev) for x in (pf, xf, wf))
        self.csr_b = tuple(x.to(dev) for x in (pb, xb, wb))
        e = lambda *s, dt=torch.float32: torch.empty(*s, dtype=dt, device=dev)   # API tensors are fp32 in both modes
        self.txt = e(B, Lt, H); self.pano = e(Bp, V, H); self.pmask = e(Bp, V, dt=torch.bool)
        self.gimg = e(B, G, H); self.gemb = e(B, G, H); self.logits = e(B, G, dt=torch.float32)
        self.dlogits = e(B, G, dt=torch.float32); self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.d_txt = e(B, Lt, H); self.d_gimg = e(B, G, H); self.d_pano = e(Bp, V, H)
        h = eng.handle
        self.st_txt = eng.buf(self.L.etp_txt_stash_bytes(h, B, Lt))
        self.st_pano = eng.buf(self.L.etp_pano_stash_bytes(h, Bp, V))
        self.st_nav = eng.buf(self.L.etp_nav_stash_bytes(h, B, Lt, G))
        # separate backward workspaces: the three backward passes may run on parallel streams
        self.ws_txt = eng.buf(self.L.etp_txt_ws_bytes(h, B, Lt))
        self.ws_pano = eng.buf(self.L.etp_pano_ws_bytes(h, Bp, V))
        self.ws_nav = eng.buf(self.L.etp_nav_ws_bytes(h, B, Lt, G))
        # side streams: `aux` carries the weight-gradient GEMMs, `s2` the panorama branch (independent of the text branch)
        self.overlap = overlap
        self.aux = self.s2 = None
        # leaf work (weight gradients; the panorama branch) runs at the lowest stream priority: whenever its workgroups and the
        # dependent chain's wait for the same CUs, the chain's are dispatched first (ETP_STREAM_PRIO=0: all default)
        low = -1 if os.environ.get("ETP_STREAM_PRIO", "1") != "0" else 0
        self._own_aux = share_side_streams is None
        if not self._own_aux:                  # MicroBatchedStep: one weight-gradient / d_txt stream for all micro-batches
            self.aux = share_side_streams.aux
        elif overlap in (True, "aux", "both"):
            self.aux = _side_stream(self.L, low)
        if overlap in (True, "s2", "both"):
            self.s2 = _side_stream(self.L, low)
        self.aux2 = None
        if not self._own_aux:
            self.aux2 = share_side_streams.aux2
        elif self.aux is not None and os.environ.get("ETP_DTXT_STREAM", "1") != "0":
            a2 = ctypes.c_void_p()
            check(self.L.etp_stream_create(ctypes.byref(a2)), "stream_create")
            self.aux2 = a2.value
        self._lazy = 1 if self.aux is not None else 0
        # schedule switches are read ONCE, here (no environment lookups on the per-step path)
        self._txt_cast_split = os.environ.get("ETP_TXT_CAST_SPLIT", "1") != "0"
        self._chain_first = os.environ.get("ETP_CHAIN_FIRST", "0") == "1"
        # node assembly (gather-mean of the view embeddings) rides behind pano_fwd on the panorama stream, off the chain (round 6: -0.35 %,
        # 4.045 against 4.059 ms, three pairs, profiles/r06_ab_runs.json r6c9); ETP_ASSEMBLE_ON_S2=0 puts it back behind the join
        self._assemble_on_s2 = os.environ.get("ETP_ASSEMBLE_ON_S2", "1") != "0"
        self._install_streams()
        self._pano_pending = False
        self.graph = None
        self.graphs = []
        self.stream = None

    def _install_streams(self):
        """The planner handle is shared by every step object of a model (PretrainDriver alternates MlmStep and several cached
        PlannerSteps): its side streams and lazy-join level are (re)installed at every enqueue, so a step never runs on another
        step's streams or silently loses its overlap schedule.  navigation / panorama weight gradients keep running on the aux
        stream while the text backward starts; the text backward's own join (or the explicit one in enqueue_main) completes
        them."""
        h = self.eng.handle
        check(self.L.etp_planner_set_aux_stream(h, self.aux), "set_aux_stream")
        check(self.L.etp_planner_set_aux2_stream(h, self.aux2), "set_aux2_stream")
        check(self.L.etp_planner_set_lazy_join(h, self._lazy), "set_lazy_join")

    def shape_key(self):
        return (self.B, self.Lt, self.Bp, self.V, self.G)

    @staticmethod
    def batch_shape_key(batch):
        B, Lt = batch["txt_ids"].shape
        Bp, V = batch["rgb_fts"].shape[:2]
        return (B, Lt, Bp, V, batch["gmap_step_ids"].shape[1])

    def load_batch(self, batch: Dict[str, torch.Tensor]):
        """Refill the preallocated device inputs with another batch of the SAME shapes (the stash / workspace buffers and the
        streams are reused; only the small CSR index arrays of the node aggregation are rebuilt)."""
        if self.batch_shape_key(batch) != self.shape_key():
            raise ValueError(f"batch shapes {self.batch_shape_key(batch)} differ from this step's {self.shape_key()}")
        if self.graphs:
            # the CSR index tensors below are re-allocated: kernel nodes of an existing graph would keep the old pointers
            raise RuntimeError("load_batch() on a step with captured / recorded graphs: close() the graphs (or build a new step) first")
        names = {"txt_ids": "txt_ids", "txt_masks": "txt_masks", "rgb": "rgb_fts", "dep": "dep_fts", "loc": "loc_fts", "nav": "nav_types",
                 "view_lens": "view_lens", "step_ids": "gmap_step_ids", "pos": "gmap_pos_fts", "gmask": "gmap_masks",
                 "visited": "gmap_visited_masks", "dists": "gmap_pair_dists", "labels": "labels"}
        for k, src in names.items():
            t = self.inp[k]
            t.copy_(batch[src].to(t.dtype), non_blocking=True)
        dev = self.eng.device
        if "traj" in batch:
            from .graph_inputs import pack_traj_csr
            tr = batch["traj"]
            f, b = pack_traj_csr(tr["traj_vp_lens"], tr["traj_vpids"], tr["traj_cand_vpids"], tr["gmap_vpids"], self.V, self.G)
        else:
            f, b = build_node_csr(batch["view_lens"].cpu(), self.V, self.G)
        self.csr_f = tuple(x.to(dev) for x in f)
        self.csr_b = tuple(x.to(dev) for x in b)

    # ------------------------------------------------------------------------------------------
    def _enqueue_pano_bwd(self, s: int):
        if not self.train_pano:
            return
        L, h, i = self.L, self.eng.handle, self.inp
        s2 = self.s2 if self.s2 is not None else s
        self._install_streams()
        self.eng.set_dropout(self._drop_state())
        check(L.etp_planner_set_grad_overwrite(h, int(self.grad_overwrite)), "set_grad_overwrite")
        check(L.etp_pano_bwd(h, ptr(self.d_pano), ptr(i["rgb"]), ptr(i["dep"]), ptr(i["loc"]), ptr(i["nav"]), self.Bp, self.V, None,
                             ptr(self.st_pano), ptr(self.ws_pano), s2), "pano_bwd")
        check(L.etp_planner_set_grad_overwrite(h, 0), "set_grad_overwrite")
        self._pano_pending = True

    def enqueue_main(self, s: int, backward: bool = True, join_pano: bool = True, defer_pano: bool = False):
        """Everything except the text-encoder backward: weight refresh, zero grads, the three forwards, loss and the
        navigation + panorama backward.  The panorama branch (forward and backward) runs on a second stream beside the
        text branch; with join_pano=False its backward is left running and joined by enqueue_txt_bwd."""
        self.enqueue_fwd(s, backward)
        if backward:
            self.enqueue_bwd_main(s, join_pano, defer_pano)

    def enqueue_fwd(self, s: int, backward: bool = True, after_prologue=None):
        """weight refresh, gradient zeroing, forward_txt || forward_panorama, node assembly, forward_navigation, loss"""
        s2 = self.s2 if self.s2 is not None else s
        L, eng, h, i = self.L, self.eng, self.eng.handle, self.inp
        B, Lt, V, G, H = self.B, self.Lt, self.V, self.G, eng.cconf.hidden
        dt = _lib.ETP_F32          # node assembly works on the fp32 API tensors
        self.step_no += 1
        self._install_streams()
        eng.set_dropout(self._drop_state())
        check(L.etp_planner_set_grad_overwrite(h, int(self.grad_overwrite)), "set_grad_overwrite")
        L.etp_stamp_mark(s, 1)                 # measurement aid: a no-op unless a stamp sink is installed (tools/chain_waits.py)
        # weight-shadow refresh and gradient zeroing ride on the stream that first needs them: the text cast on the main
        # stream, the panorama/navigation casts and the (bandwidth-bound) gradient memset on the panorama stream, whose
        # join below precedes forward_navigation and every backward kernel
        if self.refresh_weights:
            # only layer 0's cast stays in front of the first text GEMM; layers 1.. are cast on the panorama stream and
            # etp_txt_fwd waits for them after its layer 0 (ETP_TXT_CAST_SPLIT=0: the whole text cast on the main stream)
            if self.s2 is not None and self._txt_cast_split:
                check(L.etp_planner_refresh_text_split(h, s, s2), "refresh text weights")
            else:
                check(L.etp_planner_refresh_part(h, 0, s), "refresh text weights")
        check(L.etp_stream_after(s, s2), "fork")
        if self.refresh_weights:
            check(L.etp_planner_refresh_part(h, 1, s2), "refresh panorama weights")
            check(L.etp_planner_refresh_part(h, 2, s2), "refresh navigation weights")
        if backward and self.zero_grads:
            # overwrite mode: the matrix region [0, n_matrix) is fully rewritten by this step's weight-gradient stores
            # (round 6: navigation cast + this memset BEHIND the panorama branch instead of in front of it: no gain, r06_ab_runs.json r6c11;
            # the memset beside forward_navigation, the backward waiting for it: +0.7 %, 3.952 against 3.924 ms, four pairs, r6c16)
            lo = eng.n_matrix if self.grad_overwrite else 0
            check(L.etp_memset_async(eng.grads.data_ptr() + lo * 4, 0, (eng.grads.numel() - lo) * 4, s2), "memset grads")
        if after_prologue is not None:         # MicroBatchedStep: the other micro-batches' streams are ordered after the casts / memset
            after_prologue(s, s2)
        check(L.etp_txt_fwd(h, ptr(i["txt_ids"]), ptr(i["txt_masks"]), B, Lt, ptr(self.txt), ptr(self.st_txt), s), "txt_fwd")
        check(L.etp_pano_fwd(h, ptr(i["rgb"]), ptr(i["dep"]), ptr(i["loc"]), ptr(i["nav"]), ptr(i["view_lens"]), self.Bp, V,
                             ptr(self.pano), ptr(self.pmask), ptr(self.st_pano), s2), "pano_fwd")
        pf, xf, wf = self.csr_f
        if self._assemble_on_s2 and s2 != s:   # the gather-mean of the view embeddings depends on the panorama branch only: behind it, off the chain
            check(L.etp_gather_sum(dt, ptr(self.pano), ptr(pf), ptr(xf), ptr(wf), ptr(self.gimg), B * G, H, 0, s2), "node assembly")
        check(L.etp_stream_after(s2, s), "join")
        L.etp_stamp_mark(s, 2)
        if not (self._assemble_on_s2 and s2 != s):
            check(L.etp_gather_sum(dt, ptr(self.pano), ptr(pf), ptr(xf), ptr(wf), ptr(self.gimg), B * G, H, 0, s), "node assembly")
        check(L.etp_nav_fwd(h, ptr(self.txt), ptr(i["txt_masks"]), ptr(i["step_ids"]), ptr(self.gimg), ptr(i["pos"]),
                            ptr(i["gmask"]), ptr(i["visited"]), ptr(i["dists"]), B, Lt, G, ptr(self.gemb), ptr(self.logits),
                            ptr(self.st_nav), s), "nav_fwd")
        check(L.etp_sap_ce(ptr(self.logits), ptr(i["labels"]), ptr(self.loss), ptr(self.dlogits) if backward else None, B, G,
                           self.loss_scale, -100, s), "sap_ce")
        L.etp_stamp_mark(s, 3)
        check(L.etp_planner_set_grad_overwrite(h, 0), "set_grad_overwrite")     # the mode never leaks to other users of the planner

    def enqueue_bwd_main(self, s: int, join_pano: bool = True, defer_pano: bool = False):
        """navigation backward, node assembly backward, panorama backward (on the panorama stream)"""
        s2 = self.s2 if self.s2 is not None else s
        L, eng, h, i = self.L, self.eng, self.eng.handle, self.inp
        B, Lt, V, G, H = self.B, self.Lt, self.V, self.G, eng.cconf.hidden
        dt = _lib.ETP_F32
        self._install_streams()
        eng.set_dropout(self._drop_state())
        check(L.etp_planner_set_grad_overwrite(h, int(self.grad_overwrite)), "set_grad_overwrite")
        check(L.etp_nav_bwd(h, None, ptr(self.dlogits), ptr(self.txt), ptr(i["txt_masks"]), ptr(i["step_ids"]),
                            ptr(i["pos"]), ptr(i["gmask"]), ptr(i["visited"]), ptr(i["dists"]), B, Lt, G, ptr(self.d_txt),
                            ptr(self.d_gimg), ptr(self.st_nav), ptr(self.ws_nav), s), "nav_bwd")
        pb, xb, wb = self.csr_b
        L.etp_stamp_mark(s, 4)
        if not self.train_pano:    # frozen panorama embedding: the backward stops at the node features (d_gimg)
            if join_pano:
                check(L.etp_planner_join_aux(h, s), "join aux")
            check(L.etp_planner_set_grad_overwrite(h, 0), "set_grad_overwrite")
            return
        # (round 6: this gather on the panorama stream behind the fork instead -- it feeds the panorama backward only -- measured neutral,
        # 3.971 against 3.969 ms over four pairs, r06_ab_runs.json r6c14: the text backward waits for d txt_embeds meanwhile anyway)
        check(L.etp_gather_sum(dt, ptr(self.d_gimg), ptr(pb), ptr(xb), ptr(wb), ptr(self.d_pano), self.Bp * V, H, 0, s),
              "node assembly bwd")
        L.etp_stamp_mark(s, 5)
        if defer_pano:             # the caller enqueues the panorama backward later (run_eager: after the first text layers);
            check(L.etp_stream_after(s, s2), "fork")             # its stream is ordered after d_pano's producer already now
            check(L.etp_planner_set_grad_overwrite(h, 0), "set_grad_overwrite")
            return
        check(L.etp_stream_after(s, s2), "fork")
        check(L.etp_pano_bwd(h, ptr(self.d_pano), ptr(i["rgb"]), ptr(i["dep"]), ptr(i["loc"]), ptr(i["nav"]), self.Bp, V, None,
                             ptr(self.st_pano), ptr(self.ws_pano), s2), "pano_bwd")
        if join_pano:
            check(L.etp_stream_after(s2, s), "join")
            check(L.etp_planner_join_aux(h, s), "join aux")      # callers of this mode read the non-text gradients next
        else:
            self._pano_pending = True
        check(L.etp_planner_set_grad_overwrite(h, 0), "set_grad_overwrite")     # the mode never leaks to other users of the planner

    def enqueue_txt_bwd(self, s: int, layer_lo: int = 0, layer_hi: Optional[int] = None):
        """Text-encoder backward (optionally only layers [layer_lo, layer_hi), descending calls share the running gradient)."""
        L, eng, i = self.L, self.eng, self.inp
        if layer_hi is None:
            layer_hi = eng.cconf.n_l
        if not self.train_txt:     # frozen language side: only the joins this call owes its callers
            if layer_lo == 0:
                self._join_side(s)
            return
        self._install_streams()
        eng.set_dropout(self._drop_state())
        check(L.etp_planner_set_grad_overwrite(eng.handle, int(self.grad_overwrite)), "set_grad_overwrite")
        L.etp_stamp_mark(s, 6)
        check(L.etp_txt_bwd_range(eng.handle, ptr(self.d_txt), ptr(i["txt_ids"]), ptr(i["txt_masks"]), self.B, self.Lt,
                                  ptr(self.st_txt), ptr(self.ws_txt), layer_lo, layer_hi, s), "txt_bwd")
        check(L.etp_planner_set_grad_overwrite(eng.handle, 0), "set_grad_overwrite")
        if layer_lo > 0:
            return
        if self._pano_pending:
            check(L.etp_stream_after(self.s2 if self.s2 is not None else s, s), "join")
            self._pano_pending = False
        L.etp_stamp_mark(s, 7)

    def _join_side(self, s: int):
        """order `s` after the weight-gradient stream and a still-running panorama backward (what the text backward's last
        range does for a trainable text encoder)"""
        check(self.L.etp_planner_join_aux(self.eng.handle, s), "join aux")
        if self._pano_pending:
            check(self.L.etp_stream_after(self.s2 if self.s2 is not None else s, s), "join")
            self._pano_pending = False
        self.L.etp_stamp_mark(s, 7)

    def _drop_state(self):
        if self.dropout is None:
            return None
        return tuple(self.dropout) + ((self.drop_seed << 32) | (self.step_no & 0xFFFFFFFF),)

    def run_eager(self, stream: Optional[int] = None, backward: bool = True):
        """Enqueue one step on `stream` (default: torch's current stream).  With dropout on, every call draws fresh
        masks (the step counter is part of the seed); a captured graph replays the masks it was captured with.

        ETP_CHAIN_FIRST=1 enqueues the top three text-backward layers BEFORE the panorama backward (a side branch of ~45
        launches).  Measured neutral (203.7 vs 206.1 steps/s, same box): the host issues a whole step in 1.4 ms against
        4.85 ms of GPU time (tools/host_timing.py), so issue order does not matter; the default keeps the simple order."""
        s = stream if stream is not None else self.eng.stream()
        if not backward:
            self.enqueue_main(s, False, join_pano=True)
            return
        n_l = self.eng.cconf.n_l
        head = max(0, min(3, n_l - 1)) if self._chain_first else 0
        self.enqueue_main(s, True, join_pano=False, defer_pano=head > 0)   # panorama backward overlaps the text backward
        if head > 0:
            lazy = self.aux is not None
            if lazy:      # the top layers' weight gradients keep running while the chain continues (no join between the ranges)
                self._lazy = 2
            self.enqueue_txt_bwd(s, n_l - head, n_l)
            if lazy:
                self._lazy = 1
            self._enqueue_pano_bwd(s)
            self.enqueue_txt_bwd(s, 0, n_l - head)
        else:
            self.enqueue_txt_bwd(s)

    def run_data_parallel(self, groups, bucket_ready, stream: Optional[int] = None, overlapped: bool = True):
        """One step issued for data-parallel training (replaces DDP's bucket hooks, ss_trainer_ETP.py:208-212): the text
        backward runs in the layer `groups` [(lo, hi), ...] (last layers first) and `bucket_ready(i, side_streams)` is called
        as soon as bucket i's gradients are ENQUEUED -- bucket 0 = everything outside the text encoder, bucket 1 + k = text
        group k (dp.planner_buckets_layered).

        overlapped=True (a reducer whose communication stream can wait for side streams: dp.GradReducer with the library
        communicator): the step keeps the free-running single-GPU schedule -- the panorama backward and all weight gradients
        stay on their side streams, nothing is joined into the dependent chain between the groups -- and `side_streams` names
        the streams that, besides `stream`, hold producers of the bucket (the reducer orders its communication stream after
        them).  Measured on one MI355X without collectives: 4.60 ms for the joined order below against 4.24 ms for the
        free-running step (bench.py --dp-schedule).
        overlapped=False (torch.distributed collectives, which only see the main stream): everything a bucket needs is joined
        into `stream` before the callback (side_streams = ())."""
        s = stream if stream is not None else self.eng.stream()
        if not overlapped or self.aux is None:
            self.enqueue_main(s, True, join_pano=True)
            bucket_ready(0, ())
            for k, (lo, hi) in enumerate(groups):
                self.enqueue_txt_bwd(s, lo, hi)
                bucket_ready(1 + k, ())
            return
        side = tuple(x for x in (self.aux, self.s2) if x is not None)
        self.enqueue_main(s, True, join_pano=False)          # the panorama backward keeps running beside the text backward
        if not groups or not self.train_txt:                 # frozen text encoder (dp.planner_buckets_layered returns no text groups)
            bucket_ready(0, side)
            self._join_side(s)
            return
        self._lazy = 2                                       # no join between the layer groups; the last group (layer 0) joins
        try:
            for k, (lo, hi) in enumerate(groups):
                self.enqueue_txt_bwd(s, lo, hi)
                if k == 0:
                    bucket_ready(0, side)                    # issued after the first group: the panorama branch gets a head start
                bucket_ready(1 + k, (self.aux,))
        finally:
            self._lazy = 1

    # ------------------------------------------------------------------------------------------
    def _capture_one(self, fn):
        check(self.L.etp_graph_begin(self.stream), "graph_begin")
        try:
            fn(self.stream)
        finally:
            g = ctypes.c_void_p()
            rc = self.L.etp_graph_end(self.stream, ctypes.byref(g))
        check(rc, "graph_end")
        return g.value

    def capture(self, backward: bool = True, split_text_bwd: bool = False):
        """Warm up eagerly (sets kernel attributes), then capture the step into hipGraph(s) on a private stream.
        With split_text_bwd the text-encoder backward is a second graph so that a gradient all-reduce of everything
        else can be issued between the two (data-parallel overlap)."""
        if self.aux is not None and self.s2 is not None and not os.environ.get("ETP_GRAPH_FORCE"):
            raise _lib.EtpError("hipGraph capture with two side streams crashes hipStreamEndCapture on ROCm 7.2; "
                                "build the PlannerStep with overlap='s2', 'aux' or False for graph replay "
                                "(measured on MI355X: eager two-stream issue is the fastest mode anyway)")
        torch.cuda.synchronize()
        self.run_eager(backward=backward)
        torch.cuda.synchronize()
        s = ctypes.c_void_p()
        check(self.L.etp_stream_create(ctypes.byref(s)), "stream_create")
        self.stream = s.value
        self.graphs = []
        if backward and split_text_bwd:
            self.graphs.append(self._capture_one(lambda st: self.enqueue_main(st, True)))
            self.graphs.append(self._capture_one(self.enqueue_txt_bwd))
        else:
            self.graphs.append(self._capture_one(lambda st: self.run_eager(stream=st, backward=backward)))
        self.graph = self.graphs[0]
        return self

    def record(self, backward: bool = True, split_text_bwd: bool = False):
        """Build the step's hipGraph EXPLICITLY (etp_rec_begin / etp_rec_end: one kernel node per launch, dependencies from
        the per-stream order and the fork/join events) instead of by stream capture -- this keeps the full three-stream
        schedule (capture() cannot: hipStreamEndCapture crashes with two side streams on ROCm 7.2) while the host issues a
        single hipGraphLaunch per step.  Like capture(), a recorded graph replays the dropout masks it was built with."""
        torch.cuda.synchronize()
        self.run_eager(backward=backward)          # warm-up: kernel attributes, lazy allocations
        torch.cuda.synchronize()
        s = ctypes.c_void_p()
        check(self.L.etp_stream_create(ctypes.byref(s)), "stream_create")
        self.stream = s.value
        self.graphs, self.graph_stats = [], []

        def rec(fn):
            check(self.L.etp_rec_begin(), "rec_begin")
            try:
                fn(self.stream)
            except Exception:
                self.L.etp_rec_abort()
                raise
            g, nk, ne = ctypes.c_void_p(), ctypes.c_int64(), ctypes.c_int64()
            check(self.L.etp_rec_end(ctypes.byref(g), ctypes.byref(nk), ctypes.byref(ne)), "rec_end")
            self.graph_stats.append((int(nk.value), int(ne.value)))
            return g.value

        if backward and split_text_bwd:
            self.graphs.append(rec(lambda st: self.enqueue_main(st, True)))
            self.graphs.append(rec(self.enqueue_txt_bwd))
        else:
            self.graphs.append(rec(lambda st: self.run_eager(stream=st, backward=backward)))
        self.graph = self.graphs[0]
        return self

    def replay(self, part: Optional[int] = None, stream: Optional[int] = None):
        """Launch the captured graph(s) on `stream` (default: torch's current stream, so torch ops order after it)."""
        s = stream if stream is not None else self.eng.stream()
        for k, g in enumerate(self.graphs):
            if part is None or part == k:
                check(self.L.etp_graph_launch(g, s), "graph_launch")

    def sync(self):
        if self.stream is not None:
            check(self.L.etp_stream_sync(self.stream), "stream_sync")
        torch.cuda.synchronize()

    def time_replays(self, iters: int) -> float:
        """HIP-event time (ms) of `iters` back-to-back replays of the single-graph step on the capture stream."""
        assert len(self.graphs) == 1
        ms = ctypes.c_float()
        check(self.L.etp_graph_time(self.graphs[0], self.stream, iters, ctypes.byref(ms)), "graph_time")
        return float(ms.value)

    def close(self):
        for g in getattr(self, "graphs", []) or []:
            self.L.etp_graph_destroy(g)
        self.graphs, self.graph = [], None
        if self.stream is not None:
            self.L.etp_stream_destroy(self.stream); self.stream = None
        self.L.etp_planner_set_lazy_join(self.eng.handle, 0)
        self.L.etp_planner_set_aux_stream(self.eng.handle, None)
        self.L.etp_planner_set_aux2_stream(self.eng.handle, None)
        for st in ((self.aux, self.s2, getattr(self, "aux2", None)) if self._own_aux else (self.s2,)):
            if st is not None:
                self.L.etp_stream_destroy(st)
        self.aux = self.s2 = self.aux2 = None


class MicroBatchedStep:
    """The same training step (one gradient of the mean loss over the whole batch, ss_trainer_ETP.py:892,1055) issued as
    `n_micro` micro-batches -- a GRADIENT-ACCUMULATION facility for a batch whose activations do not fit, not a
    performance mode.

    Measured on MI355X (profiles/r02s_bench_micro*.json): 2 micro-batches 10.8-13.7 ms, 4 micro-batches
    25.8 ms against 4.4 ms for the single chain.  The idea that one chain's kernels would fill the other's launch / drain
    gaps does not hold on this chip: a half-batch chain takes as long as the full-batch one (its kernels are bound by the
    per-launch latency, not by work), and the chains meet on the shared weight-gradient stream, where the later
    micro-batches must accumulate behind the first one's stores in issue order.
    Gradients: the first micro-batch's weight-gradient GEMMs store (first touch), the others accumulate behind it on the
    shared weight-gradient stream (in issue order, so no atomics are needed on the matrices); vector / table gradients are
    atomics as before.  The loss of every micro-batch is scaled by 1/B of the WHOLE batch, so the arena ends up with exactly
    the full-batch gradient (fp32 summation order aside; tests/test_planner_gpu.py).  Dropout draws an independent mask
    stream per micro-batch.  The pre-training loop's gradient_accumulation_steps (train_r2r.py:231-300) is implemented in
    etpnav_amd.pretrain.PretrainDriver on whole task batches instead."""

    def __init__(self, model: GlocalTextPathNavCMT, batch: Dict[str, torch.Tensor], n_micro: int = 2, dropout=None,
                 drop_seed: int = 0):
        if "traj" in batch:
            raise ValueError("MicroBatchedStep splits along the episode axis: trajectory-list batches are not supported")
        B = batch["txt_ids"].shape[0]
        if n_micro < 1 or B % n_micro != 0:
            raise ValueError(f"batch size {B} is not a multiple of n_micro={n_micro}")
        self.model, self.eng, self.L = model, model._engine, model._engine.L
        self.B, self.n = B, n_micro
        mb = B // n_micro
        self.parts = []
        for k in range(n_micro):
            sub = {key: (v[k * mb:(k + 1) * mb] if torch.is_tensor(v) and v.shape[:1] == (B,) else v) for key, v in batch.items()}
            p = PlannerStep(model, sub, overlap=True, dropout=dropout, drop_seed=drop_seed * n_micro + k,
                            refresh_weights=(k == 0), zero_grads=(k == 0), grad_overwrite=(None if k == 0 else False),
                            share_side_streams=self.parts[0] if k > 0 else None)
            p.loss_scale = 1.0 / B
            self.parts.append(p)
        h = self.eng.handle
        # ONE weight-gradient stream and ONE d_txt stream for all micro-batches: in-order issue is what lets the later
        # micro-batches accumulate onto the first one's stores without atomics
        check(self.L.etp_planner_set_aux_stream(h, self.parts[0].aux), "set_aux_stream")
        check(self.L.etp_planner_set_aux2_stream(h, self.parts[0].aux2), "set_aux2_stream")
        check(self.L.etp_planner_set_lazy_join(h, 1), "set_lazy_join")
        self.mains = [None]
        for _ in range(1, n_micro):
            m = ctypes.c_void_p()
            check(self.L.etp_stream_create(ctypes.byref(m)), "stream_create")
            self.mains.append(m.value)

    @property
    def loss(self):
        return sum(p.loss for p in self.parts)

    @property
    def inp(self):
        return self.parts[0].inp

    def run_eager(self, stream: Optional[int] = None):
        s = stream if stream is not None else self.eng.stream()
        mains = [s] + self.mains[1:]
        L = self.L

        def fork_others(s_main, s_side):           # after micro-batch 0's weight casts and gradient memset are enqueued
            for m in mains[1:]:
                check(L.etp_stream_after(s_main, m), "fork")
                check(L.etp_stream_after(s_side, m), "fork")

        # issue order = order on the shared weight-gradient stream: all forwards (their text K/V projections run there), then
        # the navigation / panorama backward of every micro-batch, then the text backward of every micro-batch
        for k, (p, m) in enumerate(zip(self.parts, mains)):
            p.enqueue_fwd(m, True, after_prologue=fork_others if k == 0 else None)
        for p, m in zip(self.parts, mains):
            p.enqueue_bwd_main(m, join_pano=False)
        for p, m in zip(self.parts, mains):
            p.enqueue_txt_bwd(m)                   # joins this micro-batch's panorama branch and the weight-gradient stream
        for m in mains[1:]:
            check(L.etp_stream_after(m, s), "join")

    def close(self):
        torch.cuda.synchronize()
        for p in reversed(self.parts):
            p.close()
        for m in self.mains[1:]:
            self.L.etp_stream_destroy(m)
        self.mains = [None]
