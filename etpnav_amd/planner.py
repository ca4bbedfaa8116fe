"""MI355X-native drop-in for the reference planner module.

``GlocalTextPathNavCMT`` mirrors vlnce_baselines/models/etp/vilmodel_cmt.py:663-750:
same constructor contract (a config object with the attributes set by
vlnbert_init.py:32-59), same method names / argument order / return values
(``forward_txt``, ``forward_panorama``, ``forward_navigation``), and the same
state-dict names and shapes — but every parameter is a view
into ONE flat fp32 arena in HBM and all arithmetic runs in the hand-written HIP
kernels of libetpnav_hip.so through its C ABI.  There is no torch fallback: without
the shared object (or without a GPU) the compute methods raise.

Numerics modes (same kernels): ``dtype=torch.float32`` = parity mode (fp32 MFMA,
matches the fp32 reference to ~1e-5); ``dtype=torch.bfloat16`` = performance mode
(bf16 GEMM / attention operands, fp32 accumulation, fp32 residual stream, LayerNorm,
softmax statistics, logits and gradients — the MI355X counterpart of the reference's
fp16 autocast, ss_trainer_ETP.py:502).  Tensors crossing the API are fp32 in both modes.
Training mode (``model.train()``) turns on every dropout site of the reference (rates from the config, masks from a
counter-based generator that the backward recomputes); ``model.eval()`` makes them identity — see DESIGN.md §1.
"""
from __future__ import annotations

import ctypes
from types import SimpleNamespace
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import _lib
from ._lib import check, ptr


def _cfg_get(config, name, default=None):
    if isinstance(config, dict):
        return config.get(name, default)
    return getattr(config, name, default)


def make_c_config(config, dtype: torch.dtype) -> _lib.Config:
    """vis_config (vlnbert_init.py:32-59 + bert_config/*.json) -> etp_config."""
    c = _lib.Config()
    c.hidden = int(_cfg_get(config, "hidden_size", 768))
    c.heads = int(_cfg_get(config, "num_attention_heads", 12))
    c.inter = int(_cfg_get(config, "intermediate_size", 3072))
    c.n_l = int(_cfg_get(config, "num_l_layers", 9))
    c.n_p = int(_cfg_get(config, "num_pano_layers", 2))
    c.n_x = int(_cfg_get(config, "num_x_layers", 4))
    c.vocab = int(_cfg_get(config, "vocab_size", 30522))
    c.max_pos = int(_cfg_get(config, "max_position_embeddings", 512))
    c.type_vocab = int(_cfg_get(config, "type_vocab_size", 2))
    c.img_feat = int(_cfg_get(config, "image_feat_size", 512))
    c.dep_feat = int(_cfg_get(config, "depth_feat_size", 128))
    c.ang_feat = int(_cfg_get(config, "angle_feat_size", 4))
    c.max_steps = int(_cfg_get(config, "max_action_steps", 100))
    c.use_depth = 1 if _cfg_get(config, "use_depth_embedding", True) else 0
    c.use_sprels = 1 if _cfg_get(config, "graph_sprels", True) else 0
    c.ln_eps = float(_cfg_get(config, "layer_norm_eps", 1e-12))
    c.use_lang2visn = 1 if _cfg_get(config, "use_lang2visn_attn", False) else 0
    if dtype == torch.float32:
        c.dtype = _lib.ETP_F32
    elif dtype == torch.bfloat16:
        c.dtype = _lib.ETP_BF16
    else:
        raise ValueError("compute dtype must be torch.float32 or torch.bfloat16")
    if c.hidden % c.heads != 0:
        # same check / message as BertSelfAttention.__init__ (vilmodel_cmt.py:82-85)
        raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)"
                         % (c.hidden, c.heads))
    return c


class _Node(nn.Module):
    """Plain container used to reproduce the reference's module tree (state-dict names)."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("container module; call the planner's forward_* methods")


class Engine:
    """Owns the C planner handle, the flat arenas and the per-shape workspaces."""

    def __init__(self, cconf: _lib.Config, device: torch.device):
        self.L = _lib.lib()
        self.cconf = cconf
        self.device = device
        self.handle = self.L.etp_planner_create(ctypes.byref(cconf))
        if not self.handle:
            raise _lib.EtpError("etp_planner_create: " + self.L.etp_last_error().decode())
        n = self.L.etp_planner_param_count(self.handle)
        self.table = []
        info = _lib.ParamInfo()
        for i in range(n):
            check(self.L.etp_planner_param_info(self.handle, i, ctypes.byref(info)), "param_info")
            shape = tuple(int(info.shape[k]) for k in range(info.ndim))
            self.table.append((info.name.decode(), shape, int(info.offset)))
        self.total = int(self.L.etp_planner_arena_elems(self.handle))
        self.n_matrix = int(self.L.etp_planner_matrix_elems(self.handle))
        self.tdtype = torch.bfloat16 if cconf.dtype == _lib.ETP_BF16 else torch.float32
        self.params = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.grads = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.shadow = (torch.zeros(self.n_matrix, dtype=torch.bfloat16, device=device)
                       if cconf.dtype == _lib.ETP_BF16 else None)
        self._shadow_version = -1
        self.epoch = 0                  # bumped by optimizers that update the arena outside torch (FusedAdamW)
        self._ws: Dict[tuple, torch.Tensor] = {}
        self.param_views = None         # set by the module after a device move: parameters with their own version counters
        self.bind()

    def weights_version(self) -> int:
        """Changes whenever the fp32 masters change through torch.  Parameters created as views of the arena share its
        version counter; after ``module.to(device)`` they are re-pointed with ``p.data = ...``, which keeps each
        parameter's OWN counter, so in-place optimizer updates no longer bump the arena's -- the per-parameter counters
        are therefore part of the key."""
        v = self.params._version
        if self.param_views is not None:
            for p in self.param_views:
                v += p._version
        return v

    def bind(self):
        if self.device.type == "cuda":
            check(self.L.etp_planner_bind(self.handle, ptr(self.params), ptr(self.shadow), ptr(self.grads)), "bind")

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.L.etp_planner_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def require_gpu(self):
        if self.device.type != "cuda":
            raise _lib.EtpError("the ETPNav planner kernels need an MI355X (cuda/hip device); no CPU fallback exists")

    @staticmethod
    def stream() -> int:
        return torch.cuda.current_stream().cuda_stream

    def refresh_weights(self, force: bool = False):
        """bf16 mode: re-cast the GEMM weights when the fp32 masters changed (autocast's per-forward cast)."""
        if self.shadow is None:
            return
        v = self.weights_version()
        if force or v != self._shadow_version:
            check(self.L.etp_planner_refresh_weights(self.handle, self.stream()), "refresh_weights")
            self._shadow_version = v

    def mark_shadow_current(self):
        """Called by FusedAdamW: its kernel wrote the bf16 shadow together with the fp32 masters."""
        self._shadow_version = self.weights_version()
        self.epoch += 1

    def set_dropout(self, drop):
        """drop = None (eval) or (p_hidden, p_attn, p_head, p_env, seed); read by the next enqueued entry points."""
        if drop is None:
            drop = (0.0, 0.0, 0.0, 0.0, 0)
        ph, pa, pd, pe, seed = drop
        check(self.L.etp_planner_set_dropout(self.handle, ph, pa, pd, pe, seed), "set_dropout")

    def buf(self, nbytes: int) -> torch.Tensor:
        return torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)

    def ws(self, key: tuple, nbytes: int) -> torch.Tensor:
        t = self._ws.get(key)
        if t is None or t.numel() < nbytes:
            t = self.buf(nbytes)
            self._ws[key] = t
        return t


# ---- autograd bridges ------------------------------------------------------------------------
class _TxtFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, eng: Engine, drop, txt_ids, txt_masks):
        B, L = txt_ids.shape
        H = eng.cconf.hidden
        out = torch.empty(B, L, H, dtype=torch.float32, device=eng.device)
        stash = eng.buf(eng.L.etp_txt_stash_bytes(eng.handle, B, L))
        eng.set_dropout(drop)
        ctx.drop = drop
        check(eng.L.etp_txt_fwd(eng.handle, ptr(txt_ids), ptr(txt_masks), B, L, ptr(out), ptr(stash), eng.stream()),
              "etp_txt_fwd")
        ctx.eng, ctx.stash, ctx.dims = eng, stash, (B, L)
        ctx.save_for_backward(txt_ids, txt_masks)
        return out

    @staticmethod
    def backward(ctx, dout):
        eng, (B, L) = ctx.eng, ctx.dims
        txt_ids, txt_masks = ctx.saved_tensors
        dout = dout.float().contiguous()
        ws = eng.ws(("txt", B, L), eng.L.etp_txt_ws_bytes(eng.handle, B, L))
        eng.set_dropout(ctx.drop)
        check(eng.L.etp_txt_bwd(eng.handle, ptr(dout), ptr(txt_ids), ptr(txt_masks), B, L, ptr(ctx.stash), ptr(ws),
                                eng.stream()), "etp_txt_bwd")
        return None, None, None, None, None


class _PanoFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, eng: Engine, drop, rgb, dep, loc, nav_types, view_lens):
        B, V, _ = rgb.shape
        H = eng.cconf.hidden
        eng.set_dropout(drop)
        ctx.drop = drop
        out = torch.empty(B, V, H, dtype=torch.float32, device=eng.device)
        masks = torch.empty(B, V, dtype=torch.bool, device=eng.device)
        stash = eng.buf(eng.L.etp_pano_stash_bytes(eng.handle, B, V))
        check(eng.L.etp_pano_fwd(eng.handle, ptr(rgb), ptr(dep), ptr(loc), ptr(nav_types), ptr(view_lens), B, V, ptr(out),
                                 ptr(masks), ptr(stash), eng.stream()), "etp_pano_fwd")
        ctx.eng, ctx.stash, ctx.dims = eng, stash, (B, V)
        ctx.need_rgb_grad = rgb.requires_grad
        ctx.save_for_backward(rgb, dep, loc, nav_types)
        ctx.mark_non_differentiable(masks)
        return out, masks

    @staticmethod
    def backward(ctx, dout, _dmask):
        eng, (B, V) = ctx.eng, ctx.dims
        rgb, dep, loc, nav_types = ctx.saved_tensors
        dout = dout.float().contiguous()
        d_rgb = torch.empty(B, V, eng.cconf.img_feat, dtype=torch.float32, device=eng.device) if ctx.need_rgb_grad else None
        ws = eng.ws(("pano", B, V), eng.L.etp_pano_ws_bytes(eng.handle, B, V))
        eng.set_dropout(ctx.drop)
        check(eng.L.etp_pano_bwd(eng.handle, ptr(dout), ptr(rgb), ptr(dep), ptr(loc), ptr(nav_types), B, V, ptr(d_rgb),
                                 ptr(ctx.stash), ptr(ws), eng.stream()), "etp_pano_bwd")
        # a PlannerStep sharing this planner may have switched on lazy joins of the weight-gradient stream: autograd
        # consumers read .grad right after backward, so join here (no-op without an aux stream)
        check(eng.L.etp_planner_join_aux(eng.handle, eng.stream()), "join_aux")
        return None, None, None, d_rgb, None, None, None, None


class _NavFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, eng: Engine, drop, txt_embeds, txt_masks, step_ids, img_fts, pos_fts, gmasks, visited, dists):
        B, L, H = txt_embeds.shape
        eng.set_dropout(drop)
        ctx.drop = drop
        G = step_ids.shape[1]
        out = torch.empty(B, G, H, dtype=torch.float32, device=eng.device)
        logits = torch.empty(B, G, dtype=torch.float32, device=eng.device)
        stash = eng.buf(eng.L.etp_nav_stash_bytes(eng.handle, B, L, G))
        check(eng.L.etp_nav_fwd(eng.handle, ptr(txt_embeds), ptr(txt_masks), ptr(step_ids), ptr(img_fts), ptr(pos_fts),
                                ptr(gmasks), ptr(visited), ptr(dists), B, L, G, ptr(out), ptr(logits), ptr(stash),
                                eng.stream()), "etp_nav_fwd")
        ctx.eng, ctx.stash, ctx.dims = eng, stash, (B, L, G)
        ctx.save_for_backward(txt_embeds, txt_masks, step_ids, pos_fts, gmasks, visited, dists)
        return out, logits

    @staticmethod
    def backward(ctx, d_out, d_logits):
        eng, (B, L, G) = ctx.eng, ctx.dims
        txt_embeds, txt_masks, step_ids, pos_fts, gmasks, visited, dists = ctx.saved_tensors
        H = eng.cconf.hidden
        d_out = d_out.float().contiguous() if d_out is not None else None
        d_logits = d_logits.float().contiguous() if d_logits is not None else None
        d_txt = torch.empty(B, L, H, dtype=torch.float32, device=eng.device)
        d_img = torch.empty(B, G, H, dtype=torch.float32, device=eng.device)
        ws = eng.ws(("nav", B, L, G), eng.L.etp_nav_ws_bytes(eng.handle, B, L, G))
        eng.set_dropout(ctx.drop)
        check(eng.L.etp_nav_bwd(eng.handle, ptr(d_out), ptr(d_logits), ptr(txt_embeds), ptr(txt_masks),
                                ptr(step_ids), ptr(pos_fts), ptr(gmasks), ptr(visited), ptr(dists), B, L, G, ptr(d_txt),
                                ptr(d_img), ptr(ctx.stash), ptr(ws), eng.stream()), "etp_nav_bwd")
        # a PlannerStep sharing this planner may have switched on lazy joins of the weight-gradient stream: autograd
        # consumers read .grad right after backward, so join here (no-op without an aux stream)
        check(eng.L.etp_planner_join_aux(eng.handle, eng.stream()), "join_aux")
        return None, None, None, d_txt, None, None, d_img, None, None, None, None


class _NavKVFn(torch.autograd.Function):
    """txt_embeds -> K|V of every x-layer (etp_nav_kv_fwd).  Output: one tensor [n_x, B*L, 2H] in the operand dtype that
    the per-step navigation calls consume; autograd sums their d_kv and this backward projects the sum back once."""

    @staticmethod
    def forward(ctx, anchor, eng: Engine, txt_embeds):
        B, L, H = txt_embeds.shape
        cache = eng.buf(eng.L.etp_nav_kv_bytes(eng.handle, B, L))
        check(eng.L.etp_nav_kv_fwd(eng.handle, ptr(txt_embeds), B, L, ptr(cache), eng.stream()), "etp_nav_kv_fwd")
        n_x = eng.cconf.n_x
        # the K|V blocks are contiguous inside the cache buffer: expose them as a typed view (no copy)
        es = 2 if eng.tdtype == torch.bfloat16 else 4
        off = int(eng.L.etp_nav_kv_offset(eng.handle, B, L))
        kv = cache[off:off + n_x * B * L * 2 * H * es].view(eng.tdtype).view(n_x, B * L, 2 * H)
        ctx.eng, ctx.cache, ctx.dims = eng, cache, (B, L)
        ctx.save_for_backward(txt_embeds)
        return kv                        # a view into `cache`: its storage keeps the whole buffer (bf16 text included) alive

    @staticmethod
    def backward(ctx, d_kv):
        eng, (B, L) = ctx.eng, ctx.dims
        (txt_embeds,) = ctx.saved_tensors
        d_kv = d_kv.to(eng.tdtype).contiguous()
        d_txt = torch.empty(B, L, eng.cconf.hidden, dtype=torch.float32, device=eng.device)
        check(eng.L.etp_nav_kv_bwd(eng.handle, ptr(txt_embeds), ptr(d_kv), B, L, ptr(ctx.cache), ptr(d_txt), eng.stream()),
              "etp_nav_kv_bwd")
        return None, None, d_txt


class _NavKVStepsFn(torch.autograd.Function):
    """K|V of Bt instructions -> the cache of T stacked rollout steps (episode t*Bt + b reads instruction b):
    etp_nav_kv_repeat forward, etp_nav_kv_sum_steps backward (the sum over the steps that the reference's shared txt_embeds
    tensor accumulates through autograd, ss_trainer_ETP.py:819-822,1055)."""

    @staticmethod
    def forward(ctx, eng: Engine, kv, Bt: int, L: int, T: int):
        n_x, _, H2 = kv.shape
        es = kv.element_size()
        src = ctypes.c_void_p(kv.data_ptr() - int(eng.L.etp_nav_kv_offset(eng.handle, Bt, L)))
        cache = eng.buf(eng.L.etp_nav_kv_bytes(eng.handle, T * Bt, L))
        check(eng.L.etp_nav_kv_repeat(eng.handle, src, Bt, L, T, ptr(cache), eng.stream()), "etp_nav_kv_repeat")
        off = int(eng.L.etp_nav_kv_offset(eng.handle, T * Bt, L))
        out = cache[off:off + n_x * T * Bt * L * H2 * es].view(kv.dtype).view(n_x, T * Bt * L, H2)
        ctx.eng, ctx.dims, ctx.kv_shape = eng, (Bt, L, T), kv.shape
        return out                       # a view into `cache` (keeps the buffer alive), laid out for etp_nav_fwd_kv with B = T*Bt

    @staticmethod
    def backward(ctx, d_kv_steps):
        eng, (Bt, L, T) = ctx.eng, ctx.dims
        d_kv_steps = d_kv_steps.to(eng.tdtype).contiguous()
        d_kv = torch.empty(ctx.kv_shape, dtype=eng.tdtype, device=eng.device)
        check(eng.L.etp_nav_kv_sum_steps(eng.handle, ptr(d_kv_steps), Bt, L, T, ptr(d_kv), eng.stream()), "etp_nav_kv_sum_steps")
        return None, d_kv, None, None, None


class _NavCachedFn(torch.autograd.Function):
    """forward_navigation on the text K/V cache.  steps_T > 1 (batched rollout, round 6): `kv` / `txt_masks` hold the Bt = B / steps_T
    instructions ONCE and stacked episode e reads instruction e % Bt inside the cross-attention kernels (etp_nav_fwd_kv_steps /
    etp_nav_bwd_kv_steps) -- no replicated cache; the backward sums d_kv over the steps (etp_nav_kv_sum_steps), the sum the reference's
    shared txt_embeds tensor accumulates through autograd (ss_trainer_ETP.py:819-822,1055)."""

    @staticmethod
    def forward(ctx, anchor, eng: Engine, drop, kv, L, txt_masks, step_ids, img_fts, pos_fts, gmasks, visited, dists, steps_T=1):
        B, G = step_ids.shape
        H = eng.cconf.hidden
        Bt = B // steps_T
        cache = ctypes.c_void_p(kv.data_ptr() - int(eng.L.etp_nav_kv_offset(eng.handle, Bt, L)))   # base of the cache buffer
        eng.set_dropout(drop)
        ctx.drop = drop
        out = torch.empty(B, G, H, dtype=torch.float32, device=eng.device)
        logits = torch.empty(B, G, dtype=torch.float32, device=eng.device)
        stash = eng.buf(eng.L.etp_nav_stash_bytes(eng.handle, B, L, G))
        if steps_T > 1:
            check(eng.L.etp_nav_fwd_kv_steps(eng.handle, cache, ptr(txt_masks), ptr(step_ids), ptr(img_fts), ptr(pos_fts), ptr(gmasks),
                                             ptr(visited), ptr(dists), B, L, G, Bt, ptr(out), ptr(logits), ptr(stash), eng.stream()),
                  "etp_nav_fwd_kv_steps")
        else:
            check(eng.L.etp_nav_fwd_kv(eng.handle, cache, ptr(txt_masks), ptr(step_ids), ptr(img_fts), ptr(pos_fts), ptr(gmasks),
                                       ptr(visited), ptr(dists), B, L, G, ptr(out), ptr(logits), ptr(stash), eng.stream()),
                  "etp_nav_fwd_kv")
        ctx.eng, ctx.stash, ctx.dims, ctx.cache, ctx.kv_shape, ctx.steps_T = eng, stash, (B, L, G), cache, kv.shape, steps_T
        ctx.save_for_backward(txt_masks, step_ids, pos_fts, gmasks, visited, dists, kv)
        return out, logits

    @staticmethod
    def backward(ctx, d_out, d_logits):
        eng, (B, L, G), T = ctx.eng, ctx.dims, ctx.steps_T
        txt_masks, step_ids, pos_fts, gmasks, visited, dists, _kv = ctx.saved_tensors
        H = eng.cconf.hidden
        d_out = d_out.float().contiguous() if d_out is not None else None
        d_logits = d_logits.float().contiguous() if d_logits is not None else None
        d_kv = torch.empty(ctx.kv_shape, dtype=eng.tdtype, device=eng.device)
        d_img = torch.empty(B, G, H, dtype=torch.float32, device=eng.device)
        ws = eng.ws(("nav", B, L, G), eng.L.etp_nav_ws_bytes(eng.handle, B, L, G))
        eng.set_dropout(ctx.drop)
        if T > 1:
            n_x, _, H2 = ctx.kv_shape
            d_kv_steps = torch.empty(n_x, B * L, H2, dtype=eng.tdtype, device=eng.device)
            check(eng.L.etp_nav_bwd_kv_steps(eng.handle, ptr(d_out), ptr(d_logits), ctx.cache, ptr(txt_masks), ptr(step_ids),
                                             ptr(pos_fts), ptr(gmasks), ptr(visited), ptr(dists), B, L, G, B // T, ptr(d_kv_steps),
                                             ptr(d_img), ptr(ctx.stash), ptr(ws), eng.stream()), "etp_nav_bwd_kv_steps")
            check(eng.L.etp_nav_kv_sum_steps(eng.handle, ptr(d_kv_steps), B // T, L, T, ptr(d_kv), eng.stream()), "etp_nav_kv_sum_steps")
        else:
            check(eng.L.etp_nav_bwd_kv(eng.handle, ptr(d_out), ptr(d_logits), ctx.cache, ptr(txt_masks), ptr(step_ids),
                                       ptr(pos_fts), ptr(gmasks), ptr(visited), ptr(dists), B, L, G, ptr(d_kv), ptr(d_img),
                                       ptr(ctx.stash), ptr(ws), eng.stream()), "etp_nav_bwd_kv")
        # a PlannerStep sharing this planner may have switched on lazy joins of the weight-gradient stream: autograd
        # consumers read .grad right after backward, so join here (no-op without an aux stream)
        check(eng.L.etp_planner_join_aux(eng.handle, eng.stream()), "join_aux")
        return None, None, None, d_kv, None, None, None, d_img, None, None, None, None, None


class GlocalTextPathNavCMT(nn.Module):
    """Drop-in for vilmodel_cmt.py:663 ``GlocalTextPathNavCMT`` (see module docstring)."""

    def __init__(self, config, dtype: torch.dtype = torch.bfloat16, device=None):
        super().__init__()
        self.config = config
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self._engine = Engine(make_c_config(config, dtype), torch.device(device))
        self.compute_dtype = self._engine.tdtype
        self._build_tree()
        self.init_weights()
        # vilmodel_cmt.py:675-682
        if _cfg_get(config, "fix_lang_embedding", False):
            for k, v in self.named_parameters():
                if k.startswith("embeddings.") or k.startswith("lang_encoder."):
                    v.requires_grad = False
                    v.grad = None             # a frozen parameter has no .grad (the reference's optimizers skip it on that)
        if _cfg_get(config, "fix_pano_embedding", False):
            for k, v in self.named_parameters():
                if k.startswith("img_embeddings."):
                    v.requires_grad = False
                    v.grad = None
        self._anchor = torch.zeros(1, device=self._engine.device, requires_grad=True)
        # training-mode dropout (nn.Module.training, as the reference's nn.Dropout layers): rates from the config,
        # masks from a counter-based generator keyed by (seed, call counter, site, element)
        # text K/V cache across rollout steps; off = the reference's per-step re-projection
        self.cache_text_kv = False
        self.batch_steps_kv = True        # forward_navigation_steps: project the text keys/values once, not T times
        self.kv_indirection = True        # ... and let the cross-attention kernels read instruction e % B (no replicated cache; bf16, axes <= 128)
        self._kv_cache = None
        self.drop_env_prob = 0.0          # >0 fuses the policy's drop_env (Policy_ViewSelection_ETP.py:102,345) into forward_panorama
        self._drop_seed = int(torch.initial_seed()) & 0xFFFFFFFF
        self._drop_calls = 0

    def seed_dropout(self, seed: int):
        """Restart the dropout mask stream (deterministic training runs, tests)."""
        self._drop_seed = int(seed) & 0xFFFFFFFF
        self._drop_calls = 0

    def _dropout(self):
        if not self.training:
            return None
        c = self.config
        ph = float(_cfg_get(c, "hidden_dropout_prob", 0.1))
        pa = float(_cfg_get(c, "attention_probs_dropout_prob", 0.1))
        pd = float(_cfg_get(c, "pred_head_dropout_prob", 0.1))
        pe = float(self.drop_env_prob)
        if ph == 0.0 and pa == 0.0 and pd == 0.0 and pe == 0.0:
            return None
        self._drop_calls += 1
        return (ph, pa, pd, pe, (self._drop_seed << 32) | (self._drop_calls & 0xFFFFFFFF))

    # ---- parameter tree over the flat arena ---------------------------------------------------
    def _build_tree(self):
        eng = self._engine
        self._views: List[tuple] = []
        for name, shape, off in eng.table:
            n = 1
            for s in shape:
                n *= s
            p = nn.Parameter(eng.params[off:off + n].view(shape), requires_grad=True)
            parts = name.split(".")
            mod = self
            for part in parts[:-1]:
                if not hasattr(mod, part):
                    mod.add_module(part, _Node())
                mod = getattr(mod, part)
            mod.register_parameter(parts[-1], p)
            self._views.append((p, off, n, shape))
        self._attach_grads(force=True)

    def _attach_grads(self, force: bool = False):
        """Point every ``param.grad`` at its slice of the flat gradient arena.  If an optimizer reset grads to
        None (torch's default ``zero_grad(set_to_none=True)``, ss_trainer_ETP.py:499) the arena is zeroed first."""
        eng = self._engine
        lost = any(p.grad is None for p, _, _, _ in self._views if p.requires_grad)
        if not (force or lost):
            return
        if lost and not force:
            eng.grads.zero_()
        for p, off, n, shape in self._views:
            if p.requires_grad:
                p.grad = eng.grads[off:off + n].view(shape)

    def zero_grad(self, set_to_none: bool = False):
        self._engine.grads.zero_()
        self._attach_grads(force=True)

    @property
    def flat_params(self) -> torch.Tensor:
        return self._engine.params

    @property
    def flat_grads(self) -> torch.Tensor:
        return self._engine.grads

    def init_weights(self, seed: Optional[int] = None):
        """BERT-style init (normal(0,0.02) weights, zero biases, LN = (1,0)) as the reference ctor's
        ``self.init_weights()`` (vilmodel_cmt.py:673)."""
        g = None
        if seed is not None:
            g = torch.Generator(device="cpu").manual_seed(seed)
        with torch.no_grad():
            for name, p in self.named_parameters():
                is_ln = ("LayerNorm" in name or "layer_norm" in name or ".norm" in name
                         or "gmap_pos_embeddings.1" in name or "global_sap_head.net.2" in name)
                if is_ln:
                    p.fill_(1.0 if name.endswith("weight") else 0.0)
                elif name.endswith("bias"):
                    p.zero_()
                else:
                    p.copy_(torch.randn(p.shape, generator=g) * 0.02)

    def _apply(self, fn, recurse=True):
        """``.to(device)`` / ``.cuda()``: move the arenas and rebuild the parameter views (dtype casts are refused:
        the masters stay fp32, the compute dtype is chosen at construction)."""
        eng = self._engine
        new = fn(eng.params)
        if new.dtype != torch.float32:
            raise TypeError("the planner's master parameters stay fp32; choose the compute dtype at construction")
        if new.device != eng.device:
            with torch.no_grad():
                eng.device = new.device
                old_params = eng.params
                eng.params = new.contiguous()
                eng.grads = torch.zeros_like(eng.params)
                if eng.shadow is not None:
                    eng.shadow = torch.zeros(eng.n_matrix, dtype=torch.bfloat16, device=new.device)
                eng._shadow_version = -1
                eng._ws.clear()
                for p, off, n, shape in self._views:
                    p.data = eng.params[off:off + n].view(shape)
                eng.param_views = [v[0] for v in self._views]      # `.data =` keeps each parameter's own version counter
                del old_params
                self._anchor = torch.zeros(1, device=eng.device, requires_grad=True)
                self._attach_grads(force=True)
                eng.bind()
        return self

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        r = super().load_state_dict(state_dict, strict=strict, assign=False)
        self._engine._shadow_version = -1
        return r

    # ---- the three planner entry points -------------------------------------------------------
    def _prep(self):
        eng = self._engine
        eng.require_gpu()
        self._attach_grads()
        eng.refresh_weights()
        return eng

    def forward_txt(self, txt_ids, txt_masks):
        """vilmodel_cmt.py:684-688.  txt_ids [B,L] int64, txt_masks [B,L] bool -> [B,L,H]."""
        eng = self._prep()
        out = _TxtFn.apply(self._anchor, eng, self._dropout(), txt_ids.contiguous(), txt_masks.to(torch.bool).contiguous())
        if _cfg_get(self.config, "fix_lang_embedding", False):
            out = out.detach()   # LanguageEncoder.forward :431-432
        return out

    def forward_panorama(self, rgb_fts, dep_fts, loc_fts, nav_types, view_lens):
        """vilmodel_cmt.py:690-719 -> (pano_embeds [B,V,H], pano_masks [B,V] bool)."""
        eng = self._prep()
        dep = dep_fts.float().contiguous() if dep_fts is not None else None
        # fix_pano_embedding AND fix_lang_embedding: nothing behind this branch requires a gradient (token_type_embeddings(1),
        # :706-708, is frozen with the language side), so unless the features do, autograd must not enter it
        c = self.config
        if any(t is not None and t.requires_grad for t in (dep_fts, loc_fts)):
            # the explicit backward produces d rgb_fts only (etp_pano_bwd); the reference feeds precomputed depth / angle features
            # that never require a gradient (ss_trainer_ETP.py:836-839).  Refuse instead of returning None silently.
            raise NotImplementedError("forward_panorama: gradients w.r.t. dep_fts / loc_fts are not provided by etp_pano_bwd "
                                      "(only d rgb_fts); detach them")
        feats_live = rgb_fts.requires_grad
        live = feats_live or not (_cfg_get(c, "fix_pano_embedding", False) and _cfg_get(c, "fix_lang_embedding", False))
        anchor = self._anchor if live else self._anchor.detach()
        return _PanoFn.apply(anchor, eng, self._dropout(), rgb_fts.float().contiguous(), dep, loc_fts.float().contiguous(),
                             nav_types.long().contiguous(), view_lens.long().contiguous())

    def _text_kv(self, eng, txt_embeds):
        """K|V projections of `txt_embeds` for all x-layers, computed once per distinct (tensor, weights) pair.  The
        rollout passes the SAME txt_embeds tensor at every step (ss_trainer_ETP.py:801-805 computes it once, :878 reuses
        it), so identity + version of the tensor and of the parameter arena is the cache key; anything else (a sliced or
        re-computed tensor, an optimizer step) misses and re-projects -- results are identical either way."""
        key = (id(txt_embeds), txt_embeds._version, tuple(txt_embeds.shape), eng.weights_version(), eng.epoch,
               torch.is_grad_enabled() and txt_embeds.requires_grad)
        hit = self._kv_cache
        if hit is not None and hit[0] == key and hit[1]() is txt_embeds:
            return hit[2]
        import weakref
        x = txt_embeds.to(torch.float32).contiguous()
        kv = _NavKVFn.apply(self._anchor, eng, x)
        self._kv_cache = (key, weakref.ref(txt_embeds), kv)
        return kv

    def forward_navigation(self, txt_embeds, txt_masks, gmap_vpids, gmap_step_ids, gmap_img_fts, gmap_pos_fts,
                           gmap_masks, gmap_visited_masks, gmap_pair_dists):
        """vilmodel_cmt.py:721-750 (gmap_vpids is ignored, as in the reference)."""
        eng = self._prep()
        t = torch.float32
        dists = gmap_pair_dists.float().contiguous() if gmap_pair_dists is not None else None
        if self.cache_text_kv:
            kv = self._text_kv(eng, txt_embeds)
            embeds, logits = _NavCachedFn.apply(self._anchor, eng, self._dropout(), kv, txt_embeds.shape[1],
                                                txt_masks.to(torch.bool).contiguous(), gmap_step_ids.long().contiguous(),
                                                gmap_img_fts.to(t).contiguous(), gmap_pos_fts.float().contiguous(),
                                                gmap_masks.to(torch.bool).contiguous(),
                                                gmap_visited_masks.to(torch.bool).contiguous(), dists)
            return {"gmap_embeds": embeds, "global_logits": logits}
        embeds, logits = _NavFn.apply(self._anchor, eng, self._dropout(), txt_embeds.to(t).contiguous(),
                                      txt_masks.to(torch.bool).contiguous(), gmap_step_ids.long().contiguous(),
                                      gmap_img_fts.to(t).contiguous(), gmap_pos_fts.float().contiguous(),
                                      gmap_masks.to(torch.bool).contiguous(),
                                      gmap_visited_masks.to(torch.bool).contiguous(), dists)
        return {"gmap_embeds": embeds, "global_logits": logits}

    def forward_navigation_steps(self, txt_embeds, txt_masks, steps):
        """T navigation steps of one rollout in ONE batched call (second half).

        The trainer calls forward_navigation once per rollout step on the same instruction embeddings
        (ss_trainer_ETP.py:819-822,878) and sums the step losses before a single backward (:1055).  A step's node axis holds
        B * G = a few hundred rows -- every kernel of such a call is latency-bound on an MI355X, forward and
        backward.  When the inputs of several steps are known together (teacher-forced training, or replaying a finished
        rollout for its backward), the steps are independent given the text, so they run as one (T * B)-episode batch: the
        per-step graphs are padded to the largest node count (padded nodes are masked: no attention weight, -inf logit),
        stacked along the batch axis, and the text is repeated T times -- autograd's repeat backward is the sum over the steps
        that the reference's shared txt_embeds tensor produces.  One launch sequence with T times the rows instead of T
        sequences; results and gradients equal the per-step calls (tests/test_baseline_shapes_gpu.py).

        steps: list of dicts with gmap_step_ids [B,G_t], gmap_img_fts [B,G_t,H], gmap_pos_fts [B,G_t,7], gmap_masks [B,G_t],
        gmap_visited_masks [B,G_t], gmap_pair_dists [B,G_t,G_t].  Returns a list of {'gmap_embeds', 'global_logits'} per step,
        sliced back to G_t.

        Text keys/values (self.batch_steps_kv, default on): the K|V projections of the B instructions are computed ONCE
        (etp_nav_kv_fwd, shared with cache_text_kv's per-step calls), replicated for the T stacked steps by a copy
        (etp_nav_kv_repeat; round 6: not even that where the register-resident attention kernels run -- bf16, both axes <= 128 --
        the stacked episode e reads instruction e % B inside the kernels, etp_nav_fwd_kv_steps) and their gradient is summed over
        the steps (etp_nav_kv_sum_steps) before ONE projection back to
        the text (etp_nav_kv_bwd) -- instead of projecting T * B * L stacked text rows forward and backward in every x-layer,
        which is the re-projection of vilmodel_cmt.py:326-328 the reference repeats per step.  batch_steps_kv = False keeps
        the stacked re-projection (same results up to the rounding of the summed bf16 key/value gradients)."""
        T = len(steps)
        if T == 0:
            return []
        B = txt_embeds.shape[0]
        Gs = [int(st["gmap_step_ids"].shape[1]) for st in steps]
        Gm = max(Gs)

        def pad(x, dims, value=0):          # pad the node axes `dims` of x to Gm
            for d in dims:
                g = x.shape[d]
                if g < Gm:
                    shape = list(x.shape)
                    shape[d] = Gm - g
                    x = torch.cat([x, x.new_full(shape, value)], dim=d)
            return x

        cat = lambda key, dims, value=0: torch.cat([pad(st[key], dims, value) for st in steps], dim=0)
        if self.batch_steps_kv and int(_cfg_get(self.config, "num_x_layers", 4)) > 0:
            eng = self._prep()
            t = torch.float32
            L = txt_embeds.shape[1]
            kv = self._text_kv(eng, txt_embeds)
            # bf16 with both axes <= 128 (every R2R-CE shape): the cross-attention kernels read instruction e % B themselves; otherwise
            # (fp32 parity mode, RxR's 512-token instructions) the cache is replicated for the stacked steps (etp_nav_kv_repeat)
            indirect = T > 1 and eng.tdtype == torch.bfloat16 and L <= 128 and Gm <= 128 and self.kv_indirection
            if indirect:
                kv_in, masks_in, steps_T = kv, txt_masks.to(torch.bool).contiguous(), T
            else:
                kv_in, masks_in, steps_T = _NavKVStepsFn.apply(eng, kv, B, L, T), txt_masks.to(torch.bool).repeat(T, 1).contiguous(), 1
            embeds, logits = _NavCachedFn.apply(self._anchor, eng, self._dropout(), kv_in, L, masks_in,
                                                cat("gmap_step_ids", (1,)).long().contiguous(),
                                                cat("gmap_img_fts", (1,)).to(t).contiguous(),
                                                cat("gmap_pos_fts", (1,)).float().contiguous(),
                                                cat("gmap_masks", (1,), False).to(torch.bool).contiguous(),
                                                cat("gmap_visited_masks", (1,), False).to(torch.bool).contiguous(),
                                                cat("gmap_pair_dists", (1, 2)).float().contiguous(), steps_T)
            out = {"gmap_embeds": embeds, "global_logits": logits}
        else:
            out = self.forward_navigation(txt_embeds.repeat(T, 1, 1), txt_masks.repeat(T, 1), None,
                                          cat("gmap_step_ids", (1,)), cat("gmap_img_fts", (1,)), cat("gmap_pos_fts", (1,)),
                                          cat("gmap_masks", (1,), False), cat("gmap_visited_masks", (1,), False),
                                          cat("gmap_pair_dists", (1, 2)))
        res = []
        for t, g in enumerate(Gs):
            res.append({"gmap_embeds": out["gmap_embeds"][t * B:(t + 1) * B, :g],
                        "global_logits": out["global_logits"][t * B:(t + 1) * B, :g]})
        return res

    def forward(self, mode, batch, **kwargs):
        # the reference's own forward (vilmodel_cmt.py:752-771) is dead code calling non-existent methods
        raise NotImplementedError("use forward_txt / forward_panorama / forward_navigation")


def default_config(task_type: str = "r2r", **overrides) -> SimpleNamespace:
    """The vis_config of vlnbert_init.py:32-59 without HF: bert-base-uncased / xlm-roberta-base JSON values."""
    cfg = dict(hidden_size=768, num_attention_heads=12, intermediate_size=3072, hidden_act="gelu",
               hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
               vocab_size=30522, max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12,
               max_action_steps=100, image_feat_size=512, use_depth_embedding=True, depth_feat_size=128,
               angle_feat_size=4, num_l_layers=9, num_pano_layers=2, num_x_layers=4, graph_sprels=True,
               glocal_fuse="global", fix_lang_embedding=False, fix_pano_embedding=False, update_lang_bert=True,
               output_attentions=True, pred_head_dropout_prob=0.1, use_lang2visn_attn=False)
    if task_type == "rxr":
        cfg.update(vocab_size=250002, max_position_embeddings=514, type_vocab_size=2, layer_norm_eps=1e-5)
    elif task_type != "r2r":
        raise ValueError("task_type must be 'r2r' or 'rxr'")
    cfg.update(overrides)
    return SimpleNamespace(**cfg)
