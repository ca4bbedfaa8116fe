"""Whole-step roofline of the planner:  t_roof = sum_k max(flops_k / peak_mfma, bytes_k / peak_hbm)  over the
kernels of ONE fwd+bwd step, each kernel counted with its ALGORITHMIC work: 2*M*N*K per product, one read of every input and
one write of every output per fused kernel (bf16 operands / activations, fp32 residual stream and weight gradients — the
numerics contract of DESIGN.md §1).  The kernel list mirrors csrc/planner.hip's launch sequence (forward_txt /
forward_panorama / forward_navigation and their backward entry points: vilmodel_cmt.py:684-750) — it is derived from the
tensor shapes, not from timings, so `roofline_step.frac = t_roof / measured` says how far the whole step is from the
machine, independent of how the work is cut into launches.

Peaks: MI355X_MICROARCH.md — 2.5 PFLOP/s dense bf16 MFMA, 8 TB/s HBM3E.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

PEAK_BF16_FLOPS = 2.5e15
PEAK_HBM_BPS = 8.0e12

Kernel = Tuple[str, float, float]          # (class, flops, bytes)


def _gemm(M, N, K, a=2, b=2, c=2, extra=0.0) -> Kernel:
    return ("gemm", 2.0 * M * N * K, float(M * K * a + N * K * b + M * N * c + extra))


def _rows(name, M, H, reads, writes) -> Kernel:
    """row kernel (LayerNorm, embedding fuse, ...): `reads`/`writes` = bytes per element moved"""
    return (name, 0.0, float(M * H * (reads + writes)))


def _attn_fwd(B, nh, Lq, Lk) -> Kernel:
    """the probabilities never reach HBM (csrc/attn_rows.hip, streaming kernels of csrc/attn.hip): the forward leaves one
    fp32 log-sum-exp per query row and the backward recomputes P on the matrix cores"""
    dh = 64
    fl = 2.0 * B * nh * Lq * Lk * dh * 2                       # QK^T and PV
    by = B * nh * ((Lq + 2 * Lk) * dh * 2 + Lq * dh * 2 + Lq * 4)        # q,k,v in; ctx, lse out
    return ("attn", fl, float(by))


def _attn_bwd(B, nh, Lq, Lk) -> Kernel:
    dh = 64
    fl = 2.0 * B * nh * Lq * Lk * dh * 4                       # dP, dV, dQ, dK (the recomputation of QK^T is not counted)
    by = B * nh * ((Lq + 2 * Lk) * dh * 2 + Lq * dh * 2 + Lq * 4 + (Lq + 2 * Lk) * dh * 2)   # q,k,v,dO,lse in; dq,dk,dv out
    return ("attn", fl, float(by))


def _post_ln_block_fwd(M, H, I, B, nh, L) -> List[Kernel]:
    """BertAttention + BertIntermediate/BertOutput on M = B*L rows (vilmodel_cmt.py:103-193)"""
    return [_gemm(M, 3 * H, H), _attn_fwd(B, nh, L, L),
            _gemm(M, H, H, c=4, extra=M * H * 4),               # out-proj + fp32 residual -> fp32 stream
            _rows("ln", M, H, 4, 4 + 2),                        # LN: fp32 in, fp32 + bf16 out
            _gemm(M, I, H, extra=M * I * 2),                    # FFN up (+ saved pre-activation)
            _gemm(M, H, I, c=4, extra=M * H * 4), _rows("ln", M, H, 4, 4 + 2)]


def _post_ln_block_bwd(M, H, I, B, nh, L) -> List[Kernel]:
    k = [_rows("ln", M, H, 8, 4 + 2),                           # LN bwd: dy, s in; dx fp32 + bf16 out
         _gemm(M, I, H, extra=M * I * 2),                       # dgrad FFN down (GELU' reads z)
         _gemm(M, H, I, c=4, extra=M * H * 4),                  # dgrad FFN up + residual gradient
         _gemm(H, I, M, c=4), _gemm(I, H, M, c=4),              # the two FFN weight gradients (fp32 out)
         _rows("ln", M, H, 8, 4 + 2),
         _gemm(M, H, H), _attn_bwd(B, nh, L, L),
         _gemm(M, H, 3 * H, c=4, extra=M * H * 4),
         _gemm(H, H, M, c=4), _gemm(3 * H, H, M, c=4)]
    return k


def planner_step_kernels(cfg, B: int, L: int, V: int, G: int, Bp: int = None) -> List[Kernel]:
    """cfg: object with hidden_size, intermediate_size, num_attention_heads, num_{l,pano,x}_layers, image_feat_size,
    depth_feat_size.  Bp = number of panoramas (B for the rollout step, B*T for the SAP pre-training step)."""
    H, I, nh = cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads
    Fi, Fd = cfg.image_feat_size, cfg.depth_feat_size
    Bp = B if Bp is None else Bp
    Mt, Mp, Mg = B * L, Bp * V, B * G
    ks: List[Kernel] = []
    # weight shadow refresh (fp32 -> bf16) and zeroing of the vector / table tail of the gradient arena
    n_mat = cfg.num_l_layers * (4 * H * H + 2 * H * I) + cfg.num_pano_layers * (4 * H * H + 2 * H * I) + H * (Fi + Fd) \
        + cfg.num_x_layers * (8 * H * H + 2 * H * I) + H * H
    ks.append(("cast", 0.0, n_mat * 6.0))
    # ---- text: embedding + 9 post-LN layers, forward and backward ----
    ks.append(_rows("embed", Mt, H, 4, 4 + 2))
    for _ in range(cfg.num_l_layers):
        ks += _post_ln_block_fwd(Mt, H, I, B, nh, L)
        ks += _post_ln_block_bwd(Mt, H, I, B, nh, L)
    ks.append(_rows("embed", Mt, H, 4 + 4, 4))                  # embedding backward (row-sparse table gradient)
    # ---- panorama: view-embedding fuse + pre-LN layers ----
    ks += [("cast", 0.0, Mp * (Fi + Fd) * 6.0), _gemm(Mp, H, Fi), _gemm(Mp, H, Fd), _rows("embed", Mp, H, 4, 4)]
    for _ in range(cfg.num_pano_layers):
        ks += [_rows("ln", Mp, H, 4, 2), _gemm(Mp, 3 * H, H), _attn_fwd(Bp, nh, V, V), _gemm(Mp, H, H, c=4, extra=Mp * H * 4),
               _rows("ln", Mp, H, 4, 2), _gemm(Mp, I, H, extra=Mp * I * 2), _gemm(Mp, H, I, c=4, extra=Mp * H * 4)]
        ks += [_gemm(Mp, I, H, extra=Mp * I * 2), _gemm(Mp, H, I, c=4), _gemm(H, I, Mp, c=4), _gemm(I, H, Mp, c=4),
               _rows("ln", Mp, H, 12, 4 + 2), _gemm(Mp, H, H), _attn_bwd(Bp, nh, V, V), _gemm(Mp, H, 3 * H, c=4),
               _gemm(H, H, Mp, c=4), _gemm(3 * H, H, Mp, c=4), _rows("ln", Mp, H, 12, 4 + 2)]
    ks += [_rows("ln", Mp, H, 4, 4), _rows("ln", Mp, H, 8, 4 + 2), _rows("embed", Mp, H, 4 + 4, 4),
           _gemm(H, Fi, Mp, c=4), _gemm(H, Fd, Mp, c=4)]
    # ---- node assembly (CSR gather over the panorama embeddings) both ways ----
    ks += [_rows("gather", Mg, H, 4, 4), _rows("gather", Mp, H, 4, 4)]
    # ---- navigation: 4 x (cross attention + self attention + FFN) + SAP head ----
    ks.append(_rows("embed", Mg, H, 4, 4 + 2))
    for _ in range(cfg.num_x_layers):
        ks += [_gemm(Mg, H, H), _gemm(Mt, 2 * H, H), _attn_fwd(B, nh, G, L), _gemm(Mg, H, H, c=4, extra=Mg * H * 4),
               _rows("ln", Mg, H, 4, 4 + 2)]
        ks += _post_ln_block_fwd(Mg, H, I, B, nh, G)
        ks += _post_ln_block_bwd(Mg, H, I, B, nh, G)
        ks += [_rows("ln", Mg, H, 8, 4 + 2), _gemm(Mg, H, H), _attn_bwd(B, nh, G, L), _gemm(Mg, H, H, c=4, extra=Mg * H * 4),
               _gemm(Mt, H, 2 * H, c=4, extra=Mt * H * 4),                       # d txt_embeds (accumulated over the layers)
               _gemm(H, H, Mg, c=4), _gemm(H, H, Mg, c=4), _gemm(2 * H, H, Mt, c=4)]
    ks += [_gemm(Mg, H, H), _rows("head", Mg, H, 2, 0), _rows("head", Mg, H, 2, 2), _gemm(Mg, H, H, c=4), _gemm(H, H, Mg, c=4),
           _rows("embed", Mg, H, 4, 4)]
    return ks


def step_roofline(cfg, B: int, L: int, V: int, G: int, Bp: int = None, peak_flops: float = PEAK_BF16_FLOPS,
                  peak_bps: float = PEAK_HBM_BPS) -> Dict[str, float]:
    ks = planner_step_kernels(cfg, B, L, V, G, Bp)
    t = sum(max(f / peak_flops, by / peak_bps) for _, f, by in ks)
    flops = sum(f for _, f, _ in ks)
    by = sum(b for _, _, b in ks)
    t_mfma = sum(f / peak_flops for _, f, b in ks if f / peak_flops >= b / peak_bps)
    return {"t_roof_ms": t * 1e3, "flops": flops, "hbm_bytes": by, "kernels": len(ks),
            "t_mfma_bound_ms": t_mfma * 1e3, "t_hbm_bound_ms": (t - t_mfma) * 1e3,
            "peak_flops": peak_flops, "peak_hbm_bytes_per_s": peak_bps}


def fused_plan_roofline(cfg, B: int, L: int, V: int, G: int, Bp: int = None, peak_flops: float = PEAK_BF16_FLOPS,
                        peak_bps: float = PEAK_HBM_BPS) -> Dict[str, float]:
    """The byte model of SURVEY.md §8(d) ("fused-kernel plan"): what a fully fused implementation would have to move.
    Per transformer layer and direction ONE fused kernel: forward reads the layer's bf16 weights (2 B / parameter) and writes
    the saved activations once, (6H + I) * 2 B per token (x, q, k, v, ctx, LN-out, FFN pre-activation); backward reads the
    weights again (2 B), writes fp32 weight gradients (4 B), reads the saved activations once and moves an equal volume of
    gradient streams.  FLOPs as SURVEY Appendix C (2MNK, backward = 2x forward).  t_roof = sum over these fused kernels of
    max(flops / peak_mfma, bytes / peak_hbm)."""
    H, I = cfg.hidden_size, cfg.intermediate_size
    Fi, Fd = cfg.image_feat_size, cfg.depth_feat_size
    Bp = B if Bp is None else Bp
    Mt, Mp, Mg = B * L, Bp * V, B * G
    act = (6 * H + I) * 2.0                                      # saved bytes per token-layer
    ks = []

    def layer(tokens, w_params, fwd_flops):
        ks.append((fwd_flops, w_params * 2.0 + tokens * act))                          # forward
        ks.append((2.0 * fwd_flops, w_params * 6.0 + tokens * act * 2.0))              # backward

    lin = 8 * H * H + 4 * H * I                                  # per-token linear FLOPs of a BERT / pano layer
    w_layer = 4 * H * H + 2 * H * I
    for _ in range(cfg.num_l_layers):
        layer(Mt, w_layer, Mt * lin + B * 4.0 * L * L * H)
    ks.append((Mp * 2.0 * H * (Fi + Fd + 4), H * (Fi + Fd) * 2.0 + Mp * (Fi + Fd) * 4.0 + Mp * H * 2.0))
    ks.append((2.0 * Mp * 2.0 * H * (Fi + Fd + 4), H * (Fi + Fd) * 6.0 + Mp * (Fi + Fd) * 2.0 + Mp * H * 4.0))
    for _ in range(cfg.num_pano_layers):
        layer(Mp, w_layer, Mp * lin + Bp * 4.0 * V * V * H)
    for _ in range(cfg.num_x_layers):
        fl = B * G * 4.0 * H * H + B * L * 4.0 * H * H + B * 4.0 * G * L * H + B * G * 8.0 * H * H + B * 4.0 * G * G * H \
            + B * G * 4.0 * H * I
        tokens = Mg * 1.7 + Mt * (2 * H * 2.0) / act             # node sub-blocks (~1.7 layer equivalents) + the text K/V rows
        layer(tokens, 8 * H * H + 2 * H * I, fl)
    ks.append((Mg * (2.0 * H * H + 2 * H), H * H * 2.0 + Mg * H * 4.0))
    ks.append((2.0 * Mg * (2.0 * H * H + 2 * H), H * H * 6.0 + Mg * H * 6.0))
    t = sum(max(f / peak_flops, by / peak_bps) for f, by in ks)
    return {"t_roof_ms": t * 1e3, "flops": sum(f for f, _ in ks), "hbm_bytes": sum(b for _, b in ks), "kernels": len(ks)}
