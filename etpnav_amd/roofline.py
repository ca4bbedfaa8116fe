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
