/* etpnav_hip.h — C ABI of the MI355X-native ETPNav planner hot path (libetpnav_hip.so).
 *
 * The reference (MarSaKi/ETPNav) is 100 % Python/PyTorch and has no FFI; the "interface each entry point
 * replaces" is therefore the reference Python call site whose arithmetic it takes over (paths relative to
 * the reference root).  Every function:
 *   - takes raw device pointers (caller-owned, caller-allocated), plain integer sizes and a hipStream_t
 *     passed as void*; no torch / C++ types cross the boundary;
 *   - enqueues work on that stream and returns immediately (hipGraph-capturable: no allocation, no sync);
 *   - returns 0 on success, <0 for an invalid argument, >0 for a hipError_t; etp_last_error() gives text.
 * "T" below means the compute dtype selected by `dtype` (ETP_F32 parity mode, ETP_BF16 performance mode);
 * parameters, statistics, logits and losses are always fp32.
 */
#ifndef ETPNAV_HIP_H
#define ETPNAV_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ETP_OK 0
#define ETP_ERR_INVALID (-1)
#define ETP_ERR_STATE (-2)

#define ETP_F32 0
#define ETP_BF16 1

/* GEMM epilogue activations */
#define ETP_ACT_NONE 0
#define ETP_ACT_GELU 1      /* C = gelu_erf(v), aux Z = v            (BertIntermediate vilmodel_cmt.py:177-180) */
#define ETP_ACT_RELU 2      /* C = relu(v)                           (NextActionPrediction :654-655)           */
#define ETP_ACT_GELU_BWD 3  /* C = v * gelu_erf'(Z)                                                              */
#define ETP_ACT_RELU_BWD 4  /* C = v * (Z > 0)                                                                   */
/* Round 5: the GELU pair the planner's FFN blocks use.  The erf arithmetic made the GELU / GELU' epilogues VALU-bound (measured
 * with the arithmetic compiled out: FFN-up epilogue 8.2 -> 3.8 us, FFN dgrad 11.5 -> 5.5 us, profiles/r05_epilogue_valu.txt), and the
 * forward has everything the derivative needs in registers (cdf and exp(-v^2/2)): it saves gelu_erf'(v) instead of v, the backward
 * epilogue is a multiply.  Same stash footprint; nothing else reads the pre-activation. */
#define ETP_ACT_GELU_SAVEGRAD 5  /* C = gelu_erf(v), aux Z = gelu_erf'(v)   (bf16 mode: Z holds IEEE HALF values -- the      */
#define ETP_ACT_MUL_Z 6          /* C = v * Z                                derivative lies in [-0.13, 1.13], 11 bits > bf16's 8) */

typedef void* etp_stream_t; /* hipStream_t */

const char* etp_version(void);
const char* etp_last_error(void);

/* Run-time switches of the library (tile-class forcing for tests and A/B runs, schedule variants).  They are read from the
 * environment ONCE, at the first lookup (ETP_<NAME>), and afterwards change only through these calls -- no launch path calls
 * getenv (rounds 1-5 did, three times per GEMM launch).  `name` with or without the ETP_ prefix; value NULL or "" = unset.
 * The reference has no counterpart (its switches are Python config keys, vlnce_baselines/config/default.py); the names are listed
 * in csrc/options.h and every line bench.py prints carries the ones that are set (config.env_overrides). */
int etp_option_set(const char* name, const char* value);
int etp_option_get(const char* name, char* out, int cap);   /* length of the value (0 = unset), -1 = unknown switch */
int etp_option_list(char* out, int cap);                     /* "NAME=value\n" per set switch; returns the length needed */

/* ------------------------------------------------------------------------------------------------------
 * Per-operator entry points (one per implicit device op of SURVEY.md §2.1)
 * ---------------------------------------------------------------------------------------------------- */

/* C[m,n] = epi(alpha * sum_k A[m,k]*B[n,k]); replaces every nn.Linear / torch.matmul on the path
 * (vilmodel_cmt.py:108-110,117,133,151,178,190,326-328,335,348; common/transformer.py:138,140-142) and their
 * autograd backward (dgrad / wgrad).  trans_a/trans_b = 1 means the operand is stored [K][rows]. */
typedef struct etp_gemm_desc {
  const void* A; const void* B; void* C;
  int32_t M, N, K;
  int64_t lda, ldb, ldc;
  int32_t trans_a, trans_b;
  int32_t dtype;            /* operand dtype */
  int32_t c_dtype;          /* output dtype (ETP_F32 allowed with bf16 operands: weight gradients) */
  int32_t batch, batch_inner;               /* z -> (zo = z / batch_inner, zi = z % batch_inner) */
  int64_t sAo, sAi, sBo, sBi, sCo, sCi;     /* batch strides in elements */
  int32_t ksplit;           /* >1: split the reduction, needs out_mode 2 */
  float alpha;
  const float* bias;        /* [N] or NULL */
  const void* R; int64_t ldr; /* residual added after the activation, dtype T, or NULL */
  void* Z; int64_t ldz;     /* aux tensor for the activation epilogues, dtype T */
  int32_t act;              /* ETP_ACT_* */
  int32_t out_mode;         /* 0 store, 1 C += v, 2 atomicAdd (fp32 C) */
  float* a_colsum;          /* TN products (weight gradients) only, or NULL: a_colsum[m] += sum_k A[m,k] -- the bias gradient
                             * db = colsum(dY) fused into the dW = dY^T X product (K a multiple of the 128-byte slab, >= 2 slabs) */
} etp_gemm_desc;
int etp_gemm(const etp_gemm_desc* d, etp_stream_t stream);
/* n (<= 8) independent, unbatched, unsplit products of ONE (dtype, c_dtype, trans_a, trans_b) class in a single grid: the
 * four to seven weight gradients of one transformer layer (autograd of vilmodel_cmt.py:108-110,151,178,190,326-328), none
 * of which fills 256 CUs alone.  Every K must be a multiple of the 128-byte slab (64 bf16 / 32 fp32) and >= 2 slabs. */
int etp_gemm_group(const etp_gemm_desc* d, int n, etp_stream_t stream);

/* db[n] += sum_m dY[m,n]  (bias gradient of every nn.Linear). */
int etp_colsum(int dtype, const void* dy, int64_t ld, float* db, int M, int N, etp_stream_t stream);

/* y = LayerNorm(x); stats[row] = {mean, rstd}.  BertLayerNorm / nn.LayerNorm: vilmodel_cmt.py:59,147,186,459-478,
 * 571,656; common/transformer.py:144-145; common/ops.py:19-23. */
int etp_ln_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* stats, int M, int H,
               float eps, etp_stream_t stream);
/* dx = LNbwd(dy) (+ add if non-NULL); dgamma/dbeta accumulated atomically. */
int etp_ln_bwd(int dtype, const void* dy, const void* x, const float* stats, const float* gamma, const void* add, void* dx,
               float* dgamma, float* dbeta, int M, int H, etp_stream_t stream);

/* In-place masked row softmax over scores S[B,heads,Lq,ldS] (vilmodel_cmt.py:117-127,335-346,391-393,732-736):
 *   s += keymask(b,k) + (sp_w*dist[b,q,k] + sp_b);  mask_mode 0: (1-m)*-10000 (ops.py:25-34), 1: -inf (MHA key padding).
 * Columns [Lk, ldS) are written as 0. */
int etp_softmax_fwd(int dtype, void* S, const uint8_t* keymask, const float* dist, const float* sp_w, const float* sp_b,
                    int B, int heads, int Lq, int Lk, int ldS, int mask_mode, etp_stream_t stream);
int etp_softmax_bwd(int dtype, const void* P, void* dP, const float* dist, float* d_sp_w, float* d_sp_b, int B, int heads,
                    int Lq, int Lk, int ldS, etp_stream_t stream);

/* softmax(alpha*Q.K^T + mask)V for [B,heads] problems with head dim 64, head-interleaved row layouts
 * (BertSelfAttention / BertOutAttention / nn.MultiheadAttention).  P [B,heads,Lq,ldS] is the buffer the forward leaves for the
 * backward of the SAME shape/dtype: probabilities on the tile / batched-GEMM paths (fp32; bf16 with dist on an axis > 128),
 * and for bf16 otherwise only lse = rowmax + log(rowsum) (fp32, [B,heads,Lq] in the front of the buffer) -- the register-resident
 * (both axes <= 128) and streaming kernels recompute the probabilities in backward.  Callers must treat it as opaque. */
typedef struct etp_attn_desc {
  int32_t dtype, B, heads, Lq, Lk, ldS;
  const void* Q; int64_t ldq;   /* Q rows [B*Lq], head h at column h*64 */
  const void* K; int64_t ldk;
  const void* V; int64_t ldv;
  void* P;                      /* [B,heads,Lq,ldS] saved for etp_attn_bwd (opaque, see above) */
  void* ctx; int64_t ldc;       /* [B*Lq, heads*64] */
  const uint8_t* keymask;       /* [B,Lk] 1 = valid */
  int32_t mask_mode;
  const float* dist;            /* [B,Lq,Lk] or NULL (graph_sprels) */
  const float* sp_w; const float* sp_b;
  float alpha;
} etp_attn_desc;
int etp_attn_fwd(const etp_attn_desc* d, etp_stream_t stream);
/* Self-attention forward with the QKV PROJECTION folded in (round 6): Q / K / V of `d` (Lq == Lk, rows of one token block, e.g. the
 * three column blocks of a [B*L, 3*heads*64] stash) are OUTPUTS -- each (batch, head) workgroup computes
 *   [Q | K | V][b, l, h*64 : h*64+64] = x[b*L + l, :] . w_qkv[sec*heads*64 + h*64 + (0..63), :]^T + b_qkv     (sec = 0, 1, 2)
 * itself (BertSelfAttention.query / key / value, vilmodel_cmt.py:108-110; MHA in_proj_weight, common/transformer.py:138), stores them
 * (the backward's stash) and goes on with the attention, instead of reading the result of a GEMM launch.  x [B*L, heads*64] (row
 * stride ldx) and w_qkv [3*heads*64][ldw] in the operand dtype, b_qkv fp32 [3*heads*64] or NULL.  bf16, L <= 128, heads*64 == 768;
 * anything else returns ETP_ERR_INVALID and the caller issues etp_gemm + etp_attn_fwd.  Results equal that pair's (the projections
 * are rounded to bf16 exactly where the GEMM stored them). */
int etp_attn_fwd_qkv(const etp_attn_desc* d, const void* x, int64_t ldx, const void* w_qkv, int64_t ldw, const float* b_qkv,
                     etp_stream_t stream);
typedef struct etp_attn_bwd_desc {
  etp_attn_desc f;              /* same as forward; ctx / ldc MUST be the forward's output: the bf16 kernels keep only lse in P and
                                 * recompute from it (Lq or Lk > 128 also read ctx for D = rowsum(dO * O)) */
  const void* dctx; int64_t ldd;
  void* dP;                     /* scratch [B,heads,Lq,ldS] */
  void* dQ; int64_t lddq; void* dK; int64_t lddk; void* dV; int64_t lddv;
  float* d_sp_w; float* d_sp_b; /* accumulated, or NULL */
} etp_attn_bwd_desc;
int etp_attn_bwd(const etp_attn_bwd_desc* d, etp_stream_t stream);
/* The same backward with the OUT-PROJECTION's input gradient folded in (round 6): `d->dctx` is dL/d(dense output) [B*Lq, heads*64]
 * (row stride ldd) of BertSelfOutput.dense / BertOutAttention's output dense / MHA out_proj (vilmodel_cmt.py:150-154, 325-352;
 * common/transformer.py:138-142), `w_out` that projection's weight [heads*64 (out)][ldw] in the operand dtype; each (batch, head)
 * workgroup forms dctx[:, h*64:h*64+64] = dctx_in . w_out[:, h*64:h*64+64] itself instead of reading the result of a GEMM launch.
 * bf16 with both axes <= 128 and heads*64 == 768; anything else returns ETP_ERR_INVALID and the caller issues
 * etp_gemm + etp_attn_bwd.  Results equal that pair's (the tile is rounded to bf16 exactly where the GEMM stored it). */
int etp_attn_bwd_proj(const etp_attn_bwd_desc* d, const void* w_out, int64_t ldw, etp_stream_t stream);

/* Residual-stream convention: tensors that flow from one LayerNorm / residual add to the next are ALWAYS fp32 (as under
 * the reference's autocast, where LayerNorm and residual adds stay fp32); `*_lp` arguments are optional copies in the GEMM
 * operand dtype `dtype` for the next MFMA product (pass NULL in fp32 mode). */

/* LayerNorm on the fp32 stream: y (fp32, may be NULL) and/or y_lp (operand dtype, may be NULL). */
int etp_ln_stream_fwd(int dtype, const float* x, const float* gamma, const float* beta, float* y, void* y_lp, float* stats, int M,
                      int H, float eps, etp_stream_t stream);
int etp_ln_stream_bwd(int dtype, const float* dy, const float* x, const float* stats, const float* gamma, const float* add,
                      float* dx, void* dx_lp, float* dgamma, float* dbeta, int M, int H, etp_stream_t stream);
/* The same with the two-stage parameter-gradient reduction the planner uses: stage 1 (on `stream`) writes per-workgroup
 * column sums to `part` (etp_ln_bwd_part_bytes(M, H) bytes), stage 2 (on `reduce_stream`, ordered after stage 1 by the
 * caller when the streams differ) adds them into dgamma / dbeta. */
int64_t etp_ln_bwd_part_bytes(int M, int H);
int etp_ln_stream_bwd_stage1(int dtype, const float* dy, const float* x, const float* stats, const float* gamma, const float* add,
                             float* dx, void* dx_lp, float* dgamma, float* dbeta, float* part, int M, int H, etp_stream_t stream);
int etp_ln_part_reduce(const float* part, int M, int H, float* dgamma, float* dbeta, etp_stream_t reduce_stream);

/* BertEmbeddings.forward vilmodel_cmt.py:62-77 (eval): y = LN(word[id] + pos[l] + type[0]). */
int etp_text_embed_fwd(int dtype, const int64_t* ids, const float* word, const float* pos, const float* type0,
                       const float* gamma, const float* beta, float* y, void* y_lp, float* stats, int B, int L, int H, float eps,
                       etp_stream_t stream);
int etp_text_embed_bwd(int dtype, const float* dy, const int64_t* ids, const float* word, const float* pos,
                       const float* type0, const float* gamma, const float* stats, float* dword, float* dpos, float* dtype0,
                       float* dgamma, float* dbeta, int B, int L, int H, etp_stream_t stream);

/* Panorama view-embedding fuse, forward_panorama vilmodel_cmt.py:695-711:
 *   y = LN(LN_i(a) + LN_d(d) + LN_l(loc.Wl^T+bl) + nav_emb[nav] + type_emb[1]); a,d (dtype T) = MFMA projections of rgb/depth.
 * params / grads: 12 fp32 pointers in the order g_img,b_img,g_dep,b_dep,w_loc,bias_loc,g_loc,b_loc,nav_emb,type1,g_out,b_out.
 * stats: [M,8]; y / dy fp32; da, dd in dtype T. */
int etp_pano_embed_fwd(int dtype, const void* a, const void* d, const float* loc, const int64_t* nav,
                       const float* const* params, float* y, float* stats, int M, int H, etp_stream_t stream);
int etp_pano_embed_bwd(int dtype, const float* dy, const void* a, const void* d, const float* loc, const int64_t* nav,
                       const float* stats, const float* const* params, float* const* grads, void* da, void* dd, int M, int H,
                       etp_stream_t stream);

/* forward_navigation vilmodel_cmt.py:728-730: x = img + step_emb[step] + LN(pos.Wp^T+bp)  (img, x, dx fp32). */
int etp_gmap_embed_fwd(int dtype, const float* img, const int64_t* step_ids, const float* pos, const float* step_emb,
                       const float* w_pos, const float* b_pos, const float* gamma, const float* beta, float* x, void* x_lp,
                       float* stats, int M, int H, int pos_dim, etp_stream_t stream);
int etp_gmap_embed_bwd(int dtype, const float* dx, const int64_t* step_ids, const float* pos, const float* w_pos,
                       const float* b_pos, const float* gamma, const float* stats, float* d_step_emb, float* d_w_pos,
                       float* d_b_pos, float* dgamma, float* dbeta, int M, int H, int pos_dim, etp_stream_t stream);

/* NextActionPrediction tail vilmodel_cmt.py:651-661 + masked_fill_ :742-744: logits = LN(r).w2 + b2, -inf where
 * visited or !valid; r = relu(x.W1^T+b1) from etp_gemm(ETP_ACT_RELU). */
int etp_sap_tail_fwd(int dtype, const void* r, const float* gamma, const float* beta, const float* w2, const float* b2,
                     const uint8_t* visited, const uint8_t* valid, float* logits, float* stats, int M, int H,
                     etp_stream_t stream);
int etp_sap_tail_bwd(int dtype, const float* dlogits, const void* r, const float* gamma, const float* beta, const float* w2,
                     const float* stats, const uint8_t* visited, const uint8_t* valid, void* dz, float* dgamma, float* dbeta,
                     float* dw2, float* db2, int M, int H, etp_stream_t stream);

/* F.cross_entropy(reduction='sum', ignore_index) ss_trainer_ETP.py:892 scaled by `scale` (:1055):
 * *loss = scale*sum_b nll_b (stored, not accumulated) ; dlogits = scale*(softmax - onehot) (0 on ignored rows); dlogits may
 * be NULL. */
int etp_sap_ce(const float* logits, const int64_t* labels, float* loss, float* dlogits, int B, int G, float scale,
               int64_t ignore_index, etp_stream_t stream);

/* out[n,:] (+)= sum_{j in [ptr[n],ptr[n+1])} w[j]*src[idx[j],:] — node aggregation (ss_trainer_ETP.py:838-839,
 * graph_utils.py:272-276, pretrain vilmodel.py:585-619) and, with the transposed CSR, its backward. */
int etp_gather_sum(int dtype, const void* src, const int32_t* ptr, const int32_t* idx, const float* w, void* out, int N, int H,
                   int accumulate, etp_stream_t stream);

int etp_cast_f32_to_bf16(const float* src, void* dst, int64_t n, etp_stream_t stream);
int etp_cast_bf16_to_f32(const void* src, float* dst, int64_t n, float scale, etp_stream_t stream);
int etp_scale_f32(float* p, int64_t n, float scale, etp_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * Fused AdamW over flat fp32 arenas.  Replaces, in ONE HBM pass per step: torch.optim.AdamW
 * (ss_trainer_ETP.py:213,505) or the pre-training AdamW (pretrain_src/pretrain_src/optim/adamw.py:53-112),
 * GradScaler.unscale_ + its non-finite check (ss_trainer_ETP.py:463,504-506), clip_grad_norm_, the autocast weight
 * casts of the next forward (etp_planner_refresh_weights) and optimizer.zero_grad().
 *   hf_style 0: torch.optim.AdamW      p *= 1 - lr*wd;  p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
 *   hf_style 1: optim/adamw.py         p -= lr*sqrt(bc2)/bc1 * m / (sqrt(v) + eps);  p -= lr*wd*p
 *   bc1 = 1-beta1^step, bc2 = 1-beta2^step (both 1 when correct_bias == 0); step counts from 1.
 *   g = grad * grad_scale * clip, clip = min(1, max_norm / (sqrt(*sumsq)*|grad_scale| + 1e-6)) when max_norm > 0.
 * decay_mask (nullable = decay everywhere, nothing frozen): one byte per 64 consecutive elements; bit 0 = apply weight_decay
 * (planner parameters start on 64-element boundaries, so any per-parameter grouping such as optim/misc.py:12-22 fits),
 * bit 1 = FROZEN block (requires_grad = False: fix_lang_embedding / fix_pano_embedding, vilmodel_cmt.py:675-682;
 * LanguageEncoder :422-424): p / m / v / shadow are left untouched, the gradient is still zeroed.
 * CHANGELOG (round 5 -> 6): until round 4 the byte meant "any non-zero value = decay".  The byte values are now 0 = no decay,
 * 1 = decay, 2 / 3 = frozen; every other non-zero value (0xFF, a bool stored as 255, ...) still means "decay" and never "frozen", so
 * the only value whose meaning changed for an old caller is exactly 2.
 * shadow (nullable): bf16 copy written for elements [0, n_shadow).  skip (nullable, device int32): non-zero -> leave
 * p/m/v/shadow untouched (GradScaler's skipped step); gradients are still zeroed when zero_grads != 0.
 * sumsq / skip are produced by etp_grad_sqnorm (both ACCUMULATE: zero them first).  n % 4 == 0, 16-byte aligned. */
typedef struct {
  float lr, beta1, beta2, eps, weight_decay;
  int32_t step;
  int32_t hf_style;
  int32_t correct_bias;
  float grad_scale;
  float max_norm;
} etp_adamw_cfg;
int etp_adamw_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, void* shadow, int64_t n_shadow,
                   const uint8_t* decay_mask, int64_t n, const etp_adamw_cfg* cfg, const float* sumsq, const int32_t* skip,
                   int zero_grads, etp_stream_t stream);
/* The same with the optimizer's step count kept ON THE DEVICE: *step_counter is incremented only when the update is applied
 * (skip == NULL or *skip == 0) and the bias corrections use it -- GradScaler.step() does not call optimizer.step() on
 * overflow, so state['step'] must not advance on a skipped step (ss_trainer_ETP.py:504-506).  cfg->step is ignored. */
int etp_adamw_step_counted(float* params, float* grads, float* exp_avg, float* exp_avg_sq, void* shadow, int64_t n_shadow,
                           const uint8_t* decay_mask, int64_t n, const etp_adamw_cfg* cfg, const float* sumsq, const int32_t* skip,
                           int zero_grads, int32_t* step_counter, etp_stream_t stream);
int etp_grad_sqnorm(const float* grads, int64_t n, float* sumsq, int32_t* nonfinite, etp_stream_t stream);
/* The same, leaving out the blocks whose mask byte (layout of decay_mask above) has bit 1 set: frozen parameters have no .grad in
 * the reference, so clip_grad_norm_ / GradScaler's non-finite scan never see them. */
int etp_grad_sqnorm_masked(const float* grads, int64_t n, const uint8_t* mask, float* sumsq, int32_t* nonfinite,
                           etp_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * Planner engine: whole forward/backward of the three planner entry points over one flat parameter arena.
 * Replaces GlocalTextPathNavCMT.forward_txt / forward_panorama / forward_navigation (vilmodel_cmt.py:684-750)
 * and their autograd backward.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct etp_config {
  int32_t hidden, heads, inter;
  int32_t n_l, n_p, n_x;                 /* text / panorama / cross-modal layer counts (vlnbert_init.py:46-48) */
  int32_t vocab, max_pos, type_vocab;
  int32_t img_feat, dep_feat, ang_feat, max_steps;
  int32_t use_depth, use_sprels;
  float ln_eps;                          /* config.layer_norm_eps (1e-12 bert / 1e-5 xlm-r) */
  int32_t dtype;                         /* ETP_F32 | ETP_BF16 */
  int32_t use_lang2visn;                 /* pre-training variant (run_pt/r2r_model_config_dep.json use_lang2visn_attn): adds the
                                          * language-side x-layer weights (vilmodel.py:371-376) and the tied MLM head (:258-299) */
} etp_config;

typedef struct etp_param_info {
  char name[128];                        /* reference state-dict name */
  int32_t ndim; int64_t shape[2];
  int64_t offset;                        /* element offset in the fp32 arena (and grad arena, and bf16 shadow) */
} etp_param_info;

typedef struct etp_planner etp_planner;

etp_planner* etp_planner_create(const etp_config* cfg);      /* NULL on error */
void etp_planner_destroy(etp_planner* p);
int etp_planner_param_count(const etp_planner* p);
int etp_planner_param_info(const etp_planner* p, int i, etp_param_info* out);
int64_t etp_planner_arena_elems(const etp_planner* p);       /* total fp32 elements */
int64_t etp_planner_matrix_elems(const etp_planner* p);      /* leading region holding the GEMM weights */
/* params: fp32 master arena; shadow: bf16 copy of the matrix region (NULL in fp32 mode); grads: fp32 arena. */
int etp_planner_bind(etp_planner* p, float* params, void* shadow, float* grads);
/* Optional second stream for the backward entry points: weight-gradient GEMMs (leaves of the autograd graph) are issued
 * on `aux` after their dY producer and joined back before the call returns to `stream`, so they overlap the dgrad
 * chain (works eagerly and under hipGraph capture: the fork/join become graph edges).  NULL = single stream. */
int etp_planner_set_aux_stream(etp_planner* p, etp_stream_t aux);
/* Optional third stream for etp_nav_bwd: the d(txt_embeds) contributions of the text K/V projections (vilmodel_cmt.py:326-328,
 * M = B*L rows) accumulate on `aux2` beside the node chain and are joined back before the call returns.  NULL = main stream. */
int etp_planner_set_aux2_stream(etp_planner* p, etp_stream_t aux2);
/* With an aux stream: lazy = 1 lets etp_nav_bwd* / etp_pano_bwd return WITHOUT joining their weight-gradient GEMMs back
 * (nothing downstream of them reads weight gradients), so the text backward does not wait for the navigation weight
 * gradients; the gradients are complete in `stream` order only after a later joining call: etp_txt_bwd / etp_txt_bwd_range
 * / etp_nav_kv_bwd (always join) or etp_planner_join_aux.  Default 0: every backward entry point joins before it returns.
 * lazy = 2: additionally etp_txt_bwd_range calls that stop above layer 0 do not join (the next range continues the chain). */
int etp_planner_set_lazy_join(etp_planner* p, int lazy);
/* on = 1: weight-gradient products STORE into the matrix region [0, etp_planner_matrix_elems) of the gradient arena instead of
 * accumulating (torch's `.grad +=`, the default): no fp32 read of the old gradient and no per-step zeroing of that region.
 * Valid when every weight matrix receives exactly one weight-gradient product between two optimizer steps (one rollout step
 * per optimizer step, as bench.py's unit of work); the vector / embedding-table tail still accumulates and must be zeroed. */
int etp_planner_set_grad_overwrite(etp_planner* p, int on);
int etp_planner_join_aux(etp_planner* p, etp_stream_t stream);
/* Training-mode dropout (all rates 0 = eval, the default).  Masks are a counter-based hash of (seed, site, element index):
 * nothing is stored, the backward entry points recompute the masks, so a backward call must see the same rates and seed as
 * its forward (the state is read at enqueue time; change it between calls freely).  Sites mirror the reference:
 *   p_hidden  nn.Dropout(hidden_dropout_prob): BertEmbeddings vilmodel_cmt.py:76, BertSelfOutput :152, BertOutput :191,
 *             BertXAttention output :363 (BertSelfOutput), panorama embedding :711, and the panorama
 *             TransformerEncoderLayer's dropout/dropout1/dropout2 + its MultiheadAttention dropout (common/ops.py:15,
 *             common/transformer.py:138-147,178-181);
 *   p_attn    attention_probs_dropout_prob on softmax probabilities (:127 self, :346 cross);
 *   p_head    ClsPrediction dropout (:657, pred_head_dropout_prob);
 *   p_env     the policy's drop_env on the RGB features (Policy_ViewSelection_ETP.py:102,345) fused into the operand
 *             cast of forward_panorama (0 = leave it to the caller, as the reference does).
 * Sequences beyond the fused attention kernels (Lq or Lk > 128; fp32 mode > 64, e.g. the 512-token RxR instruction) take
 * the batched-GEMM attention path, whose stash then carries a second probability buffer for the dropped copy. */
int etp_planner_set_dropout(etp_planner* p, float p_hidden, float p_attn, float p_head, float p_env, uint64_t seed);
/* Host-side view of the mask generator (tests, debugging): out_host[i] = multiplier (0 or 1/(1-p)) of element i (row-major
 * index into the site's tensor) at site (mode 1=txt 2=panorama 3=navigation, layer, slot) for step seed `seed`.  Slots:
 * 0 embedding output, 1 self-attention probabilities, 2 attention-output dense, 3 FFN-output dense, 4 FFN inner
 * (panorama layers), 5 cross-attention probabilities, 6 cross-attention-output dense, 7 SAP head, 8 drop_env.
 * Needs no GPU. */
int etp_dropout_multipliers(float p, uint64_t seed, int mode, int layer, int slot, int64_t n, float* out_host);
/* bf16 mode: refresh the bf16 shadow of the GEMM weights from the fp32 masters (autocast's per-step weight cast). */
int etp_planner_refresh_weights(etp_planner* p, etp_stream_t stream);
/* The same refresh for one consumer only, so a multi-stream step can put each cast on the stream that first needs it:
 * part 0 = text encoder (forward_txt), 1 = view projections + panorama encoder (forward_panorama), 2 = x-layers + SAP
 * head (forward_navigation).  The three parts tile the matrix region exactly. */
int etp_planner_refresh_part(etp_planner* p, int part, etp_stream_t stream);
/* bf16 shadow of the text encoder, layer 0 on `main` and layers >= 1 on `side`; the next etp_txt_fwd issued on `main` waits for
 * the side cast after its layer 0 (takes ~40 us of weight casting off the head of the dependent chain of a training step). */
int etp_planner_refresh_text_split(etp_planner* p, etp_stream_t main, etp_stream_t side);

/* Activations that cross these entry points (txt_embeds, pano_embeds, gmap_img_fts, gmap_embeds and their gradients) are
 * fp32 in BOTH modes, as they are under the reference's autocast (outputs of fp32 LayerNorms); `dtype` only selects the
 * GEMM / attention operand precision inside. */
int64_t etp_txt_stash_bytes(const etp_planner* p, int B, int L);
int64_t etp_txt_ws_bytes(const etp_planner* p, int B, int L);
int etp_txt_fwd(etp_planner* p, const int64_t* txt_ids, const uint8_t* txt_masks, int B, int L, float* txt_embeds /*[B,L,H]*/,
                void* stash, etp_stream_t stream);
int etp_txt_bwd(etp_planner* p, const float* d_txt_embeds, const int64_t* txt_ids, const uint8_t* txt_masks, int B, int L,
                void* stash, void* ws, etp_stream_t stream);

/* Same, restricted to text layers [layer_lo, layer_hi) (call with descending ranges and the same ws: the running gradient is
 * kept in ws).  The first call (layer_hi = n_l) reads d_txt_embeds, the last (layer_lo = 0) also runs the embedding backward.
 * Lets a data-parallel caller all-reduce the gradients of finished layers while earlier layers are still in backward. */
int etp_txt_bwd_range(etp_planner* p, const float* d_txt_embeds, const int64_t* txt_ids, const uint8_t* txt_masks, int B, int L,
                      void* stash, void* ws, int layer_lo, int layer_hi, etp_stream_t stream);

int64_t etp_pano_stash_bytes(const etp_planner* p, int B, int V);
int64_t etp_pano_ws_bytes(const etp_planner* p, int B, int V);
int etp_pano_fwd(etp_planner* p, const float* rgb, const float* dep, const float* loc, const int64_t* nav_types,
                 const int64_t* view_lens, int B, int V, float* pano_embeds /*[B,V,H]*/, uint8_t* pano_masks /*[B,V]*/,
                 void* stash, etp_stream_t stream);
int etp_pano_bwd(etp_planner* p, const float* d_pano_embeds, const float* rgb, const float* dep, const float* loc,
                 const int64_t* nav_types, int B, int V, float* d_rgb /*[B,V,img_feat] or NULL*/, void* stash, void* ws,
                 etp_stream_t stream);

int64_t etp_nav_stash_bytes(const etp_planner* p, int B, int L, int G);
int64_t etp_nav_ws_bytes(const etp_planner* p, int B, int L, int G);
int etp_nav_fwd(etp_planner* p, const float* txt_embeds, const uint8_t* txt_masks, const int64_t* gmap_step_ids,
                const float* gmap_img_fts, const float* gmap_pos_fts, const uint8_t* gmap_masks,
                const uint8_t* gmap_visited_masks, const float* gmap_pair_dists, int B, int L, int G,
                float* gmap_embeds /*[B,G,H]*/, float* global_logits /*[B,G]*/, void* stash, etp_stream_t stream);
int etp_nav_bwd(etp_planner* p, const float* d_gmap_embeds /*or NULL*/, const float* d_logits /*or NULL*/,
                const float* txt_embeds, const uint8_t* txt_masks, const int64_t* gmap_step_ids, const float* gmap_pos_fts,
                const uint8_t* gmap_masks, const uint8_t* gmap_visited_masks, const float* gmap_pair_dists, int B, int L, int G,
                float* d_txt_embeds /*[B,L,H], overwritten*/, float* d_gmap_img_fts /*[B,G,H], overwritten*/, void* stash,
                void* ws, etp_stream_t stream);

/* Text K/V cache for rollouts.  The instruction is fixed for an episode, but BertOutAttention
 * (vilmodel_cmt.py:326-328) re-projects it to keys/values in every x-layer at every step (GraphLXRTXLayer :387-389 called
 * from the per-step loop ss_trainer_ETP.py:819-892).  Compute the projections once per episode batch and reuse them:
 *   etp_nav_kv_fwd   txt_embeds -> cache (caller-owned, etp_nav_kv_bytes): [bf16 text | K|V of x-layer 0 | ... ]
 *   etp_nav_fwd_kv   = etp_nav_fwd reading keys/values from the cache (identical results)
 *   etp_nav_bwd_kv   = etp_nav_bwd, except that dK|dV of each x-layer are WRITTEN to d_kv [n_x][B*L][2H] (operand dtype,
 *                      etp_nav_kv_grad_elems elements) instead of being projected back immediately
 *   etp_nav_kv_bwd   once per episode: d_kv summed over the steps -> gradients of the K/V weights and d_txt_embeds. */
int64_t etp_nav_kv_bytes(const etp_planner* p, int B, int L);
int64_t etp_nav_kv_grad_elems(const etp_planner* p, int B, int L);
int64_t etp_nav_kv_offset(const etp_planner* p, int B, int L);   /* byte offset of the K|V blocks (contiguous, layer-major) */
int etp_nav_kv_fwd(etp_planner* p, const float* txt_embeds, int B, int L, void* kv_cache, etp_stream_t stream);
int etp_nav_kv_bwd(etp_planner* p, const float* txt_embeds, const void* d_kv, int B, int L, const void* kv_cache,
                   float* d_txt_embeds /*[B,L,H], overwritten*/, etp_stream_t stream);
/* The cache under the batched rollout call (T steps stacked along the batch axis, episode t*Bt + b reads instruction b; the
 * reference re-projects the same instruction at every step, ss_trainer_ETP.py:819-822 -> vilmodel_cmt.py:326-328):
 *   etp_nav_kv_repeat     K|V blocks of a Bt-instruction cache replicated T times into a cache sized for T*Bt episodes
 *                         (etp_nav_kv_bytes(p, T*Bt, L)) -- a copy in place of T projections of the same rows
 *   etp_nav_kv_sum_steps  d_kv of the stacked call [n_x][T*Bt*L][2H] summed over the steps (fp32 accumulation) into
 *                         [n_x][Bt*L][2H], the operand of etp_nav_kv_bwd */
int etp_nav_kv_repeat(etp_planner* p, const void* kv_cache, int Bt, int L, int T, void* kv_cache_steps, etp_stream_t stream);
int etp_nav_kv_sum_steps(etp_planner* p, const void* d_kv_steps, int Bt, int L, int T, void* d_kv, etp_stream_t stream);
int etp_nav_fwd_kv(etp_planner* p, const void* kv_cache, const uint8_t* txt_masks, const int64_t* gmap_step_ids,
                   const float* gmap_img_fts, const float* gmap_pos_fts, const uint8_t* gmap_masks,
                   const uint8_t* gmap_visited_masks, const float* gmap_pair_dists, int B, int L, int G, float* gmap_embeds,
                   float* global_logits, void* stash, etp_stream_t stream);
int etp_nav_bwd_kv(etp_planner* p, const float* d_gmap_embeds /*or NULL*/, const float* d_logits /*or NULL*/,
                   const void* kv_cache, const uint8_t* txt_masks, const int64_t* gmap_step_ids, const float* gmap_pos_fts,
                   const uint8_t* gmap_masks, const uint8_t* gmap_visited_masks, const float* gmap_pair_dists, int B, int L,
                   int G, void* d_kv /*overwritten*/, float* d_gmap_img_fts, void* stash, void* ws, etp_stream_t stream);
/* The same two calls with PER-EPISODE INDIRECTION instead of the replicated copy (round 6; N1): B = T * Bt stacked episodes, `kv_cache`
 * (etp_nav_kv_bytes(p, Bt, L)) and `txt_masks` [Bt, L] hold the Bt instructions once, and episode e reads the keys / values / key mask
 * of instruction e % Bt inside the cross-attention kernels (vilmodel_cmt.py:326-328 with the same txt_embeds at every step,
 * ss_trainer_ETP.py:819-822).  d_kv is still per stacked episode [n_x][B*L][2H] (the caller sums it over the steps with
 * etp_nav_kv_sum_steps).  bf16 with L and G <= 128 (the register-resident attention kernels); otherwise ETP_ERR_INVALID: use
 * etp_nav_kv_repeat + the calls above. */
int etp_nav_fwd_kv_steps(etp_planner* p, const void* kv_cache, const uint8_t* txt_masks, const int64_t* gmap_step_ids,
                         const float* gmap_img_fts, const float* gmap_pos_fts, const uint8_t* gmap_masks,
                         const uint8_t* gmap_visited_masks, const float* gmap_pair_dists, int B, int L, int G, int Bt,
                         float* gmap_embeds, float* global_logits, void* stash, etp_stream_t stream);
int etp_nav_bwd_kv_steps(etp_planner* p, const float* d_gmap_embeds /*or NULL*/, const float* d_logits /*or NULL*/,
                         const void* kv_cache, const uint8_t* txt_masks, const int64_t* gmap_step_ids, const float* gmap_pos_fts,
                         const uint8_t* gmap_masks, const uint8_t* gmap_visited_masks, const float* gmap_pair_dists, int B, int L,
                         int G, int Bt, void* d_kv /*[n_x][B*L][2H], overwritten*/, float* d_gmap_img_fts, void* stash, void* ws,
                         etp_stream_t stream);

/* Device-side graph-input assembly: everything RLTrainer._nav_gmap_variable computes on the host
 * besides the node embeddings (ss_trainer_ETP.py:344-417) -- all-pairs shortest paths over the visited-node graph
 * (GraphMap.update_graph's networkx Dijkstra, graph_utils.py:256-257), nearest front of each ghost (:259-270), the 7-d
 * position features of GraphMap.get_pos_fts (:278-322), step ids, masks and the pairwise distance matrix (:371-387) --
 * from compact per-episode arrays.  One workgroup per episode; <= 64 visited nodes and <= 192 ghosts per episode.
 *   node_pos [B,Nmax,3], node_step [B,Nmax], n_nodes [B], adj [B,Nmax,Nmax] (edge length, < 0 = no edge, symmetric),
 *   ghost_pos [B,Mmax,3] (ghost_aug_pos), n_ghost [B], front_ptr [B,Mmax+1] + front_idx [B,Fmax] (CSR per episode: node
 *   indices of each ghost's fronts, in the reference's list order), cur_node [B], cur_pos [B,3], cur_heading [B] (radians;
 *   heading_from_quaternion stays with the caller).  Outputs padded to G >= 1 + n_nodes + n_ghost entries per episode,
 *   ordered [stop], visited nodes, ghosts: gmap_step_ids [B,G] i64, gmap_masks / gmap_visited_masks [B,G] u8,
 *   gmap_pos_fts [B,G,7] f32, gmap_pair_dists [B,G,G] f32. */
int etp_gmap_assemble(const float* node_pos, const int32_t* node_step, const int32_t* n_nodes, const float* adj,
                      const float* ghost_pos, const int32_t* n_ghost, const int32_t* front_ptr, const int32_t* front_idx,
                      const int32_t* cur_node, const float* cur_pos, const float* cur_heading, int B, int Nmax, int Mmax,
                      int Fmax, int G, int64_t* gmap_step_ids, uint8_t* gmap_masks, uint8_t* gmap_visited_masks,
                      float* gmap_pos_fts, float* gmap_pair_dists, etp_stream_t stream);

/* RLTrainer._vp_feature_variable (ss_trainer_ETP.py:308-342): out[b] = [candidate-view features (K_b rows) ; panorama
 * views whose index is not a candidate's image, in index order], zero-padded to V rows; nav_types 1 for the candidate rows
 * (NULL to skip), view_lens[b] = K_b + #free views (NULL to skip).  cand_fts packed [sum K, F] with cand_ptr [B+1];
 * pano_fts [B,P,F] with pano_batch_stride = P*F, or one shared [P,F] table with stride 0 (pano_angle_fts); cand_mask [B,P]. */
int etp_vp_gather(const float* cand_fts, const int32_t* cand_ptr, const float* pano_fts, int64_t pano_batch_stride,
                  const uint8_t* cand_mask, int B, int P, int F, int V, float* out_fts, int64_t* nav_types, int64_t* view_lens,
                  etp_stream_t stream);

/* Pre-training MLM task for a planner created with cfg.use_lang2visn = 1:
 * GlocalTextPathCMT.forward_mlm (pretrain vilmodel.py:708-754): the text (output of etp_txt_fwd) attends to the graph-node
 * inputs gmap_img_fts + step + position embeddings through forward_lang2visn of every x-layer (:400-411), then
 * BertOnlyMLMHead (:258-299, decoder tied to the word embeddings) on the Nm masked positions and
 * *loss += scale * sum of token cross-entropies (pretrain_cmt.py:141-163; scale = 1/Nm for the reference's .mean()).
 * The masked positions are given as the CSR of a row gather (sel_ptr = 0..Nm, sel_idx = b*L + l, sel_w = 1) and, for the
 * backward, its transpose over the B*L rows.  etp_mlm_bwd returns d txt_embeds and d gmap_img_fts and accumulates every
 * parameter gradient, including the tied decoder's into the word-embedding gradient. */
int64_t etp_mlm_stash_bytes(const etp_planner* p, int B, int L, int G, int Nm);
int64_t etp_mlm_ws_bytes(const etp_planner* p, int B, int L, int G, int Nm);
int etp_mlm_fwd(etp_planner* p, const float* txt_embeds, const uint8_t* txt_masks, const int64_t* gmap_step_ids,
                const float* gmap_img_fts, const float* gmap_pos_fts, const uint8_t* gmap_masks, const int32_t* sel_ptr,
                const int32_t* sel_idx, const float* sel_w, const int64_t* labels, int B, int L, int G, int Nm, float scale,
                float* loss, void* stash, etp_stream_t stream);
int etp_mlm_bwd(etp_planner* p, const float* txt_embeds, const uint8_t* txt_masks, const int64_t* gmap_step_ids,
                const float* gmap_pos_fts, const uint8_t* gmap_masks, const int32_t* selT_ptr, const int32_t* selT_idx,
                const float* selT_w, int B, int L, int G, int Nm, float* d_txt_embeds, float* d_gmap_img_fts, void* stash,
                void* ws, etp_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * hipGraph helpers (launch-bound inner loops are captured once and replayed) and timing.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct etp_graph etp_graph;
int etp_stream_create(etp_stream_t* out);
/* level < 0: the lowest priority the device offers (for leaf work such as weight gradients, so that the dependent chain's
 * workgroups are dispatched first whenever both wait for a CU), 0: default, > 0: highest. */
int etp_stream_create_prio(etp_stream_t* out, int level);
int etp_stream_destroy(etp_stream_t s);
int etp_stream_sync(etp_stream_t s);
/* make `to` wait for the work enqueued so far on `from` (fork / join of parallel branches; capturable) */
int etp_stream_after(etp_stream_t from, etp_stream_t to);
int etp_graph_begin(etp_stream_t s);
int etp_graph_end(etp_stream_t s, etp_graph** out);
/* ----------------------------------------------------------------------------------------------------
 * Data-parallel gradient mean: replaces DistributedDataParallel(self.policy.net) of
 * ss_trainer_ETP.py:208-212 / pretrain utils/misc.py:52-65 for the planner's flat gradient arena.  One communicator per
 * process (one process per GPU); RCCL over xGMI, bound at run time.  A bucket is reduced IN PLACE as reduce-scatter (sum) ->
 * 1/world scaling of the rank's own slice -> all-gather on a private communication stream ordered after `producer`.
 * comm_dtype ETP_F32 (default, DDP's numerics) or ETP_BF16 (opt-in: half the xGMI bytes, bf16 sums; needs
 * max_bucket_elems for the packed staging buffer).  Rank 0 creates the 128-byte id, the caller distributes it (any
 * out-of-band channel, e.g. torch.distributed's store) and every rank calls etp_allreduce_init with it.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct etp_comm etp_comm;
int etp_allreduce_unique_id(void* id_out_128_bytes);
int etp_allreduce_init(etp_comm** out, const void* unique_id, int rank, int world, int comm_dtype, int64_t max_bucket_elems);
int etp_allreduce_bucket_ready(etp_comm* c, float* grads, int64_t n, etp_stream_t producer);
/* The slice arithmetic etp_allreduce_bucket_ready applies to a bucket of n fp32 gradients over `world` ranks, as a pure host
 * function (no communicator, no GPU): out[0] = elements per rank slice, out[1] = elements covered by reduce-scatter + all-gather,
 * out[2] = tail elements that go through one all-reduce (fp32 transport only), out[3] = bf16 elements staged (bf16 transport only;
 * the pad [n, out[3]) is zeroed).  DDP's buckets (ss_trainer_ETP.py:208-212) have no counterpart of this: it exists so that the
 * multi-rank arithmetic can be unit-tested for worlds 2 / 4 / 8 on a one-GPU box. */
int etp_allreduce_plan(int64_t n, int world, int comm_dtype, int64_t* out);
int64_t etp_allreduce_staging_elems(int64_t max_bucket_elems, int world);
/* Row-sparse mean of a table gradient [n_rows, row_len] (the word-embedding table: a step touches <= B*L of its 30 522 /
 * 250 002 rows; DDP would all-reduce the dense table, ss_trainer_ETP.py:208-212).  ids[0, n_ids) = rows this rank touched (any
 * order, repeats allowed); `capacity` >= n_ids is the rank-INDEPENDENT block size (e.g. B * max_txt_len; the reference's collate
 * pads to the per-batch maximum, pretrain_src/pretrain_src/data/tasks.py:322-364, so the token count differs across ranks --
 * the capacity must not).  pack (repeats masked on the device) -> ncclAllGather of (ids, rows) -> scatter-add x 1/world, on the
 * communicator's own stream after `producer`: the same communicator and stream as the dense buckets. */
int etp_allreduce_gather_rows(etp_comm* c, float* table, int64_t n_rows, int64_t row_len, const int64_t* ids, int64_t n_ids,
                              int64_t capacity, etp_stream_t producer);
/* 1 when librccl can be bound in this process (ranks agree on it before any of them enters etp_allreduce_init) */
int etp_allreduce_available(void);
int etp_allreduce_wait(etp_comm* c, etp_stream_t consumer);
/* Host-side poll: 1 when everything issued on the communicator's stream has completed, 0 while work is in flight.  With
 * etp_allreduce_abort (ncclCommAbort; afterwards only etp_allreduce_destroy is valid) this lets a first-contact self-test give
 * up on a collective that never completes instead of blocking the job: etpnav_amd/dp.py NativeComm.self_test.  (DDP has no
 * counterpart; its watchdog is NCCL_ASYNC_ERROR_HANDLING inside torch.distributed, ss_trainer_ETP.py:208-212.) */
int etp_allreduce_idle(etp_comm* c);
int etp_allreduce_abort(etp_comm* c);
int etp_allreduce_destroy(etp_comm* c);
int etp_allreduce_rank(const etp_comm* c);
int etp_allreduce_world(const etp_comm* c);
etp_stream_t etp_allreduce_stream(const etp_comm* c);

/* Explicit graph construction (replaces stream capture for the three-stream step: hipStreamEndCapture crashes on ROCm 7.2
 * once two side streams joined a capture).  Between etp_rec_begin() and etp_rec_end() every entry point of this library
 * that takes a stream is RECORDED as hipGraph kernel / memset nodes instead of being issued: per-stream order and the event
 * edges of etp_stream_after / the planner's internal forks and joins become graph dependencies.  The result is launched,
 * timed and destroyed like a captured graph.  One recording at a time; record from one host thread. */
int etp_rec_begin(void);
int etp_rec_end(etp_graph** out, int64_t* n_kernels, int64_t* n_edges);
int etp_rec_abort(void);
int etp_graph_launch(etp_graph* g, etp_stream_t s);
int etp_graph_destroy(etp_graph* g);
int etp_memset_async(void* p, int value, int64_t bytes, etp_stream_t s);
/* HIP-event timing on the given stream: elapsed ms of `iters` graph replays. */
int etp_graph_time(etp_graph* g, etp_stream_t s, int iters, float* ms_out);

/* Per-launch HIP-event timing of the MFMA GEMM family (events recorded on the launch stream; eager launches only —
 * do not enable during graph capture).  etp_prof_report synchronises the recorded events and returns one entry per
 * kernel instantiation: launches, summed ms, summed algorithmic FLOPs and algorithmic bytes (A+B+C once). */
typedef struct etp_prof_entry {
  char name[96];
  int64_t launches;
  double ms, flops, bytes;
} etp_prof_entry;
/* Per-launch HIP-event timing of EVERY kernel the library issues (measurement aid, tools/chain_budget.py): one text line per
 * launch in launch order, "<us>\t<grid>\t<block>\t<stream>\t<kernel name>".  Events sit on the launch stream: issue the step on a
 * single stream when each pair should bracket its kernel alone. */
int etp_ktime_enable(int on);
int etp_ktime_reset(void);
int64_t etp_ktime_report(char* buf, int64_t cap);
int etp_prof_enable(int on);
int etp_prof_reset(void);
/* Bracket only launches whose name contains `name_part` (NULL / "" = all): with the step on its three streams every event pair is
 * a barrier packet the queues must process, so timing one kernel class in-step should not bracket the other 150 launches. */
int etp_prof_filter(const char* name_part);
/* Phase probe of the LDS-DMA GEMM kernels (measurement aid, tools/gemm_phase_probe.py; the reference has only host
 * time.time() counters, pretrain_src/pretrain_src/train_r2r.py:227,299-317).  With a device buffer of
 * max_launches x 4096 x 8 uint64 installed, every following eager GEMM launch of <= 4096 workgroups records per workgroup
 * { s_memrealtime at entry, at exit; s_memtime at entry, first slab visible, end of reduction, end of epilogue;
 *   XCC_ID << 32 | HW_ID; slabs } in launch order; etp_gemm_probe_meta names launch i (dims = grid, M, N, K).
 * dev_buf == NULL switches the probe off. */
int etp_gemm_probe_enable(uint64_t* dev_buf, int64_t max_launches);
int64_t etp_gemm_probe_count(void);
int etp_gemm_probe_meta(int64_t i, char* name, int cap, int32_t* dims);
int etp_prof_report(etp_prof_entry* out, int cap);
/* Device-side time stamp (measurement aid, tools/chain_waits.py): a one-thread kernel on `stream` stores s_memrealtime (100 MHz,
 * chip-wide) into *slot when the stream reaches it -- the un-profiled view of where a stream waits (the reference's single
 * backward, ss_trainer_ETP.py:504, has no joins of its own: every wait found is ours). */
int etp_stamp(uint64_t* slot, etp_stream_t stream);
/* Stamp sink: with a device buffer of `cap` uint64 installed, the planner entry points (and etp_stamp_mark, for the caller's own
 * points) append one such stamp per marked point of their issue order -- text / navigation / panorama layer boundaries, forks and
 * joins -- in enqueue order; etp_stamp_tag(i) names stamp i (tags: tools/chain_waits.py).  dev_buf == NULL removes the sink. */
int etp_stamp_sink(uint64_t* dev_buf, int64_t cap);
int64_t etp_stamp_count(void);
int etp_stamp_tag(int64_t i);
int etp_stamp_mark(etp_stream_t stream, int tag);

#ifdef __cplusplus
}
#endif
#endif /* ETPNAV_HIP_H */
