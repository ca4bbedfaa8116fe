// Internal launcher prototypes (definitions in norm.hip / embed.hip / gemm.hip).
#pragma once
#include "common.h"

namespace etp {

struct PanoEmbedParams {
  const float *g_img, *b_img, *g_dep, *b_dep, *w_loc, *bias_loc, *g_loc, *b_loc, *nav_emb, *type1, *g_out, *b_out;
};
struct PanoEmbedGrads {
  float *g_img, *b_img, *g_dep, *b_dep, *w_loc, *bias_loc, *g_loc, *b_loc, *nav_emb, *type1, *g_out, *b_out;
};

int ln_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* stats, int M, int H, float eps,
           hipStream_t st);
int ln_bwd(int dtype, const void* dy, const void* x, const float* stats, const float* gamma, const void* add, void* dx,
           float* dgamma, float* dbeta, int M, int H, hipStream_t st);
int ln_fwd_s(int dtype, const float* x, const float* gamma, const float* beta, float* y, void* yt, float* stats, int M, int H,
             float eps, hipStream_t st);
// part != NULL: two-stage dgamma/dbeta reduction -- the kernel writes per-block column sums to `part`
// (ln_bwd_part_bytes(M, H) bytes) and the caller issues ln_part_reduce(part, M, H, dgamma, dbeta) anywhere later (it is a
// leaf of the backward graph); part == NULL: per-block atomics straight into dgamma / dbeta.
constexpr int LN_BWD_MAX_BLOCKS = 1024;
size_t ln_bwd_part_bytes(int M, int H);
int ln_bwd_s(int dtype, const float* dy, const float* x, const float* stats, const float* gamma, const float* add, float* dx,
             void* dxt, float* dgamma, float* dbeta, int M, int H, hipStream_t st, Drop drop = drop_none(), float* part = nullptr);
int ln_part_reduce(const float* part, int M, int H, float* dgamma, float* dbeta, hipStream_t st);
int softmax_fwd(int dtype, void* S, const uint8_t* keymask, const float* dist, const float* sp_w, const float* sp_b, int B,
                int nh, int Lq, int Lk, int ldS, int mask_mode, hipStream_t st);
int softmax_bwd(int dtype, const void* P, void* dP, const float* dist, float* d_sp_w, float* d_sp_b, int B, int nh, int Lq,
                int Lk, int ldS, hipStream_t st);

int text_embed_fwd(int dtype, const int64_t* ids, const float* word, const float* pos, const float* type0, const float* gamma,
                   const float* beta, float* y, void* yt, float* stats, int B, int L, int H, float eps, hipStream_t st,
                   Drop drop = drop_none());
int text_embed_bwd(int dtype, const float* dy, const int64_t* ids, const float* word, const float* pos, const float* type0,
                   const float* gamma, const float* stats, float* dword, float* dpos, float* dtype0, float* dgamma, float* dbeta,
                   int B, int L, int H, hipStream_t st, Drop drop = drop_none());
int pano_embed_fwd(int dtype, const void* a, const void* d, const float* loc, const int64_t* nav, const PanoEmbedParams& p,
                   float* y, float* stats, int M, int H, hipStream_t st, Drop drop = drop_none());
int pano_embed_bwd(int dtype, const float* dy, const void* a, const void* d, const float* loc, const int64_t* nav,
                   const float* stats, const PanoEmbedParams& p, const PanoEmbedGrads& g, void* da, void* dd, int M, int H,
                   hipStream_t st, Drop drop = drop_none());
int gmap_embed_fwd(int dtype, const float* img, const int64_t* step_ids, const float* pos, const float* step_emb,
                   const float* w_pos, const float* b_pos, const float* gamma, const float* beta, float* x, void* xt, float* stats,
                   int M, int H, int PK, hipStream_t st);
int gmap_embed_bwd(int dtype, const float* dx, const int64_t* step_ids, const float* pos, const float* w_pos, const float* b_pos,
                   const float* gamma, const float* stats, float* d_step_emb, float* d_w_pos, float* d_b_pos, float* dgamma,
                   float* dbeta, int M, int H, int PK, hipStream_t st);
int sap_tail_fwd(int dtype, const void* r, const float* gamma, const float* beta, const float* w2, const float* b2,
                 const uint8_t* visited, const uint8_t* valid, float* logits, float* stats, int M, int H, hipStream_t st,
                 Drop drop = drop_none());
int sap_tail_bwd(int dtype, const float* dlogits, const void* r, const float* gamma, const float* beta, const float* w2,
                 const float* stats, const uint8_t* visited, const uint8_t* valid, void* dz, float* dgamma, float* dbeta,
                 float* dw2, float* db2, int M, int H, hipStream_t st, Drop drop = drop_none());
int sap_ce(const float* logits, const int64_t* labels, float* loss, float* dlogits, int B, int G, float scale, long ignore_index,
           hipStream_t st);
int gather_sum(int dtype, const void* src, const int32_t* ptr, const int32_t* idx, const float* w, void* out, int N, int H,
               int accumulate, hipStream_t st);
int colsum(int dtype, const void* dy, long ld, float* db, int M, int N, hipStream_t st);
int cast_f32_to_bf16(const float* src, void* dst, long n, hipStream_t st);
int cast_drop(int dtype, const float* src, void* dst, long n, Drop drop, hipStream_t st);   // dst(T) = dropout(src)
int cast_bf16_to_f32(const void* src, float* dst, long n, float scale, hipStream_t st);
int scale_f32(float* p, long n, float scale, hipStream_t st);
int copy_f32(const float* src, float* dst, long n, hipStream_t st);
int vocab_ce(int dtype, const float* logits, const int64_t* labels, float* loss, void* dl, int Nm, int V, int ldv, float scale,
             hipStream_t st);
int gelu_bwd_inplace(int dtype, void* d, const void* z, long n, hipStream_t st);
int zero_f32(float* dst, long n, hipStream_t st);
int repeat_block(const void* src, void* dst, long bytes, int T, hipStream_t st);            // dst[t][.] = src[.], t < T
int sum_steps(int dtype, const void* src, void* dst, long n, int steps, hipStream_t st);   // dst[i] = sum_t src[t][i], fp32 accumulation

// optim.hip: fused AdamW (+ bf16 shadow refresh + gradient zeroing) and the gradient norm / non-finite scan
int adamw_step(float* p, float* g, float* m, float* v, void* shadow, long n_shadow, const uint8_t* decay_mask, long n,
               const etp_adamw_cfg& c, const float* sumsq, const int32_t* skip, int zero_grads, hipStream_t st,
               int32_t* step_dev = nullptr);
int grad_sqnorm(const float* g, long n, float* sumsq, int32_t* nonfinite, hipStream_t st, const uint8_t* mask = nullptr);

// attention = batched MFMA GEMMs + masked softmax (planner.hip); head dim 64, heads interleaved in the row
struct AttnBuf {
  const void* Q; long ldq; const void* K; long ldk; const void* V; long ldv;
  int B, Lq, Lk, ldS;
  const uint8_t* keymask; int mask_mode; const float* dist; const float* sp_w; const float* sp_b;
  void* Pd = nullptr;      // unfused path only: second [B,heads,Lq,ldS] buffer for the DROPPED probabilities (training mode)
  const void* O = nullptr; long ldo = 0;   // backward only: the forward output (streaming kernels: D = rowsum(dO * O))
  // > 0 (round 6, batched rollout on the text K/V cache): episode b reads the keys / values / key mask of instruction b % kv_mod
  // (Q, ctx and every gradient stay per episode).  Register-resident kernels only (attn_rows.hip); the other families refuse it.
  int kv_mod = 0;
};
// shapes the fused kernels do not take (the batched-GEMM path runs them and needs AttnBuf::Pd for dropout)
// (bf16 with Lq or Lk > 128 normally runs the streaming kernels instead; the second buffer is still planned so that
// ETP_ATTN_FLASH=0 / an unaligned operand can fall back)
inline bool attn_needs_unfused(int dt, int Lq, int Lk) { return Lq > 128 || Lk > 128 || (dt == ETP_F32 && (Lq > 64 || Lk > 64)); }
// streaming ("flash") kernels for long key / query axes, bf16 (attn.hip): P is not materialised, the front of the P buffer
// holds lse and D (2 fp32 per query row)
bool attn_flash_ok(int dt, const AttnBuf& a, long ldc);
int attn_flash_fwd(int nh, const AttnBuf& a, void* P, void* ctx, long ldc, float alpha, hipStream_t st, Drop drop);
int attn_flash_bwd(int nh, const AttnBuf& a, const void* P, const void* dctx, long ldd, void* dQ, long lddq, void* dK, long lddk,
                   void* dV, long lddv, float alpha, hipStream_t st, Drop drop);
// register-resident kernels (attn_rows.hip): bf16, Lq and Lk <= 128 -- every R2R-CE shape of the planner.  Like the streaming
// kernels they keep lse (1 fp32 per query row) in the front of the P buffer and recompute the probabilities in backward.
bool attn_rows_ok(int dt, const AttnBuf& a, long ldc);
// qkv_x != NULL (round 6, self-attention): Q / K / V (a.Q / a.K / a.V, written) = qkv_x[B*L, nh*64] . qkv_w[3*nh*64][ldw]^T + qkv_b are
// computed in the kernel's prologue instead of read; see attn_rows.hip
bool attn_rows_qkv_ok(int nh, const AttnBuf& a, const void* X, long ldx, const void* W, long ldw);
int attn_rows_fwd(int nh, const AttnBuf& a, void* P, void* ctx, long ldc, float alpha, hipStream_t st, Drop drop,
                  const void* qkv_x = nullptr, long qkv_ldx = 0, const void* qkv_w = nullptr, long qkv_ldw = 0,
                  const float* qkv_b = nullptr);
// proj_w != NULL (round 6): `dctx` is the gradient of the OUT-PROJECTION's output [B*Lq, proj_k] and the kernel computes
// dctx_head = dctx . proj_w[0:proj_k, h*64 : h*64+64] itself (proj_w = the projection's weight [proj_k (out)][ldw]); see attn_rows.hip
bool attn_rows_proj_ok(int Kp, const void* W, long ldw);
int attn_rows_bwd(int nh, const AttnBuf& a, const void* P, const void* dctx, long ldd, void* dQ, long lddq, void* dK, long lddk,
                  void* dV, long lddv, float alpha, float* d_sp_w, float* d_sp_b, hipStream_t st, Drop drop,
                  const void* proj_w = nullptr, long proj_ldw = 0, int proj_k = 0);
int drop_rows(int dtype, const void* src, void* dst, long rows, int Lk, int ldS, Drop drop, hipStream_t st);
// fused single-kernel variants (attn.hip) for Lq, Lk <= 128
bool attn_fused_ok(int dt, const AttnBuf& a, long ldc);
int attn_fused_fwd(int dt, int nh, const AttnBuf& a, void* P, void* ctx, long ldc, float alpha, hipStream_t st, Drop drop);
int attn_fused_bwd(int dt, int nh, const AttnBuf& a, const void* P, const void* dctx, long ldd, void* dQ, long lddq, void* dK,
                   long lddk, void* dV, long lddv, float alpha, float* d_sp_w, float* d_sp_b, hipStream_t st, Drop drop);
// drop = dropout on the attention probabilities (vilmodel_cmt.py:127,346; MHA dropout): inside the fused kernels, or via
// drop_rows + AttnBuf::Pd on the batched-GEMM path
int attn_fwd_impl(int dt, int nh, const AttnBuf& a, void* P, void* ctx, long ldc, float alpha, hipStream_t st,
                  Drop drop = drop_none());
int attn_bwd_impl(int dt, int nh, const AttnBuf& a, const void* P, const void* dctx, long ldd, void* dP, void* dQ, long lddq,
                  void* dK, long lddk, void* dV, long lddv, float alpha, float* d_sp_w, float* d_sp_b, hipStream_t st,
                  Drop drop = drop_none());

}  // namespace etp
