// Fused small-sequence attention (head dim 64) forward / backward for gfx950.
//
//   ctx = softmax(alpha * Q K^T + keymask + (w*dist + b)) V        BertSelfAttention.forward vilmodel_cmt.py:103-141,
//                                                                  BertOutAttention.forward :325-352,
//                                                                  nn.MultiheadAttention (common/transformer.py:138)
//
// One 256-thread workgroup (4 wavefronts, 2x2) owns one (batch, head): Q, K, V (and dO, P, dS in backward) live in LDS
// as padded row-major tiles for the whole kernel, every product is an MFMA tile product over those LDS tiles, and the
// softmax never leaves the chip.  Replaces 3 (forward) / 6 (backward) launches of the batched-GEMM path and their
// HBM round trips of the score matrix; used when Lq, Lk <= 128 (every R2R-CE shape of the planner).  Longer
// sequences (RxR L=512) take the unfused path in planner.hip, which saves P in the same layout.
//
// LDS tiles are "natural": [rows][COLS elements] with a 32-byte row pad.  The same tile serves as a row operand
// (ds_read_b128 of 8 consecutive k) and as a transposed operand (ds_read_b64_tr_b16 / ds_read_b32 down the k rows),
// which is what lets P, dS, Q, K, dO be staged once and consumed by products that reduce over either of their axes.
#include <stdlib.h>
#include <string.h>

#include "kernels.h"

namespace etp {

template <typename T> struct AFrag;
template <> struct AFrag<bf16_t> { uint4 v; };
template <> struct AFrag<float> { float4 lo, hi; };

__device__ __forceinline__ void amma(f32x4_t& acc, const AFrag<bf16_t>& a, const AFrag<bf16_t>& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a.v), __builtin_bit_cast(bf16x8_t, b.v), acc, 0, 0, 0);
}
__device__ __forceinline__ void amma(f32x4_t& acc, const AFrag<float>& a, const AFrag<float>& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.x, b.lo.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.y, b.lo.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.z, b.lo.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.w, b.lo.w, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.x, b.hi.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.y, b.hi.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.z, b.hi.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.w, b.hi.w, acc, 0, 0, 0);
}

template <typename T, int COLS> struct Nat {
  static constexpr int EPC = 16 / (int)sizeof(T);
  static constexpr int CPR = COLS / EPC;
  static constexpr int PITCH = COLS * (int)sizeof(T) + 32;
};

// global [rows_valid][cols_valid] (row stride ld) -> LDS natural tile [ROWS][COLS]; zero outside.  Branch-free loads.
template <typename T, int ROWS, int COLS>
__device__ __forceinline__ void nat_load(char* lds, const T* __restrict__ g, long ld, int rows_valid, int cols_valid, int tid) {
  using N = Nat<T, COLS>;
  constexpr int TOTAL = ROWS * N::CPR;
#pragma unroll
  for (int j = 0; j < (TOTAL + 255) / 256; ++j) {
    const int q = tid + j * 256;
    if (TOTAL % 256 != 0 && q >= TOTAL) break;          // only the last pass of a 96-wide tile is partial
    const int r = q / N::CPR, c = (q % N::CPR) * N::EPC;
    const bool ok = r < rows_valid && c < cols_valid;
    const int rc = min(r, rows_valid - 1), cc = min(c, cols_valid - N::EPC);
    uint4 v = *reinterpret_cast<const uint4*>(g + (long)rc * ld + cc);
    v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
    *reinterpret_cast<uint4*>(lds + r * N::PITCH + (q % N::CPR) * 16) = v;
  }
}

// 8 k-values (k = k0 + 8*(lane>>4) + e) of tile row row16 + (lane&15)
template <typename T, int PITCH>
__device__ __forceinline__ void frag_row(AFrag<T>& f, const char* tile, int row16, int k0, int lane) {
  const char* p = tile + (row16 + (lane & 15)) * PITCH + (k0 + (lane >> 4) * 8) * (int)sizeof(T);
  if constexpr (sizeof(T) == 2) {
    f.v = *reinterpret_cast<const uint4*>(p);
  } else {
    f.lo = *reinterpret_cast<const float4*>(p);
    f.hi = *reinterpret_cast<const float4*>(p + 16);
  }
}
// same 8 k-values of tile COLUMN col16 + (lane&15), k running down the tile rows
template <typename T, int PITCH>
__device__ __forceinline__ void frag_tr(AFrag<T>& f, const char* tile, int col16, int k0, int lane) {
  const int i = lane & 15, g = lane >> 4;
  if constexpr (sizeof(T) == 2) {
    const char* p = tile + (k0 + g * 8 + (i >> 2)) * PITCH + (col16 + (i & 3) * 4) * 2;
    typedef short4_t __attribute__((address_space(3))) * lds_s4;
    const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(p));
    const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(p + 4 * PITCH));
    const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
    f.v = make_uint4(a.x, a.y, b.x, b.y);
  } else {
    const char* p = tile + (k0 + g * 8) * PITCH + (col16 + i) * 4;
    f.lo = make_float4(*reinterpret_cast<const float*>(p), *reinterpret_cast<const float*>(p + PITCH),
                       *reinterpret_cast<const float*>(p + 2 * PITCH), *reinterpret_cast<const float*>(p + 3 * PITCH));
    f.hi = make_float4(*reinterpret_cast<const float*>(p + 4 * PITCH), *reinterpret_cast<const float*>(p + 5 * PITCH),
                       *reinterpret_cast<const float*>(p + 6 * PITCH), *reinterpret_cast<const float*>(p + 7 * PITCH));
  }
}

// acc[a][b] += sum_k A(a_row0 + 16a + ., k) * B(b_row0 + 16b + ., k);  TRx selects which tile axis is k
template <typename T, int MT, int NT, int KSTEPS, bool TRA, bool TRB, int PA, int PB>
__device__ __forceinline__ void tile_mma(f32x4_t (&acc)[MT][NT], const char* A, int a_row0, const char* B, int b_row0, int lane) {
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) {
    AFrag<T> fa[MT], fb[NT];
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      if constexpr (TRA) frag_tr<T, PA>(fa[a], A, a_row0 + a * 16, s * 32, lane);
      else frag_row<T, PA>(fa[a], A, a_row0 + a * 16, s * 32, lane);
    }
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      if constexpr (TRB) frag_tr<T, PB>(fb[b], B, b_row0 + b * 16, s * 32, lane);
      else frag_row<T, PB>(fb[b], B, b_row0 + b * 16, s * 32, lane);
    }
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b) amma(acc[a][b], fa[a], fb[b]);
  }
}

template <int MT, int NT> __device__ __forceinline__ void acc_zero(f32x4_t (&acc)[MT][NT]) {
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
}

// accumulators (C layout: row = 4*(lane>>4)+r, col = lane&15 per 16x16 tile) -> fp32 LDS tile [.][CP]
template <int MT, int NT>
__device__ __forceinline__ void acc_to_lds(const f32x4_t (&acc)[MT][NT], float* ct, int CP, int row0, int col0, float scale, int lane) {
  const int i = lane & 15, g = lane >> 4;
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) ct[(row0 + a * 16 + g * 4 + r) * CP + col0 + b * 16 + i] = acc[a][b][r] * scale;
}

// fp32 LDS tile [rows][68] -> global rows (64 columns each) as 16-byte vectors
template <typename T, int ROWS>
__device__ __forceinline__ void store_rows64(const float* ct, T* __restrict__ g, long ld, int rows_valid, int tid) {
  constexpr int CP = 68;
#pragma unroll
  for (int j = 0; j < ROWS * 8 / 256; ++j) {
    const int q = tid + j * 256;
    const int r = q >> 3, c = (q & 7) * 8;
    if (r < rows_valid) {
      const float4 x0 = *reinterpret_cast<const float4*>(ct + r * CP + c), x1 = *reinterpret_cast<const float4*>(ct + r * CP + c + 4);
      float o0[4] = {x0.x, x0.y, x0.z, x0.w}, o1[4] = {x1.x, x1.y, x1.z, x1.w};
      store4(g + (long)r * ld + c, o0);
      store4(g + (long)r * ld + c + 4, o1);
    }
  }
}

struct AttnKArgs {
  const void *Q, *K, *V; long ldq, ldk, ldv;
  void* P; int ldS;
  void* ctx; long ldc;
  int nh, Lq, Lk;
  int nq;                        // forward: workgroups per (batch, head) along the query axis
  Drop drop;                     // dropout on the probabilities; element index = ((b*heads+h)*Lq + q)*Lk + k
  const uint8_t* keymask; int mask_mode; const float* dist; const float* sp_w; const float* sp_b;
  float alpha;
  // backward
  const void* dctx; long ldd;
  void *dQ, *dK, *dV; long lddq, lddk, lddv;
  float *d_sp_w, *d_sp_b;
};

template <typename T, int BQ, int BKV> struct AttnFwdLds {
  static constexpr int PQ = Nat<T, 64>::PITCH, PP = Nat<T, BKV>::PITCH;
  static constexpr int QK_BYTES = (BQ + BKV) * PQ, P_BYTES = BQ * PP, O_BYTES = BQ * 68 * 4;
  static constexpr int R01 = QK_BYTES > P_BYTES ? QK_BYTES : P_BYTES;
  static constexpr int R0 = R01 > O_BYTES ? R01 : O_BYTES;   // q,k tiles -> P tile -> fp32 output staging
  static constexpr int V_OFF = R0, RS_OFF = V_OFF + BKV * PQ;
  static constexpr int TOTAL = RS_OFF + 4 * BQ * 4;          // + row max / row sum exchange [2][2][BQ]
};

// scores never leave the registers: the softmax runs on the MFMA accumulators (row statistics by 16-lane shuffles and a
// tiny LDS exchange between the two wave columns), P is written straight into the LDS tile that feeds P.V.
template <typename T, int BQ, int BKV>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnKArgs a) {
  using L = AttnFwdLds<T, BQ, BKV>;
  constexpr int PQ = L::PQ, PP = L::PP;
  constexpr int MT = BQ / 32, NT = BKV / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* qt = smem; char* kt = smem + BQ * PQ; char* pt = smem; char* vt = smem + L::V_OFF;
  float* rmax = reinterpret_cast<float*>(smem + L::RS_OFF);   // [2][BQ]
  float* rsum = rmax + 2 * BQ;                                 // [2][BQ]
  float* ct = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  const int i = lane & 15, g = lane >> 4;
  // one workgroup per (batch, head, block of BQ queries): long query axes are split over workgroups (K/V re-read from L2)
  const int bh = blockIdx.x / a.nq, q0 = (blockIdx.x % a.nq) * BQ;
  const int b = bh / a.nh, h = bh % a.nh;
  const int Lq = min(BQ, a.Lq - q0);                     // valid query rows of this workgroup
  const T* Qg = reinterpret_cast<const T*>(a.Q) + ((long)b * a.Lq + q0) * a.ldq + h * 64;
  const T* Kg = reinterpret_cast<const T*>(a.K) + (long)b * a.Lk * a.ldk + h * 64;
  const T* Vg = reinterpret_cast<const T*>(a.V) + (long)b * a.Lk * a.ldv + h * 64;
  T* Pg = reinterpret_cast<T*>(a.P) + ((long)bh * a.Lq + q0) * a.ldS;
  T* Cg = reinterpret_cast<T*>(a.ctx) + ((long)b * a.Lq + q0) * a.ldc + h * 64;

  nat_load<T, BQ, 64>(qt, Qg, a.ldq, Lq, 64, tid);
  nat_load<T, BKV, 64>(kt, Kg, a.ldk, a.Lk, 64, tid);
  nat_load<T, BKV, 64>(vt, Vg, a.ldv, a.Lk, 64, tid);
  // additive key mask / validity of this lane's NT columns (one per 16-column tile), loaded once
  const uint8_t* km = a.keymask ? a.keymask + (long)b * a.Lk : nullptr;
  float kadd[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int col = wc * (BKV / 2) + n * 16 + i;
    float v = 0.f;
    if (col >= a.Lk) v = -INFINITY;
    else if (km && !km[col]) v = a.mask_mode ? -INFINITY : -10000.0f;
    kadd[n] = v;
  }
  const float w = a.sp_w ? a.sp_w[0] : 0.f, b0 = a.sp_b ? a.sp_b[0] : 0.f;
  __syncthreads();
  f32x4_t sc[MT][NT];
  acc_zero(sc);
  tile_mma<T, MT, NT, 2, false, false, PQ, PQ>(sc, qt, wr * (BQ / 2), kt, wc * (BKV / 2), lane);
  // scores + masks, row max over this wave's half of the keys
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wr * (BQ / 2) + m * 16 + g * 4 + r;
      const float* d = (a.dist && row < Lq) ? a.dist + ((long)b * a.Lq + q0 + row) * a.Lk : nullptr;
      float mx = -INFINITY;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int col = wc * (BKV / 2) + n * 16 + i;
        float v = sc[m][n][r] * a.alpha + kadd[n];
        if (d && col < a.Lk) v += w * d[col] + b0;
        sc[m][n][r] = v;
        mx = fmaxf(mx, v);
      }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      if (i == 0) rmax[wc * BQ + row] = mx;
    }
  __syncthreads();   // also: every wave is done with the q/k tiles, the P tile may overwrite them
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wr * (BQ / 2) + m * 16 + g * 4 + r;
      const float mx = fmaxf(rmax[row], rmax[BQ + row]);
      float sum = 0.f;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const float e = __expf(sc[m][n][r] - mx);
        sc[m][n][r] = e;
        sum += e;
      }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) sum += __shfl_xor(sum, o, 64);
      if (i == 0) rsum[wc * BQ + row] = sum;
    }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wr * (BQ / 2) + m * 16 + g * 4 + r;
      const float inv = (row < Lq) ? 1.0f / (rsum[row] + rsum[BQ + row]) : 0.f;   // unused query rows -> P = 0
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int col = wc * (BKV / 2) + n * 16 + i;
        Elem<T>::st(reinterpret_cast<T*>(pt + row * PP) + col, sc[m][n][r] * inv);
      }
    }
  __syncthreads();
  {  // save P for backward: [Lq][ldS] rows copied out of the LDS tile as 16-byte vectors (pad columns are 0)
    using N = Nat<T, BKV>;
    for (int q = tid; q < BQ * N::CPR; q += 256) {
      const int r = q / N::CPR, c = (q % N::CPR) * N::EPC;
      if (r < Lq && c < a.ldS)
        *reinterpret_cast<uint4*>(Pg + (long)r * a.ldS + c) = *reinterpret_cast<const uint4*>(pt + r * PP + (q % N::CPR) * 16);
    }
  }
  if (a.drop.p > 0.f) {   // attention dropout (vilmodel_cmt.py:127): P.V uses the dropped probabilities, backward keeps P
    __syncthreads();      // the undropped tile has been copied out
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wr * (BQ / 2) + m * 16 + g * 4 + r;
        const float inv = (row < Lq) ? 1.0f / (rsum[row] + rsum[BQ + row]) : 0.f;
        const uint32_t rbase = ((uint32_t)bh * a.Lq + q0 + row) * (uint32_t)a.Lk;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int col = wc * (BKV / 2) + n * 16 + i;
          Elem<T>::st(reinterpret_cast<T*>(pt + row * PP) + col,
                      sc[m][n][r] * inv * drop_mult(a.drop.seed, rbase + col, a.drop.p, a.drop.inv_keep));
        }
      }
    __syncthreads();
  }
  f32x4_t oc[MT][2];
  acc_zero(oc);
  tile_mma<T, MT, 2, BKV / 32, false, true, PP, PQ>(oc, pt, wr * (BQ / 2), vt, wc * 32, lane);
  __syncthreads();   // P tile dead -> fp32 output staging
  acc_to_lds(oc, ct, 68, wr * (BQ / 2), wc * 32, 1.0f, lane);
  __syncthreads();
  store_rows64<T, BQ>(ct, Cg, a.ldc, Lq, tid);
}

// Backward LDS plan (bf16 128x128: 77 KB -> two workgroups per CU): three slots that are reused as operands die:
//   S1: dO (dP, dV)  -> Q (dK)      S2: V (dP) -> K (dQ)      S3: P (softmax bwd, dV) -> dS (dK, dQ)
template <typename T, int BQ, int BKV> struct AttnBwdLds {
  static constexpr int PQ = Nat<T, 64>::PITCH, PP = Nat<T, BKV>::PITCH;
  static constexpr int S1_OFF = 0, S2_OFF = BQ * PQ, S3_OFF = S2_OFF + BKV * PQ, RD_OFF = S3_OFF + BQ * PP;
  static constexpr int STAGE_ROWS = BQ > BKV ? BQ : BKV;
  static constexpr int TILES = RD_OFF, STAGE = STAGE_ROWS * 68 * 4;
  static constexpr int RD = TILES > STAGE ? TILES : STAGE;                 // row-dot exchange sits behind tiles AND staging
  static constexpr int TOTAL = RD + 2 * BQ * 4 + 64;
};

template <typename T, int BQ, int BKV>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const AttnKArgs a) {
  using L = AttnBwdLds<T, BQ, BKV>;
  constexpr int PQ = L::PQ, PP = L::PP;
  constexpr int MTq = BQ / 32, MTk = BKV / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *s1 = smem + L::S1_OFF, *s2 = smem + L::S2_OFF, *s3 = smem + L::S3_OFF;
  float* rowdot = reinterpret_cast<float*>(smem + L::RD);      // [2][BQ]
  float* red = rowdot + 2 * BQ;                                  // [16] block reduction of the sprel gradients
  float* ct = reinterpret_cast<float*>(smem);                    // output staging (aliases the tiles once they are dead)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  const int i = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / a.nh, h = blockIdx.x % a.nh;
  const T* Qg = reinterpret_cast<const T*>(a.Q) + (long)b * a.Lq * a.ldq + h * 64;
  const T* Kg = reinterpret_cast<const T*>(a.K) + (long)b * a.Lk * a.ldk + h * 64;
  const T* Vg = reinterpret_cast<const T*>(a.V) + (long)b * a.Lk * a.ldv + h * 64;
  const T* Pg = reinterpret_cast<const T*>(a.P) + (long)blockIdx.x * a.Lq * a.ldS;
  const T* Dg = reinterpret_cast<const T*>(a.dctx) + (long)b * a.Lq * a.ldd + h * 64;

  nat_load<T, BQ, 64>(s1, Dg, a.ldd, a.Lq, 64, tid);         // dO
  nat_load<T, BKV, 64>(s2, Vg, a.ldv, a.Lk, 64, tid);        // V
  nat_load<T, BQ, BKV>(s3, Pg, a.ldS, a.Lq, a.ldS, tid);     // P
  __syncthreads();

  // dP = dO V^T, then dS = P * (dP - rowsum(dP*P)) without leaving the registers
  f32x4_t dp[MTq][MTk];
  acc_zero(dp);
  tile_mma<T, MTq, MTk, 2, false, false, PQ, PQ>(dp, s1, wr * (BQ / 2), s2, wc * (BKV / 2), lane);
  float pv[MTq][MTk][4];
  const bool dropping = a.drop.p > 0.f;
#pragma unroll
  for (int m = 0; m < MTq; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wr * (BQ / 2) + m * 16 + g * 4 + r;
      const uint32_t rbase = ((uint32_t)blockIdx.x * a.Lq + row) * (uint32_t)a.Lk;
      float s = 0.f;
#pragma unroll
      for (int n = 0; n < MTk; ++n) {
        const int col = wc * (BKV / 2) + n * 16 + i;
        const float p = Elem<T>::ld(reinterpret_cast<const T*>(s3 + row * PP) + col);
        pv[m][n][r] = p;
        if (dropping) dp[m][n][r] *= drop_mult(a.drop.seed, rbase + col, a.drop.p, a.drop.inv_keep);   // d P = d P_drop * mask/(1-p)
        s += p * dp[m][n][r];
      }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o, 64);
      if (i == 0) rowdot[wc * BQ + row] = s;
    }
  if (dropping) {   // dV needs the DROPPED probabilities: rewrite this lane's elements of the tile (all pv are in registers)
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MTq; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wr * (BQ / 2) + m * 16 + g * 4 + r;
        const uint32_t rbase = ((uint32_t)blockIdx.x * a.Lq + row) * (uint32_t)a.Lk;
#pragma unroll
        for (int n = 0; n < MTk; ++n) {
          const int col = wc * (BKV / 2) + n * 16 + i;
          Elem<T>::st(reinterpret_cast<T*>(s3 + row * PP) + col,
                      pv[m][n][r] * drop_mult(a.drop.seed, rbase + col, a.drop.p, a.drop.inv_keep));
        }
      }
    __syncthreads();
  }
  // dV = P^T dO while P and dO are still resident
  f32x4_t dv[MTk][2];
  acc_zero(dv);
  tile_mma<T, MTk, 2, BQ / 32, true, true, PP, PQ>(dv, s3, wr * (BKV / 2), s1, wc * 32, lane);
  __syncthreads();   // row dots published; every wave is done with dO, V and P
  nat_load<T, BQ, 64>(s1, Qg, a.ldq, a.Lq, 64, tid);         // Q over dO
  nat_load<T, BKV, 64>(s2, Kg, a.ldk, a.Lk, 64, tid);        // K over V
  float aw = 0.f, ab = 0.f;
#pragma unroll
  for (int m = 0; m < MTq; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wr * (BQ / 2) + m * 16 + g * 4 + r;
      const float dsum = rowdot[row] + rowdot[BQ + row];
      const float* d = (a.dist && row < a.Lq) ? a.dist + ((long)b * a.Lq + row) * a.Lk : nullptr;
#pragma unroll
      for (int n = 0; n < MTk; ++n) {
        const int col = wc * (BKV / 2) + n * 16 + i;
        const float ds = pv[m][n][r] * (dp[m][n][r] - dsum);
        Elem<T>::st(reinterpret_cast<T*>(s3 + row * PP) + col, ds);   // dS over P
        if (d && col < a.Lk) { aw += ds * d[col]; ab += ds; }
      }
    }
  __syncthreads();

  // dK = alpha dS^T Q (rows = keys), dQ = alpha dS K (rows = queries)
  f32x4_t dk[MTk][2], dq[MTq][2];
  acc_zero(dk); acc_zero(dq);
  tile_mma<T, MTk, 2, BQ / 32, true, true, PP, PQ>(dk, s3, wr * (BKV / 2), s1, wc * 32, lane);
  tile_mma<T, MTq, 2, BKV / 32, false, true, PP, PQ>(dq, s3, wr * (BQ / 2), s2, wc * 32, lane);
  __syncthreads();   // every operand tile is dead: reuse the front of LDS as the fp32 staging tile

  T* dQg = reinterpret_cast<T*>(a.dQ) + (long)b * a.Lq * a.lddq + h * 64;
  T* dKg = reinterpret_cast<T*>(a.dK) + (long)b * a.Lk * a.lddk + h * 64;
  T* dVg = reinterpret_cast<T*>(a.dV) + (long)b * a.Lk * a.lddv + h * 64;
  acc_to_lds(dv, ct, 68, wr * (BKV / 2), wc * 32, 1.0f, lane);
  __syncthreads();
  store_rows64<T, BKV>(ct, dVg, a.lddv, a.Lk, tid);
  __syncthreads();
  acc_to_lds(dk, ct, 68, wr * (BKV / 2), wc * 32, a.alpha, lane);
  __syncthreads();
  store_rows64<T, BKV>(ct, dKg, a.lddk, a.Lk, tid);
  __syncthreads();
  acc_to_lds(dq, ct, 68, wr * (BQ / 2), wc * 32, a.alpha, lane);
  __syncthreads();
  store_rows64<T, BQ>(ct, dQg, a.lddq, a.Lq, tid);

  if (a.d_sp_w != nullptr) {   // d sprel_linear.{weight,bias} (vilmodel_cmt.py:732-736)
    aw = wave_sum(aw); ab = wave_sum(ab);
    if (lane == 0) { red[wave] = aw; red[4 + wave] = ab; }
    __syncthreads();
    if (tid == 0) {
      atomicAdd(a.d_sp_w, red[0] + red[1] + red[2] + red[3]);
      atomicAdd(a.d_sp_b, red[4] + red[5] + red[6] + red[7]);
    }
  }
}

// =========================================================================================================
// Streaming ("flash") attention for long key axes (RxR: 512-token instructions, BASELINE.json configs[3];
// vilmodel_cmt.py:117-137 at L = 512, :335-348 with 512 text keys).  bf16, head dim 64, additive / -inf key masks.
//   forward   one workgroup per (batch, head, 64 queries): K/V tiles of 128 keys stream through LDS, online softmax
//             (running row max m and sum l in registers, O rescaled when m grows), dropout by the same counter hash; the
//             probabilities are never written: only lse = m + log l per row (fp32, in the front of the P buffer).
//   backward  P is RECOMPUTED from Q, K and lse (MFMA work is cheap here, HBM traffic is not):
//             flash_bwd_dq   per (batch, head, 64 queries), loops over key tiles:  D = rowsum(dO*O),  dS = P*(dP*mask - D),
//                            dQ = alpha dS K;  publishes D next to lse
//             flash_bwd_dkv  per (batch, head, 128 keys), loops over query tiles:  dV = (P*mask)^T dO,  dK = alpha dS^T Q
// Replaces, for Lq or Lk > 128, the batched-GEMM path (scores, probabilities and their gradients through HBM: ~0.9 GB per
// layer backward at B = 16, L = 512) with three kernels whose HBM traffic is Q, K, V, O, dO in and ctx / dQ, dK, dV out.
// =========================================================================================================
constexpr int FBQ = 64, FBKV = 128;
constexpr int FBKV2 = 64;       // key tile of the dK/dV kernel: half the accumulators -> two to three workgroups per CU
template <int BKV> struct FlashLdsT {
  static constexpr int PQ = Nat<bf16_t, 64>::PITCH, PP = Nat<bf16_t, BKV>::PITCH;
  static constexpr int Q_OFF = 0, DO_OFF = FBQ * PQ, K_OFF = 2 * FBQ * PQ, V_OFF = K_OFF + BKV * PQ, T_OFF = V_OFF + BKV * PQ;
  static constexpr int RS_OFF = T_OFF + FBQ * PP;              // row max / row sum exchange [2][2][FBQ] fp32 (also D)
  static constexpr int TOTAL = RS_OFF + 4 * FBQ * 4;
  static_assert(BKV * 68 * 4 <= T_OFF, "dK/dV fp32 staging [BKV][68] sits over the Q, dO, K (and V) tiles");
  static_assert(FBQ * 68 * 4 <= T_OFF - K_OFF || BKV < 128, "ctx/dQ fp32 staging [64][68] sits over the K and V tiles");
};
using FlashLds = FlashLdsT<FBKV>;

// global -> registers now, registers -> LDS tile later: the next tile's rows travel while the current tile is computed
template <typename T, int ROWS, int COLS> struct NatRegs {
  static constexpr int N = (ROWS * Nat<T, COLS>::CPR + 255) / 256;
  uint4 v[N];
};
template <typename T, int ROWS, int COLS>
__device__ __forceinline__ void nat_fetch(NatRegs<T, ROWS, COLS>& r, const T* __restrict__ g, long ld, int rows_valid, int cols_valid,
                                          int tid) {
  using N = Nat<T, COLS>;
  constexpr int TOTAL = ROWS * N::CPR;
#pragma unroll
  for (int j = 0; j < NatRegs<T, ROWS, COLS>::N; ++j) {
    const int q = tid + j * 256;
    const int qq = (TOTAL % 256 != 0 && q >= TOTAL) ? TOTAL - 1 : q;
    const int row = qq / N::CPR, c = (qq % N::CPR) * N::EPC;
    const bool ok = row < rows_valid && c < cols_valid;
    const int rc = min(row, max(rows_valid - 1, 0)), cc = min(c, cols_valid - N::EPC);
    uint4 v = *reinterpret_cast<const uint4*>(g + (long)rc * ld + cc);
    v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
    r.v[j] = v;
  }
}
template <typename T, int ROWS, int COLS>
__device__ __forceinline__ void nat_commit(char* lds, const NatRegs<T, ROWS, COLS>& r, int tid) {
  using N = Nat<T, COLS>;
  constexpr int TOTAL = ROWS * N::CPR;
#pragma unroll
  for (int j = 0; j < NatRegs<T, ROWS, COLS>::N; ++j) {
    const int q = tid + j * 256;
    if (TOTAL % 256 != 0 && q >= TOTAL) break;
    *reinterpret_cast<uint4*>(lds + (q / N::CPR) * N::PITCH + (q % N::CPR) * 16) = r.v[j];
  }
}

// additive key term of this lane's NT columns of the key tile starting at k0 (columns >= Lk: excluded)
template <int NT, int BKV>
__device__ __forceinline__ void key_terms(float (&kadd)[NT], const uint8_t* km, int k0, int Lk, int mask_mode, int wc, int i) {
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int col = k0 + wc * (BKV / 2) + n * 16 + i;
    float v = 0.f;
    if (col >= Lk) v = -INFINITY;
    else if (km && !km[col]) v = mask_mode ? -INFINITY : -10000.0f;
    kadd[n] = v;
  }
}

__global__ __launch_bounds__(256) void flash_fwd_kernel(const AttnKArgs a) {
  using T = bf16_t;
  using L = FlashLds;
  constexpr int PQ = L::PQ, PP = L::PP, MT = FBQ / 32, NT = FBKV / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *qt = smem + L::Q_OFF, *kt = smem + L::K_OFF, *vt = smem + L::V_OFF, *pt = smem + L::T_OFF;
  float* rmax = reinterpret_cast<float*>(smem + L::RS_OFF);
  float* rsum = rmax + 2 * FBQ;
  float* ct = reinterpret_cast<float*>(smem + L::K_OFF);          // output staging over the K/V tiles
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  const int i = lane & 15, g = lane >> 4;
  const int bh = blockIdx.x / a.nq, q0 = (blockIdx.x % a.nq) * FBQ;
  const int b = bh / a.nh, h = bh % a.nh;
  const int Lq = min(FBQ, a.Lq - q0);
  const T* Qg = reinterpret_cast<const T*>(a.Q) + ((long)b * a.Lq + q0) * a.ldq + h * 64;
  const T* Kg = reinterpret_cast<const T*>(a.K) + (long)b * a.Lk * a.ldk + h * 64;
  const T* Vg = reinterpret_cast<const T*>(a.V) + (long)b * a.Lk * a.ldv + h * 64;
  T* Cg = reinterpret_cast<T*>(a.ctx) + ((long)b * a.Lq + q0) * a.ldc + h * 64;
  float* lse = reinterpret_cast<float*>(a.P) + (long)bh * a.Lq + q0;
  const uint8_t* km = a.keymask ? a.keymask + (long)b * a.Lk : nullptr;

  NatRegs<T, FBKV, 64> rk, rv;
  nat_fetch<T, FBKV, 64>(rk, Kg, a.ldk, min(FBKV, a.Lk), 64, tid);
  nat_fetch<T, FBKV, 64>(rv, Vg, a.ldv, min(FBKV, a.Lk), 64, tid);
  nat_load<T, FBQ, 64>(qt, Qg, a.ldq, Lq, 64, tid);
  float m_run[MT][4], l_run[MT][4];
  f32x4_t oc[MT][2];
  acc_zero(oc);
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) { m_run[m][r] = -INFINITY; l_run[m][r] = 0.f; }

  for (int k0 = 0; k0 < a.Lk; k0 += FBKV) {
    __syncthreads();                                   // the previous tile's P.V is done with vt / pt
    nat_commit<T, FBKV, 64>(kt, rk, tid);
    nat_commit<T, FBKV, 64>(vt, rv, tid);
    if (k0 + FBKV < a.Lk) {                            // next tile's rows travel while this one is computed
      const int nn = min(FBKV, a.Lk - k0 - FBKV);
      nat_fetch<T, FBKV, 64>(rk, Kg + (long)(k0 + FBKV) * a.ldk, a.ldk, nn, 64, tid);
      nat_fetch<T, FBKV, 64>(rv, Vg + (long)(k0 + FBKV) * a.ldv, a.ldv, nn, 64, tid);
    }
    float kadd[NT];
    key_terms<NT, FBKV>(kadd, km, k0, a.Lk, a.mask_mode, wc, i);
    __syncthreads();
    f32x4_t sc[MT][NT];
    acc_zero(sc);
    tile_mma<T, MT, NT, 2, false, false, PQ, PQ>(sc, qt, wr * (FBQ / 2), kt, wc * (FBKV / 2), lane);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float mx = -INFINITY;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const float v = sc[m][n][r] * a.alpha + kadd[n];
          sc[m][n][r] = v;
          mx = fmaxf(mx, v);
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if (i == 0) rmax[wc * FBQ + wr * (FBQ / 2) + m * 16 + g * 4 + r] = mx;
      }
    __syncthreads();
    float scale_o[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wr * (FBQ / 2) + m * 16 + g * 4 + r;
        const float m_new = fmaxf(m_run[m][r], fmaxf(rmax[row], rmax[FBQ + row]));
        const float mref = (m_new == -INFINITY) ? 0.f : m_new;      // a fully excluded prefix: exp(-inf - 0) = 0, no NaN
        scale_o[m][r] = __expf(m_run[m][r] - mref);
        m_run[m][r] = m_new;
        float sum = 0.f;
        const uint32_t rbase = ((uint32_t)bh * a.Lq + q0 + row) * (uint32_t)a.Lk + (uint32_t)k0;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int col = wc * (FBKV / 2) + n * 16 + i;
          const float e = __expf(sc[m][n][r] - mref);
          sum += e;
          float pd = e;
          if (a.drop.p > 0.f) pd *= drop_mult(a.drop.seed, rbase + col, a.drop.p, a.drop.inv_keep);
          Elem<T>::st(reinterpret_cast<T*>(pt + row * PP) + col, pd);
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) sum += __shfl_xor(sum, o, 64);
        if (i == 0) rsum[wc * FBQ + row] = sum;
      }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) oc[m][c][r] *= scale_o[m][r];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wr * (FBQ / 2) + m * 16 + g * 4 + r;
        l_run[m][r] = l_run[m][r] * scale_o[m][r] + rsum[row] + rsum[FBQ + row];
      }
    tile_mma<T, MT, 2, FBKV / 32, false, true, PP, PQ>(oc, pt, wr * (FBQ / 2), vt, wc * 32, lane);
  }
  __syncthreads();                                     // K/V tiles dead -> fp32 output staging
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wr * (FBQ / 2) + m * 16 + g * 4 + r;
      const float inv = l_run[m][r] > 0.f ? 1.0f / l_run[m][r] : 0.f;
#pragma unroll
      for (int c = 0; c < 2; ++c) oc[m][c][r] *= inv;
      if (wc == 0 && i == 0 && row < Lq) lse[row] = m_run[m][r] + __logf(l_run[m][r]);
    }
  acc_to_lds(oc, ct, 68, wr * (FBQ / 2), wc * 32, 1.0f, lane);
  __syncthreads();
  store_rows64<T, FBQ>(ct, Cg, a.ldc, Lq, tid);
}

// P = exp(alpha*S + key term - lse), the dropout multiplier and dS = P*(dP*mult - D) for one 64 x BKV tile of scores;
// on return sc holds the DROPPED probabilities (dV operand) and dp the score gradient (dQ / dK operand)
template <int MT, int NT, int BKV>
__device__ __forceinline__ void flash_bwd_tile(f32x4_t (&sc)[MT][NT], f32x4_t (&dp)[MT][NT], const float (&kadd)[NT],
                                               const float (&lse_r)[MT][4], const float (&d_r)[MT][4], const bool (&row_ok)[MT][4],
                                               const AttnKArgs& a, uint32_t bh, int q0, int k0, int wr, int wc, int i, int g) {
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wr * (FBQ / 2) + m * 16 + g * 4 + r;
      const uint32_t rbase = (bh * (uint32_t)a.Lq + (uint32_t)(q0 + row)) * (uint32_t)a.Lk + (uint32_t)k0;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int col = wc * (BKV / 2) + n * 16 + i;
        float p = row_ok[m][r] ? __expf(sc[m][n][r] * a.alpha + kadd[n] - lse_r[m][r]) : 0.f;
        float mult = 1.f;
        if (a.drop.p > 0.f) mult = drop_mult(a.drop.seed, rbase + col, a.drop.p, a.drop.inv_keep);
        const float ds = p * (dp[m][n][r] * mult - d_r[m][r]);
        sc[m][n][r] = p * mult;
        dp[m][n][r] = ds;
      }
    }
}

template <int MT, int NT, int BKV>
__device__ __forceinline__ void acc_to_tile(const f32x4_t (&x)[MT][NT], char* tile, int PP, int wr, int wc, int i, int g) {
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wr * (FBQ / 2) + m * 16 + g * 4 + r;
#pragma unroll
      for (int n = 0; n < NT; ++n)
        Elem<bf16_t>::st(reinterpret_cast<bf16_t*>(tile + row * PP) + wc * (BKV / 2) + n * 16 + i, x[m][n][r]);
    }
}

__global__ __launch_bounds__(256) void flash_bwd_dq_kernel(const AttnKArgs a, const void* O, long ldo) {
  using T = bf16_t;
  using L = FlashLds;
  constexpr int PQ = L::PQ, PP = L::PP, MT = FBQ / 32, NT = FBKV / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *qt = smem + L::Q_OFF, *dot = smem + L::DO_OFF, *kt = smem + L::K_OFF, *vt = smem + L::V_OFF, *dst = smem + L::T_OFF;
  float* dD = reinterpret_cast<float*>(smem + L::RS_OFF);          // D of this block's 64 query rows
  float* ct = reinterpret_cast<float*>(smem + L::K_OFF);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  const int i = lane & 15, g = lane >> 4;
  const int bh = blockIdx.x / a.nq, q0 = (blockIdx.x % a.nq) * FBQ;
  const int b = bh / a.nh, h = bh % a.nh;
  const int Lq = min(FBQ, a.Lq - q0);
  const T* Qg = reinterpret_cast<const T*>(a.Q) + ((long)b * a.Lq + q0) * a.ldq + h * 64;
  const T* Kg = reinterpret_cast<const T*>(a.K) + (long)b * a.Lk * a.ldk + h * 64;
  const T* Vg = reinterpret_cast<const T*>(a.V) + (long)b * a.Lk * a.ldv + h * 64;
  const T* Dg = reinterpret_cast<const T*>(a.dctx) + ((long)b * a.Lq + q0) * a.ldd + h * 64;
  const T* Og = reinterpret_cast<const T*>(O) + ((long)b * a.Lq + q0) * ldo + h * 64;
  const float* lse = reinterpret_cast<const float*>(a.P) + (long)bh * a.Lq + q0;
  float* delta = reinterpret_cast<float*>(a.P) + (long)gridDim.x / a.nq * a.Lq + (long)bh * a.Lq + q0;     // D, behind all lse rows
  const uint8_t* km = a.keymask ? a.keymask + (long)b * a.Lk : nullptr;

  NatRegs<T, FBKV, 64> rk, rv;
  nat_fetch<T, FBKV, 64>(rk, Kg, a.ldk, min(FBKV, a.Lk), 64, tid);
  nat_fetch<T, FBKV, 64>(rv, Vg, a.ldv, min(FBKV, a.Lk), 64, tid);
  nat_load<T, FBQ, 64>(qt, Qg, a.ldq, Lq, 64, tid);
  nat_load<T, FBQ, 64>(dot, Dg, a.ldd, Lq, 64, tid);
  nat_load<T, FBQ, 64>(kt, Og, ldo, Lq, 64, tid);                  // O parked in the K tile for the row dots
  __syncthreads();
  {   // D[row] = sum_d dO[row,d] * O[row,d]: four threads per row, 16 columns each
    const int row = tid >> 2, c0 = (tid & 3) * 16;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c)
      s += Elem<T>::ld(reinterpret_cast<const T*>(dot + row * PQ) + c0 + c) * Elem<T>::ld(reinterpret_cast<const T*>(kt + row * PQ) + c0 + c);
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if ((tid & 3) == 0) { dD[row] = s; if (row < Lq) delta[row] = s; }
  }
  __syncthreads();
  float lse_r[MT][4], d_r[MT][4];
  bool row_ok[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wr * (FBQ / 2) + m * 16 + g * 4 + r;
      row_ok[m][r] = row < Lq;
      lse_r[m][r] = row < Lq ? lse[row] : 0.f;
      d_r[m][r] = dD[row];
    }
  f32x4_t dq[MT][2];
  acc_zero(dq);
  for (int k0 = 0; k0 < a.Lk; k0 += FBKV) {
    __syncthreads();                                   // previous tile's dQ product is done with kt / dst (first pass: the O rows)
    nat_commit<T, FBKV, 64>(kt, rk, tid);
    nat_commit<T, FBKV, 64>(vt, rv, tid);
    if (k0 + FBKV < a.Lk) {
      const int nn = min(FBKV, a.Lk - k0 - FBKV);
      nat_fetch<T, FBKV, 64>(rk, Kg + (long)(k0 + FBKV) * a.ldk, a.ldk, nn, 64, tid);
      nat_fetch<T, FBKV, 64>(rv, Vg + (long)(k0 + FBKV) * a.ldv, a.ldv, nn, 64, tid);
    }
    float kadd[NT];
    key_terms<NT, FBKV>(kadd, km, k0, a.Lk, a.mask_mode, wc, i);
    __syncthreads();
    f32x4_t sc[MT][NT], dp[MT][NT];
    acc_zero(sc); acc_zero(dp);
    tile_mma<T, MT, NT, 2, false, false, PQ, PQ>(sc, qt, wr * (FBQ / 2), kt, wc * (FBKV / 2), lane);
    tile_mma<T, MT, NT, 2, false, false, PQ, PQ>(dp, dot, wr * (FBQ / 2), vt, wc * (FBKV / 2), lane);
    flash_bwd_tile<MT, NT, FBKV>(sc, dp, kadd, lse_r, d_r, row_ok, a, (uint32_t)bh, q0, k0, wr, wc, i, g);
    acc_to_tile<MT, NT, FBKV>(dp, dst, PP, wr, wc, i, g);
    __syncthreads();
    tile_mma<T, MT, 2, FBKV / 32, false, true, PP, PQ>(dq, dst, wr * (FBQ / 2), kt, wc * 32, lane);
  }
  __syncthreads();
  T* dQg = reinterpret_cast<T*>(a.dQ) + ((long)b * a.Lq + q0) * a.lddq + h * 64;
  acc_to_lds(dq, ct, 68, wr * (FBQ / 2), wc * 32, a.alpha, lane);
  __syncthreads();
  store_rows64<T, FBQ>(ct, dQg, a.lddq, Lq, tid);
}

// one workgroup per (batch, head, BKV keys): K/V tile resident, loops over the query tiles
template <int BKV>
__global__ __launch_bounds__(256) void flash_bwd_dkv_kernel(const AttnKArgs a, int nkv) {
  using T = bf16_t;
  using L = FlashLdsT<BKV>;
  constexpr int PQ = L::PQ, PP = L::PP, MT = FBQ / 32, NT = BKV / 32, MTk = BKV / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *qt = smem + L::Q_OFF, *dot = smem + L::DO_OFF, *kt = smem + L::K_OFF, *vt = smem + L::V_OFF, *tt = smem + L::T_OFF;
  float* ct = reinterpret_cast<float*>(smem);                       // staging [BKV][68] fp32 over the Q / dO / K tiles
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  const int i = lane & 15, g = lane >> 4;
  const int bh = blockIdx.x / nkv, k0 = (blockIdx.x % nkv) * BKV;
  const int b = bh / a.nh, h = bh % a.nh;
  const int nk = min(BKV, a.Lk - k0);
  const T* Qg = reinterpret_cast<const T*>(a.Q) + (long)b * a.Lq * a.ldq + h * 64;
  const T* Kg = reinterpret_cast<const T*>(a.K) + ((long)b * a.Lk + k0) * a.ldk + h * 64;
  const T* Vg = reinterpret_cast<const T*>(a.V) + ((long)b * a.Lk + k0) * a.ldv + h * 64;
  const T* Dg = reinterpret_cast<const T*>(a.dctx) + (long)b * a.Lq * a.ldd + h * 64;
  const float* lse = reinterpret_cast<const float*>(a.P) + (long)bh * a.Lq;
  const float* delta = reinterpret_cast<const float*>(a.P) + (long)gridDim.x / nkv * a.Lq + (long)bh * a.Lq;
  const uint8_t* km = a.keymask ? a.keymask + (long)b * a.Lk : nullptr;

  NatRegs<T, FBQ, 64> rq, rd;
  nat_fetch<T, FBQ, 64>(rq, Qg, a.ldq, min(FBQ, a.Lq), 64, tid);
  nat_fetch<T, FBQ, 64>(rd, Dg, a.ldd, min(FBQ, a.Lq), 64, tid);
  nat_load<T, BKV, 64>(kt, Kg, a.ldk, nk, 64, tid);
  nat_load<T, BKV, 64>(vt, Vg, a.ldv, nk, 64, tid);
  float kadd[NT];
  key_terms<NT, BKV>(kadd, km, k0, a.Lk, a.mask_mode, wc, i);
  f32x4_t dk[MTk][2], dv[MTk][2];
  acc_zero(dk); acc_zero(dv);
  for (int q0 = 0; q0 < a.Lq; q0 += FBQ) {
    const int nq = min(FBQ, a.Lq - q0);
    float lse_r[MT][4], d_r[MT][4];
    bool row_ok[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wr * (FBQ / 2) + m * 16 + g * 4 + r;
        row_ok[m][r] = row < nq;
        lse_r[m][r] = row < nq ? lse[q0 + row] : 0.f;
        d_r[m][r] = row < nq ? delta[q0 + row] : 0.f;
      }
    __syncthreads();                                   // the previous query tile's products are done with qt / dot / tt
    nat_commit<T, FBQ, 64>(qt, rq, tid);
    nat_commit<T, FBQ, 64>(dot, rd, tid);
    if (q0 + FBQ < a.Lq) {                             // next query tile travels while this one is computed
      const int nn = min(FBQ, a.Lq - q0 - FBQ);
      nat_fetch<T, FBQ, 64>(rq, Qg + (long)(q0 + FBQ) * a.ldq, a.ldq, nn, 64, tid);
      nat_fetch<T, FBQ, 64>(rd, Dg + (long)(q0 + FBQ) * a.ldd, a.ldd, nn, 64, tid);
    }
    __syncthreads();
    f32x4_t sc[MT][NT], dp[MT][NT];
    acc_zero(sc); acc_zero(dp);
    tile_mma<T, MT, NT, 2, false, false, PQ, PQ>(sc, qt, wr * (FBQ / 2), kt, wc * (BKV / 2), lane);
    tile_mma<T, MT, NT, 2, false, false, PQ, PQ>(dp, dot, wr * (FBQ / 2), vt, wc * (BKV / 2), lane);
    flash_bwd_tile<MT, NT, BKV>(sc, dp, kadd, lse_r, d_r, row_ok, a, (uint32_t)bh, q0, k0, wr, wc, i, g);
    acc_to_tile<MT, NT, BKV>(sc, tt, PP, wr, wc, i, g);            // dropped P
    __syncthreads();
    tile_mma<T, MTk, 2, FBQ / 32, true, true, PP, PQ>(dv, tt, wr * (BKV / 2), dot, wc * 32, lane);     // dV += (P*mask)^T dO
    __syncthreads();
    acc_to_tile<MT, NT, BKV>(dp, tt, PP, wr, wc, i, g);            // dS over the same tile
    __syncthreads();
    tile_mma<T, MTk, 2, FBQ / 32, true, true, PP, PQ>(dk, tt, wr * (BKV / 2), qt, wc * 32, lane);      // dK += dS^T Q
  }
  __syncthreads();
  T* dKg = reinterpret_cast<T*>(a.dK) + ((long)b * a.Lk + k0) * a.lddk + h * 64;
  T* dVg = reinterpret_cast<T*>(a.dV) + ((long)b * a.Lk + k0) * a.lddv + h * 64;
  acc_to_lds(dv, ct, 68, wr * (BKV / 2), wc * 32, 1.0f, lane);
  __syncthreads();
  store_rows64<T, BKV>(ct, dVg, a.lddv, nk, tid);
  __syncthreads();
  acc_to_lds(dk, ct, 68, wr * (BKV / 2), wc * 32, a.alpha, lane);
  __syncthreads();
  store_rows64<T, BKV>(ct, dKg, a.lddk, nk, tid);
}

// ---- host side -----------------------------------------------------------------------------------------
static bool fused_enabled() {
  return opt_on(OPT_ATTN_FUSED, true);
}
bool attn_fused_ok(int dt, const AttnBuf& a, long ldc) {
  if (!fused_enabled()) return false;
  if (a.Lq > 128 || a.Lk > 128 || a.Lq < 1 || a.Lk < 1) return false;
  if (dt == ETP_F32 && (a.Lq > 64 || a.Lk > 64)) return false;        // fp32 tiles of 128 rows do not fit 160 KB of LDS
  const int epc = dt == ETP_BF16 ? 8 : 4;
  if (a.ldq % epc || a.ldk % epc || a.ldv % epc || ldc % epc || a.ldS % epc) return false;
  if (((uintptr_t)a.Q | (uintptr_t)a.K | (uintptr_t)a.V) % 16) return false;
  return true;
}

template <typename T, int BQ, int BKV> static int launch_fwd(const AttnKArgs& k, int blocks, hipStream_t st) {
  constexpr int smem = AttnFwdLds<T, BQ, BKV>::TOTAL;
  auto kern = attn_fwd_kernel<T, BQ, BKV>;
  ETP_CHECK_HIP(ensure_dyn_lds(reinterpret_cast<const void*>(kern), smem));
  ETP_LAUNCH(kern, dim3(blocks), dim3(256), smem, st, k);
  ETP_CHECK_LAUNCH("attn_fwd");
  return ETP_OK;
}
template <typename T, int BQ, int BKV> static int launch_bwd(const AttnKArgs& k, int blocks, hipStream_t st) {
  constexpr int smem = AttnBwdLds<T, BQ, BKV>::TOTAL;
  auto kern = attn_bwd_kernel<T, BQ, BKV>;
  ETP_CHECK_HIP(ensure_dyn_lds(reinterpret_cast<const void*>(kern), smem));
  ETP_LAUNCH(kern, dim3(blocks), dim3(256), smem, st, k);
  ETP_CHECK_LAUNCH("attn_bwd");
  return ETP_OK;
}

static AttnKArgs make_args(int nh, const AttnBuf& a, float alpha) {
  AttnKArgs k;
  memset(&k, 0, sizeof(k));
  k.Q = a.Q; k.K = a.K; k.V = a.V; k.ldq = a.ldq; k.ldk = a.ldk; k.ldv = a.ldv;
  k.ldS = a.ldS; k.nh = nh; k.Lq = a.Lq; k.Lk = a.Lk;
  k.keymask = a.keymask; k.mask_mode = a.mask_mode; k.dist = a.dist; k.sp_w = a.sp_w; k.sp_b = a.sp_b; k.alpha = alpha;
  return k;
}

// streaming kernels: bf16, a query or key axis beyond the resident-tile kernels, no pairwise-distance bias (that only exists
// on the graph self-attention, G <= 128), room for lse + D (2 floats per row) in the probability buffer
bool attn_flash_ok(int dt, const AttnBuf& a, long ldc) {
  const bool on = opt_on(OPT_ATTN_FLASH, true);
  if (!on || !fused_enabled() || dt != ETP_BF16) return false;
  if (a.Lq <= 128 && a.Lk <= 128) return false;
  if (a.dist != nullptr || a.Lq < 1 || a.Lk < 1 || a.ldS < 4) return false;
  if (a.ldq % 8 || a.ldk % 8 || a.ldv % 8 || ldc % 8) return false;
  if (((uintptr_t)a.Q | (uintptr_t)a.K | (uintptr_t)a.V) % 16) return false;
  return true;
}
static int flash_attr() {
  ETP_CHECK_HIP(ensure_dyn_lds(reinterpret_cast<const void*>(flash_fwd_kernel), FlashLds::TOTAL));      // per device (launch.h)
  ETP_CHECK_HIP(ensure_dyn_lds(reinterpret_cast<const void*>(flash_bwd_dq_kernel), FlashLds::TOTAL));
  ETP_CHECK_HIP(ensure_dyn_lds(reinterpret_cast<const void*>(flash_bwd_dkv_kernel<FBKV2>), FlashLdsT<FBKV2>::TOTAL));
  return ETP_OK;
}
int attn_flash_fwd(int nh, const AttnBuf& a, void* P, void* ctx, long ldc, float alpha, hipStream_t st, Drop drop) {
  ETP_TRY(flash_attr());
  AttnKArgs k = make_args(nh, a, alpha);
  k.drop = drop;
  k.P = P; k.ctx = ctx; k.ldc = ldc;
  k.nq = (a.Lq + FBQ - 1) / FBQ;
  ETP_LAUNCH(flash_fwd_kernel, dim3(a.B * nh * k.nq), dim3(256), FlashLds::TOTAL, st, k);
  ETP_CHECK_LAUNCH("flash_fwd");
  return ETP_OK;
}
int attn_flash_bwd(int nh, const AttnBuf& a, const void* P, const void* dctx, long ldd, void* dQ, long lddq, void* dK, long lddk,
                   void* dV, long lddv, float alpha, hipStream_t st, Drop drop) {
  ETP_REQUIRE(a.O != nullptr, "the streaming attention backward needs the forward output (AttnBuf::O)");
  ETP_TRY(flash_attr());
  AttnKArgs k = make_args(nh, a, alpha);
  k.drop = drop;
  k.P = const_cast<void*>(P); k.dctx = dctx; k.ldd = ldd;
  k.dQ = dQ; k.dK = dK; k.dV = dV; k.lddq = lddq; k.lddk = lddk; k.lddv = lddv;
  k.nq = (a.Lq + FBQ - 1) / FBQ;
  const int nkv = (a.Lk + FBKV2 - 1) / FBKV2;
  ETP_LAUNCH(flash_bwd_dq_kernel, dim3(a.B * nh * k.nq), dim3(256), FlashLds::TOTAL, st, k, a.O, a.ldo);
  ETP_CHECK_LAUNCH("flash_bwd_dq");
  ETP_LAUNCH(flash_bwd_dkv_kernel<FBKV2>, dim3(a.B * nh * nkv), dim3(256), FlashLdsT<FBKV2>::TOTAL, st, k, nkv);
  ETP_CHECK_LAUNCH("flash_bwd_dkv");
  return ETP_OK;
}

int attn_fused_fwd(int dt, int nh, const AttnBuf& a, void* P, void* ctx, long ldc, float alpha, hipStream_t st, Drop drop) {
  AttnKArgs k = make_args(nh, a, alpha);
  k.drop = drop;
  k.P = P; k.ctx = ctx; k.ldc = ldc;
  k.nq = (a.Lq + 63) / 64;                        // 64 queries per workgroup
  int blocks = a.B * nh * k.nq;
  if (dt == ETP_BF16) {     // key tile = 64 / 96 / 128 columns: the 80-token instruction takes the 96 one, not a 128 pad
    const bool q96 = opt_on(OPT_ATTN_Q96, true);
    if (q96 && a.Lq > 64 && a.Lq <= 96 && a.Lk > 64 && a.Lk <= 96) {
      // the 80-token self-attention: ONE 96-query workgroup per (batch, head) instead of a full and a quarter-full 64-query
      // one (K/V staged once, half the workgroups)
      k.nq = 1;
      return launch_fwd<bf16_t, 96, 96>(k, a.B * nh, st);
    }
    if (a.Lk > 96) return launch_fwd<bf16_t, 64, 128>(k, blocks, st);
    if (a.Lk > 64) return launch_fwd<bf16_t, 64, 96>(k, blocks, st);
    return launch_fwd<bf16_t, 64, 64>(k, blocks, st);
  }
  return launch_fwd<float, 64, 64>(k, blocks, st);
}

int attn_fused_bwd(int dt, int nh, const AttnBuf& a, const void* P, const void* dctx, long ldd, void* dQ, long lddq, void* dK,
                   long lddk, void* dV, long lddv, float alpha, float* d_sp_w, float* d_sp_b, hipStream_t st, Drop drop) {
  AttnKArgs k = make_args(nh, a, alpha);
  k.drop = drop;
  k.P = const_cast<void*>(P); k.dctx = dctx; k.ldd = ldd;
  k.dQ = dQ; k.dK = dK; k.dV = dV; k.lddq = lddq; k.lddk = lddk; k.lddv = lddv; k.d_sp_w = d_sp_w; k.d_sp_b = d_sp_b;
  const int blocks = a.B * nh;
  const bool bq = a.Lq > 64, bk = a.Lk > 64;
  if (dt == ETP_BF16) {
    if (bq && bk && a.Lq <= 96 && a.Lk <= 96) return launch_bwd<bf16_t, 96, 96>(k, blocks, st);
    if (!bq && bk && a.Lk <= 96) return launch_bwd<bf16_t, 64, 96>(k, blocks, st);
    if (bq && bk) return launch_bwd<bf16_t, 128, 128>(k, blocks, st);
    if (bq) return launch_bwd<bf16_t, 128, 64>(k, blocks, st);
    if (bk) return launch_bwd<bf16_t, 64, 128>(k, blocks, st);
    return launch_bwd<bf16_t, 64, 64>(k, blocks, st);
  }
  return launch_bwd<float, 64, 64>(k, blocks, st);
}

}  // namespace etp
