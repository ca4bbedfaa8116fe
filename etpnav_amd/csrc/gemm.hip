// MFMA GEMM family for the ETPNav planner (gfx950 / CDNA4).
//
//   C[m,n] = epilogue( alpha * sum_k A[m,k] * B[n,k] )
//
// One templated kernel covers every dense product on the path (reference sites:
// every nn.Linear / torch.matmul of vlnce_baselines/models/etp/vilmodel_cmt.py and the
// packed nn.MultiheadAttention of common/transformer.py:138, plus their autograd
// backward):
//   * operand storage: "row" = [rows][K] (K contiguous) or "trans" = [K][rows];
//     NT = forward linears / dgrad with the pre-transposed weight copy / Q.K^T / dO.V^T,
//     NN (B trans) = P.V and dS.K,  TN (A and B trans) = wgrad, P^T.dO, dS^T.Q.
//   * dtype: bf16 operands -> v_mfma_f32_16x16x32_bf16, fp32 operands ->
//     v_mfma_f32_16x16x4_f32 (exact fp32 "parity mode"); fp32 accumulation always.
//   * 256 threads = 4 wavefronts (2x2), BMxBN block tile, 128-byte K slab per stage
//     (64 bf16 / 32 fp32), register-staged double-buffered LDS, one barrier per slab.
//   * row operands sit in LDS as [row][128 B] with a 16-B-chunk XOR swizzle
//     (chunk ^= row & 7) so ds_read_b128 fragment reads spread over the banks;
//     trans operands sit as [k][rows] (+32 B pad) and are read with
//     ds_read_b64_tr_b16 (bf16) or ds_read_b32 (fp32).
//   * fused epilogues: alpha, bias, erf-GELU (+ saves pre-activation), ReLU, their
//     backward forms, residual add, fp32 accumulate / atomic split-K for wgrad.
// The lane->k assignment is the same for every read mode (group g = lane>>4, element e:
// k = 32*s + 8*g + e), so any A/B storage pairing is consistent.
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "common.h"

namespace etp {

template <typename T> struct MmaTraits;
template <> struct MmaTraits<bf16_t> {
  static constexpr int BK = 64;      // 128 B / 2
  static constexpr int EPC = 8;      // elements per 16-B chunk
};
template <> struct MmaTraits<float> {
  static constexpr int BK = 32;
  static constexpr int EPC = 4;
};

template <typename T> struct Frag;          // 8 k-values of one row/col for one MFMA step
template <> struct Frag<bf16_t> { uint4 v; };
template <> struct Frag<float> { float4 lo, hi; };

__device__ __forceinline__ void mma_step(f32x4_t& acc, const Frag<bf16_t>& a, const Frag<bf16_t>& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a.v), __builtin_bit_cast(bf16x8_t, b.v),
                                                acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_step(f32x4_t& acc, const Frag<float>& a, const Frag<float>& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.x, b.lo.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.y, b.lo.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.z, b.lo.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.w, b.lo.w, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.x, b.hi.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.y, b.hi.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.z, b.hi.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.w, b.hi.w, acc, 0, 0, 0);
}

// ---- LDS geometry of one operand tile -------------------------------------------------
template <typename T, bool TR, int ROWS> struct TileGeom {
  static constexpr int BK = MmaTraits<T>::BK;
  static constexpr int EPC = MmaTraits<T>::EPC;
  static constexpr int PITCH = TR ? (ROWS * (int)sizeof(T) + 32) : 128;      // bytes
  static constexpr int BYTES = TR ? BK * PITCH : ROWS * 128;
  static constexpr int CHUNKS = ROWS * 8;                                    // 16-B chunks per tile (both layouts)
  static constexpr int PER_THREAD = CHUNKS / 256;
  static constexpr int CPR = TR ? ROWS / EPC : 8;                            // chunks per LDS row
  static_assert(CHUNKS % 256 == 0, "tile too small for 256 threads");
};

template <int N> struct Regs { uint4 v[N]; unsigned okmask; };

// 8 consecutive elements <-> float[8] through 16-byte vectors (bf16: one uint4, fp32: two)
template <typename U> __device__ __forceinline__ void unpack8(const uint4* p, float (&f)[8]);
template <> __device__ __forceinline__ void unpack8<bf16_t>(const uint4* p, float (&f)[8]) {
  const uint32_t w[4] = {p[0].x, p[0].y, p[0].z, p[0].w};
#pragma unroll
  for (int e = 0; e < 4; ++e) { f[2 * e] = __uint_as_float(w[e] << 16); f[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
}
template <> __device__ __forceinline__ void unpack8<float>(const uint4* p, float (&f)[8]) {
  f[0] = __uint_as_float(p[0].x); f[1] = __uint_as_float(p[0].y); f[2] = __uint_as_float(p[0].z); f[3] = __uint_as_float(p[0].w);
  f[4] = __uint_as_float(p[1].x); f[5] = __uint_as_float(p[1].y); f[6] = __uint_as_float(p[1].z); f[7] = __uint_as_float(p[1].w);
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
  uint4 o;
  o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
  o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
  o.z = (uint32_t)f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16);
  o.w = (uint32_t)f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
  *reinterpret_cast<uint4*>(p) = o;
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
  reinterpret_cast<float4*>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
  reinterpret_cast<float4*>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
}

// global -> registers for one tile (zero-filled outside [rows_total) x [k_end)).
// Branch-free on purpose: a load inside a per-element `if` makes hipcc wait vmcnt(0) per element (one full
// memory latency each); here every load is issued unconditionally from a clamped in-bounds address and the
// out-of-range ones are zeroed with v_cndmask when they are written to LDS (after the MFMA block, so the
// s_waitcnt for them sits behind the math).
template <typename T, bool TR, int ROWS>
__device__ __forceinline__ void tile_load(Regs<TileGeom<T, TR, ROWS>::PER_THREAD>& r, const T* __restrict__ base, long ld,
                                          int row0, int rows_total, int k0, int k_end, int tid) {
  using G = TileGeom<T, TR, ROWS>;
  unsigned okmask = 0;
#pragma unroll
  for (int j = 0; j < G::PER_THREAD; ++j) {
    const int q = tid + j * 256;
    const int lr = q / G::CPR, c = q % G::CPR;
    bool ok;
    const T* src;
    if constexpr (!TR) {
      const int row = row0 + lr, k = k0 + c * G::EPC;
      ok = row < rows_total && k < k_end;
      const int rc = min(row, rows_total - 1), kc = min(k, (k_end - 1) / G::EPC * G::EPC);
      src = base + (long)rc * ld + kc;
    } else {
      const int k = k0 + lr, row = row0 + c * G::EPC;
      ok = k < k_end && row < rows_total;
      const int kc = min(k, k_end - 1), rc = min(row, (rows_total - 1) / G::EPC * G::EPC);
      src = base + (long)kc * ld + rc;
    }
    r.v[j] = *reinterpret_cast<const uint4*>(src);   // consumed (and masked) only in tile_store, after the MFMAs
    okmask |= (ok ? 1u : 0u) << j;
  }
  r.okmask = okmask;
}

template <typename T, bool TR, int ROWS>
__device__ __forceinline__ void tile_store(const Regs<TileGeom<T, TR, ROWS>::PER_THREAD>& r, char* lds, int tid) {
  using G = TileGeom<T, TR, ROWS>;
#pragma unroll
  for (int j = 0; j < G::PER_THREAD; ++j) {
    const int q = tid + j * 256;
    const int lr = q / G::CPR, c = q % G::CPR;
    int off;
    if constexpr (!TR) off = lr * 128 + ((c ^ (lr & 7)) << 4);
    else off = lr * G::PITCH + (c << 4);
    const bool ok = (r.okmask >> j) & 1u;
    uint4 v = r.v[j];
    v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
    *reinterpret_cast<uint4*>(lds + off) = v;
  }
}

// LDS -> fragment: 8 k-values (k = 32*s + 8*g + e) of tile row `row` (i = lane&15 already added by caller)
template <typename T, bool TR, int ROWS>
__device__ __forceinline__ void frag_load(Frag<T>& f, const char* lds, int row16 /*first row of the 16-row group*/, int s,
                                          int lane) {
  using G = TileGeom<T, TR, ROWS>;
  const int i = lane & 15, g = lane >> 4;
  if constexpr (!TR) {
    const int row = row16 + i;
    if constexpr (sizeof(T) == 2) {
      const int c = s * 4 + g;
      f.v = *reinterpret_cast<const uint4*>(lds + row * 128 + ((c ^ (row & 7)) << 4));
    } else {
      const int c = 2 * g;
      f.lo = *reinterpret_cast<const float4*>(lds + row * 128 + ((c ^ (row & 7)) << 4));
      f.hi = *reinterpret_cast<const float4*>(lds + row * 128 + (((c + 1) ^ (row & 7)) << 4));
    }
  } else {
    if constexpr (sizeof(T) == 2) {
      // ds_read_b64_tr_b16: within a 16-lane group, lane j supplies the address of 4 consecutive bf16 of
      // k-row (j>>2), columns 4*(j&3)..+3; lane i receives column i of that 4x16 block (k = 0..3).
      const int k0 = s * 32 + g * 8;
      const char* p0 = lds + (k0 + (i >> 2)) * G::PITCH + (row16 + (i & 3) * 4) * 2;
      typedef short4_t __attribute__((address_space(3))) * lds_s4;
      short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(p0));
      short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(p0 + 4 * G::PITCH));
      uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
      f.v = make_uint4(a.x, a.y, b.x, b.y);
    } else {
      const int k0 = g * 8;
      const float* p = reinterpret_cast<const float*>(lds + k0 * G::PITCH) + row16 + i;
      constexpr int PF = G::PITCH / 4;
      f.lo = make_float4(p[0], p[PF], p[2 * PF], p[3 * PF]);
      f.hi = make_float4(p[4 * PF], p[5 * PF], p[6 * PF], p[7 * PF]);
    }
  }
}

template <typename T, typename TC, bool TA, bool TB, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs g) {
  using GA = TileGeom<T, TA, BM>;
  using GB = TileGeom<T, TB, BN>;
  constexpr int BK = MmaTraits<T>::BK;
  constexpr int KS = BK / 32;
  constexpr int MT = BM / 32, NT = BN / 32;     // 16x16 MFMA tiles per wave (wave grid 2x2)
  constexpr int STAGE = GA::BYTES + GB::BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  // tile / batch / split-K coordinates.  blockIdx.x walks N fastest so that consecutive blocks share the
  // A row-panel; TODO(round 2): XCD-aware remap.
  const int tiles_n = (g.N + BN - 1) / BN;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int z = blockIdx.y / g.ksplit, ks = blockIdx.y % g.ksplit;
  const int zo = z / g.nb_inner, zi = z % g.nb_inner;
  const T* A = reinterpret_cast<const T*>(g.A) + zo * g.sAo + zi * g.sAi;
  const T* B = reinterpret_cast<const T*>(g.B) + zo * g.sBo + zi * g.sBi;
  TC* C = reinterpret_cast<TC*>(g.C) + zo * g.sCo + zi * g.sCi;

  int kbeg = 0, kend = g.K;
  if (g.ksplit > 1) {
    const int per = ((g.K + g.ksplit - 1) / g.ksplit + BK - 1) / BK * BK;
    kbeg = ks * per;
    kend = min(g.K, kbeg + per);
  }
  const int nk = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  Regs<GA::PER_THREAD> ra;
  Regs<GB::PER_THREAD> rb;
  if (nk > 0) {
    tile_load<T, TA, BM>(ra, A, g.lda, m0, g.M, kbeg, kend, tid);
    tile_load<T, TB, BN>(rb, B, g.ldb, n0, g.N, kbeg, kend, tid);
    tile_store<T, TA, BM>(ra, smem, tid);
    tile_store<T, TB, BN>(rb, smem + GA::BYTES, tid);
  }
  __syncthreads();

  for (int t = 0; t < nk; ++t) {
    const char* sa = smem + (t & 1) * STAGE;
    const char* sb = sa + GA::BYTES;
    const bool more = (t + 1 < nk);
    if (more) {  // issue next slab's global loads before the math (latency hides under the MFMAs)
      tile_load<T, TA, BM>(ra, A, g.lda, m0, g.M, kbeg + (t + 1) * BK, kend, tid);
      tile_load<T, TB, BN>(rb, B, g.ldb, n0, g.N, kbeg + (t + 1) * BK, kend, tid);
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      Frag<T> fa[MT], fb[NT];
#pragma unroll
      for (int a = 0; a < MT; ++a) frag_load<T, TA, BM>(fa[a], sa, wr * (BM / 2) + a * 16, s, lane);
#pragma unroll
      for (int b = 0; b < NT; ++b) frag_load<T, TB, BN>(fb[b], sb, wc * (BN / 2) + b * 16, s, lane);
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) mma_step(acc[a][b], fa[a], fb[b]);
    }
    if (more) {
      char* da = smem + ((t + 1) & 1) * STAGE;
      tile_store<T, TA, BM>(ra, da, tid);
      tile_store<T, TB, BN>(rb, da + GA::BYTES, tid);
    }
    __syncthreads();
  }

  // ---- epilogue ----------------------------------------------------------------------------------------
  // The MFMA C layout (lane: row = 4*(lane>>4)+r, col = lane&15) gives 2-byte scattered stores, so the tile is
  // staged through LDS as fp32 [BM][BN+4] and written back as whole 8-column chunks per thread: bias / residual /
  // activation operands and the result all move as 16-byte vectors (coalesced 256 B per 16 threads).
  const int i = lane & 15, gq = lane >> 4;
  constexpr int CP = BN + 4;
  float* ct = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        ct[(wr * (BM / 2) + a * 16 + gq * 4 + r) * CP + wc * (BN / 2) + b * 16 + i] = acc[a][b][r];
  __syncthreads();
  const T* R = reinterpret_cast<const T*>(g.R);   // residual has the OUTPUT type when TC != T (see launch checks)
  T* Z = reinterpret_cast<T*>(g.Z);
  constexpr int CPRW = BN / 8;                     // 8-column chunks per tile row
  constexpr int NCHUNK = BM * CPRW / 256;
  if (g.vec_epilogue) {
    // Phase A: issue every global read of the epilogue (residual / activation operand / old C) up front from
    // clamped in-bounds addresses -- no per-element branches, so the loads overlap instead of serialising.
    constexpr int VPC = 8 * (int)sizeof(TC) / 16;  // 16-byte vectors per 8-element chunk of the C type (1 or 2)
    constexpr int VPT = 8 * (int)sizeof(T) / 16;   // 16-byte vectors per 8-element chunk of T (bf16: 1, fp32: 2)
    const bool has_r = g.R != nullptr, has_zr = (g.act == ETP_ACT_GELU_BWD || g.act == ETP_ACT_RELU_BWD),
               has_c = (g.out_mode == 1);
    const bool has_bias = (g.bias != nullptr && ks == 0);
    uint4 rr[NCHUNK][VPC], cc[NCHUNK][VPC], zz[NCHUNK][VPT];
    const int col_last = (g.N - 1) / 8 * 8;
#pragma unroll
    for (int jj = 0; jj < NCHUNK; ++jj) {
      const int q = tid + jj * 256;
      const int lr = q / CPRW, lc = (q % CPRW) * 8;
      const int rowc = min(m0 + lr, g.M - 1), colc = min(n0 + lc, col_last);
      if (has_r) {
        const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const TC*>(g.R) + (long)rowc * g.ldr + colc);
#pragma unroll
        for (int u = 0; u < VPC; ++u) rr[jj][u] = p[u];
      }
      if (has_c) {
        const uint4* p = reinterpret_cast<const uint4*>(C + (long)rowc * g.ldc + colc);
#pragma unroll
        for (int u = 0; u < VPC; ++u) cc[jj][u] = p[u];
      }
      if (has_zr) {
        const uint4* p = reinterpret_cast<const uint4*>(Z + (long)rowc * g.ldz + colc);
#pragma unroll
        for (int u = 0; u < VPT; ++u) zz[jj][u] = p[u];
      }
    }
    // Phase B: combine and store
#pragma unroll
    for (int jj = 0; jj < NCHUNK; ++jj) {
      const int q = tid + jj * 256;
      const int lr = q / CPRW, lc = (q % CPRW) * 8;
      const int row = m0 + lr, col = n0 + lc;
      const bool ok = row < g.M && col < g.N;
      const int colc = min(col, col_last);
      float v[8];
      {
        const float4 x0 = *reinterpret_cast<const float4*>(ct + lr * CP + lc);
        const float4 x1 = *reinterpret_cast<const float4*>(ct + lr * CP + lc + 4);
        v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
      }
      if (has_bias) {
        const float4 b0 = *reinterpret_cast<const float4*>(g.bias + colc), b1 = *reinterpret_cast<const float4*>(g.bias + colc + 4);
        const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * g.alpha + bv[e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= g.alpha;
      }
      if (g.act == ETP_ACT_GELU) {
        if (ok) store8(Z + (long)row * g.ldz + col, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
      } else if (g.act == ETP_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      } else if (has_zr) {
        float zf[8];
        unpack8<T>(zz[jj], zf);
        if (g.act == ETP_ACT_GELU_BWD) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= gelu_erf_grad(zf[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = zf[e] > 0.f ? v[e] : 0.f;
        }
      }
      if (has_r) {
        float rf[8];
        unpack8<TC>(rr[jj], rf);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rf[e];
      }
      TC* dst = C + (long)row * g.ldc + col;
      if (g.out_mode == 2) {
        if constexpr (sizeof(TC) == 4) {
          if (ok) {
#pragma unroll
            for (int e = 0; e < 8; ++e) atomicAdd(reinterpret_cast<float*>(dst) + e, v[e]);
          }
        }
      } else {
        if (has_c) {
          float cf[8];
          unpack8<TC>(cc[jj], cf);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += cf[e];
        }
        if (ok) store8(dst, v);
      }
    }
    return;
  }
  // scalar fallback (odd leading dimensions / unaligned bases): one element per thread-iteration, row-major
  for (int q = tid; q < BM * BN; q += 256) {
    const int lr = q / BN, lc = q % BN;
    const int row = m0 + lr, col = n0 + lc;
    if (row >= g.M || col >= g.N) continue;
    float v = ct[lr * CP + lc] * g.alpha + ((g.bias != nullptr && ks == 0) ? g.bias[col] : 0.f);
    if (g.act == ETP_ACT_GELU) {
      Elem<T>::st(Z + (long)row * g.ldz + col, v);
      v = gelu_erf(v);
    } else if (g.act == ETP_ACT_RELU) {
      v = fmaxf(v, 0.f);
    } else if (g.act == ETP_ACT_GELU_BWD) {
      v *= gelu_erf_grad(Elem<T>::ld(Z + (long)row * g.ldz + col));
    } else if (g.act == ETP_ACT_RELU_BWD) {
      v = (Elem<T>::ld(Z + (long)row * g.ldz + col) > 0.f) ? v : 0.f;
    }
    if (g.R != nullptr) v += Elem<TC>::ld(reinterpret_cast<const TC*>(g.R) + (long)row * g.ldr + col);
    TC* dst = C + (long)row * g.ldc + col;
    if constexpr (sizeof(TC) == 4) {
      if (g.out_mode == 2) atomicAdd(reinterpret_cast<float*>(dst), v);
      else if (g.out_mode == 1) *reinterpret_cast<float*>(dst) += v;
      else *reinterpret_cast<float*>(dst) = v;
    } else {
      if (g.out_mode == 1) v += Elem<TC>::ld(dst);
      Elem<TC>::st(dst, v);
    }
  }
  (void)R;
}

// ---- optional per-launch HIP-event timing (bench.py roofline leg) ---------------------------------------
struct ProfRec { int id; hipEvent_t a, b; double flops, bytes; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof_recs;
static std::vector<std::string> g_prof_names;
static std::mutex g_prof_mu;

void prof_enable(bool on) { g_prof_on = on; }
void prof_reset() {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  g_prof_recs.clear();
}
static int prof_id(const std::string& name) {
  for (size_t i = 0; i < g_prof_names.size(); ++i)
    if (g_prof_names[i] == name) return (int)i;
  g_prof_names.push_back(name);
  return (int)g_prof_names.size() - 1;
}
int prof_report(etp_prof_entry* out, int cap) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  std::vector<etp_prof_entry> agg(g_prof_names.size());
  for (size_t i = 0; i < agg.size(); ++i) {
    memset(&agg[i], 0, sizeof(etp_prof_entry));
    strncpy(agg[i].name, g_prof_names[i].c_str(), sizeof(agg[i].name) - 1);
  }
  for (auto& r : g_prof_recs) {
    if (hipEventSynchronize(r.b) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
    agg[r.id].launches += 1; agg[r.id].ms += ms; agg[r.id].flops += r.flops; agg[r.id].bytes += r.bytes;
  }
  int n = 0;
  for (auto& e : agg)
    if (e.launches > 0 && n < cap) out[n++] = e;
  return n;
}

template <typename T, typename TC, bool TA, bool TB, int BM, int BN>
static int launch_one(const GemmArgs& g, int nbatch, hipStream_t st) {
  using GA = TileGeom<T, TA, BM>;
  using GB = TileGeom<T, TB, BN>;
  constexpr int smem_loop = 2 * (GA::BYTES + GB::BYTES), smem_c = BM * (BN + 4) * 4;
  constexpr int smem = smem_loop > smem_c ? smem_loop : smem_c;
  static bool attr_set = false;
  auto kern = gemm_kernel<T, TC, TA, TB, BM, BN>;
  if (!attr_set) {
    ETP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set = true;
  }
  const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
  dim3 grid(tiles, nbatch * g.ksplit, 1);
  ProfRec rec;
  const bool prof = g_prof_on;
  if (prof) {
    char nm[96];
    snprintf(nm, sizeof(nm), "gemm<%s,%s,%s%s,%dx%d>", sizeof(T) == 2 ? "bf16" : "f32", sizeof(TC) == 2 ? "bf16" : "f32",
             TA ? "T" : "N", TB ? "N" : "T", BM, BN);   // BLAS-style: opA,opB of C = opA(A) opB(B)
    std::lock_guard<std::mutex> lk(g_prof_mu);
    rec.id = prof_id(nm);
    rec.flops = 2.0 * g.M * g.N * g.K * nbatch;
    rec.bytes = ((double)g.M * g.K + (double)g.N * g.K) * nbatch * sizeof(T) + (double)g.M * g.N * nbatch * sizeof(TC);
    ETP_CHECK_HIP(hipEventCreate(&rec.a));
    ETP_CHECK_HIP(hipEventCreate(&rec.b));
    ETP_CHECK_HIP(hipEventRecord(rec.a, st));
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, g);
  ETP_CHECK_LAUNCH("gemm");
  if (prof) {
    ETP_CHECK_HIP(hipEventRecord(rec.b, st));
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_recs.push_back(rec);
  }
  return ETP_OK;
}

template <typename T, typename TC, bool TA, bool TB>
static int launch_tiles(const GemmArgs& g, int nbatch, hipStream_t st) {
  // Tile choice: 128x128 when it still yields >= ~1 block per CU, else 64x64 (fills 256 CUs on the
  // planner's small-M products and keeps batched attention tiles from wasting MFMA work).
  const long t128 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128) * nbatch * g.ksplit;
  if (g.M >= 128 && g.N >= 128 && t128 >= 192) return launch_one<T, TC, TA, TB, 128, 128>(g, nbatch, st);
  return launch_one<T, TC, TA, TB, 64, 64>(g, nbatch, st);
}

template <typename T, typename TC>
static int launch_trans(int ta, int tb, const GemmArgs& g, int nbatch, hipStream_t st) {
  if (!ta && !tb) return launch_tiles<T, TC, false, false>(g, nbatch, st);
  if (!ta && tb) return launch_tiles<T, TC, false, true>(g, nbatch, st);
  if (ta && tb) return launch_tiles<T, TC, true, true>(g, nbatch, st);
  return fail(ETP_ERR_INVALID, "gemm: (A trans, B row) storage pairing is not used on this path");
}

int launch_gemm(int dtype, int c_dtype, int ta, int tb, const GemmArgs& g_in, int nbatch, hipStream_t st) {
  const GemmArgs& g0 = g_in;
  ETP_REQUIRE(g0.M > 0 && g0.N > 0 && g0.K >= 0 && nbatch > 0 && g0.ksplit >= 1 && g0.nb_inner >= 1, "bad dims");
  const int epc = dtype == ETP_BF16 ? 8 : 4;
  ETP_REQUIRE(g0.lda % epc == 0 && g0.ldb % epc == 0, "lda/ldb must be multiples of a 16-byte chunk");
  ETP_REQUIRE(((uintptr_t)g0.A % 16 == 0) && ((uintptr_t)g0.B % 16 == 0), "A/B must be 16-byte aligned");
  ETP_REQUIRE((g0.sAo % epc == 0) && (g0.sAi % epc == 0) && (g0.sBo % epc == 0) && (g0.sBi % epc == 0),
              "batch strides must keep 16-byte alignment");
  ETP_REQUIRE(g0.out_mode != 2 || c_dtype == ETP_F32, "atomic accumulation needs an fp32 C");
  ETP_REQUIRE(g0.ksplit == 1 || g0.out_mode == 2, "split-K needs atomic accumulation");
  GemmArgs g = g_in;
  {  // the vectorised epilogue needs 8-column chunks to stay in-bounds and 16-byte aligned
    const size_t cs = dtype_size(c_dtype), ts = dtype_size(dtype);
    bool ok = (g.ldc % 8 == 0) && (g.ldc >= round_up(g.N, 8)) && ((uintptr_t)g.C % 16 == 0) && ((g.sCo * cs) % 16 == 0) &&
              ((g.sCi * cs) % 16 == 0);
    if (g.bias) ok = ok && ((uintptr_t)g.bias % 16 == 0) && (g.N % 8 == 0);
    if (g.R) ok = ok && (g.ldr % 8 == 0) && ((uintptr_t)g.R % 16 == 0) && (g.ldr >= round_up(g.N, 8));
    if (g.Z) ok = ok && (g.ldz % 8 == 0) && ((uintptr_t)g.Z % 16 == 0) && (g.ldz >= round_up(g.N, 8));
    (void)ts;
    g.vec_epilogue = ok ? 1 : 0;
  }
  if (dtype == ETP_F32) {
    ETP_REQUIRE(c_dtype == ETP_F32, "fp32 operands need an fp32 C");
    return launch_trans<float, float>(ta, tb, g, nbatch, st);
  }
  if (c_dtype == ETP_F32) return launch_trans<bf16_t, float>(ta, tb, g, nbatch, st);
  return launch_trans<bf16_t, bf16_t>(ta, tb, g, nbatch, st);
}

}  // namespace etp
