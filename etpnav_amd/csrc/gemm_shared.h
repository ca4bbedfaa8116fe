// Pieces shared by the GEMM kernel families (gemm.hip: 16x16x32 / fp32 tiles; gemm_mm32.hip: 32x32x16 bf16 tiles with the
// order-pinned main loop): 16-byte pack / unpack helpers, the fused epilogue (bias, erf-GELU and its backward, ReLU, dropout,
// residual, accumulate), the XCD-aware workgroup -> tile map, the in-kernel phase probe and the host-side launch bookkeeping
// (per-launch HIP-event timing, probe slots).
#pragma once
#include "common.h"

namespace etp {

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
  o.x = pack_bf16(v[0], v[1]);
  o.y = pack_bf16(v[2], v[3]);
  o.z = pack_bf16(v[4], v[5]);
  o.w = pack_bf16(v[6], v[7]);
  *reinterpret_cast<uint4*>(p) = o;
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
  reinterpret_cast<float4*>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
  reinterpret_cast<float4*>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
}

// The saved GELU derivative (ETP_ACT_GELU_SAVEGRAD / ETP_ACT_MUL_Z) travels as IEEE half in the 2-byte Z buffer of the bf16 mode: its
// values lie in [-0.13, 1.13], where fp16 keeps 11 significant bits against bf16's 8 -- the backward's factor is then MORE exact than
// gelu'(bf16(z)) was (first GPU run with a bf16-stored derivative: embeddings.LayerNorm.bias of the B = 1 fixture moved from 9 % to
// 11.6 % of its abs-max; with fp16 the factor's rounding error is 8 x smaller than either form).  fp32 mode: Z is fp32.
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
__device__ __forceinline__ uint32_t pack_f16(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}
template <typename U> __device__ __forceinline__ void unpack8_grad(const uint4* p, float (&f)[8]);
template <> __device__ __forceinline__ void unpack8_grad<bf16_t>(const uint4* p, float (&f)[8]) {
  const uint32_t w[4] = {p[0].x, p[0].y, p[0].z, p[0].w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const f16x2_t h = __builtin_bit_cast(f16x2_t, w[e]);
    f[2 * e] = (float)h[0]; f[2 * e + 1] = (float)h[1];
  }
}
template <> __device__ __forceinline__ void unpack8_grad<float>(const uint4* p, float (&f)[8]) { unpack8<float>(p, f); }
__device__ __forceinline__ void store8_grad(bf16_t* p, const float (&v)[8]) {
  uint4 o;
  o.x = pack_f16(v[0], v[1]); o.y = pack_f16(v[2], v[3]); o.z = pack_f16(v[4], v[5]); o.w = pack_f16(v[6], v[7]);
  *reinterpret_cast<uint4*>(p) = o;
}
__device__ __forceinline__ void store8_grad(float* p, const float (&v)[8]) { store8(p, v); }
__device__ __forceinline__ float ld_grad(const bf16_t* p) { return (float)__builtin_bit_cast(_Float16, *p); }
__device__ __forceinline__ float ld_grad(const float* p) { return *p; }
__device__ __forceinline__ void st_grad(bf16_t* p, float v) { *p = __builtin_bit_cast(bf16_t, (_Float16)v); }
__device__ __forceinline__ void st_grad(float* p, float v) { *p = v; }

__host__ __device__ __forceinline__ bool act_reads_z(int act) {
  return act == ETP_ACT_GELU_BWD || act == ETP_ACT_RELU_BWD || act == ETP_ACT_MUL_Z;
}

// Epilogue operands of a 64x64 tile (residual / activation operand / old C / bias: 2 chunks of 8 columns per thread) fetched
// BEFORE the main loop of the LDS-DMA kernel: for the planner's K = 768 products the loop is ~3 us, and two dependent global
// round trips behind it (operand reads, then the bias) were a fifth of the launch (K sweep: 7 us intercept,
// profiles/r02_gemm_sweep.json).  Larger tiles keep the fetch in the epilogue (too many registers).
// (No arrays in these structs on purpose: a runtime-indexed member array keeps the whole object in scratch memory.)
struct EpiChunk { uint4 r0, r1, c0, c1, z0, z1; };
template <typename T, typename TC, int BM, int BN, int NTH = 256> struct EpiPre {
  static constexpr int NCHUNK = BM * (BN / 8) / NTH;
  static constexpr bool ON = NCHUNK <= 2;
  EpiChunk k0, k1;
  bool valid;
  // bias of the thread's 8 columns: every chunk of a thread sits in the same columns (256 % (BN / 8) == 0), so it is
  // fetched once, before the reduction, for every tile size
  float4 b0, b1;
  bool bias_valid;
};
// The activation-backward operand Z (bf16, one 16-byte vector per chunk) of tiles whose other epilogue operands are NOT
// prefetched (128x128: 8 chunks per thread): the GELU-backward dgrad of the FFN spent 11 us of its 25 in the epilogue, two
// dependent global round trips behind the reduction (profiles/r03_gemm_phases.txt); fetched before the reduction instead.
// Indexed only with compile-time constants (fully unrolled loops), so it stays in registers.
// Named members + a select chain instead of an array: a member array indexed by a loop variable lands in scratch memory
// whenever the loop is not fully unrolled (the epilogue's chunk loops carry an early exit).
template <int N> struct ZPre {
  uint4 z0, z1, z2, z3, z4, z5, z6, z7;
  bool valid;
  __device__ __forceinline__ uint4 get(int i) const {
    return i == 0 ? z0 : i == 1 ? z1 : i == 2 ? z2 : i == 3 ? z3 : i == 4 ? z4 : i == 5 ? z5 : i == 6 ? z6 : z7;
  }
};
template <typename T, typename TC, int BM, int BN, int NTH>
__device__ __forceinline__ void z_prefetch(ZPre<BM * (BN / 8) / NTH>& zp, const GemmArgs& g, int m0, int n0, int tid) {
  constexpr int N = BM * (BN / 8) / NTH, CPRW = BN / 8;
  zp.valid = false;
  if constexpr (sizeof(T) == 2 && (N == 4 || N == 8)) {
    if (!g.vec_epilogue || !act_reads_z(g.act)) return;
    zp.valid = true;
#define ETP_ZFETCH(j)                                                                                             \
  (*reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(g.Z) + (long)min(m0 + (tid + (j) * NTH) / CPRW, g.M - 1) * g.ldz + \
                                   min(n0 + ((tid + (j) * NTH) % CPRW) * 8, (g.N - 1) / 8 * 8)))
    zp.z0 = ETP_ZFETCH(0); zp.z1 = ETP_ZFETCH(1); zp.z2 = ETP_ZFETCH(2); zp.z3 = ETP_ZFETCH(3);
    if constexpr (N > 4) { zp.z4 = ETP_ZFETCH(4); zp.z5 = ETP_ZFETCH(5); zp.z6 = ETP_ZFETCH(6); zp.z7 = ETP_ZFETCH(7); }
#undef ETP_ZFETCH
  }
}

template <typename T, typename TC, int BN, int NTH = 256>
__device__ __forceinline__ void epi_fetch_chunk(EpiChunk& k, int j, const GemmArgs& g, const TC* C, int m0, int n0, int ks, int tid) {
  constexpr int CPRW = BN / 8;
  constexpr int VPC = 8 * (int)sizeof(TC) / 16, VPT = 8 * (int)sizeof(T) / 16;
  const int q = tid + j * NTH;
  const int lr = q / CPRW, lc = (q % CPRW) * 8;
  const int rowc = min(m0 + lr, g.M - 1), colc = min(n0 + lc, (g.N - 1) / 8 * 8);
  if (g.R != nullptr) {
    const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const TC*>(g.R) + (long)rowc * g.ldr + colc);
    k.r0 = p[0];
    if constexpr (VPC == 2) k.r1 = p[1];
  }
  if (g.out_mode == 1) {
    const uint4* p = reinterpret_cast<const uint4*>(C + (long)rowc * g.ldc + colc);
    k.c0 = p[0];
    if constexpr (VPC == 2) k.c1 = p[1];
  }
  if (act_reads_z(g.act)) {
    const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(g.Z) + (long)rowc * g.ldz + colc);
    k.z0 = p[0];
    if constexpr (VPT == 2) k.z1 = p[1];
  }
}
template <typename T, typename TC, int BM, int BN, int NTH = 256>
__device__ __forceinline__ void epi_prefetch(EpiPre<T, TC, BM, BN, NTH>& pre, const GemmArgs& g, const TC* C, int m0, int n0, int ks,
                                             int tid) {
  using E = EpiPre<T, TC, BM, BN, NTH>;
  static_assert(NTH % (BN / 8) == 0, "a thread's chunks must share their columns");
  pre.valid = false;
  pre.bias_valid = false;
  if (!g.vec_epilogue) return;
  if (g.bias != nullptr && ks == 0) {
    const int colc = min(n0 + (tid % (BN / 8)) * 8, (g.N - 1) / 8 * 8);
    pre.b0 = *reinterpret_cast<const float4*>(g.bias + colc);
    pre.b1 = *reinterpret_cast<const float4*>(g.bias + colc + 4);
    pre.bias_valid = true;
  }
  if constexpr (E::ON) {
    pre.valid = true;
    epi_fetch_chunk<T, TC, BN, NTH>(pre.k0, 0, g, C, m0, n0, ks, tid);
    if constexpr (E::NCHUNK == 2) epi_fetch_chunk<T, TC, BN, NTH>(pre.k1, 1, g, C, m0, n0, ks, tid);
  }
}

// Epilogue shared by every GEMM kernel, second half: the fp32 accumulator tile sits in LDS as [BM][BN + 4] (staged by the
// caller, followed by a barrier); each thread combines whole 8-column chunks with bias / activation / dropout / residual /
// old C and stores them as 16-byte vectors.
// NCT > 1: NCT such tiles lie behind one another ([NCT][BM][BN + 4]: the partial sums of an intra-workgroup split of the
// reduction, gemm_mm32.hip KS = 2) and are added as they are read.
// FIX >= 0 (gemm_mm32.hip mm32::tile): the epilogue's variant is a COMPILE-TIME constant -- activation code in bits 0-7, bias in bit 8, residual in
// bit 9, dropout in bit 10, plain store (out_mode 0) -- so the eight unrolled chunks are straight-line code instead of eight copies of the whole
// activation / dropout / residual / accumulate decision chain, most of it jumped over (epi_key() below computes the key of a launch).
__host__ __device__ __forceinline__ int epi_key(const GemmArgs& g) {
  if (g.out_mode != 0 || !g.vec_epilogue) return -1;
  return g.act | (g.bias != nullptr ? 0x100 : 0) | (g.R != nullptr ? 0x200 : 0) | (g.drop.p > 0.f ? 0x400 : 0);
}
template <typename T, typename TC, int BM, int BN, int NTH = 256, int NCT = 1, int FIX = -1>
__device__ __forceinline__ void gemm_epilogue_staged(char* smem, const GemmArgs& g, TC* C, int m0, int n0, int ks, int tid,
                                                     const EpiPre<T, TC, BM, BN, NTH>& pre, const ZPre<BM * (BN / 8) / NTH> zp) {
  const int ACT = FIX < 0 ? g.act : (FIX & 0xff);
  constexpr int CP = BN + 4;
  float* ct = reinterpret_cast<float*>(smem);
  const T* R = reinterpret_cast<const T*>(g.R);   // residual has the OUTPUT type when TC != T (see launch checks)
  T* Z = reinterpret_cast<T*>(g.Z);
  constexpr int CPRW = BN / 8;                     // 8-column chunks per tile row
  constexpr int NCHUNK = BM * CPRW / NTH;
  if (FIX >= 0 || g.vec_epilogue) {
    // Phase A: issue every global read of the epilogue (residual / activation operand / old C) up front from
    // clamped in-bounds addresses -- no per-element branches, so the loads overlap instead of serialising.
    constexpr int VPC = 8 * (int)sizeof(TC) / 16;  // 16-byte vectors per 8-element chunk of the C type (1 or 2)
    constexpr int VPT = 8 * (int)sizeof(T) / 16;   // 16-byte vectors per 8-element chunk of T (bf16: 1, fp32: 2)
    const bool has_r = FIX < 0 ? g.R != nullptr : (FIX & 0x200) != 0, has_zr = act_reads_z(ACT),
               has_c = FIX < 0 && (g.out_mode == 1);
    const bool has_bias = FIX < 0 ? (g.bias != nullptr && ks == 0) : (FIX & 0x100) != 0;
    const bool has_drop = FIX < 0 ? g.drop.p > 0.f : (FIX & 0x400) != 0;
    // chunks per pass: all global reads of a pass are issued together (ONE memory round trip per pass; round 2 made four
    // dependent round trips on a 128x128 tile), at most 4 chunks per pass to bound the live epilogue registers
    constexpr int HC = NCHUNK >= 4 ? 4 : NCHUNK;
    const int col_last = (g.N - 1) / 8 * 8;
#pragma unroll
    for (int h0 = 0; h0 < NCHUNK; h0 += HC) {
    uint4 rr[HC][VPC], cc[HC][VPC], zz[HC][VPT];
    const bool use_pre = pre.valid;
    if (use_pre) {
      if constexpr (EpiPre<T, TC, BM, BN, NTH>::ON) {      // fetched before the main loop (these tiles have HC == NCHUNK <= 2: one pass)
        rr[0][0] = pre.k0.r0; cc[0][0] = pre.k0.c0; zz[0][0] = pre.k0.z0;
        if constexpr (VPC == 2) { rr[0][1] = pre.k0.r1; cc[0][1] = pre.k0.c1; }
        if constexpr (VPT == 2) zz[0][1] = pre.k0.z1;
        if constexpr (HC == 2) {
          rr[1][0] = pre.k1.r0; cc[1][0] = pre.k1.c0; zz[1][0] = pre.k1.z0;
          if constexpr (VPC == 2) { rr[1][1] = pre.k1.r1; cc[1][1] = pre.k1.c1; }
          if constexpr (VPT == 2) zz[1][1] = pre.k1.z1;
        }
      }
    }
#pragma unroll
    for (int jh = 0; jh < HC; ++jh) {
      if (use_pre) break;
      const int jj = jh, q = tid + (h0 + jh) * NTH;
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
        if (zp.valid) {
          if constexpr (VPT == 1) zz[jj][0] = zp.get(h0 + jh);
        } else {
          const uint4* p = reinterpret_cast<const uint4*>(Z + (long)rowc * g.ldz + colc);
#pragma unroll
          for (int u = 0; u < VPT; ++u) zz[jj][u] = p[u];
        }
      }
    }
    // Phase B: combine and store
#pragma unroll
    for (int jh = 0; jh < HC; ++jh) {
      const int jj = jh, q = tid + (h0 + jh) * NTH;
      const int lr = q / CPRW, lc = (q % CPRW) * 8;
      const int row = m0 + lr, col = n0 + lc;
      const bool ok = row < g.M && col < g.N;
      const int colc = min(col, col_last);
      float v[8];
      {
        const float4 x0 = *reinterpret_cast<const float4*>(ct + lr * CP + lc);
        const float4 x1 = *reinterpret_cast<const float4*>(ct + lr * CP + lc + 4);
        v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
#pragma unroll
        for (int t = 1; t < NCT; ++t) {
          const float4 y0 = *reinterpret_cast<const float4*>(ct + t * (BM * CP) + lr * CP + lc);
          const float4 y1 = *reinterpret_cast<const float4*>(ct + t * (BM * CP) + lr * CP + lc + 4);
          v[0] += y0.x; v[1] += y0.y; v[2] += y0.z; v[3] += y0.w; v[4] += y1.x; v[5] += y1.y; v[6] += y1.z; v[7] += y1.w;
        }
      }
      if (has_bias) {
        float4 b0, b1;
        if (pre.bias_valid) {
          b0 = pre.b0; b1 = pre.b1;
        } else {
          b0 = *reinterpret_cast<const float4*>(g.bias + colc); b1 = *reinterpret_cast<const float4*>(g.bias + colc + 4);
        }
        const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * g.alpha + bv[e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= g.alpha;
      }
      if (ACT == ETP_ACT_GELU) {
        if (ok) store8(Z + (long)row * g.ldz + col, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
      } else if (ACT == ETP_ACT_GELU_SAVEGRAD) {
        float dv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) gelu_erf_both(v[e], v[e], dv[e]);
        if (ok) store8_grad(Z + (long)row * g.ldz + col, dv);
      } else if (ACT == ETP_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      } else if (has_zr) {
        float zf[8];
        if (ACT == ETP_ACT_MUL_Z) unpack8_grad<T>(zz[jj], zf);
        else unpack8<T>(zz[jj], zf);
        if (ACT == ETP_ACT_GELU_BWD) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= gelu_erf_grad(zf[e]);
        } else if (ACT == ETP_ACT_MUL_Z) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= zf[e];
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = zf[e] > 0.f ? v[e] : 0.f;
        }
      }
      if (has_drop) {
        const uint32_t e0 = (uint32_t)row * (uint32_t)g.N + (uint32_t)col;
        float dm[8];
        drop_mult_run<8>(g.drop.seed, e0, g.drop.p, g.drop.inv_keep, dm);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= dm[e];
      }
      if (has_r) {
        float rf[8];
        unpack8<TC>(rr[jj], rf);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rf[e];
      }
      TC* dst = C + (long)row * g.ldc + col;
      if (FIX < 0 && g.out_mode == 2) {
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
    }
    return;
  }
  // scalar fallback (odd leading dimensions / unaligned bases): one element per thread-iteration, row-major
  for (int q = tid; q < BM * BN; q += NTH) {
    const int lr = q / BN, lc = q % BN;
    const int row = m0 + lr, col = n0 + lc;
    if (row >= g.M || col >= g.N) continue;
    float v = ct[lr * CP + lc];
#pragma unroll
    for (int t = 1; t < NCT; ++t) v += ct[t * (BM * CP) + lr * CP + lc];
    v = v * g.alpha + ((g.bias != nullptr && ks == 0) ? g.bias[col] : 0.f);
    if (g.act == ETP_ACT_GELU) {
      Elem<T>::st(Z + (long)row * g.ldz + col, v);
      v = gelu_erf(v);
    } else if (g.act == ETP_ACT_GELU_SAVEGRAD) {
      float dv;
      gelu_erf_both(v, v, dv);
      st_grad(Z + (long)row * g.ldz + col, dv);
    } else if (g.act == ETP_ACT_MUL_Z) {
      v *= ld_grad(Z + (long)row * g.ldz + col);
    } else if (g.act == ETP_ACT_RELU) {
      v = fmaxf(v, 0.f);
    } else if (g.act == ETP_ACT_GELU_BWD) {
      v *= gelu_erf_grad(Elem<T>::ld(Z + (long)row * g.ldz + col));
    } else if (g.act == ETP_ACT_RELU_BWD) {
      v = (Elem<T>::ld(Z + (long)row * g.ldz + col) > 0.f) ? v : 0.f;
    }
    if (g.drop.p > 0.f) v *= drop_mult(g.drop.seed, (uint32_t)row * (uint32_t)g.N + (uint32_t)col, g.drop.p, g.drop.inv_keep);
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

// First half for the 16x16 MFMA accumulator layout (gemm.hip): NTH threads = NTH / 64 wavefronts in a (NTH / 128) x 2 grid over
// the tile.  The MFMA C layout (lane: row = 4*(lane>>4)+r, col = lane&15) gives 2-byte scattered stores, so the tile is staged
// through LDS as fp32 [BM][BN+4] and written back as whole 8-column chunks per thread.
template <typename T, typename TC, int BM, int BN, int NTH = 256>
__device__ __forceinline__ void gemm_epilogue(f32x4_t (&acc)[BM / (NTH / 128) / 16][BN / 32], char* smem, const GemmArgs& g, TC* C,
                                              int m0, int n0, int ks, int tid, const EpiPre<T, TC, BM, BN, NTH>& pre,
                                              const ZPre<BM * (BN / 8) / NTH> zp) {
  constexpr int WM = NTH / 128;                      // wavefront rows of the (WM x 2) grid
  constexpr int MT = BM / WM / 16, NT = BN / 32;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int i = lane & 15, gq = lane >> 4;
  constexpr int CP = BN + 4;
  float* ct = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        ct[(wr * (BM / WM) + a * 16 + gq * 4 + r) * CP + wc * (BN / 2) + b * 16 + i] = acc[a][b][r];
  __syncthreads();
  gemm_epilogue_staged<T, TC, BM, BN, NTH>(smem, g, C, m0, n0, ks, tid, pre, zp);
}

// XCD-aware workgroup -> tile map.  Workgroups are dispatched round-robin over the 8 XCDs (workgroup i runs on XCD
// i % 8) and every XCD has a private 4 MiB L2: with the plain row-major map each XCD touches every row tile AND every
// column tile, so both operands are fetched from the fabric once per XCD (PMC, round 1: 60 MB fetched per weight-gradient
// launch for 16 MB of algorithmic operand bytes).  Remapped, the workgroups of one XCD own a CONTIGUOUS slab of the tile
// grid, sliced along the longer tile axis: its L2 then holds 1/8 of the long operand plus the short one.
// (bijective form of the remap: cdna_hip_programming.md, "XCD swizzle must be bijective")
__device__ __forceinline__ void tile_of_block(int bid, int nwg, int tiles_m, int tiles_n, int xcd_map, int& tm, int& tn) {
  int id = bid;
  if (xcd_map) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  if (!xcd_map || tiles_m >= tiles_n) { tm = id / tiles_n; tn = id % tiles_n; }
  else { tn = id / tiles_m; tm = id % tiles_m; }
}

// ---- phase probe (tools/gemm_phase_probe.py; profiles/r03_gemm_phases.txt) --------------------------------------------
// With GemmArgs::dbg set, thread 0 of every workgroup records s_memrealtime (100 MHz, chip-wide) at entry / exit and
// s_memtime (shader clock) at entry, first slab visible, end of the reduction and end of the epilogue, plus HW_ID / XCC_ID.
struct PhaseProbe {
  unsigned long long rt0, mt0, mt1, mt2;
  bool on;
};
__device__ __forceinline__ void probe_begin(PhaseProbe& p, const GemmArgs& g) {
  p.on = g.dbg != nullptr && threadIdx.x == 0;
  if (p.on) { p.rt0 = __builtin_amdgcn_s_memrealtime(); p.mt0 = __builtin_amdgcn_s_memtime(); p.mt1 = p.mt2 = p.mt0; }
}
__device__ __forceinline__ void probe_end(const PhaseProbe& p, const GemmArgs& g, int rec, int nk) {
  if (!p.on) return;
  unsigned long long* d = g.dbg + (size_t)rec * 8;
  d[0] = p.rt0; d[1] = __builtin_amdgcn_s_memrealtime();
  d[2] = p.mt0; d[3] = p.mt1; d[4] = p.mt2; d[5] = __builtin_amdgcn_s_memtime();
  d[6] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
  d[7] = (unsigned long long)nk;
}

// ---- host side (definitions in gemm.hip) ----------------------------------------------------------------------------
struct ProfRec { int id; hipEvent_t a, b; double flops, bytes; };
// true: the launch named `nm` is to be bracketed by events on `st` (etp_prof_enable / etp_prof_filter); records `a`
bool prof_begin(const char* nm, double flops, double bytes, hipStream_t st, ProfRec& rec);
void prof_end(const ProfRec& rec, hipStream_t st);
// device buffer slot of the next probed launch (nullptr: probe off / full / recording a graph)
unsigned long long* probe_slot(const char* name, long wgs, int M, int N, int K);

// ---- gemm_mm32.hip: bf16 tiles on 32x32x16 MFMAs with the order-pinned main loop (whole tiles only) ---------------------------
int mm32_class(const GemmArgs& g, int nbatch);       // 0: not taken; else the tile class to pass to launch_mm32 (g = prepared args)
int launch_mm32(int c_dtype, int ta, int tb, const GemmArgs& g, int cls, hipStream_t st);
bool mm32_group_ok(const GemmGroup& grp);            // TN fp32-out group of whole 128x128 tiles
int launch_mm32_group(GemmGroup& grp, hipStream_t st);

}  // namespace etp
