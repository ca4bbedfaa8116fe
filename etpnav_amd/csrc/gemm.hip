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

template <int N> struct Regs { uint4 v[N]; };

// global -> registers for one tile (zero-filled outside [rows_total) x [k_end))
template <typename T, bool TR, int ROWS>
__device__ __forceinline__ void tile_load(Regs<TileGeom<T, TR, ROWS>::PER_THREAD>& r, const T* __restrict__ base, long ld,
                                          int row0, int rows_total, int k0, int k_end, int tid) {
  using G = TileGeom<T, TR, ROWS>;
#pragma unroll
  for (int j = 0; j < G::PER_THREAD; ++j) {
    const int q = tid + j * 256;
    const int lr = q / G::CPR, c = q % G::CPR;
    uint4 v = make_uint4(0, 0, 0, 0);
    if constexpr (!TR) {
      const int row = row0 + lr, k = k0 + c * G::EPC;
      if (row < rows_total && k < k_end) v = *reinterpret_cast<const uint4*>(base + (long)row * ld + k);
    } else {
      const int k = k0 + lr, row = row0 + c * G::EPC;
      if (k < k_end && row < rows_total) v = *reinterpret_cast<const uint4*>(base + (long)k * ld + row);
    }
    r.v[j] = v;
  }
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
    *reinterpret_cast<uint4*>(lds + off) = r.v[j];
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

  // ---- epilogue: lane holds C[row = 4*(lane>>4)+r][col = lane&15] of each 16x16 tile ----
  const int i = lane & 15, gq = lane >> 4;
  const T* R = reinterpret_cast<const T*>(g.R);
  T* Z = reinterpret_cast<T*>(g.Z);
#pragma unroll
  for (int b = 0; b < NT; ++b) {
    const int col = n0 + wc * (BN / 2) + b * 16 + i;
    if (col >= g.N) continue;
    const float bias = (g.bias != nullptr && ks == 0) ? g.bias[col] : 0.f;
#pragma unroll
    for (int a = 0; a < MT; ++a) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wr * (BM / 2) + a * 16 + gq * 4 + r;
        if (row >= g.M) continue;
        float v = acc[a][b][r] * g.alpha + bias;
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
        if (R != nullptr) v += Elem<T>::ld(R + (long)row * g.ldr + col);
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
    }
  }
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
  constexpr int smem = 2 * (GA::BYTES + GB::BYTES);
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

int launch_gemm(int dtype, int c_dtype, int ta, int tb, const GemmArgs& g, int nbatch, hipStream_t st) {
  ETP_REQUIRE(g.M > 0 && g.N > 0 && g.K >= 0 && nbatch > 0 && g.ksplit >= 1 && g.nb_inner >= 1, "bad dims");
  const int epc = dtype == ETP_BF16 ? 8 : 4;
  ETP_REQUIRE(g.lda % epc == 0 && g.ldb % epc == 0, "lda/ldb must be multiples of a 16-byte chunk");
  ETP_REQUIRE(((uintptr_t)g.A % 16 == 0) && ((uintptr_t)g.B % 16 == 0), "A/B must be 16-byte aligned");
  ETP_REQUIRE((g.sAo % epc == 0) && (g.sAi % epc == 0) && (g.sBo % epc == 0) && (g.sBi % epc == 0),
              "batch strides must keep 16-byte alignment");
  ETP_REQUIRE(g.out_mode != 2 || c_dtype == ETP_F32, "atomic accumulation needs an fp32 C");
  ETP_REQUIRE(g.ksplit == 1 || g.out_mode == 2, "split-K needs atomic accumulation");
  if (dtype == ETP_F32) {
    ETP_REQUIRE(c_dtype == ETP_F32, "fp32 operands need an fp32 C");
    return launch_trans<float, float>(ta, tb, g, nbatch, st);
  }
  if (c_dtype == ETP_F32) return launch_trans<bf16_t, float>(ta, tb, g, nbatch, st);
  return launch_trans<bf16_t, bf16_t>(ta, tb, g, nbatch, st);
}

}  // namespace etp
