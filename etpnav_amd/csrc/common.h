// Shared device/host helpers for the ETPNav planner kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/etpnav_hip.h"
#include "launch.h"
#include "options.h"

namespace etp {

typedef uint16_t bf16_t;  // raw bfloat16 storage
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short short4_t;

constexpr int WAVE = 64;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// gfx950 converts in hardware (v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays a quiet NaN): one instruction per PAIR of
// values instead of the ~7 VALU operations of the integer rounding sequence
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// 4 consecutive elements <-> float[4] (16 B for f32, 8 B for bf16)
__device__ __forceinline__ void load4(const float* p, float (&v)[4]) {
  float4 t = *reinterpret_cast<const float4*>(p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void load4(const bf16_t* p, float (&v)[4]) {
  uint2 t = *reinterpret_cast<const uint2*>(p);
  v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
  v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
__device__ __forceinline__ void store4(float* p, const float (&v)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store4(bf16_t* p, const float (&v)[4]) {
  uint2 t;
  t.x = pack_bf16(v[0], v[1]);
  t.y = pack_bf16(v[2], v[3]);
  *reinterpret_cast<uint2*>(p) = t;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// erf-GELU (vilmodel_cmt.py:30-37, exact erf, not the tanh form) and its derivative.  erf by Abramowitz & Stegun 7.1.26
// (|error| <= 1.5e-7 before fp32 rounding): one v_exp, one v_rcp and six FMAs instead of the ~40 VALU operations of the
// library erff -- the GELU epilogues run 64 of these per thread on a 128x128 tile.  Both functions share
// q = exp(-x^2/2): erf(x/sqrt2) = sign(x) * (1 - poly(t) * q),  t = 1 / (1 + p |x| / sqrt2).
__device__ __forceinline__ void gelu_parts(float x, float& phi_cdf, float& q) {
  // constants folded by hand (the compiler may not reassociate): p |x| / sqrt2 = 0.23164189 |x|; exp(-x^2/2) = exp2(-0.72134752 x^2)
  // v_rcp_f32 (1 ulp; the argument is in [1, ~5]).  __frcp_rn compiled to the IEEE division sequence (2 x v_div_scale, v_rcp, 4 FMAs,
  // v_div_fmas, v_div_fixup: ten instructions per element; round 5, found in the ISA after the epilogues measured VALU-bound)
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, fabsf(x), 1.0f));
  q = __builtin_amdgcn_exp2f((x * x) * -0.72134752044448170368f);
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = 1.0f - p * t * q;                     // erf(|x| / sqrt2)
  phi_cdf = 0.5f * (1.0f + copysignf(e, x));
}
__device__ __forceinline__ float gelu_erf(float x) {
  float c, q;
  gelu_parts(x, c, q);
  return x * c;
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  float c, q;
  gelu_parts(x, c, q);
  return c + x * 0.39894228040143267794f * q;
}
// y = gelu(x) and dy = gelu'(x) from ONE evaluation of the shared parts (x may alias y)
__device__ __forceinline__ void gelu_erf_both(float x, float& y, float& dy) {
  float c, q;
  gelu_parts(x, c, q);
  dy = fmaf(x * 0.39894228040143267794f, q, c);
  y = x * c;
}

// ---- counter-based dropout: mask bit = hash(seed, element index); forward and backward recompute the same mask, none is
// stored.  `seed` already mixes the per-step seed with the dropout site id (drop_site_seed on the host).
__host__ __device__ __forceinline__ uint32_t drop_hash(uint32_t seed, uint32_t idx) {
  uint32_t x = idx * 0x9E3779B1u + seed;
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  x += seed * 0x27D4EB2Fu; x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12;
  return x;
}
// Per-element mask bits (round 4): ONE 32-bit avalanche per PAIR of consecutive elements, 16 bits each -- the epilogues hash
// every element of every hidden / attention-probability tensor of a step (~130 M per step), and the five 32-bit multiplies of
// drop_hash (quarter rate on CDNA) were a third of an fp32-stream GEMM epilogue.  keep <=> 16 bits >= round(p * 65536)
// (p = 0.1 -> 6554 / 65536 = 0.100006).  drop_hash stays the site-seed mixer (host side, once per site).
__host__ __device__ __forceinline__ uint32_t drop_pair(uint32_t seed, uint32_t pair) {
  uint32_t x = pair * 0x9E3779B1u + seed;
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ uint32_t drop_thr(float p) { return (uint32_t)(p * 65536.0f + 0.5f); }
// multiplier of element idx under dropout probability p: 0 (dropped) or 1/(1-p)
__host__ __device__ __forceinline__ float drop_mult(uint32_t seed, uint32_t idx, float p, float inv_keep) {
  const uint32_t w = drop_pair(seed, idx >> 1);
  const uint32_t h = (idx & 1u) ? (w >> 16) : (w & 0xFFFFu);
  return h >= drop_thr(p) ? inv_keep : 0.f;
}
// the same for N consecutive elements idx0 .. idx0 + N - 1 (N even): N / 2 avalanches when idx0 is even (every tensor of the
// planner has an even row length and the kernels walk rows in chunks of 4 or 8), N / 2 + 1 otherwise
template <int N>
__host__ __device__ __forceinline__ void drop_mult_run(uint32_t seed, uint32_t idx0, float p, float inv_keep, float (&m)[N]) {
  static_assert(N % 2 == 0, "runs of an even number of elements");
  const uint32_t thr = drop_thr(p), q0 = idx0 >> 1;
  if ((idx0 & 1u) == 0u) {
#pragma unroll
    for (int j = 0; j < N / 2; ++j) {
      const uint32_t w = drop_pair(seed, q0 + j);
      m[2 * j] = (w & 0xFFFFu) >= thr ? inv_keep : 0.f;
      m[2 * j + 1] = (w >> 16) >= thr ? inv_keep : 0.f;
    }
  } else {
    uint32_t w = drop_pair(seed, q0);
    m[0] = (w >> 16) >= thr ? inv_keep : 0.f;
#pragma unroll
    for (int j = 1; j <= N / 2; ++j) {
      w = drop_pair(seed, q0 + j);
      m[2 * j - 1] = (w & 0xFFFFu) >= thr ? inv_keep : 0.f;
      if (2 * j < N) m[2 * j] = (w >> 16) >= thr ? inv_keep : 0.f;
    }
  }
}
struct Drop {                          // dropout of one site; p == 0 -> identity
  float p, inv_keep; uint32_t seed;
};
inline Drop drop_none() { Drop d; d.p = 0.f; d.inv_keep = 1.f; d.seed = 0; return d; }
inline Drop drop_site(float p, uint64_t step_seed, uint32_t site) {
  Drop d;
  d.p = p; d.inv_keep = p > 0.f ? 1.0f / (1.0f - p) : 1.f;
  d.seed = drop_hash((uint32_t)(step_seed ^ (step_seed >> 32)) * 0x9E3779B1u + 0x7F4A7C15u, site * 0x632BE5ABu + 17u);
  return d;
}

// ---- host side ----
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);
int check_hip(hipError_t e, const char* what);

#define ETP_CHECK_HIP(expr)                                   \
  do {                                                        \
    int _rc = ::etp::check_hip((expr), #expr);                \
    if (_rc) return _rc;                                      \
  } while (0)
#define ETP_CHECK_LAUNCH(name) ETP_CHECK_HIP(::etp::launch_status())
#define ETP_REQUIRE(cond, msg)                                \
  do {                                                        \
    if (!(cond)) return ::etp::fail(ETP_ERR_INVALID, std::string(__func__) + ": " + (msg)); \
  } while (0)
#define ETP_TRY(expr)                                         \
  do {                                                        \
    int _rc = (expr);                                         \
    if (_rc) return _rc;                                      \
  } while (0)

inline size_t dtype_size(int dt) { return dt == ETP_BF16 ? 2 : 4; }
inline long round_up(long x, long m) { return (x + m - 1) / m * m; }

// ---- internal GEMM interface (gemm.hip) ----
struct GemmArgs {
  const void* A; const void* B; void* C;
  int M, N, K;
  long lda, ldb, ldc;
  int nb_inner;                       // batch z -> (zo = z / nb_inner, zi = z % nb_inner)
  long sAo, sAi, sBo, sBi, sCo, sCi;  // batch strides (elements)
  int ksplit;                         // split-K (needs out_mode 2)
  float alpha;
  const float* bias;                  // [N] fp32 or null
  const void* R; long ldr;            // residual / grad-add operand (dtype of C) or null (non-batched only)
  void* Z; long ldz;                  // aux tensor (T): act 1 writes, act 3/4 reads
  int act;                            // ETP_ACT_*
  int out_mode;                       // 0 store, 1 C += v, 2 atomicAdd (fp32 C only)
  int vec_epilogue;                   // set by launch_gemm: 16-byte epilogue accesses are legal
  float* a_colsum;                    // TN (wgrad) only: a_colsum[m] += sum_k A[m,k]  (bias gradient), LDS-DMA kernel only
  int xcd_map;                        // 1: XCD-aware workgroup->tile order (set by launch_gemm; env ETP_GEMM_XCD=0 disables)
  Drop drop;                          // dropout on the epilogue value (after activation / its backward, before the residual);
                                      // element index = row * N + col
  unsigned long long* dbg;            // phase-probe records (8 x u64 per workgroup, tools/gemm_phase_probe.py) or null
};
int launch_gemm(int dtype, int c_dtype, int transA, int transB, const GemmArgs& g, int nbatch, hipStream_t st);
// Several independent, unbatched, unsplit products of ONE (dtype, c_dtype, transA, transB) class in a single grid (LDS-DMA
// kernel only: every reduction length must pass gemm_uses_dma).  n <= ETP_GEMM_GROUP_MAX.
constexpr int ETP_GEMM_GROUP_MAX = 8;
struct GemmGroup { int n; int xcd_chunks; int tile_start[ETP_GEMM_GROUP_MAX + 1]; GemmArgs g[ETP_GEMM_GROUP_MAX]; };
int launch_gemm_group(int dtype, int c_dtype, int transA, int transB, const GemmArgs* gs, int n, hipStream_t st);
bool gemm_uses_dma(int dtype, int K, int ksplit);   // true when launch_gemm will take the LDS-DMA kernel for this reduction
// graphrec.hip: explicit hipGraph recording of everything issued through launch.h
int rec_begin();
int rec_end(hipGraph_t* graph, hipGraphExec_t* exec, long* n_kernels, long* n_edges);
void rec_abort();
void ktime_enable(bool on);
void ktime_reset();
long ktime_report(char* buf, long cap);
void gemm_probe_set(unsigned long long* buf, long launches);
long gemm_probe_count();
int gemm_probe_meta(long i, char* name, int cap, int* dims);
void prof_enable(bool on);
void prof_filter(const char* name_part);
void prof_reset();
int prof_report(etp_prof_entry* out, int cap);
// capi.hip: with a stamp sink installed (etp_stamp_sink), records the stream's device-side arrival time at this point of the
// planner's issue order under `tag`; a no-op otherwise (tools/chain_waits.py)
void stamp_mark(hipStream_t st, int tag);

}  // namespace etp
