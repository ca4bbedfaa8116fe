// PROTOTYPE for round 5 -- NOT part of libetpnav_hip.so; compiled here (hipcc cross-compiles) but NEVER RUN so far.
//
//   C[M,N] = epilogue( LayerNorm(S)[M,768] . W^T )          S = the fp32 pre-LN sum the previous sub-block's GEMM wrote
//
// Why.  The node-side products of the cross-modal layers (M = B*G = 512 rows) are launch- and
// latency-bound: each x-layer runs 11 kernels of 4-16 us with ~3 us of boundary between them, three of them LayerNorms whose
// only consumers are the next GEMM's A operand (bf16), the residual of the sub-block after it (fp32) and the backward (mean,
// rstd).  For these shapes a consumer workgroup reads ALL of its A rows anyway (K = H = 768), so it can do the LayerNorm itself:
//   prologue  each wavefront normalises BM/4 rows exactly as ln_fwd_s_kernel does (same operation order: bit-identical y),
//             writes the bf16 rows into an LDS panel [12 slabs][BM rows][128 B] in the swizzled layout frag_load expects, and
//             -- workgroups of the first tile column only -- writes y (fp32), the bf16 copy and (mean, rstd) to memory for the
//             residual add, the weight gradient and the backward;
//   loop      B (the weight) streams through the LDS-DMA ring as in gemm.hip's dma_tile; A fragments come from the resident
//             panel (no A traffic in the loop at all);
//   epilogue  the shared fused epilogue (bias, GELU + saved pre-activation, ReLU, ...).
// The B ring is issued before the prologue so that the weight's first slabs could land while the rows are normalised -- but
// KNOWN LIMIT of this first form: vmcnt retires in order, so the compiler's own wait for the first row's loads (younger than the
// ring's DMA pieces) also waits for the whole ring; the overlap only exists for rows 2.. of a wavefront.  To try on the GPU:
// issue the first rows' loads in front of the ring, or count the waits by hand (inline-asm loads).
// Cost model: the panel costs BM x 3 KiB of fp32 reads per workgroup (96 KiB at BM = 32, from L2: the producer just wrote it),
// repeated by the N/BN column tiles; at M = 512 that is 12-48 x 1.5 MB = 19-75 MB of L2 reads per launch against one
// launch + one boundary saved (4 + 3 us).  NOT for the M = 2560 text products (they are feed-bound).
//
// Test / timing harness: tools/experiments/r05_ln_prologue_test.py (builds this file into a small shared library and compares
// with etp_ln_stream_fwd + etp_gemm of the product library).
#include <string.h>

#include "../../etpnav_amd/csrc/gemm_tiles.h"
#include "../../etpnav_amd/csrc/launch.h"

namespace etp {

struct LnPro {
  const float* gamma; const float* beta; float eps;
  float* y;        // [M][768] fp32 LayerNorm output (the residual stream) or null
  bf16_t* yt;      // [M][768] bf16 copy (X operand of the weight gradient) or null
  float* stats;    // [M][2] mean, rstd or null
};

template <typename TC, bool TB, int BM, int BN, int STAGES>
__global__ __launch_bounds__(256, 2) void ln_gemm_kernel(const GemmArgs g, const LnPro ln) {
  using T = bf16_t;
  constexpr int H = 768, NSLAB = H / 64, NW = 4;
  using GA = TileGeom<T, false, BM, 0>;              // one 64-k slab of the panel: [BM][128 B]
  using GB = TileGeom<T, TB, BN, 0>;
  constexpr int MT = BM / 2 / 16, NT = BN / 32;
  constexpr int PER_SLAB = DmaPlan<T, TB, BN, NW>::PER_WAVE;
  constexpr int PANEL = NSLAB * GA::BYTES;
  static_assert(BM == 32 || BM == 64, "panel rows");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* panel = smem;
  char* ring = smem + PANEL;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  int tm, tn;
  tile_of_block(blockIdx.x, gridDim.x, (g.M + BM - 1) / BM, (g.N + BN - 1) / BN, g.xcd_map, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const float* S = reinterpret_cast<const float*>(g.A);
  const T* B = reinterpret_cast<const T*>(g.B);
  TC* C = reinterpret_cast<TC*>(g.C);
  constexpr int nk = NSLAB;

  // the weight's ring goes in flight first
  DmaPlan<T, TB, BN, NW> pb;
  dma_plan<T, TB, BN, NW>(pb, B, g.ldb, n0, g.N, 0, tid);
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  const unsigned ring0 = lds0 + (unsigned)PANEL;
#pragma unroll
  for (int s = 0; s < STAGES; ++s)
    if (s < nk) dma_issue<T, TB, BN, NW>(pb, ring0 + s * GB::BYTES, g.ldb);

  EpiPre<T, TC, BM, BN, 256> pre;
  epi_prefetch<T, TC, BM, BN, 256>(pre, g, C, m0, n0, 0, tid);
  ZPre<BM * (BN / 8) / 256> zp;
  zp.valid = false;

  // ---- LayerNorm prologue: same arithmetic and order as ln_fwd_s_kernel (norm.hip) --------------------------------------
  {
    float gm[3][4], bt[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      load4(ln.gamma + c * 256 + lane * 4, gm[c]);
      load4(ln.beta + c * 256 + lane * 4, bt[c]);
    }
    const bool publish = tn == 0;
    for (int r = wave; r < BM; r += 4) {
      const int row = m0 + r, rowc = min(row, g.M - 1);
      float v[3][4];
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        load4(S + (long)rowc * g.lda + c * 256 + lane * 4, v[c]);
        s += v[c][0] + v[c][1] + v[c][2] + v[c][3];
      }
      const float mean = wave_sum(s) * (1.0f / H);
      float q = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[c][e] -= mean; q += v[c][e] * v[c][e]; }
      const float rstd = rsqrtf(wave_sum(q) * (1.0f / H) + ln.eps);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int col = c * 256 + lane * 4;
        float o[4] = {v[c][0] * rstd * gm[c][0] + bt[c][0], v[c][1] * rstd * gm[c][1] + bt[c][1],
                      v[c][2] * rstd * gm[c][2] + bt[c][2], v[c][3] * rstd * gm[c][3] + bt[c][3]};
        if (publish && row < g.M) {
          if (ln.y != nullptr) store4(ln.y + (long)row * H + col, o);
          if (ln.yt != nullptr) store4(ln.yt + (long)row * H + col, o);
        }
        // bf16 into the panel: slab = col / 64, 16-byte chunk (col % 64) / 8 swizzled with the row, this lane's 4 values = half a chunk
        const int slab = col >> 6, ch = (col & 63) >> 3, half = (col >> 2) & 1;
        uint2 w;
        w.x = pack_bf16(o[0], o[1]);
        w.y = pack_bf16(o[2], o[3]);
        *reinterpret_cast<uint2*>(panel + slab * GA::BYTES + r * 128 + ((ch ^ (r & 7)) << 4) + half * 8) = w;
      }
      if (publish && row < g.M && ln.stats != nullptr && lane == 0) { ln.stats[2 * row] = mean; ln.stats[2 * row + 1] = rstd; }
    }
  }

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // ---- reduction: A fragments from the panel, B through the ring (the loop of gemm.hip's dma_tile without an A ring) ------
  Frag<T> fa0[MT], fb0[NT], fa1[MT], fb1[NT];
  wait_slabs<PER_SLAB, STAGES - 1>(min(STAGES - 1, nk - 1));       // B slab 0 landed, the rest of the ring stays in flight
  __syncthreads();                                                  // ... and the panel is complete
  load_frags<T, false, TB, BM, BN, NW>(fa0, fb0, panel, ring, 0, wr, wc, lane);
#define ETP_MMA_SET(FA, FB)                                            \
  _Pragma("unroll") for (int a = 0; a < MT; ++a)                       \
      _Pragma("unroll") for (int b = 0; b < NT; ++b) mma_step(acc[a][b], FA[a], FB[b]);
  int t = 0;
  for (; t + 1 < nk; ++t) {
    const char* sa = panel + t * GA::BYTES;
    const char* sb = ring + (t % STAGES) * GB::BYTES;
    load_frags<T, false, TB, BM, BN, NW>(fa1, fb1, sa, sb, 1, wr, wc, lane);
    ETP_MMA_SET(fa0, fb0)
    wait_slabs<PER_SLAB, STAGES - 2>(min(STAGES - 2, nk - 2 - t));  // B slab t+1 landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // my reads of ring slot t retired
    __builtin_amdgcn_s_barrier();
    if (t + STAGES < nk) dma_issue<T, TB, BN, NW>(pb, ring0 + (t % STAGES) * GB::BYTES, g.ldb);
    const char* na = panel + (t + 1) * GA::BYTES;
    const char* nb = ring + ((t + 1) % STAGES) * GB::BYTES;
    load_frags<T, false, TB, BM, BN, NW>(fa0, fb0, na, nb, 0, wr, wc, lane);
    ETP_MMA_SET(fa1, fb1)
  }
  {
    const char* sa = panel + t * GA::BYTES;
    const char* sb = ring + (t % STAGES) * GB::BYTES;
    load_frags<T, false, TB, BM, BN, NW>(fa1, fb1, sa, sb, 1, wr, wc, lane);
    ETP_MMA_SET(fa0, fb0)
    ETP_MMA_SET(fa1, fb1)
  }
#undef ETP_MMA_SET
  wait_vmcnt<0>();
  __syncthreads();                                                  // every wavefront is done with the panel: the C tile may overwrite it
  gemm_epilogue<T, TC, BM, BN, 256>(acc, smem, g, C, m0, n0, 0, tid, pre, zp);
}

// ---- backward companion: dX[M,N] = epilogue( LayerNormBackward(dY, S)[M,768] . W )   W stored [768 (reduction)][N] (NN) ----------
// The dgrad that follows a LayerNorm backward reads ALL 768 columns of its A rows as well (the reduction runs over the hidden size),
// so the same panel trick applies: the prologue does ln_bwd_s_kernel's row arithmetic (norm.hip; same order), the first tile column
// publishes dx (fp32: the residual-path gradient), the bf16 operand copy with the dropout mask of the dense output it belongs to
// (X operand of the weight gradient) and this row block's column sums of dy * xhat and dy as ONE slab [2][768] (the leaf reduction
// over slabs stays a separate launch on the weight-gradient stream, as today).
struct LnBwdPro {
  const float* dy; const float* x; const float* stats; const float* gamma; const float* add;   // add may be null
  float* dx;        // [M][768] fp32 or null
  bf16_t* dxt;      // [M][768] bf16 operand copy (masked) or null
  float* part;      // [tiles_m][2][768] fp32 slabs of dgamma / dbeta partial sums, or null
  Drop drop;
};

template <typename TC, int BM, int BN, int STAGES>
__global__ __launch_bounds__(256, 2) void ln_bwd_gemm_kernel(const GemmArgs g, const LnBwdPro ln) {
  using T = bf16_t;
  constexpr bool TB = true;
  constexpr int H = 768, NSLAB = H / 64, NW = 4;
  using GA = TileGeom<T, false, BM, 0>;
  using GB = TileGeom<T, TB, BN, 0>;
  constexpr int MT = BM / 2 / 16, NT = BN / 32;
  constexpr int PER_SLAB = DmaPlan<T, TB, BN, NW>::PER_WAVE;
  constexpr int PANEL = NSLAB * GA::BYTES, RING = STAGES * GB::BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* panel = smem;
  char* ring = smem + PANEL;
  float* red = reinterpret_cast<float*>(smem + PANEL + RING);        // [4 waves][2][H] column-sum exchange (first tile column only)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  int tm, tn;
  tile_of_block(blockIdx.x, gridDim.x, (g.M + BM - 1) / BM, (g.N + BN - 1) / BN, g.xcd_map, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const T* B = reinterpret_cast<const T*>(g.B);
  TC* C = reinterpret_cast<TC*>(g.C);
  constexpr int nk = NSLAB;

  DmaPlan<T, TB, BN, NW> pb;
  dma_plan<T, TB, BN, NW>(pb, B, g.ldb, n0, g.N, 0, tid);
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  const unsigned ring0 = lds0 + (unsigned)PANEL;
#pragma unroll
  for (int s = 0; s < STAGES; ++s)
    if (s < nk) dma_issue<T, TB, BN, NW>(pb, ring0 + s * GB::BYTES, g.ldb);

  EpiPre<T, TC, BM, BN, 256> pre;
  epi_prefetch<T, TC, BM, BN, 256>(pre, g, C, m0, n0, 0, tid);
  ZPre<BM * (BN / 8) / 256> zp;
  zp.valid = false;

  // ---- LayerNorm backward prologue: the row arithmetic of ln_bwd_s_kernel ---------------------------------------------
  {
    const bool publish = tn == 0;
    float gmv[3][4], ag[3][4], ab[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      load4(ln.gamma + c * 256 + lane * 4, gmv[c]);
#pragma unroll
      for (int e = 0; e < 4; ++e) { ag[c][e] = 0.f; ab[c][e] = 0.f; }
    }
    for (int r = wave; r < BM; r += 4) {
      const int row = m0 + r, rowc = min(row, g.M - 1);
      const bool live = row < g.M;
      const float mean = ln.stats[2 * rowc], rstd = ln.stats[2 * rowc + 1];
      float xh[3][4], gy[3][4];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int col = c * 256 + lane * 4;
        float xv[4], dv[4];
        load4(ln.x + (long)rowc * H + col, xv);
        load4(ln.dy + (long)rowc * H + col, dv);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xh[c][e] = (xv[e] - mean) * rstd;
          gy[c][e] = dv[e] * gmv[c][e];
          s1 += gy[c][e];
          s2 += gy[c][e] * xh[c][e];
          if (live) { ag[c][e] += dv[e] * xh[c][e]; ab[c][e] += dv[e]; }
        }
      }
      s1 = wave_sum(s1) * (1.0f / H);
      s2 = wave_sum(s2) * (1.0f / H);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int col = c * 256 + lane * 4;
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rstd * (gy[c][e] - s1 - xh[c][e] * s2);
        if (ln.add != nullptr) {
          float av[4];
          load4(ln.add + (long)rowc * H + col, av);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] += av[e];
        }
        if (publish && live && ln.dx != nullptr) store4(ln.dx + (long)row * H + col, o);
        if (ln.drop.p > 0.f) {   // the operand copy is the gradient of a dropped dense output: d(dense) = dx * mask / (1-p)
          float dm[4];
          drop_mult_run<4>(ln.drop.seed, (uint32_t)rowc * H + col, ln.drop.p, ln.drop.inv_keep, dm);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] *= dm[e];
        }
        if (publish && live && ln.dxt != nullptr) store4(ln.dxt + (long)row * H + col, o);
        const int slab = col >> 6, ch = (col & 63) >> 3, half = (col >> 2) & 1;
        uint2 w;
        w.x = pack_bf16(live ? o[0] : 0.f, live ? o[1] : 0.f);
        w.y = pack_bf16(live ? o[2] : 0.f, live ? o[3] : 0.f);
        *reinterpret_cast<uint2*>(panel + slab * GA::BYTES + r * 128 + ((ch ^ (r & 7)) << 4) + half * 8) = w;
      }
    }
    if (publish && ln.part != nullptr) {      // this row block's slab of the two-stage dgamma / dbeta reduction
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          red[(wave * 2 + 0) * H + c * 256 + lane * 4 + e] = ag[c][e];
          red[(wave * 2 + 1) * H + c * 256 + lane * 4 + e] = ab[c][e];
        }
      __syncthreads();
      float* slab = ln.part + (long)tm * 2 * H;
      for (int col = tid; col < 2 * H; col += 256) {
        const int which = col / H, cc = col % H;
        slab[col] = red[(0 * 2 + which) * H + cc] + red[(1 * 2 + which) * H + cc] + red[(2 * 2 + which) * H + cc] + red[(3 * 2 + which) * H + cc];
      }
    }
  }

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  Frag<T> fa0[MT], fb0[NT], fa1[MT], fb1[NT];
  wait_slabs<PER_SLAB, STAGES - 1>(min(STAGES - 1, nk - 1));
  __syncthreads();
  load_frags<T, false, TB, BM, BN, NW>(fa0, fb0, panel, ring, 0, wr, wc, lane);
#define ETP_MMA_SET(FA, FB)                                            \
  _Pragma("unroll") for (int a = 0; a < MT; ++a)                       \
      _Pragma("unroll") for (int b = 0; b < NT; ++b) mma_step(acc[a][b], FA[a], FB[b]);
  int t = 0;
  for (; t + 1 < nk; ++t) {
    const char* sa = panel + t * GA::BYTES;
    const char* sb = ring + (t % STAGES) * GB::BYTES;
    load_frags<T, false, TB, BM, BN, NW>(fa1, fb1, sa, sb, 1, wr, wc, lane);
    ETP_MMA_SET(fa0, fb0)
    wait_slabs<PER_SLAB, STAGES - 2>(min(STAGES - 2, nk - 2 - t));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + STAGES < nk) dma_issue<T, TB, BN, NW>(pb, ring0 + (t % STAGES) * GB::BYTES, g.ldb);
    const char* na = panel + (t + 1) * GA::BYTES;
    const char* nb = ring + ((t + 1) % STAGES) * GB::BYTES;
    load_frags<T, false, TB, BM, BN, NW>(fa0, fb0, na, nb, 0, wr, wc, lane);
    ETP_MMA_SET(fa1, fb1)
  }
  {
    const char* sa = panel + t * GA::BYTES;
    const char* sb = ring + (t % STAGES) * GB::BYTES;
    load_frags<T, false, TB, BM, BN, NW>(fa1, fb1, sa, sb, 1, wr, wc, lane);
    ETP_MMA_SET(fa0, fb0)
    ETP_MMA_SET(fa1, fb1)
  }
#undef ETP_MMA_SET
  wait_vmcnt<0>();
  __syncthreads();
  gemm_epilogue<T, TC, BM, BN, 256>(acc, smem, g, C, m0, n0, 0, tid, pre, zp);
}

template <typename TC, int BM>
static int launch_ln_bwd_gemm(const GemmArgs& g, const LnBwdPro& ln, hipStream_t st) {
  constexpr int BN = 64, STAGES = 3;
  constexpr int smem_loop = 12 * BM * 128 + STAGES * BN * 128 + 4 * 2 * 768 * 4, smem_c = BM * (BN + 4) * 4;
  constexpr int smem = smem_loop > smem_c ? smem_loop : smem_c;
  void (*kern)(const GemmArgs, const LnBwdPro) = ln_bwd_gemm_kernel<TC, BM, BN, STAGES>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return 1;
  const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), smem, st, g, ln);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

template <typename TC, bool TB, int BM>
static int launch_ln_gemm(const GemmArgs& g, const LnPro& ln, hipStream_t st) {
  constexpr int BN = 64, STAGES = 3;
  constexpr int smem_loop = 12 * BM * 128 + STAGES * BN * 128, smem_c = BM * (BN + 4) * 4;
  constexpr int smem = smem_loop > smem_c ? smem_loop : smem_c;
  void (*kern)(const GemmArgs, const LnPro) = ln_gemm_kernel<TC, TB, BM, BN, STAGES>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return 1;
  const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), smem, st, g, ln);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace etp

// C = act( LN(S) . W^T + bias )  with W stored [N][768] (tb = 0, forward) ; c_f32: output type; act: ETP_ACT_* ; Z: saved
// pre-activation (ETP_ACT_GELU) or null.  bm = 32 or 64.  y / yt / stats may be null.
extern "C" int r05_ln_gemm(const float* S, long lds_, const void* W, long ldw, void* C, long ldc, int c_f32, int M, int N,
                           const float* bias, const float* gamma, const float* beta, float eps, float* y, void* yt, float* stats,
                           int act, void* Z, long ldz, int bm, void* stream) {
  using namespace etp;
  if (M <= 0 || N <= 0 || N % 8 || ldc % 8 || lds_ % 4 || ldw % 8) return 10;
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = S; g.lda = lds_; g.B = W; g.ldb = ldw; g.C = C; g.ldc = ldc;
  g.M = M; g.N = N; g.K = 768; g.nb_inner = 1; g.ksplit = 1; g.alpha = 1.f;
  g.bias = bias; g.act = act; g.Z = Z; g.ldz = ldz;
  g.vec_epilogue = 1; g.xcd_map = 1;
  LnPro ln{gamma, beta, eps, y, reinterpret_cast<bf16_t*>(yt), stats};
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (bm == 32) return c_f32 ? launch_ln_gemm<float, false, 32>(g, ln, st) : launch_ln_gemm<bf16_t, false, 32>(g, ln, st);
  if (bm == 64) return c_f32 ? launch_ln_gemm<float, false, 64>(g, ln, st) : launch_ln_gemm<bf16_t, false, 64>(g, ln, st);
  return 11;
}

// dX = LayerNormBackward(dY, S; stats, gamma)(+ add) . W    with W stored [768][N] (the dgrad's NN storage).  C bf16 or fp32;
// R (optional, dtype of C, leading dimension ldr) is the epilogue's residual operand.  part: [ceil(M / bm)][2][768] fp32 slabs.
extern "C" int r05_ln_bwd_gemm(const float* dy, const float* x, const float* stats, const float* gamma, const float* add, const void* W,
                               long ldw, void* C, long ldc, int c_f32, const void* R, long ldr, int M, int N, float* dx, void* dxt,
                               float* part, int act, void* Z, long ldz, int bm, void* stream) {
  using namespace etp;
  if (M <= 0 || N <= 0 || N % 8 || ldc % 8 || ldw % 8) return 10;
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = x; g.lda = 768; g.B = W; g.ldb = ldw; g.C = C; g.ldc = ldc;
  g.M = M; g.N = N; g.K = 768; g.nb_inner = 1; g.ksplit = 1; g.alpha = 1.f;
  g.R = R; g.ldr = ldr; g.act = act; g.Z = Z; g.ldz = ldz;       // act = ETP_ACT_GELU_BWD with Z = the saved pre-activation (FFN dgrad)
  g.vec_epilogue = 1; g.xcd_map = 1;
  LnBwdPro ln{dy, x, stats, gamma, add, dx, reinterpret_cast<bf16_t*>(dxt), part, drop_none()};
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (bm == 32) return c_f32 ? launch_ln_bwd_gemm<float, 32>(g, ln, st) : launch_ln_bwd_gemm<bf16_t, 32>(g, ln, st);
  if (bm == 64) return c_f32 ? launch_ln_bwd_gemm<float, 64>(g, ln, st) : launch_ln_bwd_gemm<bf16_t, 64>(g, ln, st);
  return 11;
}
