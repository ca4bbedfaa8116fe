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
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "gemm_shared.h"
#include "gemm_tiles.h"

namespace etp {


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

  // tile / batch / split-K coordinates (XCD-aware workgroup -> tile map: tile_of_block)
  const int tiles_n = (g.N + BN - 1) / BN;
  int tm, tn;
  tile_of_block(blockIdx.x, gridDim.x, (g.M + BM - 1) / BM, tiles_n, g.xcd_map, tm, tn);
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

  __syncthreads();   // every wave is done reading the operand tiles before the C tile overwrites them
  EpiPre<T, TC, BM, BN> pre;
  pre.valid = false;
  pre.bias_valid = false;
  ZPre<BM * (BN / 8) / 256> zp;
  zp.valid = false;
  gemm_epilogue<T, TC, BM, BN>(acc, smem, g, C, m0, n0, ks, tid, pre, zp);
}


// One BM x BN output tile of problem `g` through the LDS-DMA main loop (shared by the single-problem and the grouped kernel).
//
// Software-pipelined reduction (round 3).  The round-2 loop was  wait -> barrier -> issue DMA -> ds_read -> MFMA  per
// 128-byte slab: every wavefront exposed the LDS read latency twice per slab with nothing to cover it (one or two
// wavefronts per SIMD), 440 cycles per workgroup-slab for 136 cycles of MFMA (K sweep, profiles/r02_gemm_sweep.json).
// Now the fragments are double-buffered in registers and the ONE barrier of a slab sits between its sub-steps:
//
//     frags(t, s+1) <- LDS   ||  MFMA(t, s)                      (sub-steps before the last one)
//     [slab t+1 landed: counted vmcnt] [lgkmcnt(0)] s_barrier      -> every wavefront holds all of slab t in registers,
//     issue DMA of slab t+STAGES into slab t's buffer                 so that buffer is free: the ring keeps STAGES-1
//     frags(t+1, 0) <- LDS   ||  MFMA(t, last)                        slabs in flight and one landed
//
// so each ds_read batch is covered by the previous sub-step's MFMAs and a DMA piece has STAGES-1 slab times to land.
// NW wavefronts (4 or 8) form a (NW / 2) x 2 grid over the tile: 4 for the 64- / 128-row tiles, 8 for the 256 x 128 tile (each
// wavefront keeps a 64 x 64 sub-tile; twice the FLOPs of 128 x 128 per byte that crosses the CU's LDS-DMA path).
template <typename T, typename TC, bool TA, bool TB, int BM, int BN, int STAGES, int NW>
__device__ __forceinline__ void dma_tile(const GemmArgs& g, const T* A, const T* B, TC* C, int tm, int tn, int ks, char* smem,
                                         int rec) {
  using GA = TileGeom<T, TA, BM, 0>;
  using GB = TileGeom<T, TB, BN, 0>;
  constexpr int BK = MmaTraits<T>::BK;
  constexpr int KS = BK / 32;
  constexpr int NTH = 64 * NW;
  constexpr int MT = BM / (NW / 2) / 16, NT = BN / 32;
  constexpr int STAGE = GA::BYTES + GB::BYTES;
  constexpr int PER_SLAB = DmaPlan<T, TA, BM, NW>::PER_WAVE + DmaPlan<T, TB, BN, NW>::PER_WAVE;   // DMA instrs per wave per slab

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = tm * BM, n0 = tn * BN;
  PhaseProbe probe;
  probe_begin(probe, g);

  int kbeg = 0, kend = g.K;
  if (g.ksplit > 1) {
    const int per = ((g.K + g.ksplit - 1) / g.ksplit + BK - 1) / BK * BK;
    kbeg = ks * per;
    kend = min(g.K, kbeg + per);
  }
  const int nk = (kend > kbeg) ? (kend - kbeg) / BK : 0;     // host guarantees BK | (kend-kbeg)

  DmaPlan<T, TA, BM, NW> pa;
  DmaPlan<T, TB, BN, NW> pb;
  dma_plan<T, TA, BM, NW>(pa, A, g.lda, m0, g.M, kbeg, tid);
  dma_plan<T, TB, BN, NW>(pb, B, g.ldb, n0, g.N, kbeg, tid);

  // the whole ring goes in flight before anything else
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);   // LDS byte address of the ring
#pragma unroll
  for (int s = 0; s < STAGES; ++s)
    if (s < nk) {
      dma_issue<T, TA, BM, NW>(pa, lds0 + s * STAGE, g.lda);
      dma_issue<T, TB, BN, NW>(pb, lds0 + s * STAGE + GA::BYTES, g.ldb);
    }

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  EpiPre<T, TC, BM, BN, NTH> pre;
  epi_prefetch<T, TC, BM, BN, NTH>(pre, g, C, m0, n0, ks, tid);     // epilogue operands travel while the reduction runs
  ZPre<BM * (BN / 8) / NTH> zp;
  z_prefetch<T, TC, BM, BN, NTH>(zp, g, m0, n0, tid);

  const bool do_colsum = TA && g.a_colsum != nullptr && tn == 0;
  float colsum_acc = 0.f;

  Frag<T> fa0[MT], fb0[NT], fa1[MT], fb1[NT];               // the two fragment sets (indexed statically below)
  if (nk > 0) {
    wait_slabs<PER_SLAB, STAGES - 1>(min(STAGES - 1, nk - 1));   // slab 0 landed, the rest of the ring stays in flight
    __builtin_amdgcn_s_barrier();
    load_frags<T, TA, TB, BM, BN, NW>(fa0, fb0, smem, smem + GA::BYTES, 0, wr, wc, lane);
  }
  if (probe.on) probe.mt1 = __builtin_amdgcn_s_memtime();

  auto colsum_slab = [&](int t) {
    if constexpr (TA) {
      // fused bias gradient: blocks of the first tile column also sum the A tile ([k][rows]) over k
      if (do_colsum) {
        const char* sa = smem + (t % STAGES) * STAGE;
        constexpr int KQ = NTH / BM, RPT = BK / KQ;
        const int col = tid % BM, kq = tid / BM;
        const int chunk = col / GA::EPC, within = (col % GA::EPC) * (int)sizeof(T);
#pragma unroll 4
        for (int e = 0; e < RPT; ++e) {
          const int kr = kq * RPT + e;
          colsum_acc += Elem<T>::ld(reinterpret_cast<const T*>(sa + kr * GA::PITCH + ((chunk ^ tr_swz<GA::CPR>(kr)) << 4) + within));
        }
      }
    }
  };
// ETP_GEMM_EXPT (measurement builds only, tools/r03_call2.sh): 1 = the loop issues and waits for the DMA but skips the
// fragment reads and MFMAs, 2 = fragment reads + MFMAs + barrier but no DMA inside the loop.  If t(1) + t(2) ~ t(full) the
// two halves serialise inside each wavefront; if max(t(1), t(2)) ~ t(full) they already overlap.
#ifndef ETP_GEMM_EXPT
#define ETP_GEMM_EXPT 0
#endif
#if ETP_GEMM_EXPT == 1
#define ETP_MMA_SET(FA, FB)
#define ETP_LOAD_FRAGS(...)
#else
#define ETP_MMA_SET(FA, FB)                                            \
  _Pragma("unroll") for (int a = 0; a < MT; ++a)                       \
      _Pragma("unroll") for (int b = 0; b < NT; ++b) mma_step(acc[a][b], FA[a], FB[b]);
#define ETP_LOAD_FRAGS(...) load_frags<T, TA, TB, BM, BN, NW>(__VA_ARGS__)
#endif
  // hand-over between slabs: slab t+1 landed, every wave's reads of slab t retired, next DMA into the freed buffer
#define ETP_SLAB_HANDOVER(t)                                                                             \
  if (ETP_GEMM_EXPT != 2) wait_slabs<PER_SLAB, STAGES - 2>(min(STAGES - 2, nk - 2 - (t)));               \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
  __builtin_amdgcn_s_barrier();                                                                          \
  if (ETP_GEMM_EXPT != 2 && (t) + STAGES < nk) {                                                         \
    const unsigned dst = lds0 + ((t) % STAGES) * STAGE;                                                  \
    dma_issue<T, TA, BM, NW>(pa, dst, g.lda);                                                                \
    dma_issue<T, TB, BN, NW>(pb, dst + GA::BYTES, g.ldb);                                                    \
  }

  if constexpr (KS == 2) {                // bf16: two sub-steps per slab, set 0 then set 1
    int t = 0;
    for (; t + 1 < nk; ++t) {             // steady state: slab t+1 exists
      const char* sa = smem + (t % STAGES) * STAGE;
      colsum_slab(t);
      ETP_LOAD_FRAGS(fa1, fb1, sa, sa + GA::BYTES, 1, wr, wc, lane);
      ETP_MMA_SET(fa0, fb0)
      ETP_SLAB_HANDOVER(t)
      const char* sn = smem + ((t + 1) % STAGES) * STAGE;
      ETP_LOAD_FRAGS(fa0, fb0, sn, sn + GA::BYTES, 0, wr, wc, lane);
      ETP_MMA_SET(fa1, fb1)
    }
    if (t < nk) {                         // last slab
      const char* sa = smem + (t % STAGES) * STAGE;
      colsum_slab(t);
      ETP_LOAD_FRAGS(fa1, fb1, sa, sa + GA::BYTES, 1, wr, wc, lane);
      ETP_MMA_SET(fa0, fb0)
      ETP_MMA_SET(fa1, fb1)
    }
  } else {                                // fp32: one sub-step per slab, the sets alternate from slab to slab
    static_assert(KS == 1, "slab = one or two MFMA k-steps");
    int t = 0;
    for (; t + 2 < nk; t += 2) {          // slabs t, t+1 with t+2 existing
      colsum_slab(t);
      ETP_SLAB_HANDOVER(t)
      const char* s1 = smem + ((t + 1) % STAGES) * STAGE;
      ETP_LOAD_FRAGS(fa1, fb1, s1, s1 + GA::BYTES, 0, wr, wc, lane);
      ETP_MMA_SET(fa0, fb0)
      colsum_slab(t + 1);
      ETP_SLAB_HANDOVER(t + 1)
      const char* s2 = smem + ((t + 2) % STAGES) * STAGE;
      ETP_LOAD_FRAGS(fa0, fb0, s2, s2 + GA::BYTES, 0, wr, wc, lane);
      ETP_MMA_SET(fa1, fb1)
    }
    if (t + 1 < nk) {                     // two slabs left: t (set 0), t+1 (set 1)
      colsum_slab(t);
      ETP_SLAB_HANDOVER(t)
      const char* s1 = smem + ((t + 1) % STAGES) * STAGE;
      ETP_LOAD_FRAGS(fa1, fb1, s1, s1 + GA::BYTES, 0, wr, wc, lane);
      ETP_MMA_SET(fa0, fb0)
      colsum_slab(t + 1);
      ETP_MMA_SET(fa1, fb1)
    } else if (t < nk) {                  // one slab left (set 0)
      colsum_slab(t);
      ETP_MMA_SET(fa0, fb0)
    }
  }
#undef ETP_MMA_SET
#undef ETP_LOAD_FRAGS
#undef ETP_SLAB_HANDOVER
  wait_vmcnt<0>();
  if (probe.on) probe.mt2 = __builtin_amdgcn_s_memtime();
  if constexpr (TA) {
    if (do_colsum && m0 + (tid % BM) < g.M) atomicAdd(g.a_colsum + m0 + (tid % BM), colsum_acc);
  }
  __syncthreads();
  gemm_epilogue<T, TC, BM, BN, NTH>(acc, smem, g, C, m0, n0, ks, tid, pre, zp);
  probe_end(probe, g, rec, nk);
}

// __launch_bounds__(256, w): w = wavefronts per SIMD the grid needs (2 for the 128-row tiles, 3 for 64x64).  Without it
// hipcc (ROCm 7.2) assumes a 512-register budget, splits it into VGPRs + AGPRs and then shuttles accumulators and
// fragments between the two files (~290 v_accvgpr_* moves per kernel, ~150 inside the reduction loop); with the bound
// every MFMA takes the VGPR form and accumulates in place.
template <int BM, int BN> struct TileWaves {
  static constexpr int MIN = (BM * BN <= 64 * 64) ? 3 : 2;
  static constexpr int NW = BM >= 256 ? 8 : 4;       // wavefronts per workgroup
};
template <typename T, typename TC, bool TA, bool TB, int BM, int BN, int STAGES>
__global__ __launch_bounds__(64 * (TileWaves<BM, BN>::NW), (TileWaves<BM, BN>::MIN)) void gemm_dma_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_n = (g.N + BN - 1) / BN;
  int tm, tn;
  tile_of_block(blockIdx.x, gridDim.x, (g.M + BM - 1) / BM, tiles_n, g.xcd_map, tm, tn);
  const int z = blockIdx.y / g.ksplit, ks = blockIdx.y % g.ksplit;
  const int zo = z / g.nb_inner, zi = z % g.nb_inner;
  const T* A = reinterpret_cast<const T*>(g.A) + zo * g.sAo + zi * g.sAi;
  const T* B = reinterpret_cast<const T*>(g.B) + zo * g.sBo + zi * g.sBi;
  TC* C = reinterpret_cast<TC*>(g.C) + zo * g.sCo + zi * g.sCi;
  dma_tile<T, TC, TA, TB, BM, BN, STAGES, TileWaves<BM, BN>::NW>(g, A, B, C, tm, tn, ks, smem, blockIdx.y * gridDim.x + blockIdx.x);
}

// Grouped launch: up to ETP_GEMM_GROUP_MAX independent products of one storage/dtype/tile class in ONE grid (the four
// weight gradients of a transformer layer, the text K/V projections of all x-layers).  The concatenated tile list is cut
// into 8 contiguous chunks, one per XCD (workgroup i runs on XCD i % 8), so every private L2 sees a compact slab of one or
// two problems; inside a problem tiles run along the longer tile axis first (same order as tile_of_block).
template <typename T, typename TC, bool TA, bool TB, int BM, int BN, int STAGES>
__global__ __launch_bounds__(64 * (TileWaves<BM, BN>::NW), (TileWaves<BM, BN>::MIN)) void gemm_group_kernel(const GemmGroup grp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // (A persistent form -- grid capped at one workgroup per CU, every workgroup walking several tiles, so that the dependent
  // chain always keeps half of every CU -- was measured SLOWER: 4.53 ms per step with 256 workgroups, 4.90 ms with 192 or 128,
  // against 4.38 ms; profiles/r03_ab_runs.json group c7.  The step is bound by the throughput of chain + leaf work together.)
  const int bid = blockIdx.x, nwg = gridDim.x;
  int id = bid;
  if (grp.xcd_chunks) {      // uniform reduction lengths: contiguous chunk of the tile list per XCD (L2 locality)
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }                          // mixed lengths: dispatch order = list order (longest reductions first, spread over all XCDs)
  int p = 0;
#pragma unroll
  for (int i = 1; i < ETP_GEMM_GROUP_MAX; ++i)
    if (i < grp.n && id >= grp.tile_start[i]) p = i;
  const GemmArgs& g = grp.g[p];
  const int local = id - grp.tile_start[p];
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
  int tm, tn;
  if (tiles_m >= tiles_n) { tm = local / tiles_n; tn = local % tiles_n; }
  else { tn = local / tiles_m; tm = local % tiles_m; }
  dma_tile<T, TC, TA, TB, BM, BN, STAGES, TileWaves<BM, BN>::NW>(g, reinterpret_cast<const T*>(g.A), reinterpret_cast<const T*>(g.B),
                                                               reinterpret_cast<TC*>(g.C), tm, tn, 0, smem, bid);
}

// ---- optional per-launch HIP-event timing (bench.py roofline leg) ---------------------------------------
static bool g_prof_on = false;
static std::string g_prof_only;          // when set, only launches whose name contains it are bracketed (fewer event packets)
static std::vector<ProfRec> g_prof_recs;
static std::vector<std::string> g_prof_names;
static std::mutex g_prof_mu;

void prof_enable(bool on) { g_prof_on = on; }
void prof_filter(const char* name_part) { g_prof_only = name_part ? name_part : ""; }
static bool prof_wanted(const char* nm) { return g_prof_only.empty() || strstr(nm, g_prof_only.c_str()) != nullptr; }
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
bool prof_begin(const char* nm, double flops, double bytes, hipStream_t st, ProfRec& rec) {
  if (!(g_prof_on && !rec_active() && prof_wanted(nm))) return false;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  rec.id = prof_id(nm);
  rec.flops = flops; rec.bytes = bytes;
  if (hipEventCreate(&rec.a) != hipSuccess || hipEventCreate(&rec.b) != hipSuccess) return false;
  return hipEventRecord(rec.a, st) == hipSuccess;
}
void prof_end(const ProfRec& rec, hipStream_t st) {
  (void)hipEventRecord(rec.b, st);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_recs.push_back(rec);
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

// ---- phase probe (host side): a caller-owned device buffer of launches x PROBE_WG_MAX x 8 u64 ------------------------
constexpr int PROBE_WG_MAX = 4096;
struct ProbeMeta { char name[96]; int dims[4]; };     // grid, M, N, K
static unsigned long long* g_probe_buf = nullptr;
static long g_probe_cap = 0;
static std::vector<ProbeMeta> g_probe_meta;
void gemm_probe_set(unsigned long long* buf, long launches) {
  g_probe_buf = buf; g_probe_cap = buf ? launches : 0;
  g_probe_meta.clear();
}
long gemm_probe_count() { return (long)g_probe_meta.size(); }
int gemm_probe_meta(long i, char* name, int cap, int* dims) {
  if (i < 0 || i >= (long)g_probe_meta.size() || !name || cap <= 0 || !dims) return ETP_ERR_INVALID;
  strncpy(name, g_probe_meta[i].name, cap - 1); name[cap - 1] = 0;
  memcpy(dims, g_probe_meta[i].dims, sizeof(int) * 4);
  return ETP_OK;
}
unsigned long long* probe_slot(const char* name, long wgs, int M, int N, int K) {
  if (!g_probe_buf || rec_active() || wgs > PROBE_WG_MAX || (long)g_probe_meta.size() >= g_probe_cap) return nullptr;
  ProbeMeta m;
  memset(&m, 0, sizeof(m));
  strncpy(m.name, name, sizeof(m.name) - 1);
  m.dims[0] = (int)wgs; m.dims[1] = M; m.dims[2] = N; m.dims[3] = K;
  unsigned long long* p = g_probe_buf + g_probe_meta.size() * (size_t)PROBE_WG_MAX * 8;
  g_probe_meta.push_back(m);
  return p;
}

template <typename T, typename TC, bool TA, bool TB, int BM, int BN, int STAGES /*0 = register-staged kernel*/>
static int launch_one(const GemmArgs& g_in, int nbatch, hipStream_t st) {
  constexpr int PAD = STAGES == 0 ? 32 : 0;
  using GA = TileGeom<T, TA, BM, PAD>;
  using GB = TileGeom<T, TB, BN, PAD>;
  constexpr int smem_loop = (STAGES == 0 ? 2 : STAGES) * (GA::BYTES + GB::BYTES), smem_c = BM * (BN + 4) * 4;
  constexpr int smem = smem_loop > smem_c ? smem_loop : smem_c;
  void (*kern)(const GemmArgs);
  if constexpr (STAGES == 0) kern = gemm_kernel<T, TC, TA, TB, BM, BN>;
  else kern = gemm_dma_kernel<T, TC, TA, TB, BM, BN, STAGES>;
  ETP_CHECK_HIP(ensure_dyn_lds(reinterpret_cast<const void*>(kern), smem));
  const int tiles = ((g_in.M + BM - 1) / BM) * ((g_in.N + BN - 1) / BN);
  dim3 grid(tiles, nbatch * g_in.ksplit, 1);
  GemmArgs g = g_in;
  char nm[96];
  snprintf(nm, sizeof(nm), "gemm%s<%s,%s,%s%s,%dx%d,s%d>", STAGES ? "_dma" : "", sizeof(T) == 2 ? "bf16" : "f32",
           sizeof(TC) == 2 ? "bf16" : "f32", TA ? "T" : "N", TB ? "N" : "T", BM, BN, STAGES);   // BLAS-style opA,opB
  g.dbg = (STAGES > 0 && g_probe_buf) ? probe_slot(nm, (long)tiles * nbatch * g_in.ksplit, g.M, g.N, g.K) : nullptr;
  ProfRec rec;
  const bool prof = g_prof_on && !rec_active() && prof_wanted(nm);
  if (prof) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    rec.id = prof_id(nm);
    rec.flops = 2.0 * g.M * g.N * g.K * nbatch;
    rec.bytes = ((double)g.M * g.K + (double)g.N * g.K) * nbatch * sizeof(T) + (double)g.M * g.N * nbatch * sizeof(TC);
    ETP_CHECK_HIP(hipEventCreate(&rec.a));
    ETP_CHECK_HIP(hipEventCreate(&rec.b));
    ETP_CHECK_HIP(hipEventRecord(rec.a, st));
  }
  ETP_LAUNCH(kern, grid, dim3(STAGES == 0 ? 256 : 64 * TileWaves<BM, BN>::NW), smem, st, g);
  ETP_CHECK_LAUNCH("gemm");
  if (prof) {
    ETP_CHECK_HIP(hipEventRecord(rec.b, st));
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_recs.push_back(rec);
  }
  return ETP_OK;
}

static bool dma_ok(int bk, int K, int ksplit) {
  // LDS-DMA main loop needs whole 128-byte slabs in every split of the reduction
  bool dma = (K % bk == 0) && (K >= 2 * bk);
  if (ksplit > 1) dma = dma && (K % ksplit == 0) && ((K / ksplit) % bk == 0);
  const char* force = opt_str(OPT_GEMM_TILE);
  if (force && strchr(force, 'r')) dma = false;
  return dma;
}
bool gemm_uses_dma(int dtype, int K, int ksplit) { return dma_ok(dtype == ETP_BF16 ? 64 : 32, K, ksplit); }

template <typename T, typename TC, bool TA, bool TB>
static int launch_tiles(const GemmArgs& g, int nbatch, hipStream_t st) {
  // Tile choice (round 3, after the pipelined main loop; sweep: tools/gemm_sweep.py -> profiles/r03_gemm_sweep.json).
  // What bounds a tile class on this chip is the L2 -> LDS feed (~64 B/clk/CU): per 128-byte slab a 64x64 tile moves
  // 16 KiB for 8 MFMAs per wave, 128x64 24 KiB for 16, 128x128 32 KiB for 32.  So: the largest tile that still gives
  // (nearly) every CU a workgroup.
  //   128x128 (ring 2, two workgroups per CU)   when it yields >= ~1.4 workgroups per CU,
  //   128x64  (ring 4, one workgroup per CU)    when it yields >= ~0.8 per CU (the M = 2560, N = 768 products: 240),
  //   64x64   (ring 3; ring 4 below one workgroup per CU) otherwise.
  constexpr int BK = MmaTraits<T>::BK;
  const long t128 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128) * nbatch * g.ksplit;
  const long tw = (long)((g.M + 127) / 128) * ((g.N + 63) / 64) * nbatch * g.ksplit;
  const long t64 = (long)((g.M + 63) / 64) * ((g.N + 63) / 64) * nbatch * g.ksplit;
  // 256x128 (eight wavefronts, one workgroup per CU) exists as a FORCED tile class only ("256s2" / "256s3"): measured on
  // MI355X (profiles/r03d_*, r03_ab_runs.json group c5) its loop runs at the same rate per CU as two co-resident 128x128
  // workgroups (1.0 us per 256x128x64 slab of work either way, ~50 % MFMA issue rate at two wavefronts per SIMD) while its
  // 180 / 216 / 240 tiles leave 6-30 % of the CUs idle: step 4.51 ms with it against 4.37 ms; no shape up to 8192 rows wins.
  bool huge = false;
  bool big = !huge && (g.M >= 128 && g.N >= 128 && t128 >= 360);
  bool wide = !huge && !big && g.M >= 128 && g.N >= 64 && tw >= 200 && tw <= 520 && nbatch == 1;
  // The 128x64 class is OFF by default: alone it is the faster kernel for the M = 2560, N = 768 products (25.1 vs 28.4 us at
  // K = 3072 in a single-stream step), but in the real three-stream step the 64x64 class wins, 4.30 vs 4.35 ms per step in
  // three same-box A/B pairs (profiles/r03_ab_runs.json groups c1, c8): its 48-KiB, ~130-register workgroups pack beside the
  // leaf kernels' workgroups where one 96-KiB 128x64 workgroup per CU does not.  ETP_GEMM_WIDE=1 enables it.
  const bool wide_on = opt_int(OPT_GEMM_WIDE, 0) == 1;
  if (!wide_on) wide = false;
  const bool dma = dma_ok(BK, g.K, g.ksplit);
  int stages = (big || huge) ? 2 : (wide ? 4 : 3);
  if (!huge && !big && !wide && t64 <= 320 && g.K >= 4 * BK) stages = 4;
  const char* force = opt_str(OPT_GEMM_TILE);   // tuning aid (tools/gemm_sweep.py): "128", "64", "w" + optional "s2".."s4", "64r"
  if (force && force[0]) {
    huge = force[0] == '2';                        // "256", "256s3"
    if (force[0] == '1' || force[0] == '6') { big = force[0] == '1'; wide = false; }
    if (force[0] == 'w') { wide = true; big = false; }
    if (huge) { big = false; wide = false; }
    stages = (big || huge) ? 2 : (wide ? 4 : 3);
    if (strstr(force, "s2")) stages = 2;
    if (strstr(force, "s3")) stages = 3;
    if (strstr(force, "s4")) stages = 4;
  }
  if (!dma) {
    if (big || huge) return launch_one<T, TC, TA, TB, 128, 128, 0>(g, nbatch, st);
    return launch_one<T, TC, TA, TB, 64, 64, 0>(g, nbatch, st);
  }
  if constexpr (sizeof(T) == 2 && !TA) {
    // 32x64 tiles (four wavefronts of 16x32, ring 4) for grids that would give fewer than half the CUs a 64x64 workgroup -- the
    // M = 512 node-side products of the x-layers (96 -> 192 workgroups) and everything of config 5.  A lone workgroup's slab
    // time is set by its own wait -> barrier -> read -> MFMA chain (0.24 us per 64x64x64 slab whatever the ring depth or the
    // wavefront count, profiles/r04_gemm_phases.txt), not by bytes: halving the tile rows doubles the workgroups that share
    // the reduction's work at (nearly) the same time per slab.  A row-major only (its 32-row slab is four 1-KiB DMA pieces).
    // ETP_GEMM_SMALL=0 switches the class off (A/B runs), ETP_GEMM_TILE=32 forces it.
    const bool small_on = opt_on(OPT_GEMM_SMALL, true);
    const bool forced = force && force[0];
    const bool take = forced ? force[0] == '3' : (small_on && nbatch == 1 && g.ksplit == 1 && t64 <= 128 && g.M >= 32 && g.K >= 4 * BK);
    if (take) return launch_one<T, TC, TA, TB, 32, 64, 4>(g, nbatch, st);
  }
  if (huge) {
    if constexpr (sizeof(T) == 2) {                  // bf16 only: the fp32 parity mode keeps the four-wavefront tiles
      if (stages == 3) return launch_one<T, TC, TA, TB, 256, 128, 3>(g, nbatch, st);
      return launch_one<T, TC, TA, TB, 256, 128, 2>(g, nbatch, st);
    } else {
      if (stages == 3) return launch_one<T, TC, TA, TB, 128, 128, 3>(g, nbatch, st);
      return launch_one<T, TC, TA, TB, 128, 128, 2>(g, nbatch, st);
    }
  }
  if (big) {
    if (stages == 3) return launch_one<T, TC, TA, TB, 128, 128, 3>(g, nbatch, st);
    return launch_one<T, TC, TA, TB, 128, 128, 2>(g, nbatch, st);
  }
  if (wide) {
    if (stages == 2) return launch_one<T, TC, TA, TB, 128, 64, 2>(g, nbatch, st);
    if (stages == 3) return launch_one<T, TC, TA, TB, 128, 64, 3>(g, nbatch, st);
    return launch_one<T, TC, TA, TB, 128, 64, 4>(g, nbatch, st);
  }
  if (stages == 4) return launch_one<T, TC, TA, TB, 64, 64, 4>(g, nbatch, st);
  // (no two-slab ring for 64x64 tiles: its fp32 TN instantiation spilled to scratch, tools/kernel_resources.py; a forced "64s2" runs s3)
  return launch_one<T, TC, TA, TB, 64, 64, 3>(g, nbatch, st);
}

template <typename T, typename TC>
static int launch_trans(int ta, int tb, const GemmArgs& g, int nbatch, hipStream_t st) {
  if (!ta && !tb) return launch_tiles<T, TC, false, false>(g, nbatch, st);
  if (!ta && tb) return launch_tiles<T, TC, false, true>(g, nbatch, st);
  if (ta && tb) return launch_tiles<T, TC, true, true>(g, nbatch, st);
  return fail(ETP_ERR_INVALID, "gemm: (A trans, B row) storage pairing is not used on this path");
}

// argument checks + derived fields (xcd_map, vec_epilogue) shared by the single and the grouped launcher
static int prepare_args(int dtype, int c_dtype, int ta, int tb, const GemmArgs& g0, int nbatch, GemmArgs& g) {
  ETP_REQUIRE(g0.M > 0 && g0.N > 0 && g0.K >= 0 && nbatch > 0 && g0.ksplit >= 1 && g0.nb_inner >= 1, "bad dims");
  const int epc = dtype == ETP_BF16 ? 8 : 4;
  ETP_REQUIRE(g0.lda % epc == 0 && g0.ldb % epc == 0, "lda/ldb must be multiples of a 16-byte chunk");
  ETP_REQUIRE(((uintptr_t)g0.A % 16 == 0) && ((uintptr_t)g0.B % 16 == 0), "A/B must be 16-byte aligned");
  ETP_REQUIRE((g0.sAo % epc == 0) && (g0.sAi % epc == 0) && (g0.sBo % epc == 0) && (g0.sBi % epc == 0),
              "batch strides must keep 16-byte alignment");
  ETP_REQUIRE(g0.out_mode != 2 || c_dtype == ETP_F32, "atomic accumulation needs an fp32 C");
  ETP_REQUIRE(g0.ksplit == 1 || g0.out_mode == 2, "split-K needs atomic accumulation");
  ETP_REQUIRE(g0.drop.p == 0.f || (g0.ksplit == 1 && nbatch == 1), "epilogue dropout needs an unsplit, unbatched product");
  ETP_REQUIRE(g0.a_colsum == nullptr || (ta && tb && gemm_uses_dma(dtype, g0.K, g0.ksplit)),
              "a_colsum needs the TN LDS-DMA kernel (check gemm_uses_dma first)");
  ETP_REQUIRE(dtype != ETP_F32 || c_dtype == ETP_F32, "fp32 operands need an fp32 C");
  g = g0;
  {
    const bool xcd_on = opt_on(OPT_GEMM_XCD, true);
    g.xcd_map = xcd_on;
  }
  {  // the vectorised epilogue needs 8-column chunks to stay in-bounds and 16-byte aligned
    const size_t cs = dtype_size(c_dtype);
    bool ok = (g.ldc % 8 == 0) && (g.ldc >= round_up(g.N, 8)) && ((uintptr_t)g.C % 16 == 0) && ((g.sCo * cs) % 16 == 0) &&
              ((g.sCi * cs) % 16 == 0);
    if (g.bias) ok = ok && ((uintptr_t)g.bias % 16 == 0) && (g.N % 8 == 0);
    if (g.R) ok = ok && (g.ldr % 8 == 0) && ((uintptr_t)g.R % 16 == 0) && (g.ldr >= round_up(g.N, 8));
    if (g.Z) ok = ok && (g.ldz % 8 == 0) && ((uintptr_t)g.Z % 16 == 0) && (g.ldz >= round_up(g.N, 8));
    if (g.out_mode == 2) ok = false;   // atomics: lane-consecutive fp32 columns (row-major scalar path) coalesce best
    g.vec_epilogue = ok ? 1 : 0;
  }
  return ETP_OK;
}

int launch_gemm(int dtype, int c_dtype, int ta, int tb, const GemmArgs& g_in, int nbatch, hipStream_t st) {
  GemmArgs g;
  ETP_TRY(prepare_args(dtype, c_dtype, ta, tb, g_in, nbatch, g));
  if (dtype == ETP_BF16 && !(ta && !tb)) {           // whole-tile bf16 products: the 32x32x16 family (gemm_mm32.hip)
    const int cls = mm32_class(g, nbatch);
    if (cls) return launch_mm32(c_dtype, ta, tb, g, cls, st);
  }
  if (dtype == ETP_F32) return launch_trans<float, float>(ta, tb, g, nbatch, st);
  if (c_dtype == ETP_F32) return launch_trans<bf16_t, float>(ta, tb, g, nbatch, st);
  return launch_trans<bf16_t, bf16_t>(ta, tb, g, nbatch, st);
}

// ---- grouped launch ---------------------------------------------------------------------------------------------------
template <typename T, typename TC, bool TA, bool TB, int BM, int BN, int STAGES>
static int launch_group_one(GemmGroup& grp, hipStream_t st) {
  using GA = TileGeom<T, TA, BM, 0>;
  using GB = TileGeom<T, TB, BN, 0>;
  constexpr int smem_loop = STAGES * (GA::BYTES + GB::BYTES), smem_c = BM * (BN + 4) * 4;
  constexpr int smem = smem_loop > smem_c ? smem_loop : smem_c;
  void (*kern)(const GemmGroup) = gemm_group_kernel<T, TC, TA, TB, BM, BN, STAGES>;
  ETP_CHECK_HIP(ensure_dyn_lds(reinterpret_cast<const void*>(kern), smem));
  int tiles = 0;
  double flops = 0, bytes = 0;
  for (int i = 0; i < grp.n; ++i) {
    const GemmArgs& g = grp.g[i];
    grp.tile_start[i] = tiles;
    tiles += ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    flops += 2.0 * g.M * g.N * g.K;
    bytes += ((double)g.M * g.K + (double)g.N * g.K) * sizeof(T) + (double)g.M * g.N * sizeof(TC);
  }
  for (int i = grp.n; i <= ETP_GEMM_GROUP_MAX; ++i) grp.tile_start[i] = tiles;
  const int grid = tiles;
  char nm[96];
  snprintf(nm, sizeof(nm), "gemm_group<%s,%s,%s%s,%dx%d,s%d>", sizeof(T) == 2 ? "bf16" : "f32", sizeof(TC) == 2 ? "bf16" : "f32",
           TA ? "T" : "N", TB ? "N" : "T", BM, BN, STAGES);
  {
    unsigned long long* slot = g_probe_buf ? probe_slot(nm, tiles, grp.g[0].M, grp.g[0].N, grp.g[0].K) : nullptr;
    for (int i = 0; i < grp.n; ++i) grp.g[i].dbg = slot;
  }
  ProfRec rec;
  const bool prof = g_prof_on && !rec_active() && prof_wanted(nm);
  if (prof) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    rec.id = prof_id(nm);
    rec.flops = flops; rec.bytes = bytes;
    ETP_CHECK_HIP(hipEventCreate(&rec.a));
    ETP_CHECK_HIP(hipEventCreate(&rec.b));
    ETP_CHECK_HIP(hipEventRecord(rec.a, st));
  }
  ETP_LAUNCH(kern, dim3(grid), dim3(64 * TileWaves<BM, BN>::NW), smem, st, grp);
  ETP_CHECK_LAUNCH("gemm_group");
  if (prof) {
    ETP_CHECK_HIP(hipEventRecord(rec.b, st));
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_recs.push_back(rec);
  }
  return ETP_OK;
}

template <typename T, typename TC, bool TA, bool TB>
static int launch_group_tiles(GemmGroup& grp, hipStream_t st) {
  long t128 = 0;
  bool all_big = true;
  for (int i = 0; i < grp.n; ++i) {
    t128 += (long)((grp.g[i].M + 127) / 128) * ((grp.g[i].N + 127) / 128);
    all_big = all_big && grp.g[i].M >= 128 && grp.g[i].N >= 128;
  }
  // 128x128 tiles halve the L2->LDS bytes per FLOP; they pay once the group still gives most CUs a tile
  bool big = all_big && t128 >= 160;
  // 256x128: forced only (see launch_tiles)
  bool all_huge = sizeof(T) == 2;
  for (int i = 0; i < grp.n; ++i) all_huge = all_huge && grp.g[i].M >= 256 && grp.g[i].N >= 128;
  bool huge = false;
  int stages = (big || huge) ? 2 : 3;
  const char* force = opt_str(OPT_GROUP_TILE);          // tuning aid: "256s2", "256s3", "128s2", "128s3", "64s3", "64s4"
  if (force && force[0]) {
    huge = force[0] == '2' && all_huge;
    big = force[0] == '1' && all_big;
    stages = (big || huge) ? 2 : 3;
    if (strstr(force, "s2")) stages = 2;
    if (strstr(force, "s3")) stages = 3;
    if (strstr(force, "s4")) stages = 4;
  }
  if (huge) {
    if constexpr (sizeof(T) == 2) {
      if (stages == 3) return launch_group_one<T, TC, TA, TB, 256, 128, 3>(grp, st);
      return launch_group_one<T, TC, TA, TB, 256, 128, 2>(grp, st);
    }
  }
  if (big) {
    if (stages == 3) return launch_group_one<T, TC, TA, TB, 128, 128, 3>(grp, st);
    return launch_group_one<T, TC, TA, TB, 128, 128, 2>(grp, st);
  }
  if (stages == 4) return launch_group_one<T, TC, TA, TB, 64, 64, 4>(grp, st);
  return launch_group_one<T, TC, TA, TB, 64, 64, 3>(grp, st);
}

template <typename T, typename TC>
static int launch_group_trans(int ta, int tb, GemmGroup& grp, hipStream_t st) {
  if (!ta && !tb) return launch_group_tiles<T, TC, false, false>(grp, st);
  if (!ta && tb) return launch_group_tiles<T, TC, false, true>(grp, st);
  if (ta && tb) return launch_group_tiles<T, TC, true, true>(grp, st);
  return fail(ETP_ERR_INVALID, "gemm group: (A trans, B row) storage pairing is not used on this path");
}

int launch_gemm_group(int dtype, int c_dtype, int ta, int tb, const GemmArgs* gs, int n, hipStream_t st) {
  ETP_REQUIRE(gs && n >= 1 && n <= ETP_GEMM_GROUP_MAX, "1..ETP_GEMM_GROUP_MAX problems per group");
  if (n == 1) return launch_gemm(dtype, c_dtype, ta, tb, gs[0], 1, st);
  GemmGroup grp;
  memset(&grp, 0, sizeof(grp));
  grp.n = n;
  // longest reduction first: a tile's run time grows with K, and the last-dispatched tiles set the tail of the launch
  int order[ETP_GEMM_GROUP_MAX];
  for (int i = 0; i < n; ++i) order[i] = i;
  std::stable_sort(order, order + n, [&](int a, int b) { return gs[a].K > gs[b].K; });
  bool uniform = true;
  for (int i = 0; i < n; ++i) {
    const GemmArgs& gi = gs[order[i]];
    uniform = uniform && gi.K == gs[order[0]].K;
    ETP_REQUIRE(gi.ksplit == 1 && gemm_uses_dma(dtype, gi.K, 1), "grouped products need unsplit LDS-DMA-able reductions");
    ETP_TRY(prepare_args(dtype, c_dtype, ta, tb, gi, 1, grp.g[i]));
  }
  grp.xcd_chunks = (uniform && grp.g[0].xcd_map) ? 1 : 0;
  if (dtype == ETP_BF16 && c_dtype == ETP_F32 && ta && tb && mm32_group_ok(grp)) return launch_mm32_group(grp, st);
  if (dtype == ETP_F32) return launch_group_trans<float, float>(ta, tb, grp, st);
  if (c_dtype == ETP_F32) return launch_group_trans<bf16_t, float>(ta, tb, grp, st);
  return launch_group_trans<bf16_t, bf16_t>(ta, tb, grp, st);
}

}  // namespace etp
