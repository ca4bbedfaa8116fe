// Fused embedding / head / loss / gather kernels of the ETPNav planner (gfx950).
//
// All of these are HBM-bound row kernels with hidden size H = NCH*256 (768 on this path): one 64-lane
// wavefront owns one row, each lane owns NCH groups of 4 consecutive columns (8/16-byte vector accesses,
// 512 B / 1 KiB contiguous per wave instruction), row statistics are fp32 wave-shuffle reductions.
// Parameter gradients that reduce over rows are accumulated in LDS (ds_add_f32) per block and flushed with
// one global atomic per column per block.
//
// Reference sites are cited at each kernel.
#include <algorithm>

#include <stdlib.h>

#include "kernels.h"

namespace etp {

template <int NCH> struct Row {  // per-lane slice of one H-wide row
  float v[NCH][4];
};

template <int NCH> __device__ __forceinline__ float row_sum(const Row<NCH>& r) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) s += r.v[c][0] + r.v[c][1] + r.v[c][2] + r.v[c][3];
  return wave_sum(s);
}
template <int NCH> __device__ __forceinline__ float row_dot(const Row<NCH>& a, const Row<NCH>& b) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) s += a.v[c][e] * b.v[c][e];
  return wave_sum(s);
}
template <int NCH, typename T> __device__ __forceinline__ void row_load(Row<NCH>& r, const T* p, int lane) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) load4(p + c * 256 + lane * 4, r.v[c]);
}
template <int NCH, typename T> __device__ __forceinline__ void row_store(const Row<NCH>& r, T* p, int lane) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) store4(p + c * 256 + lane * 4, r.v[c]);
}
template <int NCH> __device__ __forceinline__ void row_add(Row<NCH>& a, const Row<NCH>& b) {
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) a.v[c][e] += b.v[c][e];
}
// in place: x -> xhat = (x-mean)*rstd ; returns mean/rstd (two-pass variance, matches torch.nn.LayerNorm)
template <int NCH> __device__ __forceinline__ void row_normalize(Row<NCH>& x, float eps, float& mean, float& rstd) {
  constexpr float invH = 1.0f / (NCH * 256);
  mean = row_sum<NCH>(x) * invH;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) { x.v[c][e] -= mean; q += x.v[c][e] * x.v[c][e]; }
  rstd = rsqrtf(wave_sum(q) * invH + eps);
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) x.v[c][e] *= rstd;
}
template <int NCH> __device__ __forceinline__ void row_affine(Row<NCH>& y, const Row<NCH>& xhat, const float* gamma,
                                                              const float* beta, int lane) {
  Row<NCH> g, b;
  row_load<NCH>(g, gamma, lane);
  row_load<NCH>(b, beta, lane);
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) y.v[c][e] = xhat.v[c][e] * g.v[c][e] + b.v[c][e];
}
// LayerNorm backward for one row: dy (in) -> dx (out, in place); xhat given
template <int NCH> __device__ __forceinline__ void row_ln_bwd(Row<NCH>& d, const Row<NCH>& xhat, const float* gamma,
                                                              float rstd, int lane) {
  constexpr float invH = 1.0f / (NCH * 256);
  Row<NCH> g;
  row_load<NCH>(g, gamma, lane);
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) d.v[c][e] *= g.v[c][e];
  const float c1 = row_sum<NCH>(d) * invH;
  const float c2 = row_dot<NCH>(d, xhat) * invH;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) d.v[c][e] = rstd * (d.v[c][e] - c1 - xhat.v[c][e] * c2);
}
// Per-lane register accumulators for the parameter gradients that reduce over rows (dgamma = sum dy*xhat, ...).
// Each wave accumulates over the rows it owns; block_flush then sums the block's 4 waves through a [4][H] LDS
// scratch and issues ONE global atomic per column per block.  (LDS float atomics were measured 60x slower here.)
template <int NCH> __device__ __forceinline__ void row_zero(Row<NCH>& a) {
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) a.v[c][e] = 0.f;
}
template <int NCH> __device__ __forceinline__ void acc_mul(Row<NCH>& acc, const Row<NCH>& a, const Row<NCH>& b) {
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc.v[c][e] += a.v[c][e] * b.v[c][e];
}
template <int NCH> __device__ __forceinline__ void acc_scaled(Row<NCH>& acc, const Row<NCH>& a, float s) {
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc.v[c][e] += a.v[c][e] * s;
}
// dst[col*stride + off] += sum over the block's waves of acc[col];  scratch: 4*H floats of LDS
template <int NCH> __device__ __forceinline__ void block_flush(float* scratch, const Row<NCH>& acc, float* dst, int stride,
                                                               int off) {
  constexpr int H = NCH * 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) scratch[wave * H + c * 256 + lane * 4 + e] = acc.v[c][e];
  __syncthreads();
  for (int col = threadIdx.x; col < H; col += 256)
    atomicAdd(dst + (long)col * stride + off, scratch[col] + scratch[H + col] + scratch[2 * H + col] + scratch[3 * H + col]);
  __syncthreads();
}
// x[col] *= mask(row*H + col) / (1-p)
template <int NCH> __device__ __forceinline__ void row_dropout(Row<NCH>& x, const Drop& d, int row, int lane) {
  if (d.p > 0.f) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      float dm[4];
      drop_mult_run<4>(d.seed, (uint32_t)row * (NCH * 256) + c * 256 + lane * 4, d.p, d.inv_keep, dm);
#pragma unroll
      for (int e = 0; e < 4; ++e) x.v[c][e] *= dm[e];
    }
  }
}
template <int NCH> __device__ __forceinline__ void global_acc(float* dst, const Row<NCH>& a, int lane) {
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicAdd(dst + c * 256 + lane * 4 + e, a.v[c][e]);
}

// --------------------------------------------------------------------------------------
// Text embedding: y = LN(word[id] + pos[l] + type[0])           BertEmbeddings.forward vilmodel_cmt.py:62-77
// --------------------------------------------------------------------------------------
template <typename T, int NCH>
__global__ __launch_bounds__(256) void text_embed_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ word,
                                                             const float* __restrict__ pos, const float* __restrict__ type0,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ y, T* __restrict__ yt, float* __restrict__ stats,
                                                             int M, int L, float eps, Drop drop) {
  constexpr int H = NCH * 256;
  const int lane = threadIdx.x & 63;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += gridDim.x * 4) {
    const long id = ids[row];
    const int l = row % L;
    Row<NCH> x, t;
    row_load<NCH>(x, word + id * H, lane);
    row_load<NCH>(t, pos + (long)l * H, lane);
    row_add<NCH>(x, t);
    row_load<NCH>(t, type0, lane);
    row_add<NCH>(x, t);
    float mean, rstd;
    row_normalize<NCH>(x, eps, mean, rstd);
    row_affine<NCH>(t, x, gamma, beta, lane);
    row_dropout<NCH>(t, drop, row, lane);             // BertEmbeddings.dropout vilmodel_cmt.py:76
    row_store<NCH>(t, y + (long)row * H, lane);
    if (yt != nullptr) row_store<NCH>(t, yt + (long)row * H, lane);
    if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
  }
}

// backward: scatter-add into word rows (padding_idx 0 gets none: vilmodel_cmt.py:53), pos rows, type row 0.
// One workgroup per POSITION l, its 4 waves split the batch (rows b*L + l): the position-embedding gradient of l is a
// segment sum over the batch held in registers and flushed once per column (B-way same-address atomics before: every sample
// hit the same L rows), the word rows -- distinct ids with few repeats -- stay atomics.
template <typename T, int NCH>
__global__ __launch_bounds__(256) void text_embed_bwd_kernel(const float* __restrict__ dy, const int64_t* __restrict__ ids,
                                                             const float* __restrict__ word, const float* __restrict__ pos,
                                                             const float* __restrict__ type0, const float* __restrict__ gamma,
                                                             const float* __restrict__ stats, float* __restrict__ dword,
                                                             float* __restrict__ dpos, float* __restrict__ dtype0,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta, int M, int L,
                                                             Drop drop) {
  constexpr int H = NCH * 256;
  __shared__ float scratch[4 * H];
  Row<NCH> a_g, a_b, a_t;   // dgamma, dbeta, dtype0
  row_zero<NCH>(a_g); row_zero<NCH>(a_b); row_zero<NCH>(a_t);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, B = M / L;
  for (int l = blockIdx.x; l < L; l += gridDim.x) {
    Row<NCH> a_p, pl, ty;     // d pos[l]; pos[l] and type[0] are the same for every row of this segment
    row_zero<NCH>(a_p);
    row_load<NCH>(pl, pos + (long)l * H, lane);
    row_load<NCH>(ty, type0, lane);
    for (int b = wave; b < B; b += 4) {
      const int row = b * L + l;
      const long id = ids[row];
      Row<NCH> x, d;
      row_load<NCH>(x, word + id * H, lane);
      row_add<NCH>(x, pl);                                // same association as the forward: (word + pos) + type
      row_add<NCH>(x, ty);
      const float mean = stats[2 * row], rstd = stats[2 * row + 1];
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) x.v[c][e] = (x.v[c][e] - mean) * rstd;
      row_load<NCH>(d, dy + (long)row * H, lane);
      row_dropout<NCH>(d, drop, row, lane);
      acc_mul<NCH>(a_g, d, x);
      acc_scaled<NCH>(a_b, d, 1.0f);
      row_ln_bwd<NCH>(d, x, gamma, rstd, lane);
      acc_scaled<NCH>(a_p, d, 1.0f);
      if (id != 0) global_acc<NCH>(dword + id * H, d, lane);
    }
    block_flush<NCH>(scratch, a_p, dpos + (long)l * H, 1, 0);
    acc_scaled<NCH>(a_t, a_p, 1.0f);                      // d type[0] = sum over all rows
  }
  block_flush<NCH>(scratch, a_g, dgamma, 1, 0);
  block_flush<NCH>(scratch, a_b, dbeta, 1, 0);
  block_flush<NCH>(scratch, a_t, dtype0, 1, 0);
}

// --------------------------------------------------------------------------------------
// Panorama view-embedding fuse                       forward_panorama vilmodel_cmt.py:695-711
//   y = LN( LN_i(a) + LN_d(d) + LN_l(loc.Wl^T + bl) + nav_emb[nav] + type_emb[1] )      all eps 1e-12
//   a = rgb.Wi^T + bi and d = dep.Wd^T + bd come from the MFMA GEMM; the K=4 angle projection is done here.
//   stats[row] = {mean,rstd} x {a, d, loc-proj, sum}
// --------------------------------------------------------------------------------------
// CHUNKWISE: a scheduling fence behind every 256-column chunk keeps only that chunk's 4 x float4 of w_loc in flight (the backward
// launches carry up to six accumulator rows: with all twelve float4 hoisted to the front the PART 2 launch spilled two registers once
// the packed-fp32 forms were gone, round 6)
template <int NCH, bool CHUNKWISE = false>
__device__ __forceinline__ void loc_project(Row<NCH>& t, const float* loc4, const float* w_loc, const float* bias_loc, int lane) {
  const float l0 = loc4[0], l1 = loc4[1], l2 = loc4[2], l3 = loc4[3];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int col = c * 256 + lane * 4 + e;
      const float4 w = *reinterpret_cast<const float4*>(w_loc + col * 4);
      t.v[c][e] = bias_loc[col] + l0 * w.x + l1 * w.y + l2 * w.z + l3 * w.w;
    }
    if constexpr (CHUNKWISE) __builtin_amdgcn_sched_barrier(0);
  }
}

template <typename T, int NCH>
__global__ __launch_bounds__(256) void pano_embed_fwd_kernel(const T* __restrict__ a, const T* __restrict__ d,
                                                             const float* __restrict__ loc, const int64_t* __restrict__ nav,
                                                             PanoEmbedParams p, float* __restrict__ y, float* __restrict__ stats,
                                                             int M, Drop drop) {
  constexpr int H = NCH * 256;
  const int lane = threadIdx.x & 63;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += gridDim.x * 4) {
    float* st = stats + (long)row * 8;
    Row<NCH> e, x, t;
    float mean, rstd;
    row_load<NCH>(x, a + (long)row * H, lane);
    row_normalize<NCH>(x, 1e-12f, mean, rstd);
    row_affine<NCH>(e, x, p.g_img, p.b_img, lane);
    if (lane == 0) { st[0] = mean; st[1] = rstd; }
    if (d != nullptr) {
      row_load<NCH>(x, d + (long)row * H, lane);
      row_normalize<NCH>(x, 1e-12f, mean, rstd);
      row_affine<NCH>(t, x, p.g_dep, p.b_dep, lane);
      row_add<NCH>(e, t);
      if (lane == 0) { st[2] = mean; st[3] = rstd; }
    }
    loc_project<NCH>(x, loc + (long)row * 4, p.w_loc, p.bias_loc, lane);
    row_normalize<NCH>(x, 1e-12f, mean, rstd);
    row_affine<NCH>(t, x, p.g_loc, p.b_loc, lane);
    row_add<NCH>(e, t);
    if (lane == 0) { st[4] = mean; st[5] = rstd; }
    row_load<NCH>(t, p.nav_emb + nav[row] * H, lane);
    row_add<NCH>(e, t);
    row_load<NCH>(t, p.type1, lane);
    row_add<NCH>(e, t);
    row_normalize<NCH>(e, 1e-12f, mean, rstd);
    row_affine<NCH>(t, e, p.g_out, p.b_out, lane);
    row_dropout<NCH>(t, drop, row, lane);             // img_embeddings.dropout vilmodel_cmt.py:711
    row_store<NCH>(t, y + (long)row * H, lane);
    if (lane == 0) { st[6] = mean; st[7] = rstd; }
  }
}

// Backward of the fuse, in THREE launches (round 5).  The single-launch form kept sixteen row-sized accumulators per lane: 482 registers,
// 226 of them AGPRs (tools/kernel_resources.py) -- the register class DESIGN.md §3.6 bans outside the matrix-core kernels -- and one
// wavefront per SIMD.  A first rewrite moved eight accumulators into per-wavefront LDS regions (108 KB): no AGPRs, but its sums then
// deviated from run to run whenever other kernels' workgroups shared the CU (race screen of tests/test_variants_gpu.py; clean with the
// CU's whole LDS to itself, which cost the step 0.1 ms; profiles/r05_pano_embed_race.txt).  Now each launch rebuilds the row's branch sum
// and carries only its own accumulators in registers:
//   PART 0  the outer LayerNorm: gamma_out, beta_out, sum(de), nav_emb[0]            (4 rows; sum(de) serves type1, the three inner
//           biases and nav_emb[0] + nav_emb[1]: they all receive the same column sum of the branch-sum gradient `de`)
//   PART 1  the RGB and depth branches: gamma_img, gamma_dep (2 rows) and the GEMM-side gradients da / dd
//   PART 2  the angle branch: gamma_loc, bias_loc, w_loc[., 0..3]                    (6 rows)
// (the row state itself -- e, x, t, de, the operand rows in flight -- is ~175 registers: at most six accumulator rows fit beside it)
// The base pointers of the parameter vectors pass through an empty asm every row trip: the compiler used to hoist all eleven vectors
// (plus the [H, 4] angle projection) out of the three-trip loop, ~180 registers of loop-invariant operands.
template <typename T, int NCH, int PART>
__global__ __launch_bounds__(256, 2) void pano_embed_bwd_kernel(const float* __restrict__ dy, const T* __restrict__ a,
                                                                const T* __restrict__ d, const float* __restrict__ loc,
                                                                const int64_t* __restrict__ nav, const float* __restrict__ stats,
                                                                PanoEmbedParams p, PanoEmbedGrads g, T* __restrict__ da,
                                                                T* __restrict__ dd, int M, Drop drop) {
  constexpr int H = NCH * 256;
  extern __shared__ __attribute__((aligned(16))) float scratch[];   // 4*H floats (block_flush staging)
  Row<NCH> A0, A1, A2, A3, A4, A5;       // PART 0: go, bo, ty, nav0 | PART 1: gi, gd | PART 2: gl, lb, lw0..3
  row_zero<NCH>(A0); row_zero<NCH>(A1);
  if constexpr (PART != 1) { row_zero<NCH>(A2); row_zero<NCH>(A3); }
  if constexpr (PART == 2) { row_zero<NCH>(A4); row_zero<NCH>(A5); }
  const int lane = threadIdx.x & 63;
  const PanoEmbedParams p0 = p;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += gridDim.x * 4) {
    PanoEmbedParams p = p0;
    asm volatile("" : "+s"(p.g_img), "+s"(p.b_img), "+s"(p.g_dep), "+s"(p.b_dep), "+s"(p.w_loc), "+s"(p.bias_loc));
    asm volatile("" : "+s"(p.g_loc), "+s"(p.b_loc), "+s"(p.nav_emb), "+s"(p.type1), "+s"(p.g_out), "+s"(p.b_out));
    const float* st = stats + (long)row * 8;
    Row<NCH> e, x, t, de;
    // rebuild e = sum of the branches, normalised (outer xhat)
    row_load<NCH>(x, a + (long)row * H, lane);
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int k = 0; k < 4; ++k) x.v[c][k] = (x.v[c][k] - st[0]) * st[1];
    row_affine<NCH>(e, x, p.g_img, p.b_img, lane);
    if (d != nullptr) {
      row_load<NCH>(x, d + (long)row * H, lane);
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int k = 0; k < 4; ++k) x.v[c][k] = (x.v[c][k] - st[2]) * st[3];
      row_affine<NCH>(t, x, p.g_dep, p.b_dep, lane);
      row_add<NCH>(e, t);
    }
    loc_project<NCH, PART == 2>(x, loc + (long)row * 4, p.w_loc, p.bias_loc, lane);
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int k = 0; k < 4; ++k) x.v[c][k] = (x.v[c][k] - st[4]) * st[5];
    row_affine<NCH>(t, x, p.g_loc, p.b_loc, lane);
    row_add<NCH>(e, t);
    const long nv = nav[row];
    row_load<NCH>(t, p.nav_emb + nv * H, lane);
    row_add<NCH>(e, t);
    row_load<NCH>(t, p.type1, lane);
    row_add<NCH>(e, t);
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int k = 0; k < 4; ++k) e.v[c][k] = (e.v[c][k] - st[6]) * st[7];
    // outer LN backward
    row_load<NCH>(de, dy + (long)row * H, lane);
    row_dropout<NCH>(de, drop, row, lane);
    if constexpr (PART == 0) {
      acc_mul<NCH>(A0, de, e);                        // gamma_out
      acc_scaled<NCH>(A1, de, 1.0f);                  // beta_out
    }
    row_ln_bwd<NCH>(de, e, p.g_out, st[7], lane);     // de = grad wrt the branch sum
    if constexpr (PART == 0) {
      acc_scaled<NCH>(A2, de, 1.0f);                  // = d type1 = d beta_{img, dep, loc} = d nav_emb[0] + d nav_emb[1]
      acc_scaled<NCH>(A3, de, nv == 0 ? 1.0f : 0.0f);
    } else if constexpr (PART == 2) {
      // loc branch (x still holds its xhat)
      t = de;
      acc_mul<NCH>(A0, t, x);
      row_ln_bwd<NCH>(t, x, p.g_loc, st[5], lane);
      acc_scaled<NCH>(A1, t, 1.0f);
      {
        const float* l4 = loc + (long)row * 4;
        acc_scaled<NCH>(A2, t, l4[0]); acc_scaled<NCH>(A3, t, l4[1]);
        acc_scaled<NCH>(A4, t, l4[2]); acc_scaled<NCH>(A5, t, l4[3]);
      }
    } else {
      // img branch
      row_load<NCH>(x, a + (long)row * H, lane);
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int k = 0; k < 4; ++k) x.v[c][k] = (x.v[c][k] - st[0]) * st[1];
      t = de;
      acc_mul<NCH>(A0, t, x);
      row_ln_bwd<NCH>(t, x, p.g_img, st[1], lane);
      row_store<NCH>(t, da + (long)row * H, lane);
      if (d != nullptr) {
        row_load<NCH>(x, d + (long)row * H, lane);
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
          for (int k = 0; k < 4; ++k) x.v[c][k] = (x.v[c][k] - st[2]) * st[3];
        t = de;
        acc_mul<NCH>(A1, t, x);
        row_ln_bwd<NCH>(t, x, p.g_dep, st[3], lane);
        row_store<NCH>(t, dd + (long)row * H, lane);
      }
    }
  }
  if constexpr (PART == 0) {
    block_flush<NCH>(scratch, A0, g.g_out, 1, 0); block_flush<NCH>(scratch, A1, g.b_out, 1, 0);
    block_flush<NCH>(scratch, A2, g.b_img, 1, 0);
    if (d != nullptr) block_flush<NCH>(scratch, A2, g.b_dep, 1, 0);
    block_flush<NCH>(scratch, A2, g.b_loc, 1, 0);
    block_flush<NCH>(scratch, A2, g.type1, 1, 0);
    block_flush<NCH>(scratch, A3, g.nav_emb, 1, 0);
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int k = 0; k < 4; ++k) A3.v[c][k] = A2.v[c][k] - A3.v[c][k];           // rows with nav_type 1
    block_flush<NCH>(scratch, A3, g.nav_emb + H, 1, 0);
  } else if constexpr (PART == 1) {
    block_flush<NCH>(scratch, A0, g.g_img, 1, 0);
    if (d != nullptr) block_flush<NCH>(scratch, A1, g.g_dep, 1, 0);
  } else {
    block_flush<NCH>(scratch, A0, g.g_loc, 1, 0);
    block_flush<NCH>(scratch, A1, g.bias_loc, 1, 0);
    block_flush<NCH>(scratch, A2, g.w_loc, 4, 0); block_flush<NCH>(scratch, A3, g.w_loc, 4, 1);
    block_flush<NCH>(scratch, A4, g.w_loc, 4, 2); block_flush<NCH>(scratch, A5, g.w_loc, 4, 3);
  }
}

// --------------------------------------------------------------------------------------
// Graph-node embedding: x = img + step_emb[step] + LN(pos.Wp^T + bp)      forward_navigation vilmodel_cmt.py:728-730
// --------------------------------------------------------------------------------------
template <int NCH, int PK> __device__ __forceinline__ void pos_project(Row<NCH>& t, const float* pos, const float* w,
                                                                       const float* bias, int lane) {
  float pv[PK];
#pragma unroll
  for (int j = 0; j < PK; ++j) pv[j] = pos[j];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int col = c * 256 + lane * 4 + e;
      float s = bias[col];
#pragma unroll
      for (int j = 0; j < PK; ++j) s += pv[j] * w[col * PK + j];
      t.v[c][e] = s;
    }
}

template <typename T, int NCH, int PK>
__global__ __launch_bounds__(256) void gmap_embed_fwd_kernel(const float* __restrict__ img, const int64_t* __restrict__ step_ids,
                                                             const float* __restrict__ pos, const float* __restrict__ step_emb,
                                                             const float* __restrict__ w_pos, const float* __restrict__ b_pos,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ x, T* __restrict__ xt, float* __restrict__ stats,
                                                             int M) {
  constexpr int H = NCH * 256;
  const int lane = threadIdx.x & 63;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += gridDim.x * 4) {
    Row<NCH> t, o, u;
    pos_project<NCH, PK>(t, pos + (long)row * PK, w_pos, b_pos, lane);
    float mean, rstd;
    row_normalize<NCH>(t, 1e-12f, mean, rstd);
    row_affine<NCH>(o, t, gamma, beta, lane);
    row_load<NCH>(u, step_emb + step_ids[row] * H, lane);
    row_add<NCH>(o, u);
    row_load<NCH>(u, img + (long)row * H, lane);
    row_add<NCH>(o, u);
    row_store<NCH>(o, x + (long)row * H, lane);
    if (xt != nullptr) row_store<NCH>(o, xt + (long)row * H, lane);
    if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
  }
}

// Register budget (round 4): the first form of this kernel kept seven per-lane accumulators for d w_pos (7 x NCH x 4 floats) beside
// the four others, the hoisted projection weights and two rows in flight: 256 VGPRs + 52 AGPRs, with the row's position features
// parked in AGPRs between their two uses -- and that parked copy was sporadically wrong whenever a grouped weight-gradient GEMM
// ran beside this kernel on the side stream (d w_pos off by up to 2 % in ~40 % of the steps of configuration 2, every other
// output exact; found by tests/test_planner_gpu.py in file order, present since round 3; profiles/r04_gmap_pos_race.txt).  Now
// d w_pos is accumulated per THREAD for the thread's own columns from the four rows a block has just finished (they pass through
// the LDS scratch): 3 x 7 accumulators instead of 84, no cross-wave reduction for them, and the kernel stays far below 256
// registers (no AGPRs, no scratch).
template <typename T, int NCH, int PK>
__global__ __launch_bounds__(256) void gmap_embed_bwd_kernel(const float* __restrict__ dx, const int64_t* __restrict__ step_ids,
                                                             const float* __restrict__ pos, const float* __restrict__ w_pos,
                                                             const float* __restrict__ b_pos, const float* __restrict__ gamma,
                                                             const float* __restrict__ stats, float* __restrict__ d_step_emb,
                                                             float* __restrict__ d_w_pos, float* __restrict__ d_b_pos,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta, int M) {
  constexpr int H = NCH * 256;
  extern __shared__ __attribute__((aligned(16))) float scratch[];   // 4*H floats
  Row<NCH> a_g, a_b, a_bp, a_s0;                                     // dgamma, dbeta, d_b_pos, d_step_emb[0]
  row_zero<NCH>(a_g); row_zero<NCH>(a_b); row_zero<NCH>(a_bp); row_zero<NCH>(a_s0);
  float a_w[NCH][PK];                                                // d w_pos[col][j] of this thread's columns tid + 256 k
#pragma unroll
  for (int k = 0; k < NCH; ++k)
#pragma unroll
    for (int j = 0; j < PK; ++j) a_w[k][j] = 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll 1
  for (int base = blockIdx.x * 4; base < M; base += gridDim.x * 4) {   // four rows per block and iteration, one per wavefront
    const int row = base + wave;
    if (row < M) {
      Row<NCH> t, d;
      pos_project<NCH, PK>(t, pos + (long)row * PK, w_pos, b_pos, lane);
      const float mean = stats[2 * row], rstd = stats[2 * row + 1];
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) t.v[c][e] = (t.v[c][e] - mean) * rstd;
      row_load<NCH>(d, dx + (long)row * H, lane);
      // step id 0 marks the [stop] token and every unvisited (ghost) node (ss_trainer_ETP.py:364-366,393): most rows hit
      // table row 0, and per-row atomics on one row serialise in L2 -- those rows go through the block accumulator instead
      const int64_t sid = step_ids[row];
      if (sid == 0) acc_scaled<NCH>(a_s0, d, 1.0f);
      else global_acc<NCH>(d_step_emb + sid * H, d, lane);
      acc_mul<NCH>(a_g, d, t);
      acc_scaled<NCH>(a_b, d, 1.0f);
      row_ln_bwd<NCH>(d, t, gamma, rstd, lane);
      acc_scaled<NCH>(a_bp, d, 1.0f);
      row_store<NCH>(d, scratch + wave * H, lane);                   // d (LayerNorm input gradient) of this row for the d w_pos pass
    }
    __syncthreads();
    const int nrows = min(4, M - base);
    for (int r = 0; r < nrows; ++r) {
      const float* pr = pos + (long)(base + r) * PK;
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        const float dv = scratch[r * H + k * 256 + threadIdx.x];
#pragma unroll
        for (int j = 0; j < PK; ++j) a_w[k][j] += dv * pr[j];
      }
    }
    __syncthreads();
  }
  block_flush<NCH>(scratch, a_g, dgamma, 1, 0);
  block_flush<NCH>(scratch, a_b, dbeta, 1, 0);
  block_flush<NCH>(scratch, a_bp, d_b_pos, 1, 0);
  block_flush<NCH>(scratch, a_s0, d_step_emb, 1, 0);
#pragma unroll
  for (int k = 0; k < NCH; ++k)
#pragma unroll
    for (int j = 0; j < PK; ++j) atomicAdd(d_w_pos + (long)(k * 256 + threadIdx.x) * PK + j, a_w[k][j]);
}

// --------------------------------------------------------------------------------------
// SAP head tail: logit = LN(r).w2 + b2, -inf where visited or padded     NextActionPrediction vilmodel_cmt.py:651-661,
//   r = relu(x.W1^T + b1) comes from the MFMA GEMM (ReLU epilogue).                                    :742-744
// --------------------------------------------------------------------------------------
template <typename T, int NCH>
__global__ __launch_bounds__(256) void sap_tail_fwd_kernel(const T* __restrict__ r, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ w2,
                                                           const float* __restrict__ b2, const uint8_t* __restrict__ visited,
                                                           const uint8_t* __restrict__ valid, float* __restrict__ logits,
                                                           float* __restrict__ stats, int M, Drop drop) {
  constexpr int H = NCH * 256;
  const int lane = threadIdx.x & 63;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += gridDim.x * 4) {
    Row<NCH> x, n, w;
    row_load<NCH>(x, r + (long)row * H, lane);
    float mean, rstd;
    row_normalize<NCH>(x, 1e-12f, mean, rstd);
    row_affine<NCH>(n, x, gamma, beta, lane);
    row_dropout<NCH>(n, drop, row, lane);             // NextActionPrediction Dropout vilmodel_cmt.py:657
    row_load<NCH>(w, w2, lane);
    const float v = row_dot<NCH>(n, w) + b2[0];
    if (lane == 0) {
      const bool masked = (visited && visited[row]) || (valid && !valid[row]);
      logits[row] = masked ? -INFINITY : v;
      stats[2 * row] = mean; stats[2 * row + 1] = rstd;
    }
  }
}

// dz (written over the ReLU mask) = LNbwd(dlogit*w2) * (r > 0)
template <typename T, int NCH>
__global__ __launch_bounds__(256) void sap_tail_bwd_kernel(const float* __restrict__ dlogits, const T* __restrict__ r,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ w2, const float* __restrict__ stats,
                                                           const uint8_t* __restrict__ visited, const uint8_t* __restrict__ valid,
                                                           T* __restrict__ dz, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, float* __restrict__ dw2,
                                                           float* __restrict__ db2, int M, Drop drop) {
  constexpr int H = NCH * 256;
  __shared__ float scratch[4 * H];
  __shared__ float sb2[4];
  Row<NCH> a_g, a_b, a_w;   // dgamma, dbeta, dw2
  row_zero<NCH>(a_g); row_zero<NCH>(a_b); row_zero<NCH>(a_w);
  float a_b2 = 0.f;
  const int lane = threadIdx.x & 63;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += gridDim.x * 4) {
    const bool masked = (visited && visited[row]) || (valid && !valid[row]);
    const float dl = masked ? 0.f : dlogits[row];
    Row<NCH> x, rr, n, d;
    row_load<NCH>(rr, r + (long)row * H, lane);
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) x.v[c][e] = (rr.v[c][e] - mean) * rstd;
    row_affine<NCH>(n, x, gamma, beta, lane);
    row_dropout<NCH>(n, drop, row, lane);
    acc_scaled<NCH>(a_w, n, dl);                     // dw2 += dl * dropout(n)
    a_b2 += dl;                                      // db2 (same value on every lane)
    row_load<NCH>(d, w2, lane);
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) d.v[c][e] *= dl;
    row_dropout<NCH>(d, drop, row, lane);            // d n = dl * w2 * mask/(1-p)
    acc_mul<NCH>(a_g, d, x);
    acc_scaled<NCH>(a_b, d, 1.0f);
    row_ln_bwd<NCH>(d, x, gamma, rstd, lane);
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) d.v[c][e] = rr.v[c][e] > 0.f ? d.v[c][e] : 0.f;
    row_store<NCH>(d, dz + (long)row * H, lane);
  }
  if (lane == 0) sb2[threadIdx.x >> 6] = a_b2;
  block_flush<NCH>(scratch, a_g, dgamma, 1, 0);
  block_flush<NCH>(scratch, a_b, dbeta, 1, 0);
  block_flush<NCH>(scratch, a_w, dw2, 1, 0);
  if (threadIdx.x == 0) atomicAdd(db2, sb2[0] + sb2[1] + sb2[2] + sb2[3]);
}

// --------------------------------------------------------------------------------------
// Cross entropy (sum, ignore_index) + its gradient        F.cross_entropy ss_trainer_ETP.py:892
//   loss = scale * sum_b [lse(logits_b) - logits_b[y_b]],  dlogits = scale * (softmax - onehot)
// One workgroup (B is the episode batch: tens of rows): the loss is reduced in LDS and STORED, so the caller needs no
// zeroing launch in front of it.
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void sap_ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                      float* __restrict__ loss, float* __restrict__ dlogits, int B, int G,
                                                      float scale, long ignore_index) {
  __shared__ float red[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;      // 16 waves: two rows each at B = 32
  float acc = 0.f;
  for (int row = wave; row < B; row += 16) {
    const float* s = logits + (long)row * G;
    const long y = labels[row];
    const bool keep = (y != ignore_index);
    float mx = -INFINITY;
    for (int k = lane; k < G; k += 64) mx = fmaxf(mx, s[k]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int k = lane; k < G; k += 64) sum += expf(s[k] - mx);
    sum = wave_sum(sum);
    const float lse = mx + logf(sum);
    if (dlogits != nullptr)
      for (int k = lane; k < G; k += 64) {
        float gk = 0.f;
        if (keep) gk = scale * (expf(s[k] - lse) - (k == y ? 1.f : 0.f));
        dlogits[(long)row * G + k] = gk;
      }
    if (keep) acc += scale * (lse - s[y]);
  }
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w];
    *loss = t;
  }
}

// Masked-LM cross-entropy over the vocabulary (pretrain_cmt.py:155-159, reduction 'none' then .mean()): one workgroup per
// masked token.  logits fp32 [Nm, ldv] (columns >= V are padding); writes dlogits = scale*(softmax - onehot) in the
// operand dtype (padding columns 0) and accumulates scale * nll into *loss.
template <typename T>
__global__ __launch_bounds__(256) void vocab_ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                       float* __restrict__ loss, T* __restrict__ dl, int V, int ldv, float scale) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* s = logits + (long)row * ldv;
  float mx = -INFINITY;
  for (int k = tid; k < V; k += 256) mx = fmaxf(mx, s[k]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int k = tid; k < V; k += 256) sum += expf(s[k] - mx);
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  const float lse = mx + logf(red[0] + red[1] + red[2] + red[3]);
  const long y = labels[row];
  T* d = dl + (long)row * ldv;
  for (int k = tid; k < ldv; k += 256) {
    float g = 0.f;
    if (k < V) g = scale * (expf(s[k] - lse) - (k == y ? 1.f : 0.f));
    Elem<T>::st(d + k, g);
  }
  if (tid == 0) atomicAdd(loss, scale * (lse - s[y]));
}

// d <- d * gelu'(z)   (BertPredictionHeadTransform's activation, between a LayerNorm backward and a weight gradient)
template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(T* __restrict__ d, const T* __restrict__ z, long n) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float x = Elem<T>::ld(z + i);
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.39894228040143268f * expf(-0.5f * x * x);
    Elem<T>::st(d + i, Elem<T>::ld(d + i) * (cdf + x * pdf));
  }
}

// --------------------------------------------------------------------------------------
// Weighted gather-sum (node aggregation and its transpose for backward):
//   out[n,:] (+)= sum_{j in [ptr[n],ptr[n+1])} w[j] * src[idx[j],:]
// covers the masked panorama mean (ss_trainer_ETP.py:838-839), ghost-node means (graph_utils.py:272-276) and
// pretrain _aggregate_gmap_features (pretrain_src/.../vilmodel.py:585-619).
// --------------------------------------------------------------------------------------
template <typename T, int NCH>
__global__ __launch_bounds__(256) void gather_sum_kernel(const T* __restrict__ src, const int32_t* __restrict__ ptr,
                                                         const int32_t* __restrict__ idx, const float* __restrict__ w,
                                                         T* __restrict__ out, int N, int accumulate) {
  constexpr int H = NCH * 256;
  const int lane = threadIdx.x & 63;
  for (int n = blockIdx.x * 4 + (threadIdx.x >> 6); n < N; n += gridDim.x * 4) {
    Row<NCH> o, t;
    if (accumulate) row_load<NCH>(o, out + (long)n * H, lane);
    else {
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) o.v[c][e] = 0.f;
    }
    for (int j = ptr[n]; j < ptr[n + 1]; ++j) {
      row_load<NCH>(t, src + (long)idx[j] * H, lane);
      const float wj = w[j];
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) o.v[c][e] += wj * t.v[c][e];
    }
    row_store<NCH>(o, out + (long)n * H, lane);
  }
}

// --------------------------------------------------------------------------------------
// Column sum (bias gradients): db[n] += sum_m dY[m,n]
// --------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ dy, long ld, float* __restrict__ db, int M, int N,
                                                     int rows_per_block) {
  __shared__ float red[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 256 + lane * 4;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  if (col < N)
    for (int r = r0 + wave; r < r1; r += 4) {
      float v[4];
      load4(dy + (long)r * ld + col, v);
      a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3];
    }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[wave][lane * 4 + e] = a[e];
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < N) atomicAdd(db + c, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// --------------------------------------------------------------------------------------
// dtype conversion (fp32 master weights / inputs -> bf16 operands) and fp32 axpy-free helpers
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
  const long stride = (long)gridDim.x * blockDim.x * 8;
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
    if (i + 8 <= n) {
      const float4 a = *reinterpret_cast<const float4*>(src + i), b = *reinterpret_cast<const float4*>(src + i + 4);
      uint4 o;
      o.x = pack_bf16(a.x, a.y);
      o.y = pack_bf16(a.z, a.w);
      o.z = pack_bf16(b.x, b.y);
      o.w = pack_bf16(b.z, b.w);
      *reinterpret_cast<uint4*>(dst + i) = o;
    } else {
      for (long j = i; j < n; ++j) dst[j] = f32_to_bf16(src[j]);
    }
  }
}
// dst(T) = dropout(src): ETP.forward drop_env on the RGB view features (Policy_ViewSelection_ETP.py:102,345) fused with the
// fp32 -> operand-dtype conversion
template <typename T>
__global__ __launch_bounds__(256) void cast_drop_kernel(const float* __restrict__ src, T* __restrict__ dst, long n, Drop drop) {
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i + 4 <= n; i += stride) {
    float v[4];
    load4(src + i, v);
    if (drop.p > 0.f) {
      float dm[4];
      drop_mult_run<4>(drop.seed, (uint32_t)i, drop.p, drop.inv_keep, dm);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= dm[e];
    }
    store4(dst + i, v);
  }
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, long n,
                                                            float scale) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = bf16_to_f32(src[i]) * scale;
}
__global__ __launch_bounds__(256) void copy_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, long n) {
  const long n4 = n / 4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
    reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
  for (long i = n4 * 4 + blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void scale_f32_kernel(float* __restrict__ p, long n, float scale) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] *= scale;
}

// ---- host launchers -------------------------------------------------------------------
#define ETP_DISPATCH_H(H, CALL)                                                         \
  switch ((H) / 256) {                                                                  \
    case 1: { constexpr int NCH = 1; CALL; } break;                                     \
    case 2: { constexpr int NCH = 2; CALL; } break;                                     \
    case 3: { constexpr int NCH = 3; CALL; } break;                                     \
    default: return fail(ETP_ERR_INVALID, "hidden size must be 256, 512 or 768");       \
  }

static inline int row_grid(int M, int cap) { return (int)std::min<long>(((long)M + 3) / 4, cap); }

// Activations that cross kernels of the "residual stream" are fp32 (y / dy below); `yt` / `xt` are optional copies in
// the GEMM operand dtype `dtype` (NULL in fp32 mode, where the fp32 tensor itself is the operand).
int text_embed_fwd(int dtype, const int64_t* ids, const float* word, const float* pos, const float* type0, const float* gamma,
                   const float* beta, float* y, void* yt, float* stats, int B, int L, int H, float eps, hipStream_t st, Drop drop) {
  ETP_REQUIRE(B > 0 && L > 0 && H % 256 == 0, "bad dims");
  const int M = B * L, grid = row_grid(M, 4096);
  if (dtype == ETP_BF16) { ETP_DISPATCH_H(H, ETP_LAUNCH((text_embed_fwd_kernel<bf16_t, NCH>), dim3(grid), dim3(256), 0, st, ids, word, pos, type0, gamma, beta, y, (bf16_t*)yt, stats, M, L, eps, drop)); }
  else { ETP_DISPATCH_H(H, ETP_LAUNCH((text_embed_fwd_kernel<float, NCH>), dim3(grid), dim3(256), 0, st, ids, word, pos, type0, gamma, beta, y, (float*)yt, stats, M, L, eps, drop)); }
  ETP_CHECK_LAUNCH("text_embed_fwd");
  return ETP_OK;
}

int text_embed_bwd(int dtype, const float* dy, const int64_t* ids, const float* word, const float* pos, const float* type0,
                   const float* gamma, const float* stats, float* dword, float* dpos, float* dtype0, float* dgamma, float* dbeta,
                   int B, int L, int H, hipStream_t st, Drop drop) {
  ETP_REQUIRE(B > 0 && L > 0 && H % 256 == 0, "bad dims");
  (void)dtype;
  const int M = B * L, grid = L < 1024 ? L : 1024;      // one workgroup per position (segment sum of its gradient over the batch)
  ETP_DISPATCH_H(H, ETP_LAUNCH_ROW(ROWF_TEXT_BWD, (text_embed_bwd_kernel<float, NCH>), dim3(grid), dim3(256), 0, st, dy, ids, word, pos, type0, gamma, stats, dword, dpos, dtype0, dgamma, dbeta, M, L, drop));
  ETP_CHECK_LAUNCH("text_embed_bwd");
  return ETP_OK;
}

int pano_embed_fwd(int dtype, const void* a, const void* d, const float* loc, const int64_t* nav, const PanoEmbedParams& p,
                   float* y, float* stats, int M, int H, hipStream_t st, Drop drop) {
  ETP_REQUIRE(M > 0 && H % 256 == 0, "bad dims");
  const int grid = row_grid(M, 4096);
  if (dtype == ETP_BF16) { ETP_DISPATCH_H(H, ETP_LAUNCH((pano_embed_fwd_kernel<bf16_t, NCH>), dim3(grid), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)d, loc, nav, p, y, stats, M, drop)); }
  else { ETP_DISPATCH_H(H, ETP_LAUNCH((pano_embed_fwd_kernel<float, NCH>), dim3(grid), dim3(256), 0, st, (const float*)a, (const float*)d, loc, nav, p, y, stats, M, drop)); }
  ETP_CHECK_LAUNCH("pano_embed_fwd");
  return ETP_OK;
}

template <typename TT, int NCH, int PART>
static int launch_pano_embed_bwd(int grid, size_t smem, hipStream_t st, const float* dy, const void* a, const void* d, const float* loc,
                                 const int64_t* nav, const float* stats, const PanoEmbedParams& p, const PanoEmbedGrads& g, void* da,
                                 void* dd, int M, Drop drop) {
  void (*kern)(const float*, const TT*, const TT*, const float*, const int64_t*, const float*, PanoEmbedParams, PanoEmbedGrads, TT*, TT*,
               int, Drop) = pano_embed_bwd_kernel<TT, NCH, PART>;
  ETP_LAUNCH_ROW(ROWF_PANO_BWD, kern, dim3(grid), dim3(256), (unsigned)smem, st, dy, (const TT*)a, (const TT*)d, loc, nav, stats, p, g,
                 (TT*)da, (TT*)dd, M, drop);
  ETP_CHECK_LAUNCH("pano_embed_bwd");
  return ETP_OK;
}

int pano_embed_bwd(int dtype, const float* dy, const void* a, const void* d, const float* loc, const int64_t* nav,
                   const float* stats, const PanoEmbedParams& p, const PanoEmbedGrads& g, void* da, void* dd, int M, int H,
                   hipStream_t st, Drop drop) {
  ETP_REQUIRE(M > 0 && H % 256 == 0, "bad dims");
  const int grid = row_grid(M, 96);   // ~3 rows per wave; each block flushes 16*H global atomics
  // The launches use 12 KB of LDS and ASK for the CU's whole LDS (ROWF_PANO_BWD in the ROW_EXCLUSIVE mask, launch.h; the size comes
  // from the device's properties and the attribute is set per device): no workgroup of another kernel can then share the CU.
  // Every form of this backward that shared SIMDs with other kernels' wavefronts returned sums that moved from run to run (race
  // screen of tests/test_variants_gpu.py, profiles/r05_pano_embed_race.txt): the round-4 kernel was only stable because its 482
  // registers kept each SIMD to itself.  The mechanism is open; the exclusivity is explicit.
  const size_t smem = 4 * (size_t)H * sizeof(float);
  if (dtype == ETP_BF16) {
    ETP_DISPATCH_H(H, ETP_TRY((launch_pano_embed_bwd<bf16_t, NCH, 0>(grid, smem, st, dy, a, d, loc, nav, stats, p, g, da, dd, M, drop))));
    ETP_DISPATCH_H(H, ETP_TRY((launch_pano_embed_bwd<bf16_t, NCH, 1>(grid, smem, st, dy, a, d, loc, nav, stats, p, g, da, dd, M, drop))));
    ETP_DISPATCH_H(H, ETP_TRY((launch_pano_embed_bwd<bf16_t, NCH, 2>(grid, smem, st, dy, a, d, loc, nav, stats, p, g, da, dd, M, drop))));
  } else {
    ETP_DISPATCH_H(H, ETP_TRY((launch_pano_embed_bwd<float, NCH, 0>(grid, smem, st, dy, a, d, loc, nav, stats, p, g, da, dd, M, drop))));
    ETP_DISPATCH_H(H, ETP_TRY((launch_pano_embed_bwd<float, NCH, 1>(grid, smem, st, dy, a, d, loc, nav, stats, p, g, da, dd, M, drop))));
    ETP_DISPATCH_H(H, ETP_TRY((launch_pano_embed_bwd<float, NCH, 2>(grid, smem, st, dy, a, d, loc, nav, stats, p, g, da, dd, M, drop))));
  }
  ETP_CHECK_LAUNCH("pano_embed_bwd");
  return ETP_OK;
}

int gmap_embed_fwd(int dtype, const float* img, const int64_t* step_ids, const float* pos, const float* step_emb,
                   const float* w_pos, const float* b_pos, const float* gamma, const float* beta, float* x, void* xt, float* stats,
                   int M, int H, int PK, hipStream_t st) {
  ETP_REQUIRE(M > 0 && H % 256 == 0 && PK == 7, "bad dims (pos feature width must be 7)");
  const int grid = row_grid(M, 4096);
  if (dtype == ETP_BF16) { ETP_DISPATCH_H(H, ETP_LAUNCH((gmap_embed_fwd_kernel<bf16_t, NCH, 7>), dim3(grid), dim3(256), 0, st, img, step_ids, pos, step_emb, w_pos, b_pos, gamma, beta, x, (bf16_t*)xt, stats, M)); }
  else { ETP_DISPATCH_H(H, ETP_LAUNCH((gmap_embed_fwd_kernel<float, NCH, 7>), dim3(grid), dim3(256), 0, st, img, step_ids, pos, step_emb, w_pos, b_pos, gamma, beta, x, (float*)xt, stats, M)); }
  ETP_CHECK_LAUNCH("gmap_embed_fwd");
  return ETP_OK;
}

int gmap_embed_bwd(int dtype, const float* dx, const int64_t* step_ids, const float* pos, const float* w_pos, const float* b_pos,
                   const float* gamma, const float* stats, float* d_step_emb, float* d_w_pos, float* d_b_pos, float* dgamma,
                   float* dbeta, int M, int H, int PK, hipStream_t st) {
  ETP_REQUIRE(M > 0 && H % 256 == 0 && PK == 7, "bad dims (pos feature width must be 7)");
  (void)dtype;
  const int grid = row_grid(M, 64);
  const size_t smem = 4 * (size_t)H * sizeof(float);
  ETP_DISPATCH_H(H, ETP_LAUNCH_ROW(ROWF_GMAP_BWD, (gmap_embed_bwd_kernel<float, NCH, 7>), dim3(grid), dim3(256), smem, st, dx, step_ids, pos, w_pos, b_pos, gamma, stats, d_step_emb, d_w_pos, d_b_pos, dgamma, dbeta, M));
  ETP_CHECK_LAUNCH("gmap_embed_bwd");
  return ETP_OK;
}

int sap_tail_fwd(int dtype, const void* r, const float* gamma, const float* beta, const float* w2, const float* b2,
                 const uint8_t* visited, const uint8_t* valid, float* logits, float* stats, int M, int H, hipStream_t st, Drop drop) {
  ETP_REQUIRE(M > 0 && H % 256 == 0, "bad dims");
  const int grid = row_grid(M, 4096);
  if (dtype == ETP_BF16) { ETP_DISPATCH_H(H, ETP_LAUNCH((sap_tail_fwd_kernel<bf16_t, NCH>), dim3(grid), dim3(256), 0, st, (const bf16_t*)r, gamma, beta, w2, b2, visited, valid, logits, stats, M, drop)); }
  else { ETP_DISPATCH_H(H, ETP_LAUNCH((sap_tail_fwd_kernel<float, NCH>), dim3(grid), dim3(256), 0, st, (const float*)r, gamma, beta, w2, b2, visited, valid, logits, stats, M, drop)); }
  ETP_CHECK_LAUNCH("sap_tail_fwd");
  return ETP_OK;
}

int sap_tail_bwd(int dtype, const float* dlogits, const void* r, const float* gamma, const float* beta, const float* w2,
                 const float* stats, const uint8_t* visited, const uint8_t* valid, void* dz, float* dgamma, float* dbeta,
                 float* dw2, float* db2, int M, int H, hipStream_t st, Drop drop) {
  ETP_REQUIRE(M > 0 && H % 256 == 0, "bad dims");
  const int grid = row_grid(M, 64);
  if (dtype == ETP_BF16) { ETP_DISPATCH_H(H, ETP_LAUNCH_ROW(ROWF_SAP_BWD, (sap_tail_bwd_kernel<bf16_t, NCH>), dim3(grid), dim3(256), 0, st, dlogits, (const bf16_t*)r, gamma, beta, w2, stats, visited, valid, (bf16_t*)dz, dgamma, dbeta, dw2, db2, M, drop)); }
  else { ETP_DISPATCH_H(H, ETP_LAUNCH_ROW(ROWF_SAP_BWD, (sap_tail_bwd_kernel<float, NCH>), dim3(grid), dim3(256), 0, st, dlogits, (const float*)r, gamma, beta, w2, stats, visited, valid, (float*)dz, dgamma, dbeta, dw2, db2, M, drop)); }
  ETP_CHECK_LAUNCH("sap_tail_bwd");
  return ETP_OK;
}

int sap_ce(const float* logits, const int64_t* labels, float* loss, float* dlogits, int B, int G, float scale, long ignore_index,
           hipStream_t st) {
  ETP_REQUIRE(B > 0 && G > 0, "bad dims");
  ETP_LAUNCH(sap_ce_kernel, dim3(1), dim3(1024), 0, st, logits, labels, loss, dlogits, B, G, scale, ignore_index);
  ETP_CHECK_LAUNCH("sap_ce");
  return ETP_OK;
}

int vocab_ce(int dtype, const float* logits, const int64_t* labels, float* loss, void* dl, int Nm, int V, int ldv, float scale,
             hipStream_t st) {
  ETP_REQUIRE(logits && labels && loss && dl && Nm > 0 && V > 0 && ldv >= V, "bad arguments");
  if (dtype == ETP_BF16) ETP_LAUNCH((vocab_ce_kernel<bf16_t>), dim3(Nm), dim3(256), 0, st, logits, labels, loss, (bf16_t*)dl, V, ldv, scale);
  else ETP_LAUNCH((vocab_ce_kernel<float>), dim3(Nm), dim3(256), 0, st, logits, labels, loss, (float*)dl, V, ldv, scale);
  ETP_CHECK_LAUNCH("vocab_ce");
  return ETP_OK;
}
int gelu_bwd_inplace(int dtype, void* d, const void* z, long n, hipStream_t st) {
  if (n <= 0) return ETP_OK;
  const int grid = (int)std::min<long>((n + 255) / 256, 2048);
  if (dtype == ETP_BF16) ETP_LAUNCH((gelu_bwd_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, (bf16_t*)d, (const bf16_t*)z, n);
  else ETP_LAUNCH((gelu_bwd_kernel<float>), dim3(grid), dim3(256), 0, st, (float*)d, (const float*)z, n);
  ETP_CHECK_LAUNCH("gelu_bwd");
  return ETP_OK;
}

int gather_sum(int dtype, const void* src, const int32_t* ptr, const int32_t* idx, const float* w, void* out, int N, int H,
               int accumulate, hipStream_t st) {
  ETP_REQUIRE(N > 0 && H % 256 == 0, "bad dims");
  const int grid = row_grid(N, 4096);
  if (dtype == ETP_BF16) { ETP_DISPATCH_H(H, ETP_LAUNCH((gather_sum_kernel<bf16_t, NCH>), dim3(grid), dim3(256), 0, st, (const bf16_t*)src, ptr, idx, w, (bf16_t*)out, N, accumulate)); }
  else { ETP_DISPATCH_H(H, ETP_LAUNCH((gather_sum_kernel<float, NCH>), dim3(grid), dim3(256), 0, st, (const float*)src, ptr, idx, w, (float*)out, N, accumulate)); }
  ETP_CHECK_LAUNCH("gather_sum");
  return ETP_OK;
}

int colsum(int dtype, const void* dy, long ld, float* db, int M, int N, hipStream_t st) {
  ETP_REQUIRE(M > 0 && N > 0 && N % 4 == 0 && ld % 4 == 0, "bad dims");
  const int rpb = 64;
  dim3 grid((N + 255) / 256, (M + rpb - 1) / rpb);
  if (dtype == ETP_BF16) ETP_LAUNCH((colsum_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)dy, ld, db, M, N, rpb);
  else ETP_LAUNCH((colsum_kernel<float>), grid, dim3(256), 0, st, (const float*)dy, ld, db, M, N, rpb);
  ETP_CHECK_LAUNCH("colsum");
  return ETP_OK;
}

int cast_f32_to_bf16(const float* src, void* dst, long n, hipStream_t st) {
  if (n <= 0) return ETP_OK;
  const int grid = (int)std::min<long>((n / 8 + 255) / 256 + 1, 4096);
  ETP_LAUNCH(cast_f32_bf16_kernel, dim3(grid), dim3(256), 0, st, src, (bf16_t*)dst, n);
  ETP_CHECK_LAUNCH("cast_f32_bf16");
  return ETP_OK;
}
int cast_drop(int dtype, const float* src, void* dst, long n, Drop drop, hipStream_t st) {
  if (n <= 0) return ETP_OK;
  ETP_REQUIRE(n % 4 == 0, "cast_drop: element count must be a multiple of 4");
  const int grid = (int)std::min<long>((n / 4 + 255) / 256, 4096);
  if (dtype == ETP_BF16) ETP_LAUNCH((cast_drop_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, src, (bf16_t*)dst, n, drop);
  else ETP_LAUNCH((cast_drop_kernel<float>), dim3(grid), dim3(256), 0, st, src, (float*)dst, n, drop);
  ETP_CHECK_LAUNCH("cast_drop");
  return ETP_OK;
}
int cast_bf16_to_f32(const void* src, float* dst, long n, float scale, hipStream_t st) {
  if (n <= 0) return ETP_OK;
  const int grid = (int)std::min<long>((n + 255) / 256, 4096);
  ETP_LAUNCH(cast_bf16_f32_kernel, dim3(grid), dim3(256), 0, st, (const bf16_t*)src, dst, n, scale);
  ETP_CHECK_LAUNCH("cast_bf16_f32");
  return ETP_OK;
}
__global__ __launch_bounds__(256) void zero_f32_kernel(float* __restrict__ dst, long n) {
  const long n4 = n / 4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
    reinterpret_cast<float4*>(dst)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long i = n4 * 4 + blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = 0.f;
}
int zero_f32(float* dst, long n, hipStream_t st) {
  if (n <= 0) return ETP_OK;
  const int grid = (int)std::min<long>((n / 4 + 255) / 256 + 1, 4096);
  ETP_LAUNCH(zero_f32_kernel, dim3(grid), dim3(256), 0, st, dst, n);
  ETP_CHECK_LAUNCH("zero_f32");
  return ETP_OK;
}
// device-to-device copy as an ordinary kernel: hipMemcpyAsync(D2D) goes through the runtime's blit path, which stalled
// the issuing stream for 100-300 us per copy in the step's kernel trace (tools/timeline.py, round 1)
int copy_f32(const float* src, float* dst, long n, hipStream_t st) {
  if (n <= 0 || src == dst) return ETP_OK;
  ETP_REQUIRE(((uintptr_t)src | (uintptr_t)dst) % 16 == 0, "copy_f32: 16-byte aligned buffers required");
  const int grid = (int)std::min<long>((n / 4 + 255) / 256 + 1, 2048);
  ETP_LAUNCH(copy_f32_kernel, dim3(grid), dim3(256), 0, st, src, dst, n);
  ETP_CHECK_LAUNCH("copy_f32");
  return ETP_OK;
}
// dst[t][i] = src[i], t < T  (16-byte vectors): the text K|V cache of Bt instructions replicated for T stacked rollout steps
__global__ __launch_bounds__(256) void repeat_block_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long n16, int T) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
    const uint4 v = src[i];
    for (int t = 0; t < T; ++t) dst[(long)t * n16 + i] = v;
  }
}
int repeat_block(const void* src, void* dst, long bytes, int T, hipStream_t st) {
  if (bytes <= 0 || T <= 0) return ETP_OK;
  ETP_REQUIRE(bytes % 16 == 0 && ((uintptr_t)src | (uintptr_t)dst) % 16 == 0, "repeat_block: 16-byte aligned blocks required");
  const long n16 = bytes / 16;
  const int grid = (int)std::min<long>((n16 + 255) / 256, 4096);
  ETP_LAUNCH(repeat_block_kernel, dim3(grid), dim3(256), 0, st, (const uint4*)src, (uint4*)dst, n16, T);
  ETP_CHECK_LAUNCH("repeat_block");
  return ETP_OK;
}
// dst[i] = sum_t src[t][i] accumulated in fp32 (4 elements per thread and iteration; n % 4 == 0)
template <typename T>
__global__ __launch_bounds__(256) void sum_steps_kernel(const T* __restrict__ src, T* __restrict__ dst, long n, int steps) {
  const long n4 = n / 4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < steps; ++t) {
      float v[4];
      load4(src + (long)t * n + i * 4, v);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] += v[e];
    }
    store4(dst + i * 4, a);
  }
}
int sum_steps(int dtype, const void* src, void* dst, long n, int steps, hipStream_t st) {
  if (n <= 0) return ETP_OK;
  ETP_REQUIRE(n % 4 == 0 && steps > 0 && ((uintptr_t)src | (uintptr_t)dst) % 16 == 0, "sum_steps: n % 4 == 0 and 16-byte aligned buffers");
  const int grid = (int)std::min<long>((n / 4 + 255) / 256, 4096);
  if (dtype == ETP_BF16) ETP_LAUNCH(sum_steps_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, n, steps);
  else ETP_LAUNCH(sum_steps_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)src, (float*)dst, n, steps);
  ETP_CHECK_LAUNCH("sum_steps");
  return ETP_OK;
}
int scale_f32(float* p, long n, float scale, hipStream_t st) {
  if (n <= 0) return ETP_OK;
  const int grid = (int)std::min<long>((n + 255) / 256, 4096);
  ETP_LAUNCH(scale_f32_kernel, dim3(grid), dim3(256), 0, st, p, n, scale);
  ETP_CHECK_LAUNCH("scale_f32");
  return ETP_OK;
}

}  // namespace etp
