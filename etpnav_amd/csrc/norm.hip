// LayerNorm forward/backward and masked row-softmax forward/backward (gfx950).
//
// Reference sites: every BertLayerNorm / nn.LayerNorm on the planner path
// (vilmodel_cmt.py:59,147,186,459-478,571,656; common/transformer.py:144-145; ops.py:19-23) and
// nn.Softmax(dim=-1) in BertSelfAttention / BertOutAttention (vilmodel_cmt.py:117-127, :335-346) plus the
// softmax inside nn.MultiheadAttention (common/transformer.py:138).
//
// Both are HBM-bound row kernels: one 64-lane wavefront owns one row, statistics are reduced with
// wave shuffles in fp32, I/O is 8/16-byte vectors per lane (coalesced 512 B / 1 KiB per wave instruction).
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "kernels.h"

namespace etp {

// --------------------------------------------------------------------------------------
// LayerNorm
// --------------------------------------------------------------------------------------
template <typename T, int NCH>   // H = NCH * 256
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ stats, int M, float eps) {
  constexpr int H = NCH * 256;
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < M; row += gridDim.x * wpb) {
    float v[NCH][4];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      load4(x + (long)row * H + c * 256 + lane * 4, v[c]);
      s += v[c][0] + v[c][1] + v[c][2] + v[c][3];
    }
    const float mean = wave_sum(s) * (1.0f / H);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[c][e] -= mean; q += v[c][e] * v[c][e]; }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / H) + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      const float4 gm = *reinterpret_cast<const float4*>(gamma + col);
      const float4 bt = *reinterpret_cast<const float4*>(beta + col);
      float o[4] = {v[c][0] * rstd * gm.x + bt.x, v[c][1] * rstd * gm.y + bt.y, v[c][2] * rstd * gm.z + bt.z,
                    v[c][3] * rstd * gm.w + bt.w};
      store4(y + (long)row * H + col, o);
    }
    if (stats != nullptr && lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*gamma ; dgamma += dy*xhat ; dbeta += dy
template <typename T, int NCH>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ stats, const float* __restrict__ gamma,
                                                     const T* __restrict__ add, T* __restrict__ dx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, int M) {
  constexpr int H = NCH * 256;
  __shared__ float red[4][2][H];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float ag[NCH][4], ab[NCH][4];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) { ag[c][e] = 0.f; ab[c][e] = 0.f; }

  for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    float xh[NCH][4], gy[NCH][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      float xv[4], dv[4];
      load4(x + (long)row * H + col, xv);
      load4(dy + (long)row * H + col, dv);
      const float4 gm = *reinterpret_cast<const float4*>(gamma + col);
      const float gmv[4] = {gm.x, gm.y, gm.z, gm.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xh[c][e] = (xv[e] - mean) * rstd;
        gy[c][e] = dv[e] * gmv[e];
        s1 += gy[c][e];
        s2 += gy[c][e] * xh[c][e];
        ag[c][e] += dv[e] * xh[c][e];
        ab[c][e] += dv[e];
      }
    }
    s1 = wave_sum(s1) * (1.0f / H);
    s2 = wave_sum(s2) * (1.0f / H);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = rstd * (gy[c][e] - s1 - xh[c][e] * s2);
      if (add != nullptr) {
        float av[4];
        load4(add + (long)row * H + col, av);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += av[e];
      }
      store4(dx + (long)row * H + col, o);
    }
  }
  if (dgamma == nullptr) return;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[wave][0][c * 256 + lane * 4 + e] = ag[c][e];
      red[wave][1][c * 256 + lane * 4 + e] = ab[c][e];
    }
  __syncthreads();
  for (int col = threadIdx.x; col < H; col += 256) {
    atomicAdd(dgamma + col, red[0][0][col] + red[1][0][col] + red[2][0][col] + red[3][0][col]);
    atomicAdd(dbeta + col, red[0][1][col] + red[1][1][col] + red[2][1][col] + red[3][1][col]);
  }
}

template <typename T>
static int ln_fwd_t(const void* x, const float* gamma, const float* beta, void* y, float* stats, int M, int H, float eps,
                    hipStream_t st) {
  const int grid = (int)std::min<long>((M + 3) / 4, 4096);
  switch (H / 256) {
    case 1: ETP_LAUNCH((ln_fwd_kernel<T, 1>), dim3(grid), dim3(256), 0, st, (const T*)x, gamma, beta, (T*)y, stats, M, eps); break;
    case 2: ETP_LAUNCH((ln_fwd_kernel<T, 2>), dim3(grid), dim3(256), 0, st, (const T*)x, gamma, beta, (T*)y, stats, M, eps); break;
    case 3: ETP_LAUNCH((ln_fwd_kernel<T, 3>), dim3(grid), dim3(256), 0, st, (const T*)x, gamma, beta, (T*)y, stats, M, eps); break;
    case 4: ETP_LAUNCH((ln_fwd_kernel<T, 4>), dim3(grid), dim3(256), 0, st, (const T*)x, gamma, beta, (T*)y, stats, M, eps); break;
    default: return fail(ETP_ERR_INVALID, "layer norm: hidden size must be 256, 512, 768 or 1024");
  }
  ETP_CHECK_LAUNCH("ln_fwd");
  return ETP_OK;
}

template <typename T>
static int ln_bwd_t(const void* dy, const void* x, const float* stats, const float* gamma, const void* add, void* dx,
                    float* dgamma, float* dbeta, int M, int H, hipStream_t st) {
  const int grid = (int)std::min<long>((M + 7) / 8, 128);   // 2 rows per wave minimum; each block flushes 2*H atomics
  switch (H / 256) {
    case 1: ETP_LAUNCH((ln_bwd_kernel<T, 1>), dim3(grid), dim3(256), 0, st, (const T*)dy, (const T*)x, stats, gamma, (const T*)add, (T*)dx, dgamma, dbeta, M); break;
    case 2: ETP_LAUNCH((ln_bwd_kernel<T, 2>), dim3(grid), dim3(256), 0, st, (const T*)dy, (const T*)x, stats, gamma, (const T*)add, (T*)dx, dgamma, dbeta, M); break;
    case 3: ETP_LAUNCH((ln_bwd_kernel<T, 3>), dim3(grid), dim3(256), 0, st, (const T*)dy, (const T*)x, stats, gamma, (const T*)add, (T*)dx, dgamma, dbeta, M); break;
    case 4: ETP_LAUNCH((ln_bwd_kernel<T, 4>), dim3(grid), dim3(256), 0, st, (const T*)dy, (const T*)x, stats, gamma, (const T*)add, (T*)dx, dgamma, dbeta, M); break;
    default: return fail(ETP_ERR_INVALID, "layer norm: hidden size must be 256, 512, 768 or 1024");
  }
  ETP_CHECK_LAUNCH("ln_bwd");
  return ETP_OK;
}

// ---- "stream" LayerNorm: the residual stream is always fp32 (as under the reference's autocast, where LayerNorm and
// the residual adds stay fp32); y/dx are the fp32 results, yt/dxt optional copies in the GEMM operand dtype T.
template <typename T, int NCH>
__global__ __launch_bounds__(256) void ln_fwd_s_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ y, T* __restrict__ yt,
                                                       float* __restrict__ stats, int M, float eps) {
  constexpr int H = NCH * 256;
  const int lane = threadIdx.x & 63;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += gridDim.x * 4) {
    float v[NCH][4];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      load4(x + (long)row * H + c * 256 + lane * 4, v[c]);
      s += v[c][0] + v[c][1] + v[c][2] + v[c][3];
    }
    const float mean = wave_sum(s) * (1.0f / H);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[c][e] -= mean; q += v[c][e] * v[c][e]; }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / H) + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      const float4 gm = *reinterpret_cast<const float4*>(gamma + col);
      const float4 bt = *reinterpret_cast<const float4*>(beta + col);
      float o[4] = {v[c][0] * rstd * gm.x + bt.x, v[c][1] * rstd * gm.y + bt.y, v[c][2] * rstd * gm.z + bt.z,
                    v[c][3] * rstd * gm.w + bt.w};
      if (y != nullptr) store4(y + (long)row * H + col, o);
      if (yt != nullptr) store4(yt + (long)row * H + col, o);
    }
    if (stats != nullptr && lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
  }
}

template <typename T, int NCH>
__global__ __launch_bounds__(256) void ln_bwd_s_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                       const float* __restrict__ stats, const float* __restrict__ gamma,
                                                       const float* __restrict__ add, float* __restrict__ dx, T* __restrict__ dxt,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                       float* __restrict__ part, int M, Drop drop) {
  constexpr int H = NCH * 256;
  __shared__ float red[4][2][H];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float ag[NCH][4], ab[NCH][4];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) { ag[c][e] = 0.f; ab[c][e] = 0.f; }
  for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    float xh[NCH][4], gy[NCH][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      float xv[4], dv[4];
      load4(x + (long)row * H + col, xv);
      load4(dy + (long)row * H + col, dv);
      const float4 gm = *reinterpret_cast<const float4*>(gamma + col);
      const float gmv[4] = {gm.x, gm.y, gm.z, gm.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xh[c][e] = (xv[e] - mean) * rstd;
        gy[c][e] = dv[e] * gmv[e];
        s1 += gy[c][e];
        s2 += gy[c][e] * xh[c][e];
        ag[c][e] += dv[e] * xh[c][e];
        ab[c][e] += dv[e];
      }
    }
    s1 = wave_sum(s1) * (1.0f / H);
    s2 = wave_sum(s2) * (1.0f / H);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = rstd * (gy[c][e] - s1 - xh[c][e] * s2);
      if (add != nullptr) {
        float av[4];
        load4(add + (long)row * H + col, av);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += av[e];
      }
      if (dx != nullptr) store4(dx + (long)row * H + col, o);
      if (dxt != nullptr) {
        if (drop.p > 0.f) {   // the operand copy is the gradient of a dropped dense output: d(dense) = dx * mask / (1-p)
          float dm[4];
          drop_mult_run<4>(drop.seed, (uint32_t)row * H + col, drop.p, drop.inv_keep, dm);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] *= dm[e];
        }
        store4(dxt + (long)row * H + col, o);
      }
    }
  }
  if (dgamma == nullptr) return;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[wave][0][c * 256 + lane * 4 + e] = ag[c][e];
      red[wave][1][c * 256 + lane * 4 + e] = ab[c][e];
    }
  __syncthreads();
  if (part != nullptr) {
    // two-stage parameter-gradient reduction: this block's column sums go to its own slab [2][H]; ln_part_reduce_kernel
    // (a leaf of the backward graph, issued off the dependent chain) adds the slabs into dgamma / dbeta
    float* slab = part + (long)blockIdx.x * 2 * H;
    for (int col = threadIdx.x; col < 2 * H; col += 256) {
      const int which = col / H, c = col % H;
      slab[col] = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
    }
    return;
  }
  for (int col = threadIdx.x; col < H; col += 256) {
    atomicAdd(dgamma + col, red[0][0][col] + red[1][0][col] + red[2][0][col] + red[3][0][col]);
    atomicAdd(dbeta + col, red[0][1][col] + red[1][1][col] + red[2][1][col] + red[3][1][col]);
  }
}

// dgamma[c] += sum_b part[b][0][c], dbeta[c] += sum_b part[b][1][c].  Grid (2H/256, slab chunks): one thread per column and
// chunk of LN_PART_CHUNK slabs (coalesced rows, independent loads in flight), one atomic per thread -- 2H x ~10 atomics in
// total instead of 2H per workgroup of the backward kernel.
constexpr int LN_PART_CHUNK = 32;
__global__ __launch_bounds__(256) void ln_part_reduce_kernel(const float* __restrict__ part, int nblk, int H,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= 2 * H) return;
  const int b0 = blockIdx.y * LN_PART_CHUNK, b1 = min(nblk, b0 + LN_PART_CHUNK);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int b = b0;
  for (; b + 4 <= b1; b += 4) {
    a0 += part[(long)(b + 0) * 2 * H + col]; a1 += part[(long)(b + 1) * 2 * H + col];
    a2 += part[(long)(b + 2) * 2 * H + col]; a3 += part[(long)(b + 3) * 2 * H + col];
  }
  for (; b < b1; ++b) a0 += part[(long)b * 2 * H + col];
  const float v = (a0 + a1) + (a2 + a3);
  if (col < H) atomicAdd(dgamma + col, v);
  else atomicAdd(dbeta + col - H, v);
}

#define ETP_LN_DISPATCH(FAM, KERN, T, GRID, ...)                                                                        \
  switch (H / 256) {                                                                                                 \
    case 1: ETP_LAUNCH_ROW(FAM, (KERN<T, 1>), dim3(GRID), dim3(256), 0, st, __VA_ARGS__); break;                     \
    case 2: ETP_LAUNCH_ROW(FAM, (KERN<T, 2>), dim3(GRID), dim3(256), 0, st, __VA_ARGS__); break;                     \
    case 3: ETP_LAUNCH_ROW(FAM, (KERN<T, 3>), dim3(GRID), dim3(256), 0, st, __VA_ARGS__); break;                     \
    case 4: ETP_LAUNCH_ROW(FAM, (KERN<T, 4>), dim3(GRID), dim3(256), 0, st, __VA_ARGS__); break;                     \
    default: return fail(ETP_ERR_INVALID, "layer norm: hidden size must be 256, 512, 768 or 1024");                 \
  }

// yt / dxt: copy in the operand dtype `dtype` (ignored when NULL; pass NULL in fp32 mode where y itself is the operand)
// MEASUREMENT ONLY (tools/r03_call19.sh): ETP_SKIP_LN names launches to drop -- "fwd" (ln_fwd_s), "bwd" (ln_bwd_s), "red"
// (ln_part_reduce) -- so that the step time shows what fusing them away could buy at most (results are wrong in that mode),
// like ETP_SKIP_WGRAD in planner.hip.
static bool skip_ln(const char* what) {
#ifdef ETP_EXPERIMENTS       // measurement builds only (tools/build_variant.sh ... -DETP_EXPERIMENTS): never in the shipped library
  const char* e = opt_str(OPT_SKIP_LN);
  return e && strstr(e, what) != nullptr;
#else
  (void)what;
  return false;
#endif
}

int ln_fwd_s(int dtype, const float* x, const float* gamma, const float* beta, float* y, void* yt, float* stats, int M, int H,
             float eps, hipStream_t st) {
  ETP_REQUIRE(M > 0 && H % 256 == 0 && (y || yt), "bad arguments");
  if (skip_ln("fwd")) return ETP_OK;
  const int grid = (int)std::min<long>((M + 3) / 4, 4096);
  if (dtype == ETP_BF16) { ETP_LN_DISPATCH(ROWF_LN_FWD, ln_fwd_s_kernel, bf16_t, grid, x, gamma, beta, y, (bf16_t*)yt, stats, M, eps) }
  else { ETP_LN_DISPATCH(ROWF_LN_FWD, ln_fwd_s_kernel, float, grid, x, gamma, beta, y, (float*)yt, stats, M, eps) }
  ETP_CHECK_LAUNCH("ln_fwd_s");
  return ETP_OK;
}
// number of workgroups ln_bwd_s launches for M rows (= slabs of the two-stage reduction): every wave gets the same number
// of rows, at most LN_BWD_MAX_BLOCKS blocks (one row per wavefront up to 4096 rows: 640 blocks at M = 2560; round 4, was 512 / two rows per wavefront)
static int ln_bwd_blocks(int M) {
  const int groups = (M + 3) / 4;
  const int cap = std::max(1, opt_int(OPT_LNBWD_GRID, LN_BWD_MAX_BLOCKS));
  const int rounds = (groups + cap - 1) / cap;
  return (groups + rounds - 1) / rounds;
}
size_t ln_bwd_part_bytes(int M, int H) { return (size_t)std::min(ln_bwd_blocks(M), LN_BWD_MAX_BLOCKS) * 2 * H * sizeof(float); }

int ln_bwd_s(int dtype, const float* dy, const float* x, const float* stats, const float* gamma, const float* add, float* dx,
             void* dxt, float* dgamma, float* dbeta, int M, int H, hipStream_t st, Drop drop, float* part) {
  ETP_REQUIRE(M > 0 && H % 256 == 0 && (dx || dxt), "bad arguments");
  ETP_REQUIRE(drop.p == 0.f || (dxt != nullptr && dxt != (void*)dx), "dropout needs a separate operand copy");
  if (skip_ln("bwd")) return ETP_OK;
  // without a slab buffer the blocks flush 2*H atomics each: keep their number low (128, the round-1 setting)
  const int grid = part ? ln_bwd_blocks(M) : (int)std::min<long>((M + 3) / 4, 128);
  if (dgamma == nullptr) part = nullptr;
  if (dtype == ETP_BF16) { ETP_LN_DISPATCH(ROWF_LN_BWD, ln_bwd_s_kernel, bf16_t, grid, dy, x, stats, gamma, add, dx, (bf16_t*)dxt, dgamma, dbeta, part, M, drop) }
  else { ETP_LN_DISPATCH(ROWF_LN_BWD, ln_bwd_s_kernel, float, grid, dy, x, stats, gamma, add, dx, (float*)dxt, dgamma, dbeta, part, M, drop) }
  ETP_CHECK_LAUNCH("ln_bwd_s");
  return ETP_OK;
}
int ln_part_reduce(const float* part, int M, int H, float* dgamma, float* dbeta, hipStream_t st) {
  ETP_REQUIRE(part && dgamma && dbeta && M > 0 && H % 256 == 0, "bad arguments");
  if (skip_ln("red")) return ETP_OK;
  const int nblk = ln_bwd_blocks(M);
  ETP_LAUNCH(ln_part_reduce_kernel, dim3((2 * H + 255) / 256, (nblk + LN_PART_CHUNK - 1) / LN_PART_CHUNK), dim3(256), 0, st, part, nblk, H,
             dgamma, dbeta);
  ETP_CHECK_LAUNCH("ln_part_reduce");
  return ETP_OK;
}

int ln_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* stats, int M, int H, float eps,
           hipStream_t st) {
  ETP_REQUIRE(M > 0 && H % 256 == 0, "bad dims");
  return dtype == ETP_BF16 ? ln_fwd_t<bf16_t>(x, gamma, beta, y, stats, M, H, eps, st)
                           : ln_fwd_t<float>(x, gamma, beta, y, stats, M, H, eps, st);
}
int ln_bwd(int dtype, const void* dy, const void* x, const float* stats, const float* gamma, const void* add, void* dx,
           float* dgamma, float* dbeta, int M, int H, hipStream_t st) {
  ETP_REQUIRE(M > 0 && H % 256 == 0, "bad dims");
  return dtype == ETP_BF16 ? ln_bwd_t<bf16_t>(dy, x, stats, gamma, add, dx, dgamma, dbeta, M, H, st)
                           : ln_bwd_t<float>(dy, x, stats, gamma, add, dx, dgamma, dbeta, M, H, st);
}

// --------------------------------------------------------------------------------------
// masked softmax over attention scores  S[b, h, q, 0:Lk]  (row pitch ldS >= Lk, pad written as 0)
//   s = S + keymask(b,k) + (w*dist[b,q,k] + b0)           (vilmodel_cmt.py:120, :391-393, :732-736)
//   keymask: mode 0 -> (1-m)*-10000 (ops.py:25-34);  mode 1 -> -inf where !m (nn.MultiheadAttention
//   key_padding_mask).  In place: S becomes P.
// --------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(T* __restrict__ S, const uint8_t* __restrict__ keymask,
                                                          const float* __restrict__ dist, const float* __restrict__ sp_w,
                                                          const float* __restrict__ sp_b, int rows, int nh, int Lq, int Lk,
                                                          int ldS, int mask_mode) {
  const int lane = threadIdx.x & 63;
  const float w = sp_w ? sp_w[0] : 0.f, b0 = sp_b ? sp_b[0] : 0.f;
  for (long row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long)gridDim.x * 4) {
    const int q = row % Lq;
    const int b = (row / Lq) / nh;
    T* s = S + row * ldS;
    const uint8_t* km = keymask ? keymask + (long)b * Lk : nullptr;
    const float* d = dist ? dist + ((long)b * Lq + q) * Lk : nullptr;
    float mx = -INFINITY;
    for (int k = lane; k < Lk; k += 64) {
      float v = Elem<T>::ld(s + k);
      if (km && !km[k]) v = mask_mode ? -INFINITY : v - 10000.0f;
      if (d) v += w * d[k] + b0;
      mx = fmaxf(mx, v);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int k = lane; k < Lk; k += 64) {
      float v = Elem<T>::ld(s + k);
      if (km && !km[k]) v = mask_mode ? -INFINITY : v - 10000.0f;
      if (d) v += w * d[k] + b0;
      sum += __expf(v - mx);
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int k = lane; k < ldS; k += 64) {
      float p = 0.f;
      if (k < Lk) {
        float v = Elem<T>::ld(s + k);
        if (km && !km[k]) v = mask_mode ? -INFINITY : v - 10000.0f;
        if (d) v += w * d[k] + b0;
        p = __expf(v - mx) * inv;
      }
      Elem<T>::st(s + k, p);
    }
  }
}

// dS = P * (dP - sum_k dP*P)  in place over dP;  d sprel_linear.{weight,bias} accumulated with atomics.
template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const T* __restrict__ P, T* __restrict__ dP,
                                                          const float* __restrict__ dist, float* __restrict__ d_sp_w,
                                                          float* __restrict__ d_sp_b, int rows, int nh, int Lq, int Lk,
                                                          int ldS) {
  __shared__ float red[2][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float aw = 0.f, ab = 0.f;
  for (long row = blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const int q = row % Lq;
    const int b = (row / Lq) / nh;
    const T* p = P + row * ldS;
    T* dp = dP + row * ldS;
    const float* d = dist ? dist + ((long)b * Lq + q) * Lk : nullptr;
    float dot = 0.f;
    for (int k = lane; k < Lk; k += 64) dot += Elem<T>::ld(p + k) * Elem<T>::ld(dp + k);
    dot = wave_sum(dot);
    for (int k = lane; k < ldS; k += 64) {
      float ds = 0.f;
      if (k < Lk) {
        ds = Elem<T>::ld(p + k) * (Elem<T>::ld(dp + k) - dot);
        if (d) { aw += ds * d[k]; ab += ds; }
      }
      Elem<T>::st(dp + k, ds);
    }
  }
  if (d_sp_w == nullptr) return;
  aw = wave_sum(aw); ab = wave_sum(ab);
  if (lane == 0) { red[0][wave] = aw; red[1][wave] = ab; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(d_sp_w, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(d_sp_b, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

int softmax_fwd(int dtype, void* S, const uint8_t* keymask, const float* dist, const float* sp_w, const float* sp_b, int B,
                int nh, int Lq, int Lk, int ldS, int mask_mode, hipStream_t st) {
  ETP_REQUIRE(B > 0 && nh > 0 && Lq > 0 && Lk > 0 && ldS >= Lk, "bad dims");
  const long rows = (long)B * nh * Lq;
  const int grid = (int)std::min<long>((rows + 3) / 4, 8192);
  if (dtype == ETP_BF16)
    ETP_LAUNCH((softmax_fwd_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, (bf16_t*)S, keymask, dist, sp_w, sp_b, (int)rows, nh, Lq, Lk, ldS, mask_mode);
  else
    ETP_LAUNCH((softmax_fwd_kernel<float>), dim3(grid), dim3(256), 0, st, (float*)S, keymask, dist, sp_w, sp_b, (int)rows, nh, Lq, Lk, ldS, mask_mode);
  ETP_CHECK_LAUNCH("softmax_fwd");
  return ETP_OK;
}

int softmax_bwd(int dtype, const void* P, void* dP, const float* dist, float* d_sp_w, float* d_sp_b, int B, int nh, int Lq,
                int Lk, int ldS, hipStream_t st) {
  ETP_REQUIRE(B > 0 && nh > 0 && Lq > 0 && Lk > 0 && ldS >= Lk, "bad dims");
  const long rows = (long)B * nh * Lq;
  const int grid = (int)std::min<long>((rows + 3) / 4, 1024);
  if (dtype == ETP_BF16)
    ETP_LAUNCH((softmax_bwd_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)P, (bf16_t*)dP, dist, d_sp_w, d_sp_b, (int)rows, nh, Lq, Lk, ldS);
  else
    ETP_LAUNCH((softmax_bwd_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)P, (float*)dP, dist, d_sp_w, d_sp_b, (int)rows, nh, Lq, Lk, ldS);
  ETP_CHECK_LAUNCH("softmax_bwd");
  return ETP_OK;
}

// dst[row, k] = src[row, k] * mask(row*Lk + k)/(1-p) for k < Lk (pad columns copied): attention-probability dropout of the
// unfused (Lq or Lk > 128) path; dst may alias src.  Element index = the fused kernels' ((b*heads+h)*Lq + q)*Lk + k.
template <typename T>
__global__ __launch_bounds__(256) void drop_rows_kernel(const T* __restrict__ src, T* __restrict__ dst, long rows, int Lk, int ldS,
                                                        Drop drop) {
  const int lane = threadIdx.x & 63;
  for (long row = blockIdx.x * 4L + (threadIdx.x >> 6); row < rows; row += (long)gridDim.x * 4) {
    const T* s = src + row * ldS;
    T* d = dst + row * ldS;
    const uint32_t base = (uint32_t)(row * Lk);
    for (int k = lane; k < ldS; k += 64) {
      float v = Elem<T>::ld(s + k);
      if (k < Lk) v *= drop_mult(drop.seed, base + k, drop.p, drop.inv_keep);
      Elem<T>::st(d + k, v);
    }
  }
}
int drop_rows(int dtype, const void* src, void* dst, long rows, int Lk, int ldS, Drop drop, hipStream_t st) {
  ETP_REQUIRE(src && dst && rows > 0 && Lk > 0 && ldS >= Lk, "bad arguments");
  const int grid = (int)std::min<long>((rows + 3) / 4, 2048);
  if (dtype == ETP_BF16)
    ETP_LAUNCH((drop_rows_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, rows, Lk, ldS, drop);
  else
    ETP_LAUNCH((drop_rows_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)src, (float*)dst, rows, Lk, ldS, drop);
  ETP_CHECK_LAUNCH("drop_rows");
  return ETP_OK;
}

}  // namespace etp
