// Fused AdamW over the planner's flat fp32 arenas ("closes the step on device").
//
// One pass over [params | grads | exp_avg | exp_avg_sq] does what the reference spreads over torch's multi-tensor
// AdamW (ss_trainer_ETP.py:213,505), GradScaler's unscale/inf-check (:463,504-506), clip_grad_norm_ (pre-training), the
// next forward's autocast weight casts and optimizer.zero_grad():
//   g      = grad * grad_scale * clip_coef                      (clip_coef from the device-side squared norm)
//   m, v   = EMA updates                                        (torch.optim.AdamW / pretrain optim/adamw.py:88-92)
//   p      = decoupled-weight-decay Adam update, in either of the reference's two formulations (below)
//   shadow = bf16(p) for the GEMM-weight region                 (replaces etp_planner_refresh_weights)
//   grad   = 0                                                  (replaces the gradient memset of the next step)
// HBM-bound: 16 B read + 12..18 B written per parameter; nothing else to optimise but the access pattern (16-byte
// vectors, grid-stride, every array touched exactly once).
#include "kernels.h"

namespace etp {

struct AdamwK {
  float lr, beta1, beta2, eps, wd;
  float bc1, bc2_sqrt;        // 1 - beta1^t, sqrt(1 - beta2^t)   (1, 1 when bias correction is off)
  int hf_style;               // 0: torch.optim.AdamW   1: pretrain_src/optim/adamw.py
  float grad_scale;           // multiplies the raw gradient (1/S of a GradScaler; 1/world for a summed all-reduce)
  float max_norm;             // > 0: clip to this global L2 norm using *sumsq (torch.nn.utils.clip_grad_norm_)
};

typedef float f4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ntload(const float* p) {
  const f4_t x = __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(p));
  return make_float4(x[0], x[1], x[2], x[3]);
}
__device__ __forceinline__ void ntstore(float* p, float4 v) {
  f4_t x = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(x, reinterpret_cast<f4_t*>(p));
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ shadow, long n_shadow,
                                                    const uint8_t* __restrict__ decay_mask, long n, AdamwK k,
                                                    const float* __restrict__ sumsq, const int32_t* __restrict__ skip,
                                                    int zero_grads, const int32_t* __restrict__ step_dev) {
  const bool skipped = skip != nullptr && skip[0] != 0;      // GradScaler: non-finite gradients -> no update this step
  if (step_dev != nullptr && k.bc1 < 0.f) {
    // device-side step counter (GradScaler semantics: a skipped step does not advance optimizer.state['step']): the bias
    // corrections follow the number of APPLIED updates, already bumped by adamw_bump_kernel for this call
    const float t = (float)step_dev[0];
    k.bc1 = 1.0f - powf(k.beta1, t);
    k.bc2_sqrt = sqrtf(1.0f - powf(k.beta2, t));
  }
  float gs = k.grad_scale;
  if (k.max_norm > 0.f && sumsq != nullptr) {
    const float norm = sqrtf(sumsq[0]) * fabsf(k.grad_scale);
    gs *= fminf(1.0f, k.max_norm / (norm + 1e-6f));          // clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to 1
  }
  const long n4 = n / 4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long e0 = i * 4;
    // every array is streamed exactly once: non-temporal accesses keep 4.7 GB of optimizer traffic out of L2/MALL
    float4 gv = ntload(g + e0);
    // mask byte of this 64-element block: bit 0 = weight decay applies, bit 1 = FROZEN (requires_grad = False:
    // vilmodel_cmt.py:675-682 fix_lang_embedding / fix_pano_embedding) -- p / m / v / shadow are left alone, the gradient is
    // still zeroed
    // Values (header changelog): 0 no decay, 1 decay, 2 frozen, 3 frozen (decay bit irrelevant); every OTHER non-zero
    // byte keeps the pre-round-5 meaning "decay" (callers that pass 0xFF / true-as-255), never "frozen"
    const uint8_t mraw = decay_mask == nullptr ? (uint8_t)1 : decay_mask[e0 >> 6];
    const uint8_t mb = mraw > 3 ? (uint8_t)1 : mraw;
    if (!skipped && !(mb & 2)) {
      float4 pv = ntload(p + e0);
      float4 mv = ntload(m + e0);
      float4 vv = ntload(v + e0);
      const float wd = (mb & 1) ? k.wd : 0.f;
      float pp[4] = {pv.x, pv.y, pv.z, pv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w};
      float mm[4] = {mv.x, mv.y, mv.z, mv.w}, vq[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gr = gg[e] * gs;
        mm[e] = k.beta1 * mm[e] + (1.0f - k.beta1) * gr;
        vq[e] = k.beta2 * vq[e] + (1.0f - k.beta2) * gr * gr;
        if (k.hf_style) {     // optim/adamw.py:93-112: denom = sqrt(v)+eps; step = lr*sqrt(bc2)/bc1; decay AFTER the update
          const float upd = pp[e] - (k.lr * k.bc2_sqrt / k.bc1) * mm[e] / (sqrtf(vq[e]) + k.eps);
          pp[e] = upd - k.lr * wd * upd;
        } else {              // torch.optim.AdamW: decay first, denom = sqrt(v)/sqrt(bc2) + eps, step = lr/bc1
          const float dec = pp[e] * (1.0f - k.lr * wd);
          pp[e] = dec - (k.lr / k.bc1) * mm[e] / (sqrtf(vq[e]) / k.bc2_sqrt + k.eps);
        }
      }
      ntstore(p + e0, make_float4(pp[0], pp[1], pp[2], pp[3]));
      ntstore(m + e0, make_float4(mm[0], mm[1], mm[2], mm[3]));
      ntstore(v + e0, make_float4(vq[0], vq[1], vq[2], vq[3]));
      if (shadow != nullptr && e0 < n_shadow) store4(shadow + e0, pp);
    }
    if (zero_grads) ntstore(g + e0, make_float4(0.f, 0.f, 0.f, 0.f));
  }
}

__global__ void adamw_bump_kernel(int32_t* __restrict__ step_dev, const int32_t* __restrict__ skip) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && !(skip != nullptr && skip[0] != 0)) step_dev[0] += 1;
}

// sum of squares (+ count of non-finite values) of a gradient arena; both outputs ACCUMULATE (zero them first)
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, long n, float* __restrict__ sumsq,
                                                     int32_t* __restrict__ nonfinite, const uint8_t* __restrict__ mask) {
  __shared__ float red[4];
  __shared__ int bad[4];
  float s = 0.f;
  int nf = 0;
  const long n4 = n / 4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    if (mask != nullptr && (mask[(i * 4) >> 6] == 2 || mask[(i * 4) >> 6] == 3)) continue;      // frozen block: no .grad in the reference, not in the norm
    const float4 x = *reinterpret_cast<const float4*>(g + i * 4);
    s += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
    nf += !isfinite(x.x) + !isfinite(x.y) + !isfinite(x.z) + !isfinite(x.w);
  }
  s = wave_sum(s);
  for (int o = 32; o > 0; o >>= 1) nf += __shfl_xor(nf, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[wave] = s; bad[wave] = nf; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(sumsq, red[0] + red[1] + red[2] + red[3]);
    const int b = bad[0] + bad[1] + bad[2] + bad[3];
    if (nonfinite != nullptr && b) atomicAdd(nonfinite, b);
  }
}

int adamw_step(float* p, float* g, float* m, float* v, void* shadow, long n_shadow, const uint8_t* decay_mask, long n,
               const etp_adamw_cfg& c, const float* sumsq, const int32_t* skip, int zero_grads, hipStream_t st,
               int32_t* step_dev) {
  ETP_REQUIRE(p && g && m && v && n > 0 && n % 4 == 0, "arena pointers / length (multiple of 4) required");
  ETP_REQUIRE(((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0, "arenas must be 16-byte aligned");
  ETP_REQUIRE((c.step >= 1 || step_dev != nullptr) && c.beta1 >= 0.f && c.beta1 < 1.f && c.beta2 >= 0.f && c.beta2 < 1.f && c.eps >= 0.f, "bad hyper-parameters");
  ETP_REQUIRE(shadow == nullptr || (n_shadow >= 0 && n_shadow <= n && n_shadow % 4 == 0 && (uintptr_t)shadow % 8 == 0), "bad shadow region");
  AdamwK k;
  k.lr = c.lr; k.beta1 = c.beta1; k.beta2 = c.beta2; k.eps = c.eps; k.wd = c.weight_decay;
  k.hf_style = c.hf_style; k.grad_scale = c.grad_scale; k.max_norm = c.max_norm;
  if (c.correct_bias && step_dev != nullptr) {
    k.bc1 = -1.f; k.bc2_sqrt = 1.f;           // computed in the kernel from the device counter
    ETP_LAUNCH(adamw_bump_kernel, dim3(1), dim3(64), 0, st, step_dev, skip);
  } else if (c.correct_bias) {
    k.bc1 = (float)(1.0 - pow((double)c.beta1, (double)c.step));
    k.bc2_sqrt = (float)sqrt(1.0 - pow((double)c.beta2, (double)c.step));
  } else {
    k.bc1 = 1.f; k.bc2_sqrt = 1.f;
  }
  const int grid = (int)std::min<long>((n / 4 + 255) / 256, 256L * 16);
  if (!c.correct_bias && step_dev != nullptr) ETP_LAUNCH(adamw_bump_kernel, dim3(1), dim3(64), 0, st, step_dev, skip);
  ETP_LAUNCH(adamw_kernel, dim3(grid), dim3(256), 0, st, p, g, m, v, (bf16_t*)shadow, n_shadow, decay_mask, n, k, sumsq,
                     skip, zero_grads, (const int32_t*)step_dev);
  ETP_CHECK_LAUNCH("adamw");
  return ETP_OK;
}

int grad_sqnorm(const float* g, long n, float* sumsq, int32_t* nonfinite, hipStream_t st, const uint8_t* mask) {
  ETP_REQUIRE(g && sumsq && n > 0 && n % 4 == 0 && (uintptr_t)g % 16 == 0, "bad arguments");
  const int grid = (int)std::min<long>((n / 4 + 255) / 256, 256L * 8);
  ETP_LAUNCH(sqnorm_kernel, dim3(grid), dim3(256), 0, st, g, n, sumsq, nonfinite, mask);
  ETP_CHECK_LAUNCH("grad_sqnorm");
  return ETP_OK;
}

}  // namespace etp
