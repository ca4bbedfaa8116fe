// round 5: the IEEE forms (__frcp_rn, __expf) against the raw-instruction forms (v_rcp_f32, v_exp_f32 through the amdgcn builtins)
// of the A&S erf-GELU, evaluated on the GPU over a range of inputs, scalar and with the polynomial on float2 vectors.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>
__device__ __forceinline__ void parts_ieee(float x, float& c, float& q) {
  const float t = __frcp_rn(fmaf(0.3275911f, fabsf(x) * 0.70710678118654752440f, 1.0f));
  q = __expf(-0.5f * x * x);
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f); p = fmaf(p, t, -0.284496736f); p = fmaf(p, t, 0.254829592f);
  c = 0.5f * (1.0f + copysignf(1.0f - p * t * q, x));
}
__device__ __forceinline__ void parts_raw(float x, float& c, float& q, float& tt, float& arg) {
  arg = (x * x) * -0.72134752044448170368f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, fabsf(x), 1.0f));
  q = __builtin_amdgcn_exp2f(arg);
  tt = t;
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f); p = fmaf(p, t, -0.284496736f); p = fmaf(p, t, 0.254829592f);
  c = 0.5f * (1.0f + copysignf(1.0f - p * t * q, x));
}
__global__ void k(const float* x, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float c0, q0, c1, q1, t1, a1;
  parts_ieee(x[i], c0, q0);
  parts_raw(x[i], c1, q1, t1, a1);
  out[6 * i] = c0; out[6 * i + 1] = q0; out[6 * i + 2] = c1; out[6 * i + 3] = q1; out[6 * i + 4] = t1; out[6 * i + 5] = a1;
}
int main() {
  const int n = 4001;
  std::vector<float> hx(n), ho(6 * n);
  for (int i = 0; i < n; ++i) hx[i] = -10.f + 20.f * i / (n - 1);
  float *dx, *dout;
  hipMalloc(&dx, n * 4); hipMalloc(&dout, 6 * n * 4);
  hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
  k<<<(n + 255) / 256, 256>>>(dx, dout, n);
  hipMemcpy(ho.data(), dout, 6 * n * 4, hipMemcpyDeviceToHost);
  double mc = 0, mq = 0; int ic = 0, iq = 0;
  for (int i = 0; i < n; ++i) {
    const double dc = fabs(ho[6 * i] - ho[6 * i + 2]), dq = fabs(ho[6 * i + 1] - ho[6 * i + 3]);
    if (dc > mc) { mc = dc; ic = i; }
    if (dq > mq) { mq = dq; iq = i; }
  }
  printf("max |cdf_ieee - cdf_raw| %.3e at x = %.4f (ieee %.7f raw %.7f, t %.7f); max |q diff| %.3e at x = %.4f (ieee %.7e raw %.7e, exp2 arg %.5f)\n",
         mc, hx[ic], ho[6 * ic], ho[6 * ic + 2], ho[6 * ic + 4], mq, hx[iq], ho[6 * iq + 1], ho[6 * iq + 3], ho[6 * iq + 5]);
  for (float xv : {-3.f, -1.f, -0.25f, 0.f, 0.5f, 2.f, 5.f}) {
    int i = (int)((xv + 10.f) / 20.f * (n - 1) + 0.5f);
    printf("x %.3f: cdf ieee %.7f raw %.7f | q ieee %.7e raw %.7e (exact %.7e)\n", hx[i], ho[6 * i], ho[6 * i + 2], ho[6 * i + 1], ho[6 * i + 3],
           exp(-0.5 * hx[i] * hx[i]));
  }
  return 0;
}
