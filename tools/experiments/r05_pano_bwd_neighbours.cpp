// round 5, the last seconds of GPU budget: the REAL pano_embed_bwd launches (experiment library, 12 KB LDS = shares CUs) beside the
// library's REAL GEMM kernels on a second stream, without the planner around them.  Which neighbour makes its results move?
//   usage: r05_pano_bwd_neighbours <libetp_*.so> [reps]
// neighbours: none | mm32 128x128 (bf16 2560x3072x768 NT: LDS-DMA x4 + ds_read_b64_tr_b16 + 32x32x16 MFMA) | mm32 128x64
// (2560x768x768) | gemm.hip classes (ETP_MM32=0: LDS-DMA + 16x16 MFMA) | fp32 GEMM (no bf16 path) | device-to-device copies.
// Victim inputs are fixed; reference = the launches alone.  Reported per neighbour: repetitions whose da / dd differ bitwise,
// rows, and gradients off by more than 2e-5 of their abs-max with the first columns (lane = col % 256 / 4, element = col % 4).
// build: hipcc -O2 -o r05_pano_bwd_neighbours r05_pano_bwd_neighbours.cpp -ldl
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <string>
#include "../../include/etpnav_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static uint64_t rng_s = 0x9E3779B97F4A7C15ull;
static float urand() { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return (float)((rng_s >> 11) * (1.0 / 9007199254740992.0)); }
static float nrand() { float u = urand() + 1e-7f, v = urand(); return sqrtf(-2.f * logf(u)) * cosf(6.2831853f * v); }
static uint16_t bf16(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
template <typename T> static T* dev(const std::vector<T>& h) { T* p; CK(hipMalloc(&p, h.size() * sizeof(T))); CK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return p; }

typedef int (*fwd_t)(int, const void*, const void*, const float*, const int64_t*, const float* const*, float*, float*, int, int, void*);
typedef int (*bwd_t)(int, const float*, const void*, const void*, const float*, const int64_t*, const float*, const float* const*, float* const*, void*, void*, int, int, void*);
typedef int (*gemm_t)(const etp_gemm_desc*, void*);
typedef const char* (*err_t)();

int main(int argc, char** argv) {
  if (argc < 2) { printf("usage: %s lib.so [reps]\n", argv[0]); return 2; }
  const int reps = argc > 2 ? atoi(argv[2]) : 24;
  void* L = dlopen(argv[1], RTLD_NOW);
  if (!L) { printf("dlopen: %s\n", dlerror()); return 1; }
  fwd_t fwd = (fwd_t)dlsym(L, "etp_pano_embed_fwd"); bwd_t bwd = (bwd_t)dlsym(L, "etp_pano_embed_bwd");
  gemm_t gemm = (gemm_t)dlsym(L, "etp_gemm"); err_t lasterr = (err_t)dlsym(L, "etp_last_error");
  if (!fwd || !bwd || !gemm) { printf("missing symbols\n"); return 1; }
  setenv("ETP_PANO_BWD_LDS", "12288", 1);
  const int M = 1152, H = 768;
  std::vector<uint16_t> a((size_t)M * H), d((size_t)M * H);
  std::vector<float> loc((size_t)M * 4), dy((size_t)M * H);
  std::vector<int64_t> nav(M);
  for (auto& x : a) x = bf16(nrand()); for (auto& x : d) x = bf16(nrand());
  for (int r = 0; r < M; ++r) { float h = 6.2831853f * (r % 12) / 12.f, e = ((r / 12) % 3 - 1) * 0.5236f; loc[r * 4] = sinf(h); loc[r * 4 + 1] = cosf(h); loc[r * 4 + 2] = sinf(e); loc[r * 4 + 3] = cosf(e); nav[r] = (r % 36) < 4; }
  for (auto& x : dy) x = 0.01f * nrand();
  const int psz[12] = {H, H, H, H, 4 * H, H, H, H, 2 * H, H, H, H};
  const char* pname[12] = {"g_img", "b_img", "g_dep", "b_dep", "w_loc", "bias_loc", "g_loc", "b_loc", "nav_emb", "type1", "g_out", "b_out"};
  const float* params[12]; float* grads[12];
  for (int i = 0; i < 12; ++i) {
    std::vector<float> v(psz[i]);
    const bool gamma = (i == 0 || i == 2 || i == 6 || i == 10);
    for (auto& x : v) x = gamma ? 1.f + 0.1f * nrand() : (i == 4 ? 0.3f * nrand() : 0.02f * nrand());
    params[i] = dev(v);
    CK(hipMalloc(&grads[i], psz[i] * 4));
  }
  uint16_t *da_, *dd_; float *y, *stats;
  CK(hipMalloc(&da_, (size_t)M * H * 2)); CK(hipMalloc(&dd_, (size_t)M * H * 2)); CK(hipMalloc(&y, (size_t)M * H * 4)); CK(hipMalloc(&stats, (size_t)M * 8 * 4));
  uint16_t *A = dev(a), *D = dev(d); float *LOC = dev(loc), *DY = dev(dy); int64_t* NAV = dev(nav);
  hipStream_t s1, s2; int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, hi)); CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, lo));
  if (fwd(ETP_BF16, A, D, LOC, NAV, params, y, stats, M, H, s2)) { printf("fwd: %s\n", lasterr ? lasterr() : "?"); return 1; }
  CK(hipDeviceSynchronize());
  // neighbour operands (bf16 2560 x 3072 x 768 and friends; fp32 variant)
  const int GM = 2560, GN = 3072, GK = 768;
  std::vector<uint16_t> ga((size_t)GM * GN), gb((size_t)GN * GN);      // big enough for every shape below
  for (auto& x : ga) x = bf16(nrand()); for (auto& x : gb) x = bf16(0.05f * nrand());
  uint16_t *GA = dev(ga), *GB = dev(gb); void* GC; CK(hipMalloc(&GC, (size_t)GM * GN * 4));
  std::vector<float> fa((size_t)GM * GK), fb((size_t)GN * GK); for (auto& x : fa) x = nrand(); for (auto& x : fb) x = 0.05f * nrand();
  float *FA = dev(fa), *FB = dev(fb);
  void *cp0, *cp1; CK(hipMalloc(&cp0, 64 << 20)); CK(hipMalloc(&cp1, 64 << 20));
  auto desc = [&](int m, int n, int k, int dt, int cdt, const void* pa, const void* pb) {
    etp_gemm_desc g; memset(&g, 0, sizeof(g));
    g.A = pa; g.B = pb; g.C = GC; g.M = m; g.N = n; g.K = k; g.lda = k; g.ldb = k; g.ldc = n; g.trans_a = 0; g.trans_b = 0;
    g.dtype = dt; g.c_dtype = cdt; g.batch = 1; g.batch_inner = 1; g.ksplit = 1; g.alpha = 1.f; return g; };
  struct Nb { const char* name; int kind; };
  const Nb nbs[] = {{"none", 0}, {"mm32 128x128 (bf16 2560x3072x768 NT)", 1}, {"mm32 128x64 (bf16 2560x768x768 NT)", 2},
                    {"gemm.hip kernels, same shapes (ETP_MM32=0)", 3}, {"fp32 GEMM 2560x768x768", 4}, {"device-to-device copies", 5}, {"none (again)", 0}};
  std::vector<uint16_t> ref_da((size_t)M * H), ref_dd((size_t)M * H), h_da((size_t)M * H), h_dd((size_t)M * H);
  std::vector<std::vector<float>> ref_g(12), h_g(12);
  auto victim = [&]() {
    for (int i = 0; i < 12; ++i) CK(hipMemsetAsync(grads[i], 0, psz[i] * 4, s2));
    if (bwd(ETP_BF16, DY, A, D, LOC, NAV, stats, params, grads, da_, dd_, M, H, s2)) { printf("bwd: %s\n", lasterr ? lasterr() : "?"); exit(1); }
  };
  auto fetch = [&](std::vector<uint16_t>& xa, std::vector<uint16_t>& xd, std::vector<std::vector<float>>& g) {
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(xa.data(), da_, xa.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(xd.data(), dd_, xd.size() * 2, hipMemcpyDeviceToHost));
    for (int i = 0; i < 12; ++i) { g[i].resize(psz[i]); CK(hipMemcpy(g[i].data(), grads[i], psz[i] * 4, hipMemcpyDeviceToHost)); }
  };
  victim(); fetch(ref_da, ref_dd, ref_g);
  for (const Nb& nb : nbs) {
    if (nb.kind == 3) setenv("ETP_MM32", "0", 1); else unsetenv("ETP_MM32");
    int reps_bad = 0, rows_max = 0, g_bad[12] = {0}; float g_worst[12] = {0}; std::string pat[12];
    for (int r = 0; r < reps; ++r) {
      for (int j = 0; j < 24 && nb.kind; ++j) {
        if (nb.kind == 1 || nb.kind == 3) { etp_gemm_desc g = desc(GM, GN, GK, ETP_BF16, ETP_BF16, GA, GB); if (gemm(&g, s1)) { printf("gemm: %s\n", lasterr()); return 1; } }
        if (nb.kind == 2 || nb.kind == 3) { etp_gemm_desc g = desc(GM, GK, GK, ETP_BF16, ETP_BF16, GA, GB); if (gemm(&g, s1)) { printf("gemm: %s\n", lasterr()); return 1; } }
        if (nb.kind == 4) { etp_gemm_desc g = desc(GM, GK, GK, ETP_F32, ETP_F32, FA, FB); if (gemm(&g, s1)) { printf("gemm: %s\n", lasterr()); return 1; } }
        if (nb.kind == 5) CK(hipMemcpyAsync(cp1, cp0, 64 << 20, hipMemcpyDeviceToDevice, s1));
        if (j == 2) victim();                      // the victim goes out while the neighbour stream is busy and stays busy
      }
      if (!nb.kind) victim();
      fetch(h_da, h_dd, h_g);
      int rows = 0;
      for (int row = 0; row < M; ++row)
        if (memcmp(&h_da[(size_t)row * H], &ref_da[(size_t)row * H], H * 2) || memcmp(&h_dd[(size_t)row * H], &ref_dd[(size_t)row * H], H * 2)) ++rows;
      reps_bad += rows > 0; rows_max = rows > rows_max ? rows : rows_max;
      for (int i = 0; i < 12; ++i) {
        float amax = 1e-20f, worst = 0; for (float v : ref_g[i]) amax = fmaxf(amax, fabsf(v));
        std::string cols; int nc = 0;
        for (int c = 0; c < psz[i]; ++c) { float e = fabsf(h_g[i][c] - ref_g[i][c]) / amax; worst = fmaxf(worst, e);
          if (e > 2e-5f && nc++ < 6) cols += std::to_string(c) + " "; }
        if (worst > 2e-5f) { ++g_bad[i]; if (worst > g_worst[i]) { g_worst[i] = worst; pat[i] = cols + "(" + std::to_string(nc) + " columns)"; } }
      }
    }
    printf("neighbour %-48s: da/dd differ in %d of %d repetitions (max %d rows)", nb.name, reps_bad, reps, rows_max);
    for (int i = 0; i < 12; ++i) if (g_bad[i]) printf(" | d %s %d reps, worst %.1e, cols %s", pname[i], g_bad[i], g_worst[i], pat[i].c_str());
    printf("\n"); fflush(stdout);
  }
  return 0;
}
