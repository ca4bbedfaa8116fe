// L2 -> CU feed microbenchmark (round-4 experiment): how many bytes per clock can one CU pull in GEMM-shaped access
// (128-byte row segments of a row-major bf16 matrix, rows `ld` apart) through
//   mode 0: LDS-DMA (global_load_lds_dwordx4, SADDR form), counted vmcnt, ring of 2 slabs
//   mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128 (register staging, next slab's loads in flight under this slab's writes)
//   mode 2: global_load_dwordx4 -> VGPR only (xor-accumulated), no LDS
// Each workgroup (256 threads) reads the A panel of its row tile (128 rows) and the B panel of its column tile (128 rows) over
// K/64 slabs exactly like the 128x128 GEMM tile does: 32 KiB per slab.
//   hipcc --offload-arch=gfx950 -O3 feed_bench.hip -o feed_bench && ./feed_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ void glds(unsigned voff, const char* sbase, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void feed(const char* A, const char* B, long lda, long ldb, int tiles_n, int nk, unsigned* sink, int map, int tiles_m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bid = blockIdx.x;
  int tm = bid / tiles_n, tn = bid % tiles_n;
  if (map == 1) { tm = 0; tn = 0; }                                   // every workgroup reads the same panels: cache-hit ceiling
  if (map == 2) {                                                     // workgroups b and b + 256 (assumed co-resident) share the B panel
    const int p = bid & 255, h = bid >> 8;
    tn = p % tiles_n; tm = (2 * (p / tiles_n) + h) % tiles_m;
  }
  if (map == 3) {                                                     // XCD-contiguous chunks (workgroup b runs on XCD b % 8)
    const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3;
    const int id = xcd * q + (bid >> 3);
    tm = id / tiles_n; tn = id % tiles_n;
  }
  if (map == 4) {                                                     // XCD chunks + co-resident pairs share the B panel inside the chunk
    const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3;            // q tiles per XCD; local index l = bid >> 3
    const int l = bid >> 3, half = q >> 1;
    const int ll = l < half ? 2 * l : 2 * (l - half) + 1;             // l and l + half -> adjacent tiles of the chunk (column-major below)
    const int id = xcd * q + ll;
    tn = id / tiles_m; tm = id % tiles_m;
  }
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  // piece p = 4j + wave: rows 8p .. 8p+7, lane -> row 8p + (lane>>3), chunk lane&7 (no swizzle needed here)
  const int lr = 8 * wave + (lane >> 3);
  const unsigned voa = (unsigned)(lr * lda * 2 + (lane & 7) * 16);
  const unsigned vob = (unsigned)(lr * ldb * 2 + (lane & 7) * 16);
  const char* sa = A + (long)tm * 128 * lda * 2;
  const char* sb = B + (long)tn * 128 * ldb * 2;
  const long psa = 32 * lda * 2, psb = 32 * ldb * 2;
  unsigned acc = 0;
  if constexpr (MODE == 0) {
    for (int t = 0; t < nk; ++t) {
      const unsigned slot = lds0 + (t & 1) * 32768 + wave * 1024;
#pragma unroll
      for (int j = 0; j < 4; ++j) glds(voa, sa + j * psa, slot + j * 4096);
#pragma unroll
      for (int j = 0; j < 4; ++j) glds(vob, sb + j * psb, slot + 16384 + j * 4096);
      sa += 128; sb += 128;
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // the previous slab landed
      __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    u32x4 r[8], q[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      r[j] = *reinterpret_cast<const u32x4*>(sa + j * psa + voa);
      r[4 + j] = *reinterpret_cast<const u32x4*>(sb + j * psb + vob);
    }
    for (int t = 0; t < nk; ++t) {
      sa += 128; sb += 128;
      if (t + 1 < nk) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          q[j] = *reinterpret_cast<const u32x4*>(sa + j * psa + voa);
          q[4 + j] = *reinterpret_cast<const u32x4*>(sb + j * psb + vob);
        }
      }
      if constexpr (MODE == 1) {
        char* slot = smem + (t & 1) * 32768 + wave * 1024 + lane * 16;
#pragma unroll
        for (int j = 0; j < 8; ++j) *reinterpret_cast<u32x4*>(slot + j * 4096) = r[j];
        __builtin_amdgcn_s_barrier();
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= r[j].x ^ r[j].y ^ r[j].z ^ r[j].w;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = q[j];
    }
  }
  if (MODE != 2) acc = *reinterpret_cast<unsigned*>(smem + tid * 4);
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE> float run(const char* A, const char* B, long lda, long ldb, int tiles_m, int tiles_n, int nk, unsigned* sink, int reps, int map, int grid = 0) {
  if (!grid) grid = tiles_m * tiles_n;
  const int smem = 65536;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(feed<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(feed<MODE>, dim3(grid), dim3(256), smem, 0, A, B, lda, ldb, tiles_n, nk, sink, map, tiles_m);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(feed<MODE>, dim3(grid), dim3(256), smem, 0, A, B, lda, ldb, tiles_n, nk, sink, map, tiles_m);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps * 1e3f;
}

int main() {
  // FFN-up shape: A [2560 x 768] bf16, B [3072 x 768] bf16, 20 x 24 = 480 tiles, 12 slabs;  and a long-K shape: K = 3072
  char *A, *B; unsigned* sink;
  CK(hipMalloc(&A, 8192L * 3072 * 2)); CK(hipMalloc(&B, 3072L * 3072 * 2)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(A, 1, 8192L * 3072 * 2)); CK(hipMemset(B, 2, 3072L * 3072 * 2));
  struct Case { const char* name; int tm, tn; long K; } cases[] = {
      {"2560x3072 K=768  (480 WGs, 2/CU)", 20, 24, 768}, {"2560x1536 K=3072 (240 WGs, 1/CU)", 20, 12, 3072},
      {"2560x3072 K=3072 (480 WGs, 2/CU)", 20, 24, 3072}, {"8192x3072 K=768 (1536 WGs)", 64, 24, 768}};
  for (auto& c : cases) {
    const int nk = (int)(c.K / 64);
    for (int map = 0; map <= 4; ++map) {
      const int grid = map == 2 ? 512 : c.tm * c.tn;
      if (map == 2 && c.tm * c.tn > 512) continue;
      const double bytes = (double)grid * nk * 32768.0;
      float t0 = run<0>(A, B, c.K, c.K, c.tm, c.tn, nk, sink, 20, map, grid);
      float t2 = run<2>(A, B, c.K, c.K, c.tm, c.tn, nk, sink, 20, map, grid);
      printf("%-36s map %d  lds-dma %7.2f us %6.2f TB/s | reg only %7.2f us %6.2f TB/s\n", c.name, map, t0, bytes / t0 / 1e6, t2,
             bytes / t2 / 1e6);
    }
  }
  return 0;
}
