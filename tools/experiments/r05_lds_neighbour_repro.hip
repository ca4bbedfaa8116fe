// round 5: reproducer for the cross-workgroup LDS corruption behind DESIGN.md §3.6 (round 4's "AGPR" corruption) and this round's
// pano_embed_bwd finding: a kernel that keeps long-lived, lane-private data in a LARGE dynamic LDS allocation returns wrong sums when
// workgroups of OTHER kernels share its CU, and is exact when it owns the CU's LDS (profiles/r05_pano_embed_race.txt; this reproducer came out NEGATIVE: 0 corrupted runs in every cell).
//
//   victim    256 threads, `vbytes` of dynamic LDS: every lane read-modify-writes (+1) its own 16-byte slots ITER times, then counts the
//             slots that do not hold ITER.  No barrier, no sharing between lanes: any mismatch was written by somebody else.
//   aggressor 256 threads, `abytes` of dynamic LDS on a second stream, one of
//               0  none
//               1  ds_write_b128 of a poison pattern over its own allocation, in a loop
//               2  LDS-DMA (global_load_lds_dwordx4, M0 = LDS byte address) of a poison buffer over its own allocation, in a loop
//
// build: hipcc --offload-arch=gfx950 -O3 -o r05_lds_neighbour_repro r05_lds_neighbour_repro.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void victim(int slots_per_lane, int iters, unsigned* bad, unsigned* first_bad) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float4* mine = reinterpret_cast<float4*>(lds) + threadIdx.x;             // slot s of this lane: mine[s * 256]
  for (int s = 0; s < slots_per_lane; ++s) mine[s * 256] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int it = 0; it < iters; ++it)
    for (int s = 0; s < slots_per_lane; ++s) {
      float4 v = mine[s * 256];
      v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
      mine[s * 256] = v;
      asm volatile("" ::: "memory");
    }
  unsigned nb = 0;
  const float want = (float)iters;
  for (int s = 0; s < slots_per_lane; ++s) {
    const float4 v = mine[s * 256];
    if (v.x != want || v.y != want || v.z != want || v.w != want) {
      if (nb == 0) atomicMin(first_bad, (unsigned)((s * 256 + threadIdx.x) * 16));
      ++nb;
    }
  }
  if (nb) atomicAdd(bad, nb);
}

__global__ __launch_bounds__(256) void aggressor(int mode, int bytes, int iters, const float4* poison, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int n16 = bytes / 16;
  if (mode == 1) {
    float4* p = reinterpret_cast<float4*>(lds);
    for (int it = 0; it < iters; ++it)
      for (int i = threadIdx.x; i < n16; i += 256) { p[i] = make_float4(1e30f, 1e30f, 1e30f, 1e30f); asm volatile("" ::: "memory"); }
  } else if (mode == 2) {
    const unsigned base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)lds);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int it = 0; it < iters; ++it)
      for (int piece = wave; piece * 1024 < bytes; piece += 4) {          // one 1-KiB piece per wavefront and trip (64 lanes x 16 B)
        const unsigned lds_addr = __builtin_amdgcn_readfirstlane(base + (unsigned)piece * 1024u);
        const unsigned voff = (unsigned)((threadIdx.x & 63) * 16);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(poison), "s"(lds_addr) : "memory");
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(sink, (unsigned)lds[0]);
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 40;
  unsigned *bad, *first, *sink;
  float4* poison;
  CHECK(hipMalloc(&bad, 4)); CHECK(hipMalloc(&first, 4)); CHECK(hipMalloc(&sink, 4));
  CHECK(hipMalloc(&poison, 1 << 20));
  CHECK(hipMemset(poison, 0x7f, 1 << 20));                                  // 0x7f7f7f7f = 3.39e38: unmistakable in a float sum
  hipStream_t s1, s2;
  CHECK(hipStreamCreate(&s1)); CHECK(hipStreamCreate(&s2));
  const int vsizes[] = {12 * 1024, 48 * 1024, 108 * 1024, 160 * 1024};
  const int asizes[] = {16 * 1024, 48 * 1024};
  for (int vb : vsizes) {
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(victim), hipFuncAttributeMaxDynamicSharedMemorySize, vb));
    for (int mode = 0; mode <= 2; ++mode)
      for (int ab : asizes) {
        if (mode == 0 && ab != asizes[0]) continue;
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(aggressor), hipFuncAttributeMaxDynamicSharedMemorySize, ab));
        unsigned total = 0, runs_bad = 0, first_h = 0xffffffffu;
        for (int r = 0; r < reps; ++r) {
          CHECK(hipMemsetAsync(bad, 0, 4, s1));
          CHECK(hipMemsetAsync(first, 0xff, 4, s1));
          CHECK(hipStreamSynchronize(s1));
          if (mode) hipLaunchKernelGGL(aggressor, dim3(1024), dim3(256), ab, s2, mode, ab, 40, poison, sink);
          hipLaunchKernelGGL(victim, dim3(256), dim3(256), vb, s1, vb / 16 / 256, 300, bad, first);
          if (mode) hipLaunchKernelGGL(aggressor, dim3(1024), dim3(256), ab, s2, mode, ab, 40, poison, sink);
          CHECK(hipDeviceSynchronize());
          unsigned h, f;
          CHECK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
          CHECK(hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost));
          total += h; runs_bad += h != 0;
          if (h && f < first_h) first_h = f;
        }
        printf("victim LDS %6d B, aggressor mode %d (%s) LDS %6d B: %u of %d runs corrupted, %u bad slots, lowest bad byte offset %d\n", vb, mode,
               mode == 0 ? "none" : mode == 1 ? "ds_write" : "LDS-DMA", mode ? ab : 0, runs_bad, reps, total, first_h == 0xffffffffu ? -1 : (int)first_h);
      }
  }
  return 0;
}
