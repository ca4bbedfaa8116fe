// Round 6: synthetic reproducer for the co-residency corruption of DESIGN.md §3.6.
//
// What the round-6 bisect showed (profiles/r06_neighbour_bisect.txt): the three pano_embed_bwd launches return different results
// beside a 128x128-tile GEMM in 24 of 24 repetitions; with the GEMM's MFMAs compiled out (LDS-DMA only) 0 of 24; with the VICTIM
// compiled without packed-fp32 instructions (-fno-slp-vectorize) 0 of 24.  The round-5 isolation had located the wrong values in
// elements 0 and 2 of a lane's four, lanes 48-63 -- the LOW halves of v_pk_*_f32 results in the last quarter of the wavefront --
// and the only packed form that kernel has and the clean row kernels lack is
//        v_pk_mul_f32 v[a:a+1], v[a:a+1], v[s:s+1] op_sel:[0,1]          ((x - mean) * rstd with (mean, rstd) in one register pair:
//                                                                           the LOW product reads the HIGH register of src1)
// This file tests that instruction (and its siblings) directly: every lane multiplies known operands with one packed instruction and
// compares the two results bit for bit with two scalar v_mul_f32 of the same operands, in a loop, while a neighbour stream runs
// MFMA-dense work on the same CUs.  Mismatches are counted per lane and per half.
//   build + run: tools/experiments/r06_pk_opsel_repro.py
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ float smul(float a, float b) {
  float r;
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <int V> __device__ __forceinline__ f2 pk(f2 a, f2 b) {
  f2 r;
  if constexpr (V == 0) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  if constexpr (V == 1) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  if constexpr (V == 2) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(a), "v"(b));
  if constexpr (V == 3) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
  if constexpr (V == 4) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  if constexpr (V == 5) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  if constexpr (V == 6) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  if constexpr (V == 7) asm volatile("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,1,0]" : "=v"(r) : "v"(a), "v"(b));      // a.lo * b.hi + a.lo | a.hi * b.hi + a.hi
  if constexpr (V == 8) asm volatile("v_pk_fma_f32 %0, %1, %1, %2 op_sel:[0,0,1]" : "=v"(r) : "v"(a), "v"(b));      // a.lo * a.lo + b.hi | a.hi * a.hi + b.hi
  if constexpr (V == 9) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1]" : "=v"(r) : "v"(a), "v"(b));
  // v_pk_mov_b32: D.lo = src0[op_sel[0]], D.hi = src1[op_sel[1]].  op_sel:[1,0] is the form the compiler emits in the mm32 GEMM epilogues
  // (132 of them in gemm_mm32.o); [0,1], [1,1] for completeness
  if constexpr (V == 10) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(a), "v"(b));
  if constexpr (V == 11) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  if constexpr (V == 12) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// expected (lo, hi) of variant V: op_sel picks the source register of the LOW product, op_sel_hi of the HIGH product (default 1)
__device__ __forceinline__ float sadd(float a, float b) {
  float r;
  asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float sfma(float a, float b, float c) {
  float r;
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
template <int V> __device__ __forceinline__ f2 expect(f2 a, f2 b) {
  f2 r;
  if constexpr (V == 6) { r.x = sadd(a.x, b.y); r.y = sadd(a.y, b.y); return r; }
  if constexpr (V == 7) { r.x = sfma(a.x, b.y, a.x); r.y = sfma(a.y, b.y, a.y); return r; }
  if constexpr (V == 8) { r.x = sfma(a.x, a.x, b.y); r.y = sfma(a.y, a.y, b.y); return r; }
  if constexpr (V == 10) { r.x = a.y; r.y = b.x; return r; }
  if constexpr (V == 11) { r.x = a.x; r.y = b.y; return r; }
  if constexpr (V == 12) { r.x = a.y; r.y = b.y; return r; }
  constexpr int sl0 = (V == 2 || V == 9) ? 1 : 0, sl1 = (V == 1 || V == 5 || V == 9) ? 1 : 0;
  constexpr int sh0 = (V == 4 || V == 5) ? 0 : 1, sh1 = (V == 3) ? 0 : 1;
  r.x = smul(sl0 ? a.y : a.x, sl1 ? b.y : b.x);
  r.y = smul(sh0 ? a.y : a.x, sh1 ? b.y : b.x);
  return r;
}

// out: [64 lanes][2 halves] mismatch counts, then [4] = {total checks lo, first bad lane, first bad got, first bad want}
template <int V>
__global__ __launch_bounds__(256) void pk_victim(unsigned* __restrict__ out, int iters, unsigned seed) {
  const unsigned tid = blockIdx.x * 256 + threadIdx.x;
  unsigned s = tid * 2654435761u + seed;
  unsigned bad_lo = 0, bad_hi = 0;
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    f2 a, b;
    a.x = __uint_as_float(0x3f800000u | (s >> 9));              // [1, 2)
    a.y = __uint_as_float(0x3f800000u | ((s * 7u) >> 9));
    b.x = __uint_as_float(0x3f800000u | ((s * 13u) >> 9));
    b.y = __uint_as_float(0x3f800000u | ((s * 29u) >> 9));
    const f2 got = pk<V>(a, b), want = expect<V>(a, b);
    const bool blo = __float_as_uint(got.x) != __float_as_uint(want.x), bhi = __float_as_uint(got.y) != __float_as_uint(want.y);
    if ((blo || bhi) && atomicAdd(out + 128, 1u) == 0) {
      out[129] = threadIdx.x & 63; out[130] = __float_as_uint(blo ? got.x : got.y); out[131] = __float_as_uint(blo ? want.x : want.y);
      out[132] = blo ? 0 : 1; out[133] = __float_as_uint(a.x); out[134] = __float_as_uint(a.y); out[135] = __float_as_uint(b.x);
      out[136] = __float_as_uint(b.y);
    }
    bad_lo += blo; bad_hi += bhi;
  }
  if (bad_lo) atomicAdd(out + 2 * (threadIdx.x & 63), bad_lo);
  if (bad_hi) atomicAdd(out + 2 * (threadIdx.x & 63) + 1, bad_hi);
}

// MFMA-dense neighbour: NACC independent 32x32x16 bf16 accumulators per wavefront (NACC * 16 registers), no memory traffic in the loop
template <int NACC>
__global__ __launch_bounds__(256) void mfma_neighbour(float* __restrict__ sink, int iters) {
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(1.0f + 0.001f * (threadIdx.x & 63)); b[e] = (__bf16)(0.5f); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) sink[0] = s;       // keeps the accumulators alive
}

extern "C" {
int pk_run(int variant, unsigned* out, int iters, int blocks, unsigned seed, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (variant) {
    case 0: hipLaunchKernelGGL(pk_victim<0>, dim3(blocks), dim3(256), 0, st, out, iters, seed); break;
    case 1: hipLaunchKernelGGL(pk_victim<1>, dim3(blocks), dim3(256), 0, st, out, iters, seed); break;
    case 2: hipLaunchKernelGGL(pk_victim<2>, dim3(blocks), dim3(256), 0, st, out, iters, seed); break;
    case 3: hipLaunchKernelGGL(pk_victim<3>, dim3(blocks), dim3(256), 0, st, out, iters, seed); break;
    case 4: hipLaunchKernelGGL(pk_victim<4>, dim3(blocks), dim3(256), 0, st, out, iters, seed); break;
    case 5: hipLaunchKernelGGL(pk_victim<5>, dim3(blocks), dim3(256), 0, st, out, iters, seed); break;
    case 6: hipLaunchKernelGGL(pk_victim<6>, dim3(blocks), dim3(256), 0, st, out, iters, seed); break;
    case 7: hipLaunchKernelGGL(pk_victim<7>, dim3(blocks), dim3(256), 0, st, out, iters, seed); break;
    case 8: hipLaunchKernelGGL(pk_victim<8>, dim3(blocks), dim3(256), 0, st, out, iters, seed); break;
    case 9: hipLaunchKernelGGL(pk_victim<9>, dim3(blocks), dim3(256), 0, st, out, iters, seed); break;
    case 10: hipLaunchKernelGGL(pk_victim<10>, dim3(blocks), dim3(256), 0, st, out, iters, seed); break;
    case 11: hipLaunchKernelGGL(pk_victim<11>, dim3(blocks), dim3(256), 0, st, out, iters, seed); break;
    case 12: hipLaunchKernelGGL(pk_victim<12>, dim3(blocks), dim3(256), 0, st, out, iters, seed); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}
int mfma_run(int nacc, float* sink, int iters, int blocks, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (nacc == 4) hipLaunchKernelGGL(mfma_neighbour<4>, dim3(blocks), dim3(256), 0, st, sink, iters);
  else if (nacc == 2) hipLaunchKernelGGL(mfma_neighbour<2>, dim3(blocks), dim3(256), 0, st, sink, iters);
  else hipLaunchKernelGGL(mfma_neighbour<1>, dim3(blocks), dim3(256), 0, st, sink, iters);
  return (int)hipGetLastError();
}
}
