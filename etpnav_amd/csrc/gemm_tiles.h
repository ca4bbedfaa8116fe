// Tile-level building blocks of the 16x16x32 (bf16) / 16x16x4 (fp32) GEMM family of gemm.hip, in a header so that other
// translation units (prototypes under tools/experiments, future fused kernels) can build on the same LDS layouts:
// MFMA step, LDS geometry of an operand tile and its swizzles, register-staged tile load / store, fragment reads, and the
// LDS-DMA plan / issue / counted waits of the global_load_lds ring.  Moved verbatim from gemm.hip (round 4): the device code
// of gemm.hip is unchanged by the move (checked on the generated ISA).
#pragma once
#include "gemm_shared.h"

namespace etp {

template <typename T> struct MmaTraits;
template <> struct MmaTraits<bf16_t> {
  static constexpr int BK = 64;      // 128 B / 2
  static constexpr int EPC = 8;      // elements per 16-B chunk
};
template <> struct MmaTraits<float> {
  static constexpr int BK = 32;
  static constexpr int EPC = 4;
};

template <typename T> struct Frag;          // 8 k-values of one row/col for one MFMA step
template <> struct Frag<bf16_t> { uint4 v; };
template <> struct Frag<float> { float4 lo, hi; };

__device__ __forceinline__ void mma_step(f32x4_t& acc, const Frag<bf16_t>& a, const Frag<bf16_t>& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a.v), __builtin_bit_cast(bf16x8_t, b.v),
                                                acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_step(f32x4_t& acc, const Frag<float>& a, const Frag<float>& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.x, b.lo.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.y, b.lo.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.z, b.lo.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.w, b.lo.w, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.x, b.hi.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.y, b.hi.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.z, b.hi.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.w, b.hi.w, acc, 0, 0, 0);
}

// ---- LDS geometry of one operand tile -------------------------------------------------
template <typename T, bool TR, int ROWS, int PAD = 32> struct TileGeom {
  static constexpr int BK = MmaTraits<T>::BK;
  static constexpr int EPC = MmaTraits<T>::EPC;
  static constexpr int PITCH = TR ? (ROWS * (int)sizeof(T) + PAD) : 128;     // bytes
  static constexpr int BYTES = TR ? BK * PITCH : ROWS * 128;
  static constexpr int CHUNKS = ROWS * 8;                                    // 16-B chunks per tile (both layouts)
  static constexpr int PER_THREAD = CHUNKS / 256;
  static constexpr int CPR = TR ? ROWS / EPC : 8;                            // chunks per LDS row
  static_assert(CHUNKS % 256 == 0, "tile too small for 256 threads");
};

template <int N> struct Regs { uint4 v[N]; unsigned okmask; };

// 16-byte-chunk XOR swizzle of unpadded transposed tiles ([k][rows], LDS-DMA layout).  One ds_read_b64_tr_b16 is served in two
// groups of 32 lanes (MI355X_MICROARCH.md, LDS); with frag_load's lane -> address map a group reads EIGHT k-rows -- kr0 + {0..3} from
// its first 16 lanes, kr0 + 8 + {0..3} from the other 16 (the second read of the pair: + 4) -- 32 contiguous bytes (two chunks) each,
// all at the same column offset.  Bank of a byte address: (a / 4) mod 64.
//   * 256-byte rows (128-row tiles, CPR = 16): every k-row starts at bank 0; the eight rows need eight different chunk pairs:
//     pair index ^= (k & 3) | ((k >> 3) & 1) << 2.
//   * 128-byte rows (64-row tiles, CPR = 8): even k-rows start at bank 0, odd ones at bank 32, so the four rows of one parity --
//     k, k + 2, k + 8, k + 10 -- need four different chunk pairs: pair index ^= ((k >> 1) & 1) | ((k >> 3) & 1) << 1.
//     Round 6: until round 5 this layout used (k & 3), under which k and k + 8 (and k + 2, k + 10) share their
//     banks -- the 33-40 % SQ_LDS_BANK_CONFLICT of the NN 32x64 / 64x64 classes (profiles/r05_gemm_counters.txt).
template <int CPR> __device__ __forceinline__ int tr_swz(int krow) {
  if constexpr (CPR >= 16) return ((krow & 3) << 1) ^ (((krow >> 3) & 1) << 3);
  else return ((((krow >> 1) & 1) | (((krow >> 3) & 1) << 1)) << 1);
}

// global -> registers for one tile (zero-filled outside [rows_total) x [k_end)).
// Branch-free on purpose: a load inside a per-element `if` makes hipcc wait vmcnt(0) per element (one full
// memory latency each); here every load is issued unconditionally from a clamped in-bounds address and the
// out-of-range ones are zeroed with v_cndmask when they are written to LDS (after the MFMA block, so the
// s_waitcnt for them sits behind the math).
template <typename T, bool TR, int ROWS>
__device__ __forceinline__ void tile_load(Regs<TileGeom<T, TR, ROWS>::PER_THREAD>& r, const T* __restrict__ base, long ld,
                                          int row0, int rows_total, int k0, int k_end, int tid) {
  using G = TileGeom<T, TR, ROWS>;
  unsigned okmask = 0;
#pragma unroll
  for (int j = 0; j < G::PER_THREAD; ++j) {
    const int q = tid + j * 256;
    const int lr = q / G::CPR, c = q % G::CPR;
    bool ok;
    const T* src;
    if constexpr (!TR) {
      const int row = row0 + lr, k = k0 + c * G::EPC;
      ok = row < rows_total && k < k_end;
      const int rc = min(row, rows_total - 1), kc = min(k, (k_end - 1) / G::EPC * G::EPC);
      src = base + (long)rc * ld + kc;
    } else {
      const int k = k0 + lr, row = row0 + c * G::EPC;
      ok = k < k_end && row < rows_total;
      const int kc = min(k, k_end - 1), rc = min(row, (rows_total - 1) / G::EPC * G::EPC);
      src = base + (long)kc * ld + rc;
    }
    r.v[j] = *reinterpret_cast<const uint4*>(src);   // consumed (and masked) only in tile_store, after the MFMAs
    okmask |= (ok ? 1u : 0u) << j;
  }
  r.okmask = okmask;
}

template <typename T, bool TR, int ROWS>
__device__ __forceinline__ void tile_store(const Regs<TileGeom<T, TR, ROWS>::PER_THREAD>& r, char* lds, int tid) {
  using G = TileGeom<T, TR, ROWS>;
#pragma unroll
  for (int j = 0; j < G::PER_THREAD; ++j) {
    const int q = tid + j * 256;
    const int lr = q / G::CPR, c = q % G::CPR;
    int off;
    if constexpr (!TR) off = lr * 128 + ((c ^ (lr & 7)) << 4);
    else off = lr * G::PITCH + (c << 4);
    const bool ok = (r.okmask >> j) & 1u;
    uint4 v = r.v[j];
    v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
    *reinterpret_cast<uint4*>(lds + off) = v;
  }
}

// LDS -> fragment: 8 k-values (k = 32*s + 8*g + e) of tile row `row` (i = lane&15 already added by caller)
template <typename T, bool TR, int ROWS, int PAD = 32>
__device__ __forceinline__ void frag_load(Frag<T>& f, const char* lds, int row16 /*first row of the 16-row group*/, int s,
                                          int lane) {
  using G = TileGeom<T, TR, ROWS, PAD>;
  const int i = lane & 15, g = lane >> 4;
  if constexpr (!TR) {
    const int row = row16 + i;
    if constexpr (sizeof(T) == 2) {
      const int c = s * 4 + g;
      f.v = *reinterpret_cast<const uint4*>(lds + row * 128 + ((c ^ (row & 7)) << 4));
    } else {
      const int c = 2 * g;
      f.lo = *reinterpret_cast<const float4*>(lds + row * 128 + ((c ^ (row & 7)) << 4));
      f.hi = *reinterpret_cast<const float4*>(lds + row * 128 + (((c + 1) ^ (row & 7)) << 4));
    }
  } else {
    if constexpr (sizeof(T) == 2) {
      // ds_read_b64_tr_b16: within a 16-lane group, lane j supplies the address of 4 consecutive bf16 of
      // k-row (j>>2), columns 4*(j&3)..+3; lane i receives column i of that 4x16 block (k = 0..3).
      const int k0 = s * 32 + g * 8;
      const int kr = k0 + (i >> 2);
      int coff = (row16 + (i & 3) * 4) * 2;                      // byte offset of this lane's 8 bytes inside the k-row
      int coff_hi = coff;
      if constexpr (PAD == 0) {                                  // LDS-DMA layout: chunk swizzle instead of padding
        coff ^= tr_swz<G::CPR>(kr) << 4;
        coff_hi ^= tr_swz<G::CPR>(kr + 4) << 4;
      }
      typedef short4_t __attribute__((address_space(3))) * lds_s4;
      short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(lds + kr * G::PITCH + coff));
      short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(lds + (kr + 4) * G::PITCH + coff_hi));
      uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
      f.v = make_uint4(a.x, a.y, b.x, b.y);
    } else {
      const int k0 = g * 8;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        int boff = (row16 + i) * 4;
        if constexpr (PAD == 0) boff ^= tr_swz<G::CPR>(k0 + e) << 4;
        v[e] = *reinterpret_cast<const float*>(lds + (k0 + e) * G::PITCH + boff);
      }
      f.lo = make_float4(v[0], v[1], v[2], v[3]);
      f.hi = make_float4(v[4], v[5], v[6], v[7]);
    }
  }
}

// =========================================================================================================
// LDS-DMA main loop (global_load_lds_dwordx4, no VGPR staging, no ds_write pass), STAGES-deep LDS ring.
//   Used when the reduction length is a multiple of the 128-byte slab (every linear layer of the planner);
//   ragged reductions fall back to the register-staged kernel above.
//   * each wave-instruction moves 1 KiB: LDS destination = wave-uniform base + lane*16 (lane-linear), so the
//     XOR swizzle of row operands is applied to the per-lane SOURCE address (logical chunk = phys ^ (row&7)),
//     the same involution frag_load applies on the read side;
//   * out-of-range rows/columns are clamped to valid addresses (their products are never stored);
//   * one barrier per slab; `s_waitcnt vmcnt(N)` leaves the younger slabs' DMA in flight across it.
// =========================================================================================================
template <typename T, bool TR, int ROWS, int NW = 4>
struct DmaPlan {
  static_assert(ROWS % (8 * NW) == 0, "tile rows must split into whole 1-KiB pieces per wavefront");
  static constexpr int PER_WAVE = ROWS / (8 * NW);    // 1-KiB pieces per wave per slab (ROWS / 8 pieces per tile)
  const T* src[PER_WAVE];                             // per-lane source address of piece j (advanced per slab)
  int lds_off[PER_WAVE];                              // wave-uniform LDS byte offset of piece j within the tile
};

template <typename T, bool TR, int ROWS, int NW>
__device__ __forceinline__ void dma_plan(DmaPlan<T, TR, ROWS, NW>& p, const T* base, long ld, int row0, int rows_total, int k0,
                                         int tid) {
  using G = TileGeom<T, TR, ROWS, 0>;
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int j = 0; j < DmaPlan<T, TR, ROWS, NW>::PER_WAVE; ++j) {
    const int piece = j * NW + wave;
    const int idx = piece * 64 + lane;               // 16-byte chunk index inside the tile == LDS position
    p.lds_off[j] = __builtin_amdgcn_readfirstlane(piece * 1024);
    if constexpr (!TR) {
      const int lr = idx >> 3, pch = idx & 7;
      const int c = pch ^ (lr & 7);
      const int rc = min(row0 + lr, rows_total - 1);
      p.src[j] = base + (long)rc * ld + k0 + c * G::EPC;
    } else {
      const int kr = idx / G::CPR, cch = (idx % G::CPR) ^ tr_swz<G::CPR>(idx / G::CPR);
      const int rc = min(row0 + cch * G::EPC, (rows_total - 1) / G::EPC * G::EPC);
      p.src[j] = base + (long)(k0 + kr) * ld + rc;
    }
  }
}

// One LDS-DMA piece, issued from inline asm on purpose: with the builtin hipcc (ROCm 7.2) treats the DMA as a
// pending LDS write that may alias the fragment reads and drains it with s_waitcnt vmcnt(0) before the first ds_read
// of the slab, which removes all overlap.  Hidden in asm, the copy stays in flight under the MFMAs; completion is
// tracked by our own counted s_waitcnt vmcnt(N) + barrier (cdna_hip_programming.md §5.7).  M0 = LDS byte address of
// the piece (wave-uniform), written in the same statement that consumes it and restored afterwards.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_addr)
      : "memory");
}

template <typename T, bool TR, int ROWS, int NW>
__device__ __forceinline__ void dma_issue(DmaPlan<T, TR, ROWS, NW>& p, unsigned lds_tile_addr, long ld) {
  constexpr int BK = MmaTraits<T>::BK;
#pragma unroll
  for (int j = 0; j < DmaPlan<T, TR, ROWS, NW>::PER_WAVE; ++j) {
    glds16(p.src[j], lds_tile_addr + (unsigned)p.lds_off[j]);
    if constexpr (!TR) p.src[j] += BK;
    else p.src[j] += (long)BK * ld;
  }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wait until at most `younger` whole slabs (PER_SLAB DMA instructions each) are still in flight; younger in [0, MAXS]
template <int PER_SLAB, int MAXS> __device__ __forceinline__ void wait_slabs(int younger) {
  if constexpr (MAXS <= 0) {
    wait_vmcnt<0>();
  } else {
    if (younger >= MAXS) wait_vmcnt<MAXS * PER_SLAB>();
    else wait_slabs<PER_SLAB, MAXS - 1>(younger);
  }
}

template <typename T, bool TA, bool TB, int BM, int BN, int NW>
__device__ __forceinline__ void load_frags(Frag<T> (&fa)[BM / (NW / 2) / 16], Frag<T> (&fb)[BN / 32], const char* sa, const char* sb,
                                           int s, int wr, int wc, int lane) {
#pragma unroll
  for (int a = 0; a < BM / (NW / 2) / 16; ++a) frag_load<T, TA, BM, 0>(fa[a], sa, wr * (BM / (NW / 2)) + a * 16, s, lane);
#pragma unroll
  for (int b = 0; b < BN / 32; ++b) frag_load<T, TB, BN, 0>(fb[b], sb, wc * (BN / 2) + b * 16, s, lane);
}

}  // namespace etp
