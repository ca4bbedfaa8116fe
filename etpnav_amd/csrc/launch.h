// Kernel-launch layer of libetpnav_hip.so: every launch, event edge and memset of the library goes through here, so that a
// whole planner step can either be issued eagerly or RECORDED into an explicitly built hipGraph.
//
// Why not stream capture: the step runs on three streams (dependent chain | weight gradients | panorama branch) and
// hipStreamEndCapture crashes on ROCm 7.2 once two side streams have been pulled into a capture (round 1).  The recorder
// builds the same DAG by hand instead: one kernel node per launch, dependencies = (previous node of the logical stream) +
// (nodes of the events that stream waited for since).  Replay costs one hipGraphLaunch on the host instead of ~350
// launches + ~80 event calls.
#pragma once
#include <hip/hip_runtime.h>

#include <tuple>
#include <type_traits>
#include <utility>

namespace etp {

bool rec_active();
int rec_kernel(const void* fn, dim3 grid, dim3 block, unsigned smem, hipStream_t st, void** args);   // adds a kernel node
hipError_t event_record(hipEvent_t e, hipStream_t s);        // eager: hipEventRecord; recording: remembers the stream's node
hipError_t stream_wait_event(hipStream_t s, hipEvent_t e);   // eager: hipStreamWaitEvent; recording: adds a dependency
hipError_t memset_async(void* p, int value, size_t bytes, hipStream_t s);
// optional per-launch HIP-event timing of EVERY kernel (tools/chain_budget.py): events on the launch stream, so use a
// single-stream issue when the numbers should bracket each kernel alone
bool ktime_active();
void ktime_begin(const void* fn, dim3 grid, dim3 block, hipStream_t st);
void ktime_end(hipStream_t st);
// Dynamic LDS above 64 KB needs hipFuncAttributeMaxDynamicSharedMemorySize on the kernel -- per DEVICE: the attribute lives on the
// device's copy of the function (a process-wide `static bool attr_set` left the second device of a process without it).
// Cached per (kernel, current device); a larger request than the cached one is set again.
hipError_t ensure_dyn_lds(const void* kern, int bytes);
// LDS bytes of one CU of the current device (hipDeviceProp.maxSharedMemoryPerMultiProcessor; 160 KB on gfx950): what a launch asks
// for when it must have the CU to itself
int cu_lds_bytes();
int cu_count();                                              // compute units of the current device (256 on MI355X)
// Row-kernel families that can be launched with the CU to themselves (a workgroup that asks for the CU's whole LDS
// shares it with no other kernel's wavefronts).  The switch ROW_EXCLUSIVE is a bit mask over these families; default
// ROWF_DEFAULT = none since round 6: the corruption the exclusivity of round 5 papered over was one packed-fp32 instruction form
// (v_pk_mul_f32 ... op_sel:[0,1]) misbehaving beside another kernel's MFMAs, and the row kernels no longer contain packed fp32
// (etpnav_amd/build.py NO_PACKED_FP32; build-time audit).  tests/test_neighbours_gpu.py runs every family SHARED beside the 128x128
// GEMM tile classes; the mask stays as a switch.
enum RowFamily {
  ROWF_PANO_BWD = 1, ROWF_GMAP_BWD = 2, ROWF_TEXT_BWD = 4, ROWF_SAP_BWD = 8, ROWF_LN_BWD = 16, ROWF_LN_FWD = 32, ROWF_ATTN_BWD = 64,
  ROWF_ATTN_FWD = 128,
  ROWF_DEFAULT = 0
};
// dynamic LDS bytes to launch `kern` with: `smem` itself, or (CU LDS - the kernel's static LDS) when the family's bit is set
unsigned row_launch_lds(const void* kern, int family, unsigned smem);
hipError_t launch_status();                                  // error of the last launch (eager: hipGetLastError)
void set_launch_error(hipError_t e);

template <typename Tuple, size_t... I>
inline void tuple_ptrs(Tuple& t, void** out, std::index_sequence<I...>) {
  ((out[I] = const_cast<void*>(static_cast<const void*>(&std::get<I>(t)))), ...);
}

template <typename... KArgs, typename... Args>
inline void launch_kernel(void (*kern)(KArgs...), dim3 grid, dim3 block, unsigned smem, hipStream_t st, Args&&... args) {
  static_assert(sizeof...(KArgs) == sizeof...(Args), "kernel argument count mismatch");
  std::tuple<std::remove_cv_t<std::remove_reference_t<KArgs>>...> vals(
      static_cast<std::remove_cv_t<std::remove_reference_t<KArgs>>>(std::forward<Args>(args))...);
  void* ptrs[sizeof...(KArgs) > 0 ? sizeof...(KArgs) : 1];
  tuple_ptrs(vals, ptrs, std::index_sequence_for<KArgs...>{});
  if (rec_active()) {
    const int rc = rec_kernel(reinterpret_cast<const void*>(kern), grid, block, smem, st, ptrs);
    if (rc) set_launch_error(hipErrorUnknown);
    return;
  }
  const bool timed = ktime_active();
  if (timed) ktime_begin(reinterpret_cast<const void*>(kern), grid, block, st);
  const hipError_t e = hipLaunchKernel(reinterpret_cast<const void*>(kern), grid, block, ptrs, smem, st);
  if (e != hipSuccess) set_launch_error(e);
  if (timed) ktime_end(st);
}

}  // namespace etp

#define ETP_LAUNCH(kern, grid, block, smem, st, ...) ::etp::launch_kernel(kern, grid, block, smem, st, __VA_ARGS__)
#define ETP_LAUNCH_ROW(family, kern, grid, block, smem, st, ...) \
  ::etp::launch_kernel(kern, grid, block, ::etp::row_launch_lds(reinterpret_cast<const void*>(kern), family, smem), st, __VA_ARGS__)
