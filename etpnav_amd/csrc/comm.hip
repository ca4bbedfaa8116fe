// Data-parallel gradient mean over RCCL / xGMI behind the C ABI (replaces the DistributedDataParallel
// reducer of ss_trainer_ETP.py:208-212 and pretrain utils/misc.py:52-65 for the planner's flat gradient arena).
//
// One communicator per process (one process per GPU).  A bucket = a contiguous fp32 range of the gradient arena that a
// backward segment has just completed.  etp_allreduce_bucket_ready() orders a private communication stream after the
// producer stream and issues, IN PLACE on the arena,
//       reduce-scatter (sum)  ->  scale the rank's own 1/world slice by 1/world  ->  all-gather
// i.e. the bandwidth-optimal decomposition of the all-reduce: every GPU receives and reduces 1/world of the bucket from each
// of its 7 xGMI peers and then broadcasts its slice back; the mean's scaling touches only 1/world of the data.  The tail
// that does not divide by world x 64 elements goes through one small all-reduce.  fp32 transport by default (bit-compatible
// with DDP's fp32 mean up to summation order); bf16 transport is opt-in (halves the xGMI bytes, sums in bf16).
// etp_allreduce_wait() orders a consumer stream (optimizer) after everything issued so far.
//
// RCCL is bound at run time (dlopen of the librccl the process already carries -- torch ships one -- else ROCm's), so
// libetpnav_hip.so has no link-time dependency on it and single-GPU users never load it.
#include <dlfcn.h>
#include <string.h>

#include <vector>

#include "kernels.h"

namespace {

// the slice of rccl.h this file needs (types are ABI-stable: NCCL 2.x)
typedef struct { char internal[128]; } rcclUniqueId;
typedef void* rcclComm_t;
enum { RCCL_FLOAT32 = 7, RCCL_BFLOAT16 = 9, RCCL_SUM = 0 };
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(rcclUniqueId*) = nullptr;
  int (*CommInitRank)(rcclComm_t*, int, rcclUniqueId, int) = nullptr;
  int (*CommDestroy)(rcclComm_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
  int (*ReduceScatter)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, rcclComm_t, hipStream_t) = nullptr;
};
Rccl g_rccl;

int load_rccl() {
  if (g_rccl.lib) return ETP_OK;
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {          // prefer the copy already mapped into the process (torch's)
    h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    if (h) break;
  }
  for (int i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
  if (!h) return etp::fail(ETP_ERR_INVALID, std::string("cannot load librccl: ") + dlerror());
#define ETP_RCCL_SYM(field, sym)                                                                   \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, sym));                          \
  if (!g_rccl.field) return etp::fail(ETP_ERR_INVALID, std::string("librccl lacks ") + sym);
  ETP_RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
  ETP_RCCL_SYM(CommInitRank, "ncclCommInitRank")
  ETP_RCCL_SYM(CommDestroy, "ncclCommDestroy")
  ETP_RCCL_SYM(GetErrorString, "ncclGetErrorString")
  ETP_RCCL_SYM(AllReduce, "ncclAllReduce")
  ETP_RCCL_SYM(ReduceScatter, "ncclReduceScatter")
  ETP_RCCL_SYM(AllGather, "ncclAllGather")
#undef ETP_RCCL_SYM
  g_rccl.lib = h;
  return ETP_OK;
}

int rccl_check(int rc, const char* what) {
  if (rc == 0) return ETP_OK;
  return etp::fail(ETP_ERR_INVALID, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "rccl error"));
}
#define ETP_CHECK_RCCL(expr) ETP_TRY(rccl_check((expr), #expr))

}  // namespace

struct etp_comm {
  rcclComm_t comm = nullptr;
  int rank = 0, world = 1, comm_dtype = ETP_F32;
  hipStream_t stream = nullptr;             // private communication stream
  std::vector<hipEvent_t> events;
  size_t ev_next = 0;
  void* staging = nullptr;                  // bf16 transport: packed copy of the bucket in flight
  int64_t staging_elems = 0;
  hipEvent_t next_event() {
    hipEvent_t e = events[ev_next];
    ev_next = (ev_next + 1) % events.size();
    return e;
  }
};

using namespace etp;

extern "C" {

int etp_allreduce_unique_id(void* id_out) {
  ETP_REQUIRE(id_out, "null pointer");
  ETP_TRY(load_rccl());
  rcclUniqueId id;
  ETP_CHECK_RCCL(g_rccl.GetUniqueId(&id));
  memcpy(id_out, id.internal, sizeof(id.internal));
  return ETP_OK;
}

int etp_allreduce_init(etp_comm** out, const void* unique_id, int rank, int world, int comm_dtype, int64_t max_bucket_elems) {
  ETP_REQUIRE(out && unique_id && world >= 1 && rank >= 0 && rank < world, "bad arguments");
  ETP_REQUIRE(comm_dtype == ETP_F32 || comm_dtype == ETP_BF16, "comm dtype must be ETP_F32 or ETP_BF16");
  ETP_TRY(load_rccl());
  etp_comm* c = new etp_comm();
  c->rank = rank; c->world = world; c->comm_dtype = comm_dtype;
  rcclUniqueId id;
  memcpy(id.internal, unique_id, sizeof(id.internal));
  int rc = rccl_check(g_rccl.CommInitRank(&c->comm, world, id, rank), "ncclCommInitRank");
  if (rc) { delete c; return rc; }
  rc = check_hip(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), "hipStreamCreateWithFlags");
  if (rc) { (void)g_rccl.CommDestroy(c->comm); delete c; return rc; }
  c->events.resize(64);
  for (auto& e : c->events) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
  if (comm_dtype == ETP_BF16 && max_bucket_elems > 0) {
    c->staging_elems = round_up(max_bucket_elems, (long)world * 64);
    rc = check_hip(hipMalloc(&c->staging, (size_t)c->staging_elems * 2), "hipMalloc(staging)");
    if (rc) { etp_allreduce_destroy(c); return rc; }
  }
  *out = c;
  return ETP_OK;
}

int etp_allreduce_rank(const etp_comm* c) { return c ? c->rank : -1; }
int etp_allreduce_world(const etp_comm* c) { return c ? c->world : 0; }
etp_stream_t etp_allreduce_stream(const etp_comm* c) { return c ? (etp_stream_t)c->stream : nullptr; }

// grads[0, n) <- mean over ranks, in place.  Everything enqueued on `producer` so far completes before the bucket is read.
int etp_allreduce_bucket_ready(etp_comm* c, float* grads, int64_t n, etp_stream_t producer) {
  ETP_REQUIRE(c && grads && n >= 0 && ((uintptr_t)grads % 16 == 0), "bad arguments");
  if (n == 0) return ETP_OK;
  hipStream_t prod = (hipStream_t)producer, cs = c->stream;
  hipEvent_t e = c->next_event();
  ETP_CHECK_HIP(hipEventRecord(e, prod));
  ETP_CHECK_HIP(hipStreamWaitEvent(cs, e, 0));
  const int W = c->world;
  const float inv = 1.0f / (float)W;
  if (c->comm_dtype == ETP_BF16) {
    ETP_REQUIRE(c->staging && n <= c->staging_elems, "bucket larger than max_bucket_elems given to etp_allreduce_init");
    // pack -> reduce-scatter + all-gather on the packed copy -> unpack with the 1/world scaling
    const long per = round_up(n, (long)W * 8) / W;                 // elements per rank (16-byte slices); pad region is garbage-free:
    ETP_TRY(cast_f32_to_bf16(grads, c->staging, n, cs));
    if (per * W > n) ETP_CHECK_HIP(hipMemsetAsync((char*)c->staging + n * 2, 0, (size_t)(per * W - n) * 2, cs));
    char* mine = (char*)c->staging + (size_t)c->rank * per * 2;
    ETP_CHECK_RCCL(g_rccl.ReduceScatter(c->staging, mine, (size_t)per, RCCL_BFLOAT16, RCCL_SUM, c->comm, cs));
    ETP_CHECK_RCCL(g_rccl.AllGather(mine, c->staging, (size_t)per, RCCL_BFLOAT16, c->comm, cs));
    return cast_bf16_to_f32(c->staging, grads, n, inv, cs);
  }
  const int64_t per = (n / ((int64_t)W * 64)) * 64;                // 256-byte-aligned slice per rank
  const int64_t body = per * W;
  if (per > 0) {
    float* mine = grads + (int64_t)c->rank * per;
    ETP_CHECK_RCCL(g_rccl.ReduceScatter(grads, mine, (size_t)per, RCCL_FLOAT32, RCCL_SUM, c->comm, cs));
    ETP_TRY(scale_f32(mine, per, inv, cs));                        // the mean: only this rank's 1/world slice is scaled
    ETP_CHECK_RCCL(g_rccl.AllGather(mine, grads, (size_t)per, RCCL_FLOAT32, c->comm, cs));
  }
  if (body < n) {
    ETP_CHECK_RCCL(g_rccl.AllReduce(grads + body, grads + body, (size_t)(n - body), RCCL_FLOAT32, RCCL_SUM, c->comm, cs));
    ETP_TRY(scale_f32(grads + body, n - body, inv, cs));
  }
  return ETP_OK;
}

// order `consumer` after every bucket issued so far
int etp_allreduce_wait(etp_comm* c, etp_stream_t consumer) {
  ETP_REQUIRE(c, "null communicator");
  hipEvent_t e = c->next_event();
  ETP_CHECK_HIP(hipEventRecord(e, c->stream));
  ETP_CHECK_HIP(hipStreamWaitEvent((hipStream_t)consumer, e, 0));
  return ETP_OK;
}

int etp_allreduce_destroy(etp_comm* c) {
  if (!c) return ETP_OK;
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  for (auto& e : c->events) (void)hipEventDestroy(e);
  if (c->staging) (void)hipFree(c->staging);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return ETP_OK;
}

}  // extern "C"
