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
  int (*CommAbort)(rcclComm_t) = nullptr;             // optional: only used to get out of a collective that never completes
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
  g_rccl.CommAbort = reinterpret_cast<decltype(g_rccl.CommAbort)>(dlsym(h, "ncclCommAbort"));
  g_rccl.lib = h;
  return ETP_OK;
}

int rccl_check(int rc, const char* what) {
  if (rc == 0) return ETP_OK;
  return etp::fail(ETP_ERR_INVALID, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "rccl error"));
}
#define ETP_CHECK_RCCL(expr) ETP_TRY(rccl_check((expr), #expr))

}  // namespace

// ---- row-sparse table exchange (word-embedding gradient): pack -> all-gather -> scatter-add ------------------------------
// slot j of this rank's block: id (or -1: padding / a repeated id) and the table row (zeros for -1).  The packed row is
// REMOVED from the table (zeroed): the rank's own rows come back inside the gathered block like everybody else's.
// ids outside [0, n_rows) (an out-of-vocabulary token on ANY rank) are treated as padding on both sides: they must never become
// addresses.
template <typename R>
__global__ void rows_pack_kernel(float* __restrict__ table, long n_rows, long row_len, const int64_t* __restrict__ ids, long n_ids,
                                 int64_t* __restrict__ out_ids, R* __restrict__ out_rows) {
  const long j = blockIdx.x;
  __shared__ int dup;
  int64_t id = j < n_ids ? ids[j] : -1;
  if (id >= n_rows) id = -1;
  if (threadIdx.x == 0) dup = 0;
  __syncthreads();
  if (id >= 0) {                      // a repeated id contributes its row once: the first slot that names it
    int found = 0;
    for (long i = threadIdx.x; i < j; i += blockDim.x) found |= (ids[i] == id);
    if (found) dup = 1;               // benign race: every writer stores 1
  }
  __syncthreads();
  if (dup) id = -1;
  if (threadIdx.x == 0) out_ids[j] = id;
  R* dst = out_rows + j * row_len;
  if (id < 0) {
    for (long c = threadIdx.x; c < row_len; c += blockDim.x) etp::Elem<R>::st(dst + c, 0.f);
    return;
  }
  float* src = table + id * row_len;
  for (long c = threadIdx.x; c < row_len; c += blockDim.x) {
    etp::Elem<R>::st(dst + c, src[c]);
    src[c] = 0.f;
  }
}
template <typename R>
__global__ void rows_scatter_kernel(float* __restrict__ table, long n_rows, long row_len, const int64_t* __restrict__ all_ids,
                                    const R* __restrict__ all_rows, float scale) {
  const long j = blockIdx.x;
  const int64_t id = all_ids[j];
  if (id < 0 || id >= n_rows) return;
  const R* src = all_rows + j * row_len;
  float* dst = table + id * row_len;
  for (long c = threadIdx.x; c < row_len; c += blockDim.x) atomicAdd(dst + c, etp::Elem<R>::ld(src + c) * scale);
}

struct etp_comm {
  void* rows_ids = nullptr; void* rows_buf = nullptr;   // gather_rows staging: [world + 1][capacity] ids / rows (slot 0 = send block)
  void* rows_in = nullptr;                              // the caller's ids, copied on the producer stream (the caller may free its tensor)
  hipEvent_t rows_read = nullptr;                       // recorded behind rows_pack: the next call's copy into rows_in waits for it
  bool rows_pending = false;
  int64_t rows_cap = 0, rows_len = 0;
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
    c->staging_elems = etp_allreduce_staging_elems(max_bucket_elems, world);
    rc = check_hip(hipMalloc(&c->staging, (size_t)c->staging_elems * 2), "hipMalloc(staging)");
    if (rc) { etp_allreduce_destroy(c); return rc; }
  }
  *out = c;
  return ETP_OK;
}

int etp_allreduce_rank(const etp_comm* c) { return c ? c->rank : -1; }
int etp_allreduce_world(const etp_comm* c) { return c ? c->world : 0; }
etp_stream_t etp_allreduce_stream(const etp_comm* c) { return c ? (etp_stream_t)c->stream : nullptr; }

// Slice arithmetic of one dense bucket of n fp32 gradients over `world` ranks (a pure function: host-side unit tests cover worlds
// 2 / 4 / 8 without a second GPU, tests/test_dp_gloo.py):
//   fp32 transport: reduce-scatter + all-gather over body = per * world elements, per a multiple of 64 (256-byte slices), and one
//                   all-reduce over the tail [body, n) (fewer than world * 64 elements);
//   bf16 transport: the bucket is staged as bf16, padded with zeros to per * world elements (per a multiple of 8: 16-byte slices),
//                   reduce-scatter + all-gather over the whole staged copy, no tail.
// out[0] = per, out[1] = body (elements covered by the scatter / gather), out[2] = tail elements, out[3] = staged elements.
int etp_allreduce_plan(int64_t n, int world, int comm_dtype, int64_t* out) {
  ETP_REQUIRE(out && n >= 0 && world >= 1 && (comm_dtype == ETP_F32 || comm_dtype == ETP_BF16), "bad arguments");
  if (comm_dtype == ETP_BF16) {
    const int64_t per = round_up(n, (long)world * 8) / world;
    out[0] = per; out[1] = per * world; out[2] = 0; out[3] = per * world;
  } else {
    const int64_t per = (n / ((int64_t)world * 64)) * 64;
    out[0] = per; out[1] = per * world; out[2] = n - per * world; out[3] = 0;
  }
  return ETP_OK;
}
// staging capacity (bf16 elements) etp_allreduce_init allocates for buckets of up to max_bucket_elems
int64_t etp_allreduce_staging_elems(int64_t max_bucket_elems, int world) {
  return (max_bucket_elems > 0 && world >= 1) ? round_up(max_bucket_elems, (long)world * 64) : 0;
}

// grads[0, n) <- mean over ranks, in place.  Everything enqueued on `producer` so far completes before the bucket is read.
int etp_allreduce_bucket_ready(etp_comm* c, float* grads, int64_t n, etp_stream_t producer) {
  ETP_REQUIRE(c && grads && n >= 0 && ((uintptr_t)grads % 16 == 0), "bad arguments");
  ETP_REQUIRE(c->comm, "communicator was aborted (only etp_allreduce_destroy is valid now)");
  if (n == 0) return ETP_OK;
  hipStream_t prod = (hipStream_t)producer, cs = c->stream;
  hipEvent_t e = c->next_event();
  ETP_CHECK_HIP(hipEventRecord(e, prod));
  ETP_CHECK_HIP(hipStreamWaitEvent(cs, e, 0));
  const int W = c->world;
  const float inv = 1.0f / (float)W;
  int64_t plan[4];
  ETP_TRY(etp_allreduce_plan(n, W, c->comm_dtype, plan));
  if (c->comm_dtype == ETP_BF16) {
    ETP_REQUIRE(c->staging && plan[3] <= c->staging_elems, "bucket larger than max_bucket_elems given to etp_allreduce_init");
    // pack -> reduce-scatter + all-gather on the packed copy -> unpack with the 1/world scaling
    const long per = plan[0];                                      // elements per rank (16-byte slices); pad region is garbage-free:
    ETP_TRY(cast_f32_to_bf16(grads, c->staging, n, cs));
    if (per * W > n) ETP_CHECK_HIP(hipMemsetAsync((char*)c->staging + n * 2, 0, (size_t)(per * W - n) * 2, cs));
    char* mine = (char*)c->staging + (size_t)c->rank * per * 2;
    ETP_CHECK_RCCL(g_rccl.ReduceScatter(c->staging, mine, (size_t)per, RCCL_BFLOAT16, RCCL_SUM, c->comm, cs));
    ETP_CHECK_RCCL(g_rccl.AllGather(mine, c->staging, (size_t)per, RCCL_BFLOAT16, c->comm, cs));
    return cast_bf16_to_f32(c->staging, grads, n, inv, cs);
  }
  const int64_t per = plan[0], body = plan[1];                     // 256-byte-aligned slice per rank
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

// Row-sparse mean of a table gradient [n_rows, row_len] (the word-embedding table: <= B*L of 30 522 / 250 002 rows are
// non-zero; replaces DDP's dense all-reduce of that table, ss_trainer_ETP.py:208-212).  ids[0, n_ids) = the rows this rank
// touched (any order, may repeat); `capacity` >= n_ids must be THE SAME on every rank (e.g. B * max_txt_len): every rank
// contributes a fixed-size block, so nothing depends on how many distinct rows a rank has and there is no host
// synchronisation.  Runs on the communicator's stream after `producer`, i.e. on the SAME communicator and stream as the
// dense buckets (one communicator in flight).  Result: table = mean over ranks of the dense table.
int etp_allreduce_gather_rows(etp_comm* c, float* table, int64_t n_rows, int64_t row_len, const int64_t* ids, int64_t n_ids,
                              int64_t capacity, etp_stream_t producer) {
  ETP_REQUIRE(c && table && n_rows > 0 && row_len > 0 && capacity > 0 && n_ids >= 0 && n_ids <= capacity && (ids || n_ids == 0),
              "bad arguments (n_ids must not exceed the rank-independent capacity)");
  ETP_REQUIRE(c->comm, "communicator was aborted (only etp_allreduce_destroy is valid now)");
  const int W = c->world;
  const size_t rs = c->comm_dtype == ETP_BF16 ? 2 : 4;
  if (c->rows_cap < capacity || c->rows_len != row_len) {          // (re)allocate the staging blocks: first call / larger batch
    ETP_CHECK_HIP(hipStreamSynchronize(c->stream));
    if (c->rows_ids) (void)hipFree(c->rows_ids);
    if (c->rows_buf) (void)hipFree(c->rows_buf);
    if (c->rows_in) (void)hipFree(c->rows_in);
    c->rows_ids = c->rows_buf = c->rows_in = nullptr;
    ETP_CHECK_HIP(hipMalloc(&c->rows_in, (size_t)capacity * sizeof(int64_t)));
    ETP_CHECK_HIP(hipMalloc(&c->rows_ids, (size_t)(W + 1) * capacity * sizeof(int64_t)));
    ETP_CHECK_HIP(hipMalloc(&c->rows_buf, (size_t)(W + 1) * capacity * row_len * rs));
    c->rows_cap = capacity; c->rows_len = row_len;
  }
  hipStream_t prod = (hipStream_t)producer, cs = c->stream;
  // the ids are copied into the communicator's own block ON THE PRODUCER STREAM: the caller's tensor (often a temporary: a cast or
  // a concatenation) may be freed and its memory recycled as soon as this call returns
  // (a second exchange before etp_allreduce_wait -- another table, the next accumulation window -- must not overwrite the ids the
  // first one's pack kernel has not read yet: ADVICE r4)
  if (c->rows_pending) ETP_CHECK_HIP(hipStreamWaitEvent(prod, c->rows_read, 0));
  if (n_ids > 0) ETP_CHECK_HIP(hipMemcpyAsync(c->rows_in, ids, (size_t)n_ids * sizeof(int64_t), hipMemcpyDeviceToDevice, prod));
  ids = (const int64_t*)c->rows_in;
  hipEvent_t e = c->next_event();
  ETP_CHECK_HIP(hipEventRecord(e, prod));
  ETP_CHECK_HIP(hipStreamWaitEvent(cs, e, 0));
  int64_t* send_ids = (int64_t*)c->rows_ids;
  int64_t* all_ids = send_ids + capacity;
  char* send_rows = (char*)c->rows_buf;
  char* all_rows = send_rows + (size_t)capacity * row_len * rs;
  const float inv = 1.0f / (float)W;
  if (c->comm_dtype == ETP_BF16)
    ETP_LAUNCH(rows_pack_kernel<bf16_t>, dim3((unsigned)capacity), dim3(256), 0, cs, table, (long)n_rows, (long)row_len, ids, (long)n_ids, send_ids,
               (bf16_t*)send_rows);
  else
    ETP_LAUNCH(rows_pack_kernel<float>, dim3((unsigned)capacity), dim3(256), 0, cs, table, (long)n_rows, (long)row_len, ids, (long)n_ids, send_ids,
               (float*)send_rows);
  ETP_CHECK_LAUNCH("rows_pack");
  if (!c->rows_read) ETP_CHECK_HIP(hipEventCreateWithFlags(&c->rows_read, hipEventDisableTiming));
  ETP_CHECK_HIP(hipEventRecord(c->rows_read, cs));
  c->rows_pending = true;
  ETP_CHECK_RCCL(g_rccl.AllGather(send_ids, all_ids, (size_t)capacity * 2, RCCL_FLOAT32, c->comm, cs));   // int64 ids as 2 x 32-bit words
  ETP_CHECK_RCCL(g_rccl.AllGather(send_rows, all_rows, (size_t)capacity * row_len, c->comm_dtype == ETP_BF16 ? RCCL_BFLOAT16 : RCCL_FLOAT32,
                                  c->comm, cs));
  if (c->comm_dtype == ETP_BF16)
    ETP_LAUNCH(rows_scatter_kernel<bf16_t>, dim3((unsigned)(capacity * W)), dim3(256), 0, cs, table, (long)n_rows, (long)row_len, all_ids,
               (const bf16_t*)all_rows, inv);
  else
    ETP_LAUNCH(rows_scatter_kernel<float>, dim3((unsigned)(capacity * W)), dim3(256), 0, cs, table, (long)n_rows, (long)row_len, all_ids,
               (const float*)all_rows, inv);
  ETP_CHECK_LAUNCH("rows_scatter");
  return ETP_OK;
}

// 1 when librccl can be bound in this process (ranks agree on this BEFORE anybody enters ncclCommInitRank)
int etp_allreduce_available(void) { return load_rccl() == ETP_OK ? 1 : 0; }

// order `consumer` after every bucket issued so far
int etp_allreduce_wait(etp_comm* c, etp_stream_t consumer) {
  ETP_REQUIRE(c, "null communicator");
  hipEvent_t e = c->next_event();
  ETP_CHECK_HIP(hipEventRecord(e, c->stream));
  ETP_CHECK_HIP(hipStreamWaitEvent((hipStream_t)consumer, e, 0));
  return ETP_OK;
}

// 1 when everything issued on the communicator's stream so far has completed, 0 while work is in flight (host-side poll for
// the first-contact self-test, etpnav_amd/dp.py: a collective that never completes must not block the caller)
int etp_allreduce_idle(etp_comm* c) {
  if (!c || !c->stream) return 1;
  return hipStreamQuery(c->stream) == hipSuccess ? 1 : 0;
}

// Give up on the communicator without waiting for work in flight (ncclCommAbort): after a timed-out self-test every rank
// aborts, destroys the handle and falls back to torch.distributed.  The handle stays valid for etp_allreduce_destroy only.
int etp_allreduce_abort(etp_comm* c) {
  ETP_REQUIRE(c, "null communicator");
  if (c->comm) {
    if (g_rccl.CommAbort) (void)g_rccl.CommAbort(c->comm);
    c->comm = nullptr;                                  // never ncclCommDestroy an aborted communicator
  }
  return ETP_OK;
}

int etp_allreduce_destroy(etp_comm* c) {
  if (!c) return ETP_OK;
  if (c->stream && c->comm) (void)hipStreamSynchronize(c->stream);
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  for (auto& e : c->events) (void)hipEventDestroy(e);
  if (c->staging) (void)hipFree(c->staging);
  if (c->rows_ids) (void)hipFree(c->rows_ids);
  if (c->rows_buf) (void)hipFree(c->rows_buf);
  if (c->rows_in) (void)hipFree(c->rows_in);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return ETP_OK;
}

}  // extern "C"
