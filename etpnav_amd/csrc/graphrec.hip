// Explicit hipGraph construction of a recorded planner step (see launch.h).
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace etp {

namespace {
struct Recorder {
  hipGraph_t graph = nullptr;
  std::unordered_map<hipStream_t, hipGraphNode_t> last;                   // last node of each logical stream
  std::unordered_map<hipStream_t, std::vector<hipGraphNode_t>> pending;   // extra dependencies of its NEXT node
  std::unordered_map<hipEvent_t, hipGraphNode_t> events;                  // node an event stands for (absent: nothing)
  size_t n_kernels = 0, n_edges = 0;
};
Recorder* g_rec = nullptr;
thread_local hipError_t g_launch_err = hipSuccess;

// dependencies of the next node on stream s (consumes the pending list)
std::vector<hipGraphNode_t> take_deps(Recorder& r, hipStream_t s) {
  std::vector<hipGraphNode_t> d;
  auto it = r.last.find(s);
  if (it != r.last.end()) d.push_back(it->second);
  auto pt = r.pending.find(s);
  if (pt != r.pending.end()) {
    for (hipGraphNode_t n : pt->second) {
      bool dup = false;
      for (hipGraphNode_t m : d) dup = dup || (m == n);
      if (!dup) d.push_back(n);
    }
    pt->second.clear();
  }
  r.n_edges += d.size();
  return d;
}
}  // namespace

bool rec_active() { return g_rec != nullptr; }
void set_launch_error(hipError_t e) { g_launch_err = e; }
hipError_t launch_status() {
  hipError_t e = g_launch_err;
  g_launch_err = hipSuccess;
  if (e != hipSuccess) return e;
  return rec_active() ? hipSuccess : hipGetLastError();
}

int rec_kernel(const void* fn, dim3 grid, dim3 block, unsigned smem, hipStream_t st, void** args) {
  Recorder& r = *g_rec;
  std::vector<hipGraphNode_t> deps = take_deps(r, st);
  hipKernelNodeParams p;
  memset(&p, 0, sizeof(p));
  p.func = const_cast<void*>(fn);
  p.gridDim = grid; p.blockDim = block; p.sharedMemBytes = smem; p.kernelParams = args; p.extra = nullptr;
  hipGraphNode_t node;
  ETP_CHECK_HIP(hipGraphAddKernelNode(&node, r.graph, deps.empty() ? nullptr : deps.data(), deps.size(), &p));
  r.last[st] = node;
  ++r.n_kernels;
  return ETP_OK;
}

hipError_t event_record(hipEvent_t e, hipStream_t s) {
  if (!rec_active()) return hipEventRecord(e, s);
  Recorder& r = *g_rec;
  auto pt = r.pending.find(s);
  if (pt != r.pending.end() && !pt->second.empty()) {
    // the stream has waited for other streams since its last node: the event must stand for those too -> join node
    std::vector<hipGraphNode_t> deps = take_deps(r, s);
    hipGraphNode_t node;
    const hipError_t rc = hipGraphAddEmptyNode(&node, r.graph, deps.data(), deps.size());
    if (rc != hipSuccess) return rc;
    r.last[s] = node;
  }
  auto it = r.last.find(s);
  if (it != r.last.end()) r.events[e] = it->second;
  else r.events.erase(e);                                   // nothing recorded on s yet: the event is already complete
  return hipSuccess;
}

hipError_t stream_wait_event(hipStream_t s, hipEvent_t e) {
  if (!rec_active()) return hipStreamWaitEvent(s, e, 0);
  Recorder& r = *g_rec;
  auto it = r.events.find(e);
  if (it != r.events.end()) r.pending[s].push_back(it->second);
  return hipSuccess;
}

hipError_t memset_async(void* p, int value, size_t bytes, hipStream_t s) {
  if (!rec_active()) return hipMemsetAsync(p, value, bytes, s);
  Recorder& r = *g_rec;
  std::vector<hipGraphNode_t> deps = take_deps(r, s);
  hipMemsetParams m;
  memset(&m, 0, sizeof(m));
  m.dst = p; m.value = (unsigned)value & 0xffu; m.elementSize = 1; m.width = bytes; m.height = 1; m.pitch = bytes;
  hipGraphNode_t node;
  const hipError_t rc = hipGraphAddMemsetNode(&node, r.graph, deps.empty() ? nullptr : deps.data(), deps.size(), &m);
  if (rc == hipSuccess) r.last[s] = node;
  return rc;
}

int rec_begin() {
  ETP_REQUIRE(g_rec == nullptr, "a recording is already active");
  Recorder* r = new Recorder();
  const hipError_t e = hipGraphCreate(&r->graph, 0);
  if (e != hipSuccess) { delete r; return check_hip(e, "hipGraphCreate"); }
  g_rec = r;
  return ETP_OK;
}

int rec_end(hipGraph_t* graph, hipGraphExec_t* exec, long* n_kernels, long* n_edges) {
  ETP_REQUIRE(g_rec != nullptr, "no active recording");
  Recorder* r = g_rec;
  g_rec = nullptr;
  *graph = r->graph;
  if (n_kernels) *n_kernels = (long)r->n_kernels;
  if (n_edges) *n_edges = (long)r->n_edges;
  delete r;
  ETP_CHECK_HIP(hipGraphInstantiate(exec, *graph, nullptr, nullptr, 0));
  return ETP_OK;
}

// ---- per-launch timing of every kernel ------------------------------------------------------------------------------
namespace {
struct KRec { const void* fn; unsigned grid, block; hipStream_t st; hipEvent_t a, b; };
bool g_ktime = false;
std::vector<KRec> g_krecs;
}  // namespace
bool ktime_active() { return g_ktime && !rec_active(); }
void ktime_begin(const void* fn, dim3 grid, dim3 block, hipStream_t st) {
  KRec r{fn, grid.x * grid.y * grid.z, block.x * block.y * block.z, st, nullptr, nullptr};
  (void)hipEventCreate(&r.a);
  (void)hipEventCreate(&r.b);
  (void)hipEventRecord(r.a, st);
  g_krecs.push_back(r);
}
void ktime_end(hipStream_t st) {
  if (!g_krecs.empty()) (void)hipEventRecord(g_krecs.back().b, st);
}
void ktime_enable(bool on) { g_ktime = on; }
void ktime_reset() {
  for (auto& r : g_krecs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  g_krecs.clear();
}
// one text line per launch, in launch order: "<us>\t<grid>\t<block>\t<stream>\t<kernel name>"
long ktime_report(char* buf, long cap) {
  long n = 0;
  for (auto& r : g_krecs) {
    if (hipEventSynchronize(r.b) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
    const char* name = hipKernelNameRefByPtr(r.fn, r.st);
    char line[512];
    const int len = snprintf(line, sizeof(line), "%.2f\t%u\t%u\t%p\t%s\n", ms * 1e3, r.grid, r.block, (void*)r.st,
                             name ? name : "?");
    if (n + len >= cap) break;
    memcpy(buf + n, line, len);
    n += len;
  }
  if (n < cap) buf[n] = 0;
  return n;
}

void rec_abort() {
  if (!g_rec) return;
  (void)hipGraphDestroy(g_rec->graph);
  delete g_rec;
  g_rec = nullptr;
}

}  // namespace etp
