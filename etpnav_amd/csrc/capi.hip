// extern "C" surface of libetpnav_hip.so: per-operator entry points, hipGraph helpers, error state.
#include <string.h>

#include <vector>

#include <stdlib.h>

#include <atomic>
#include <mutex>

#include "kernels.h"
#include "options.h"

namespace etp {

static thread_local std::string g_last_error;

// ---- per-device launch attributes (launch.h) ----------------------------------------------------------------------------------
static std::mutex g_attr_mu;
static std::vector<std::pair<std::pair<const void*, int>, int>> g_attr_set;     // ((kernel, device), bytes)
hipError_t ensure_dyn_lds(const void* kern, int bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lk(g_attr_mu);
  for (auto& it : g_attr_set)
    if (it.first.first == kern && it.first.second == dev) {
      if (it.second >= bytes) return hipSuccess;
      e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
      if (e == hipSuccess) it.second = bytes;
      return e;
    }
  e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) g_attr_set.push_back({{kern, dev}, bytes});
  return e;
}
int cu_lds_bytes() {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 160 * 1024;
  int v = cache[dev].load(std::memory_order_relaxed);
  if (v > 0) return v;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 160 * 1024;
  v = (int)prop.maxSharedMemoryPerMultiProcessor;
  if (v <= 0) v = 160 * 1024;
  cache[dev].store(v, std::memory_order_relaxed);
  return v;
}

int cu_count() {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int v = cache[dev].load(std::memory_order_relaxed);
  if (v > 0) return v;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
  v = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  cache[dev].store(v, std::memory_order_relaxed);
  return v;
}

unsigned row_launch_lds(const void* kern, int family, unsigned smem) {
  if (!(opt_int(OPT_ROW_EXCLUSIVE, ROWF_DEFAULT) & family)) return smem;
  static std::mutex mu;
  static std::vector<std::pair<const void*, int>> stat;       // static LDS of each kernel seen (the same on every device)
  int st_bytes = -1;
  {
    std::lock_guard<std::mutex> lk(mu);
    for (auto& it : stat) if (it.first == kern) st_bytes = it.second;
    if (st_bytes < 0) {
      hipFuncAttributes fa;
      st_bytes = hipFuncGetAttributes(&fa, kern) == hipSuccess ? (int)fa.sharedSizeBytes : 0;
      stat.push_back({kern, st_bytes});
    }
  }
  const int want = cu_lds_bytes() - st_bytes;
  if (want <= (int)smem) return smem;
  const hipError_t e = ensure_dyn_lds(kern, want);
  if (e != hipSuccess) { set_launch_error(e); return smem; }
  return (unsigned)want;
}

// ---- run-time switches (options.h) -------------------------------------------------------------------------------------------
static const char* const kOptNames[OPT_COUNT] = {
    "MM32", "MM32_GROUP", "MM32_K2", "GEMM_TILE", "GROUP_TILE", "GEMM_WIDE", "GEMM_SMALL", "GEMM_XCD", "ATTN_FUSED", "ATTN_FLASH", "ATTN_Q96",
    "ATTN_ROWS", "LNBWD_GRID", "LNBWD_TWO_STAGE", "WGRAD_GROUP", "FLUSH_DELAY", "FLUSH_EVERY", "ROW_EXCLUSIVE",
    "ATTN_PROJ", "NAV_TAIL", "TXT_LAST_SPLIT", "TXT_TAIL", "ATTN_QKV",
#ifdef ETP_EXPERIMENTS
    "SKIP_LN", "SKIP_ATTN", "SKIP_WGRAD",
#endif
};
// two value slots per switch: a setter writes the slot that is NOT published and then flips the index, so a concurrent reader never
// sees a half-written string (readers hold no lock; setters are serialised)
static char g_opt_val[OPT_COUNT][2][32];
static std::atomic<int> g_opt_slot[OPT_COUNT];      // -1 unset, else the published slot
static std::once_flag g_opt_once;
static std::mutex g_opt_mu;
static void opt_store(int i, const char* v) {
  if (!v || !v[0]) { g_opt_slot[i].store(-1, std::memory_order_release); return; }
  const int cur = g_opt_slot[i].load(std::memory_order_acquire), nxt = cur == 0 ? 1 : 0;
  strncpy(g_opt_val[i][nxt], v, sizeof(g_opt_val[i][nxt]) - 1);
  g_opt_val[i][nxt][sizeof(g_opt_val[i][nxt]) - 1] = 0;
  g_opt_slot[i].store(nxt, std::memory_order_release);
}
static void opt_init() {
  std::call_once(g_opt_once, [] {
    for (int i = 0; i < OPT_COUNT; ++i) {
      g_opt_slot[i].store(-1);
      const std::string env = std::string("ETP_") + kOptNames[i];
      opt_store(i, getenv(env.c_str()));
    }
  });
}
const char* opt_str(Opt o) {
  opt_init();
  const int s = g_opt_slot[o].load(std::memory_order_acquire);
  return s < 0 ? nullptr : g_opt_val[o][s];
}
int opt_int(Opt o, int dflt) {
  const char* s = opt_str(o);
  return s ? atoi(s) : dflt;
}
static int opt_index(const char* name) {
  if (!name) return -1;
  if (!strncmp(name, "ETP_", 4)) name += 4;
  for (int i = 0; i < OPT_COUNT; ++i)
    if (!strcmp(name, kOptNames[i])) return i;
  return -1;
}
int opt_set(const char* name, const char* value) {
  opt_init();
  const int i = opt_index(name);
  if (i < 0) return -1;
  std::lock_guard<std::mutex> lk(g_opt_mu);
  opt_store(i, value);
  return 0;
}
int opt_get(const char* name, char* out, int cap) {
  const int i = opt_index(name);
  if (i < 0) return -1;
  const char* v = opt_str((Opt)i);
  const int n = v ? (int)strlen(v) : 0;
  if (out && cap > 0) { strncpy(out, v ? v : "", cap - 1); out[cap - 1] = 0; }
  return n;
}
int opt_list(char* out, int cap) {
  opt_init();
  std::string s;
  for (int i = 0; i < OPT_COUNT; ++i) {
    const char* v = opt_str((Opt)i);
    if (v) s += std::string(kOptNames[i]) + "=" + v + "\n";
  }
  if (out && cap > 0) { strncpy(out, s.c_str(), cap - 1); out[cap - 1] = 0; }
  return (int)s.size();
}

void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) { g_last_error = msg; return code; }
int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return 0;
  g_last_error = std::string(what) + ": " + hipGetErrorString(e);
  return (int)e > 0 ? (int)e : 999;
}

}  // namespace etp

using namespace etp;

struct etp_graph { hipGraph_t graph; hipGraphExec_t exec; };

// etp_stamp: the stream's arrival time at this point, taken on the device (no host event, no profiler)
__global__ void stamp_kernel(unsigned long long* slot) {
  if (threadIdx.x == 0) *slot = __builtin_amdgcn_s_memrealtime();
}
namespace etp {
static unsigned long long* g_stamp_buf = nullptr;
static long g_stamp_cap = 0;
static std::vector<int> g_stamp_tags;
void stamp_mark(hipStream_t st, int tag) {
  if (!g_stamp_buf || rec_active() || (long)g_stamp_tags.size() >= g_stamp_cap) return;
  ETP_LAUNCH(stamp_kernel, dim3(1), dim3(64), 0, st, g_stamp_buf + g_stamp_tags.size());
  g_stamp_tags.push_back(tag);
}
}  // namespace etp

extern "C" {

const char* etp_version(void) { return "etpnav_hip 0.1.0 (gfx950)"; }
const char* etp_last_error(void) { return g_last_error.c_str(); }

int etp_option_set(const char* name, const char* value) {
  if (opt_set(name, value) != 0) return fail(ETP_ERR_INVALID, std::string("etp_option_set: unknown switch ") + (name ? name : "(null)"));
  return ETP_OK;
}
int etp_option_get(const char* name, char* out, int cap) { return opt_get(name, out, cap); }
int etp_option_list(char* out, int cap) { return opt_list(out, cap); }

static int desc_to_args(const etp_gemm_desc* d, GemmArgs& g) {
  ETP_REQUIRE(d && d->A && d->B && d->C, "null descriptor/operand");
  ETP_REQUIRE(d->act >= ETP_ACT_NONE && d->act <= ETP_ACT_MUL_Z, "bad activation");
  ETP_REQUIRE((d->act == ETP_ACT_NONE || d->act == ETP_ACT_RELU) || d->Z, "activation needs Z");
  memset(&g, 0, sizeof(g));
  g.A = d->A; g.B = d->B; g.C = d->C; g.M = d->M; g.N = d->N; g.K = d->K;
  g.lda = d->lda; g.ldb = d->ldb; g.ldc = d->ldc;
  g.nb_inner = d->batch_inner > 0 ? d->batch_inner : 1;
  g.sAo = d->sAo; g.sAi = d->sAi; g.sBo = d->sBo; g.sBi = d->sBi; g.sCo = d->sCo; g.sCi = d->sCi;
  g.ksplit = d->ksplit > 0 ? d->ksplit : 1;
  g.alpha = d->alpha; g.bias = d->bias; g.R = d->R; g.ldr = d->ldr; g.Z = d->Z; g.ldz = d->ldz; g.act = d->act;
  g.out_mode = d->out_mode;
  g.a_colsum = d->a_colsum;
  return ETP_OK;
}
int etp_gemm(const etp_gemm_desc* d, etp_stream_t stream) {
  GemmArgs g;
  ETP_TRY(desc_to_args(d, g));
  return launch_gemm(d->dtype, d->c_dtype, d->trans_a, d->trans_b, g, d->batch > 0 ? d->batch : 1, (hipStream_t)stream);
}
int etp_gemm_group(const etp_gemm_desc* d, int n, etp_stream_t stream) {
  ETP_REQUIRE(d && n >= 1 && n <= ETP_GEMM_GROUP_MAX, "1..8 descriptors");
  GemmArgs gs[ETP_GEMM_GROUP_MAX];
  for (int i = 0; i < n; ++i) {
    ETP_TRY(desc_to_args(d + i, gs[i]));
    ETP_REQUIRE(d[i].dtype == d[0].dtype && d[i].c_dtype == d[0].c_dtype && d[i].trans_a == d[0].trans_a &&
                    d[i].trans_b == d[0].trans_b && d[i].batch <= 1 && d[i].ksplit <= 1,
                "grouped products share dtype / storage class and are unbatched, unsplit");
  }
  return launch_gemm_group(d[0].dtype, d[0].c_dtype, d[0].trans_a, d[0].trans_b, gs, n, (hipStream_t)stream);
}

int etp_colsum(int dtype, const void* dy, int64_t ld, float* db, int M, int N, etp_stream_t s) {
  ETP_REQUIRE(dy && db, "null pointer");
  return colsum(dtype, dy, ld, db, M, N, (hipStream_t)s);
}
int etp_ln_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* stats, int M, int H, float eps,
               etp_stream_t s) {
  ETP_REQUIRE(x && gamma && beta && y, "null pointer");
  return ln_fwd(dtype, x, gamma, beta, y, stats, M, H, eps, (hipStream_t)s);
}
int etp_ln_bwd(int dtype, const void* dy, const void* x, const float* stats, const float* gamma, const void* add, void* dx,
               float* dgamma, float* dbeta, int M, int H, etp_stream_t s) {
  ETP_REQUIRE(dy && x && stats && gamma && dx && ((dgamma == nullptr) == (dbeta == nullptr)), "null pointer");
  return ln_bwd(dtype, dy, x, stats, gamma, add, dx, dgamma, dbeta, M, H, (hipStream_t)s);
}
int etp_ln_stream_fwd(int dtype, const float* x, const float* gamma, const float* beta, float* y, void* y_lp, float* stats, int M,
                      int H, float eps, etp_stream_t s) {
  ETP_REQUIRE(x && gamma && beta && (y || y_lp), "null pointer");
  return ln_fwd_s(dtype, x, gamma, beta, y, y_lp, stats, M, H, eps, (hipStream_t)s);
}
int etp_ln_stream_bwd(int dtype, const float* dy, const float* x, const float* stats, const float* gamma, const float* add,
                      float* dx, void* dx_lp, float* dgamma, float* dbeta, int M, int H, etp_stream_t s) {
  ETP_REQUIRE(dy && x && stats && gamma && (dx || dx_lp) && ((dgamma == nullptr) == (dbeta == nullptr)), "null pointer");
  return ln_bwd_s(dtype, dy, x, stats, gamma, add, dx, dx_lp, dgamma, dbeta, M, H, (hipStream_t)s);
}
int64_t etp_ln_bwd_part_bytes(int M, int H) { return (M > 0 && H > 0) ? (int64_t)ln_bwd_part_bytes(M, H) : 0; }
int etp_ln_stream_bwd_stage1(int dtype, const float* dy, const float* x, const float* stats, const float* gamma, const float* add,
                             float* dx, void* dx_lp, float* dgamma, float* dbeta, float* part, int M, int H, etp_stream_t s) {
  ETP_REQUIRE(dy && x && stats && gamma && (dx || dx_lp) && dgamma && dbeta && part, "null pointer");
  return ln_bwd_s(dtype, dy, x, stats, gamma, add, dx, dx_lp, dgamma, dbeta, M, H, (hipStream_t)s, drop_none(), part);
}
int etp_ln_part_reduce(const float* part, int M, int H, float* dgamma, float* dbeta, etp_stream_t s) {
  return ln_part_reduce(part, M, H, dgamma, dbeta, (hipStream_t)s);
}
int etp_softmax_fwd(int dtype, void* S, const uint8_t* keymask, const float* dist, const float* sp_w, const float* sp_b, int B,
                    int heads, int Lq, int Lk, int ldS, int mask_mode, etp_stream_t s) {
  ETP_REQUIRE(S, "null pointer");
  return softmax_fwd(dtype, S, keymask, dist, sp_w, sp_b, B, heads, Lq, Lk, ldS, mask_mode, (hipStream_t)s);
}
int etp_softmax_bwd(int dtype, const void* P, void* dP, const float* dist, float* d_sp_w, float* d_sp_b, int B, int heads, int Lq,
                    int Lk, int ldS, etp_stream_t s) {
  ETP_REQUIRE(P && dP, "null pointer");
  return softmax_bwd(dtype, P, dP, dist, d_sp_w, d_sp_b, B, heads, Lq, Lk, ldS, (hipStream_t)s);
}

static AttnBuf to_buf(const etp_attn_desc& f) {
  AttnBuf a{f.Q, f.ldq, f.K, f.ldk, f.V, f.ldv, f.B, f.Lq, f.Lk, f.ldS, f.keymask, f.mask_mode, f.dist, f.sp_w, f.sp_b};
  return a;
}
int etp_attn_fwd(const etp_attn_desc* d, etp_stream_t s) {
  ETP_REQUIRE(d && d->Q && d->K && d->V && d->P && d->ctx, "null pointer");
  ETP_REQUIRE(d->ldS >= d->Lk && d->ldS % 8 == 0, "ldS must be a multiple of 8 and >= Lk");
  return attn_fwd_impl(d->dtype, d->heads, to_buf(*d), d->P, d->ctx, d->ldc, d->alpha, (hipStream_t)s);
}
int etp_attn_bwd(const etp_attn_bwd_desc* d, etp_stream_t s) {
  ETP_REQUIRE(d && d->f.Q && d->f.K && d->f.V && d->f.P && d->dctx && d->dP && d->dQ && d->dK && d->dV, "null pointer");
  ETP_REQUIRE(d->f.ldS >= d->f.Lk && d->f.ldS % 8 == 0, "ldS must be a multiple of 8 and >= Lk");
  AttnBuf ab = to_buf(d->f);
  ab.O = d->f.ctx; ab.ldo = d->f.ldc;          // the forward output (the streaming kernels need it)
  return attn_bwd_impl(d->f.dtype, d->f.heads, ab, d->f.P, d->dctx, d->ldd, d->dP, d->dQ, d->lddq, d->dK, d->lddk,
                       d->dV, d->lddv, d->f.alpha, d->d_sp_w, d->d_sp_b, (hipStream_t)s);
}

int etp_attn_fwd_qkv(const etp_attn_desc* d, const void* x, int64_t ldx, const void* w_qkv, int64_t ldw, const float* b_qkv, etp_stream_t s) {
  ETP_REQUIRE(d && d->Q && d->K && d->V && d->P && d->ctx && x && w_qkv, "null pointer");
  ETP_REQUIRE(d->ldS >= d->Lk && d->ldS % 8 == 0, "ldS must be a multiple of 8 and >= Lk");
  const AttnBuf ab = to_buf(*d);
  if (!(d->dtype == ETP_BF16 && attn_rows_ok(d->dtype, ab, d->ldc) && attn_rows_qkv_ok(d->heads, ab, x, ldx, w_qkv, ldw))) {
    set_error("etp_attn_fwd_qkv: shape / dtype outside the fused kernel (bf16 self-attention, L <= 128, heads*64 == 768)");
    return ETP_ERR_INVALID;
  }
  return attn_rows_fwd(d->heads, ab, d->P, d->ctx, d->ldc, d->alpha, (hipStream_t)s, drop_none(), x, ldx, w_qkv, ldw, b_qkv);
}
int etp_attn_bwd_proj(const etp_attn_bwd_desc* d, const void* w_out, int64_t ldw, etp_stream_t s) {
  ETP_REQUIRE(d && d->f.Q && d->f.K && d->f.V && d->f.P && d->dctx && d->dQ && d->dK && d->dV && w_out, "null pointer");
  ETP_REQUIRE(d->f.ldS >= d->f.Lk && d->f.ldS % 8 == 0, "ldS must be a multiple of 8 and >= Lk");
  AttnBuf ab = to_buf(d->f);
  ab.O = d->f.ctx; ab.ldo = d->f.ldc;
  const int H = d->f.heads * 64;
  if (!(d->f.dtype == ETP_BF16 && attn_rows_ok(d->f.dtype, ab, d->f.ldc) && attn_rows_proj_ok(H, w_out, ldw))) {
    set_error("etp_attn_bwd_proj: shape / dtype outside the fused kernel (bf16, Lq and Lk <= 128, heads*64 == 768)");
    return ETP_ERR_INVALID;
  }
  return attn_rows_bwd(d->f.heads, ab, d->f.P, d->dctx, d->ldd, d->dQ, d->lddq, d->dK, d->lddk, d->dV, d->lddv, d->f.alpha, d->d_sp_w,
                       d->d_sp_b, (hipStream_t)s, drop_none(), w_out, ldw, H);
}

int etp_text_embed_fwd(int dtype, const int64_t* ids, const float* word, const float* pos, const float* type0, const float* gamma,
                       const float* beta, float* y, void* y_lp, float* stats, int B, int L, int H, float eps, etp_stream_t s) {
  ETP_REQUIRE(ids && word && pos && type0 && gamma && beta && y && stats, "null pointer");
  return text_embed_fwd(dtype, ids, word, pos, type0, gamma, beta, y, y_lp, stats, B, L, H, eps, (hipStream_t)s);
}
int etp_text_embed_bwd(int dtype, const float* dy, const int64_t* ids, const float* word, const float* pos, const float* type0,
                       const float* gamma, const float* stats, float* dword, float* dpos, float* dtype0, float* dgamma,
                       float* dbeta, int B, int L, int H, etp_stream_t s) {
  ETP_REQUIRE(dy && ids && word && pos && type0 && gamma && stats && dword && dpos && dtype0 && dgamma && dbeta, "null pointer");
  return text_embed_bwd(dtype, dy, ids, word, pos, type0, gamma, stats, dword, dpos, dtype0, dgamma, dbeta, B, L, H, (hipStream_t)s);
}

static PanoEmbedParams to_params(const float* const* p) {
  PanoEmbedParams q{p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10], p[11]};
  return q;
}
int etp_pano_embed_fwd(int dtype, const void* a, const void* d, const float* loc, const int64_t* nav, const float* const* params,
                       float* y, float* stats, int M, int H, etp_stream_t s) {
  ETP_REQUIRE(a && loc && nav && params && y && stats, "null pointer");
  return pano_embed_fwd(dtype, a, d, loc, nav, to_params(params), y, stats, M, H, (hipStream_t)s);
}
int etp_pano_embed_bwd(int dtype, const float* dy, const void* a, const void* d, const float* loc, const int64_t* nav,
                       const float* stats, const float* const* params, float* const* grads, void* da, void* dd, int M, int H,
                       etp_stream_t s) {
  ETP_REQUIRE(dy && a && loc && nav && stats && params && grads && da && (d == nullptr || dd != nullptr), "null pointer");
  PanoEmbedGrads g{grads[0], grads[1], grads[2], grads[3], grads[4], grads[5], grads[6], grads[7], grads[8], grads[9], grads[10],
                   grads[11]};
  return pano_embed_bwd(dtype, dy, a, d, loc, nav, stats, to_params(params), g, da, dd, M, H, (hipStream_t)s);
}
int etp_gmap_embed_fwd(int dtype, const float* img, const int64_t* step_ids, const float* pos, const float* step_emb,
                       const float* w_pos, const float* b_pos, const float* gamma, const float* beta, float* x, void* x_lp,
                       float* stats, int M, int H, int pos_dim, etp_stream_t s) {
  ETP_REQUIRE(img && step_ids && pos && step_emb && w_pos && b_pos && gamma && beta && x && stats, "null pointer");
  return gmap_embed_fwd(dtype, img, step_ids, pos, step_emb, w_pos, b_pos, gamma, beta, x, x_lp, stats, M, H, pos_dim, (hipStream_t)s);
}
int etp_gmap_embed_bwd(int dtype, const float* dx, const int64_t* step_ids, const float* pos, const float* w_pos, const float* b_pos,
                       const float* gamma, const float* stats, float* d_step_emb, float* d_w_pos, float* d_b_pos, float* dgamma,
                       float* dbeta, int M, int H, int pos_dim, etp_stream_t s) {
  ETP_REQUIRE(dx && step_ids && pos && w_pos && b_pos && gamma && stats && d_step_emb && d_w_pos && d_b_pos && dgamma && dbeta,
              "null pointer");
  return gmap_embed_bwd(dtype, dx, step_ids, pos, w_pos, b_pos, gamma, stats, d_step_emb, d_w_pos, d_b_pos, dgamma, dbeta, M, H,
                        pos_dim, (hipStream_t)s);
}
int etp_sap_tail_fwd(int dtype, const void* r, const float* gamma, const float* beta, const float* w2, const float* b2,
                     const uint8_t* visited, const uint8_t* valid, float* logits, float* stats, int M, int H, etp_stream_t s) {
  ETP_REQUIRE(r && gamma && beta && w2 && b2 && logits && stats, "null pointer");
  return sap_tail_fwd(dtype, r, gamma, beta, w2, b2, visited, valid, logits, stats, M, H, (hipStream_t)s);
}
int etp_sap_tail_bwd(int dtype, const float* dlogits, const void* r, const float* gamma, const float* beta, const float* w2,
                     const float* stats, const uint8_t* visited, const uint8_t* valid, void* dz, float* dgamma, float* dbeta,
                     float* dw2, float* db2, int M, int H, etp_stream_t s) {
  ETP_REQUIRE(dlogits && r && gamma && beta && w2 && stats && dz && dgamma && dbeta && dw2 && db2, "null pointer");
  return sap_tail_bwd(dtype, dlogits, r, gamma, beta, w2, stats, visited, valid, dz, dgamma, dbeta, dw2, db2, M, H, (hipStream_t)s);
}
int etp_sap_ce(const float* logits, const int64_t* labels, float* loss, float* dlogits, int B, int G, float scale,
               int64_t ignore_index, etp_stream_t s) {
  ETP_REQUIRE(logits && labels && loss, "null pointer");
  return sap_ce(logits, labels, loss, dlogits, B, G, scale, (long)ignore_index, (hipStream_t)s);
}
int etp_adamw_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, void* shadow, int64_t n_shadow,
                   const uint8_t* decay_mask, int64_t n, const etp_adamw_cfg* cfg, const float* sumsq, const int32_t* skip,
                   int zero_grads, etp_stream_t s) {
  ETP_REQUIRE(cfg, "null config");
  return adamw_step(params, grads, exp_avg, exp_avg_sq, shadow, (long)n_shadow, decay_mask, (long)n, *cfg, sumsq, skip,
                    zero_grads, (hipStream_t)s);
}
int etp_adamw_step_counted(float* params, float* grads, float* exp_avg, float* exp_avg_sq, void* shadow, int64_t n_shadow,
                           const uint8_t* decay_mask, int64_t n, const etp_adamw_cfg* cfg, const float* sumsq, const int32_t* skip,
                           int zero_grads, int32_t* step_counter, etp_stream_t s) {
  ETP_REQUIRE(cfg && step_counter, "null config / step counter");
  return adamw_step(params, grads, exp_avg, exp_avg_sq, shadow, (long)n_shadow, decay_mask, (long)n, *cfg, sumsq, skip,
                    zero_grads, (hipStream_t)s, step_counter);
}
int etp_grad_sqnorm(const float* grads, int64_t n, float* sumsq, int32_t* nonfinite, etp_stream_t s) {
  return grad_sqnorm(grads, (long)n, sumsq, nonfinite, (hipStream_t)s);
}
int etp_grad_sqnorm_masked(const float* grads, int64_t n, const uint8_t* mask, float* sumsq, int32_t* nonfinite, etp_stream_t s) {
  return grad_sqnorm(grads, (long)n, sumsq, nonfinite, (hipStream_t)s, mask);
}
int etp_gather_sum(int dtype, const void* src, const int32_t* ptr, const int32_t* idx, const float* w, void* out, int N, int H,
                   int accumulate, etp_stream_t s) {
  ETP_REQUIRE(src && ptr && idx && w && out, "null pointer");
  return gather_sum(dtype, src, ptr, idx, w, out, N, H, accumulate, (hipStream_t)s);
}
int etp_cast_f32_to_bf16(const float* src, void* dst, int64_t n, etp_stream_t s) {
  ETP_REQUIRE(src && dst && ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0), "null / unaligned pointer");
  return cast_f32_to_bf16(src, dst, n, (hipStream_t)s);
}
int etp_cast_bf16_to_f32(const void* src, float* dst, int64_t n, float scale, etp_stream_t s) {
  ETP_REQUIRE(src && dst, "null pointer");
  return cast_bf16_to_f32(src, dst, n, scale, (hipStream_t)s);
}
int etp_scale_f32(float* p, int64_t n, float scale, etp_stream_t s) {
  ETP_REQUIRE(p, "null pointer");
  return scale_f32(p, n, scale, (hipStream_t)s);
}

// ---- streams / graphs -----------------------------------------------------------------
int etp_stream_create(etp_stream_t* out) {
  ETP_REQUIRE(out, "null pointer");
  hipStream_t s;
  ETP_CHECK_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *out = (etp_stream_t)s;
  return ETP_OK;
}
int etp_stream_create_prio(etp_stream_t* out, int level) {
  // level < 0: lowest priority the device offers (leaf work: weight gradients), 0: default, > 0: highest
  ETP_REQUIRE(out, "null pointer");
  int least = 0, greatest = 0;
  ETP_CHECK_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
  const int prio = level < 0 ? least : (level > 0 ? greatest : (least + greatest) / 2);
  hipStream_t s;
  ETP_CHECK_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio));
  *out = (etp_stream_t)s;
  return ETP_OK;
}
int etp_stream_destroy(etp_stream_t s) { ETP_CHECK_HIP(hipStreamDestroy((hipStream_t)s)); return ETP_OK; }
int etp_stream_sync(etp_stream_t s) { ETP_CHECK_HIP(hipStreamSynchronize((hipStream_t)s)); return ETP_OK; }
int etp_stream_after(etp_stream_t from, etp_stream_t to) {
  // order `to` after everything enqueued so far on `from`; capturable (becomes a graph edge and pulls `to` into the capture)
  static std::vector<hipEvent_t> pool;
  static size_t next = 0;
  if (from == to) return ETP_OK;
  if (pool.empty()) {
    pool.resize(64);
    for (auto& e : pool) ETP_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  hipEvent_t e = pool[next];
  next = (next + 1) % pool.size();
  ETP_CHECK_HIP(event_record(e, (hipStream_t)from));
  ETP_CHECK_HIP(stream_wait_event((hipStream_t)to, e));
  return ETP_OK;
}
int etp_graph_begin(etp_stream_t s) {
  ETP_CHECK_HIP(hipStreamBeginCapture((hipStream_t)s, hipStreamCaptureModeThreadLocal));
  return ETP_OK;
}
int etp_graph_end(etp_stream_t s, etp_graph** out) {
  ETP_REQUIRE(out, "null pointer");
  hipGraph_t g;
  ETP_CHECK_HIP(hipStreamEndCapture((hipStream_t)s, &g));
  hipGraphExec_t e;
  ETP_CHECK_HIP(hipGraphInstantiate(&e, g, nullptr, nullptr, 0));
  *out = new etp_graph{g, e};
  return ETP_OK;
}
// Explicitly built graph (launch.h / graphrec.hip): between etp_rec_begin and etp_rec_end every call of this library that
// takes a stream is RECORDED as graph nodes instead of being issued; stream handles only name the logical streams
// (dependencies follow the same per-stream order + event edges the eager issue would have had).
int etp_rec_begin(void) { return rec_begin(); }
int etp_rec_end(etp_graph** out, int64_t* n_kernels, int64_t* n_edges) {
  ETP_REQUIRE(out, "null pointer");
  hipGraph_t g = nullptr;
  hipGraphExec_t e = nullptr;
  long nk = 0, ne = 0;
  const int rc = rec_end(&g, &e, &nk, &ne);
  if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
  if (n_kernels) *n_kernels = nk;
  if (n_edges) *n_edges = ne;
  *out = new etp_graph{g, e};
  return ETP_OK;
}
int etp_rec_abort(void) { rec_abort(); return ETP_OK; }
int etp_graph_launch(etp_graph* g, etp_stream_t s) {
  ETP_REQUIRE(g, "null graph");
  ETP_CHECK_HIP(hipGraphLaunch(g->exec, (hipStream_t)s));
  return ETP_OK;
}
int etp_graph_destroy(etp_graph* g) {
  if (!g) return ETP_OK;
  (void)hipGraphExecDestroy(g->exec);
  (void)hipGraphDestroy(g->graph);
  delete g;
  return ETP_OK;
}
int etp_memset_async(void* p, int value, int64_t bytes, etp_stream_t s) {
  ETP_REQUIRE(p && bytes >= 0, "bad arguments");
  if (value == 0 && bytes % 4 == 0 && (uintptr_t)p % 16 == 0)      // zeroing (loss, gradient arena): our own kernel, no runtime blit
    return zero_f32(reinterpret_cast<float*>(p), bytes / 4, (hipStream_t)s);
  ETP_CHECK_HIP(memset_async(p, value, (size_t)bytes, (hipStream_t)s));
  return ETP_OK;
}
int etp_graph_time(etp_graph* g, etp_stream_t s, int iters, float* ms_out) {
  ETP_REQUIRE(g && ms_out && iters > 0, "bad arguments");
  hipEvent_t a, b;
  ETP_CHECK_HIP(hipEventCreate(&a));
  ETP_CHECK_HIP(hipEventCreate(&b));
  ETP_CHECK_HIP(hipEventRecord(a, (hipStream_t)s));
  for (int i = 0; i < iters; ++i) ETP_CHECK_HIP(hipGraphLaunch(g->exec, (hipStream_t)s));
  ETP_CHECK_HIP(hipEventRecord(b, (hipStream_t)s));
  ETP_CHECK_HIP(hipEventSynchronize(b));
  ETP_CHECK_HIP(hipEventElapsedTime(ms_out, a, b));
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  return ETP_OK;
}

int etp_ktime_enable(int on) { ktime_enable(on != 0); return ETP_OK; }
int etp_ktime_reset(void) { ktime_reset(); return ETP_OK; }
int64_t etp_ktime_report(char* buf, int64_t cap) { return (buf && cap > 0) ? ktime_report(buf, (long)cap) : 0; }
int etp_gemm_probe_enable(uint64_t* dev_buf, int64_t max_launches) {
  ETP_REQUIRE(dev_buf == nullptr || max_launches > 0, "max_launches must be positive");
  gemm_probe_set(reinterpret_cast<unsigned long long*>(dev_buf), (long)max_launches);
  return ETP_OK;
}
int64_t etp_gemm_probe_count(void) { return gemm_probe_count(); }
int etp_gemm_probe_meta(int64_t i, char* name, int cap, int32_t* dims) { return gemm_probe_meta((long)i, name, cap, dims); }
int etp_prof_enable(int on) { prof_enable(on != 0); return ETP_OK; }
int etp_prof_reset(void) { prof_reset(); return ETP_OK; }
int etp_prof_filter(const char* name_part) { prof_filter(name_part); return ETP_OK; }
int etp_prof_report(etp_prof_entry* out, int cap) {
  if (!out || cap <= 0) return 0;
  return prof_report(out, cap);
}
int etp_stamp_sink(uint64_t* dev_buf, int64_t cap) {
  ETP_REQUIRE(dev_buf == nullptr || cap > 0, "cap must be positive");
  g_stamp_buf = reinterpret_cast<unsigned long long*>(dev_buf);
  g_stamp_cap = dev_buf ? (long)cap : 0;
  g_stamp_tags.clear();
  return ETP_OK;
}
int64_t etp_stamp_count(void) { return (int64_t)g_stamp_tags.size(); }
int etp_stamp_tag(int64_t i) { return (i >= 0 && i < (int64_t)g_stamp_tags.size()) ? g_stamp_tags[(size_t)i] : -1; }
int etp_stamp_mark(etp_stream_t s, int tag) { stamp_mark((hipStream_t)s, tag); return ETP_OK; }
int etp_stamp(uint64_t* slot, etp_stream_t s) {
  ETP_REQUIRE(slot, "null pointer");
  ETP_LAUNCH(stamp_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, reinterpret_cast<unsigned long long*>(slot));
  ETP_CHECK_LAUNCH("stamp");
  return ETP_OK;
}

}  // extern "C"
