// Planner engine: forward/backward of forward_txt / forward_panorama / forward_navigation
// (vlnce_baselines/models/etp/vilmodel_cmt.py:684-750) over one flat fp32 parameter arena.
//
// Host-side orchestration only: every product is an MFMA GEMM launch (gemm.hip) with a fused epilogue,
// everything else a row kernel (norm.hip / embed.hip).  No allocation and no synchronisation happen here, so a
// whole training step can be captured into one hipGraph by the caller.
//
// HBM layout
//   params  fp32 arena  [ GEMM weight matrices | vectors & small projections | embedding tables ]
//   shadow  bf16 copy of the leading matrix region (same element offsets), refreshed once per step
//   grads   fp32 arena with the same offsets (accumulated: wgrad RMW / split-K atomics, row-kernel atomics)
//   q/k/v weights of one attention are adjacent so QKV (self) and KV (cross) projections are single GEMMs.
//   stash   per forward call: saved activations in compute dtype T (+ fp32 LN statistics)
//   ws      per backward call: gradient scratch reused across layers
#include <string.h>

#include <algorithm>
#include <functional>
#include <vector>

#include "kernels.h"

namespace etp {

struct AttnP { int qkv_w, qkv_b, o_w, o_b, ln_g, ln_b; };
struct FfnP { int i_w, i_b, o_w, o_b, ln_g, ln_b; };
struct TxtLayerP { AttnP att; FfnP ffn; };
struct PanoLayerP { int in_w, in_b, out_w, out_b, l1_w, l1_b, l2_w, l2_b, n1_g, n1_b, n2_g, n2_b; };
struct XLayerP { int q_w, q_b, kv_w, kv_b, xo_w, xo_b, xln_g, xln_b; AttnP self; FfnP ffn;
                 AttnP lself; FfnP lffn; };      // language side (forward_lang2visn), pre-training variant only

struct PInfo { std::string name; int ndim; long shape[2]; int region; long offset; long numel; };

}  // namespace etp

struct etp_planner {
  etp_config cfg;
  std::vector<etp::PInfo> params;
  long total = 0, n_matrix = 0;
  // named indices
  int word, pos, type, emb_g, emb_b;
  std::vector<etp::TxtLayerP> txt;
  int img_w, img_b, img_g, img_bb, dep_w, dep_b, dep_g, dep_bb, loc_w, loc_b, loc_g, loc_bb, nav_emb, pe_g, pe_b;
  std::vector<etp::PanoLayerP> pano;
  int pn_g, pn_b;
  int gpos_w, gpos_b, gpos_g, gpos_bb, step_emb, sp_w, sp_b;
  std::vector<etp::XLayerP> xl;
  int sap0_w, sap0_b, sap2_g, sap2_b, sap4_w, sap4_b;
  int mlm_w, mlm_b, mlm_g, mlm_bb, mlm_vb;       // MLM head (pre-training variant), -1 otherwise
  // bound arenas
  float* P = nullptr; void* S = nullptr; float* G = nullptr;
  // training-mode dropout (0 = eval): hidden / attention-probability / RGB-feature ("drop_env") rates and the step seed
  float p_hidden = 0.f, p_attn = 0.f, p_env = 0.f, p_head = 0.f;
  // 1: etp_nav_bwd / etp_pano_bwd leave their weight-gradient GEMMs running on the aux stream instead of joining them
  // before returning; the caller joins later (etp_txt_bwd* always joins, or etp_planner_join_aux)
  int lazy_join = 0;          // 0: every backward entry point joins; 1: nav / pano leave their leaf work running; 2: also text ranges that stop above layer 0
  // 1: weight-gradient GEMMs STORE into the matrix region of the gradient arena instead of accumulating (no fp32 read of
  // C, and the caller zeroes only the vector/table tail [n_matrix, total) per step).  Valid when every matrix is touched by
  // exactly one weight-gradient product between two optimizer steps (one planner step per optimizer step).
  bool grad_overwrite = false;
  uint64_t drop_seed = 0;
  // optional second stream: weight-gradient GEMMs (leaves of the backward graph) run beside the dgrad chain
  hipStream_t aux = nullptr;
  // optional third stream: the d(txt_embeds) contributions of the x-layers' text K/V projections (M = B*L rows, the largest
  // GEMMs of the navigation backward) form a serial accumulate chain of their own that nothing on the node chain reads
  hipStream_t aux2 = nullptr;
  // etp_planner_refresh_text_split: event after which the bf16 shadow of text layers >= 1 is valid (consumed once by the next
  // etp_txt_fwd, after it has enqueued layer 0)
  hipEvent_t txt_w_ready = nullptr;
  bool txt_w_pending = false;
  // etp_nav_bwd under lazy joins (round 6): d txt_embeds is complete on the aux2 stream at this event; the consumers -- the text backward,
  // etp_planner_join_aux -- wait for it, the navigation backward itself no longer does (its tail and the node-assembly backward overlap it)
  hipEvent_t dtxt_ready = nullptr;
  std::vector<hipStream_t> dtxt_waiters;   // main streams whose etp_nav_bwd deferred the join and that have not consumed it yet
  bool dtxt_owed(hipStream_t st) {         // true once per deferral of `st`
    for (size_t i = 0; i < dtxt_waiters.size(); ++i)
      if (dtxt_waiters[i] == st) { dtxt_waiters.erase(dtxt_waiters.begin() + i); return true; }
    return false;
  }
  std::vector<hipEvent_t> events;
  size_t ev_next = 0;
  hipEvent_t next_event() {
    if (events.empty()) {
      events.resize(256);
      for (auto& e : events) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    }
    hipEvent_t e = events[ev_next];
    ev_next = (ev_next + 1) % events.size();
    return e;
  }

  long off(int i) const { return params[i].offset; }
  const float* pf(int i) const { return P + params[i].offset; }          // fp32 parameter
  float* gf(int i) const { return G + params[i].offset; }                // fp32 gradient
  const void* pw(int i) const {                                          // GEMM operand in compute dtype
    if (cfg.dtype == ETP_BF16) return reinterpret_cast<const uint16_t*>(S) + params[i].offset;
    return P + params[i].offset;
  }
};

namespace etp {

static int add_param(etp_planner* pl, const std::string& name, long r, long c, int region) {
  PInfo p;
  p.name = name; p.ndim = c > 0 ? 2 : 1; p.shape[0] = r; p.shape[1] = c > 0 ? c : 0; p.region = region;
  p.numel = c > 0 ? r * c : r; p.offset = -1;
  pl->params.push_back(p);
  return (int)pl->params.size() - 1;
}

// region 0: GEMM matrices, 1: vectors / small, 2: embedding tables
static void build_layout(etp_planner* pl) {
  const etp_config& c = pl->cfg;
  const long H = c.hidden, I = c.inter;
  auto mat = [&](const std::string& n, long r, long k) { return add_param(pl, n, r, k, 0); };
  auto vec = [&](const std::string& n, long r) { return add_param(pl, n, r, 0, 1); };
  auto small = [&](const std::string& n, long r, long k) { return add_param(pl, n, r, k, 1); };
  auto table = [&](const std::string& n, long r, long k) { return add_param(pl, n, r, k, 2); };

  pl->word = table("embeddings.word_embeddings.weight", c.vocab, H);
  pl->pos = table("embeddings.position_embeddings.weight", c.max_pos, H);
  pl->type = table("embeddings.token_type_embeddings.weight", c.type_vocab, H);
  pl->emb_g = vec("embeddings.LayerNorm.weight", H);
  pl->emb_b = vec("embeddings.LayerNorm.bias", H);

  auto self_att = [&](const std::string& p) {
    AttnP a;
    a.qkv_w = mat(p + ".self.query.weight", H, H);
    mat(p + ".self.key.weight", H, H);
    mat(p + ".self.value.weight", H, H);
    a.qkv_b = vec(p + ".self.query.bias", H);
    vec(p + ".self.key.bias", H);
    vec(p + ".self.value.bias", H);
    a.o_w = mat(p + ".output.dense.weight", H, H);
    a.o_b = vec(p + ".output.dense.bias", H);
    a.ln_g = vec(p + ".output.LayerNorm.weight", H);
    a.ln_b = vec(p + ".output.LayerNorm.bias", H);
    return a;
  };
  auto ffn = [&](const std::string& pi, const std::string& po) {
    FfnP f;
    f.i_w = mat(pi + ".dense.weight", I, H);
    f.i_b = vec(pi + ".dense.bias", I);
    f.o_w = mat(po + ".dense.weight", H, I);
    f.o_b = vec(po + ".dense.bias", H);
    f.ln_g = vec(po + ".LayerNorm.weight", H);
    f.ln_b = vec(po + ".LayerNorm.bias", H);
    return f;
  };
  for (int l = 0; l < c.n_l; ++l) {
    const std::string p = "lang_encoder.layer." + std::to_string(l);
    TxtLayerP t;
    t.att = self_att(p + ".attention");
    t.ffn = ffn(p + ".intermediate", p + ".output");
    pl->txt.push_back(t);
  }
  const std::string e = "img_embeddings";
  pl->img_w = mat(e + ".img_linear.weight", H, c.img_feat);
  pl->img_b = vec(e + ".img_linear.bias", H);
  pl->img_g = vec(e + ".img_layer_norm.weight", H);
  pl->img_bb = vec(e + ".img_layer_norm.bias", H);
  pl->loc_w = small(e + ".loc_linear.weight", H, c.ang_feat);
  pl->loc_b = vec(e + ".loc_linear.bias", H);
  pl->loc_g = vec(e + ".loc_layer_norm.weight", H);
  pl->loc_bb = vec(e + ".loc_layer_norm.bias", H);
  if (c.use_depth) {
    pl->dep_w = mat(e + ".dep_linear.weight", H, c.dep_feat);
    pl->dep_b = vec(e + ".dep_linear.bias", H);
    pl->dep_g = vec(e + ".dep_layer_norm.weight", H);
    pl->dep_bb = vec(e + ".dep_layer_norm.bias", H);
  } else {
    pl->dep_w = pl->dep_b = pl->dep_g = pl->dep_bb = -1;
  }
  pl->nav_emb = table(e + ".nav_type_embedding.weight", 2, H);
  pl->pe_g = vec(e + ".layer_norm.weight", H);
  pl->pe_b = vec(e + ".layer_norm.bias", H);
  for (int l = 0; l < c.n_p; ++l) {
    const std::string p = e + ".pano_encoder.layers." + std::to_string(l);
    PanoLayerP q;
    q.in_w = mat(p + ".self_attn.in_proj_weight", 3 * H, H);
    q.in_b = vec(p + ".self_attn.in_proj_bias", 3 * H);
    q.out_w = mat(p + ".self_attn.out_proj.weight", H, H);
    q.out_b = vec(p + ".self_attn.out_proj.bias", H);
    q.l1_w = mat(p + ".linear1.weight", I, H);
    q.l1_b = vec(p + ".linear1.bias", I);
    q.l2_w = mat(p + ".linear2.weight", H, I);
    q.l2_b = vec(p + ".linear2.bias", H);
    q.n1_g = vec(p + ".norm1.weight", H);
    q.n1_b = vec(p + ".norm1.bias", H);
    q.n2_g = vec(p + ".norm2.weight", H);
    q.n2_b = vec(p + ".norm2.bias", H);
    pl->pano.push_back(q);
  }
  if (c.n_p > 0) {
    pl->pn_g = vec(e + ".pano_encoder.norm.weight", H);
    pl->pn_b = vec(e + ".pano_encoder.norm.bias", H);
  } else {
    pl->pn_g = pl->pn_b = -1;
  }
  const std::string g = "global_encoder";
  pl->gpos_w = small(g + ".gmap_pos_embeddings.0.weight", H, c.ang_feat + 3);
  pl->gpos_b = vec(g + ".gmap_pos_embeddings.0.bias", H);
  pl->gpos_g = vec(g + ".gmap_pos_embeddings.1.weight", H);
  pl->gpos_bb = vec(g + ".gmap_pos_embeddings.1.bias", H);
  pl->step_emb = table(g + ".gmap_step_embeddings.weight", c.max_steps, H);
  for (int l = 0; l < c.n_x; ++l) {
    const std::string p = g + ".encoder.x_layers." + std::to_string(l);
    XLayerP x;
    x.self = self_att(p + ".visn_self_att");
    x.ffn = ffn(p + ".visn_inter", p + ".visn_output");
    x.q_w = mat(p + ".visual_attention.att.query.weight", H, H);
    x.q_b = vec(p + ".visual_attention.att.query.bias", H);
    x.kv_w = mat(p + ".visual_attention.att.key.weight", H, H);
    mat(p + ".visual_attention.att.value.weight", H, H);
    x.kv_b = vec(p + ".visual_attention.att.key.bias", H);
    vec(p + ".visual_attention.att.value.bias", H);
    x.xo_w = mat(p + ".visual_attention.output.dense.weight", H, H);
    x.xo_b = vec(p + ".visual_attention.output.dense.bias", H);
    x.xln_g = vec(p + ".visual_attention.output.LayerNorm.weight", H);
    x.xln_b = vec(p + ".visual_attention.output.LayerNorm.bias", H);
    if (c.use_lang2visn) {     // GraphLXRTXLayer.__init__ pretrain vilmodel.py:371-376
      x.lself = self_att(p + ".lang_self_att");
      x.lffn = ffn(p + ".lang_inter", p + ".lang_output");
    }
    pl->xl.push_back(x);
  }
  if (c.use_sprels) {
    pl->sp_w = small(g + ".sprel_linear.weight", 1, 1);
    pl->sp_b = vec(g + ".sprel_linear.bias", 1);
  } else {
    pl->sp_w = pl->sp_b = -1;
  }
  pl->sap0_w = mat("global_sap_head.net.0.weight", H, H);
  pl->sap0_b = vec("global_sap_head.net.0.bias", H);
  pl->sap2_g = vec("global_sap_head.net.2.weight", H);
  pl->sap2_b = vec("global_sap_head.net.2.bias", H);
  pl->sap4_w = small("global_sap_head.net.4.weight", 1, H);
  pl->sap4_b = vec("global_sap_head.net.4.bias", 1);
  if (c.use_lang2visn) {       // BertOnlyMLMHead vilmodel.py:258-299; the decoder weight is the word-embedding table (tied)
    pl->mlm_w = mat("mlm_head.predictions.transform.dense.weight", H, H);
    pl->mlm_b = vec("mlm_head.predictions.transform.dense.bias", H);
    pl->mlm_g = vec("mlm_head.predictions.transform.LayerNorm.weight", H);
    pl->mlm_bb = vec("mlm_head.predictions.transform.LayerNorm.bias", H);
    pl->mlm_vb = vec("mlm_head.predictions.bias", c.vocab);
  } else {
    pl->mlm_w = pl->mlm_b = pl->mlm_g = pl->mlm_bb = pl->mlm_vb = -1;
  }

  long off = 0;
  for (int region = 0; region < 3; ++region) {
    for (auto& p : pl->params)
      if (p.region == region) { p.offset = off; off += round_up(p.numel, 64); }
    if (region == 0) pl->n_matrix = off;
  }
  pl->total = off;
}

// ---- bump allocator over caller-provided stash / workspace -----------------------------
struct Bump {
  char* base; size_t off = 0;
  explicit Bump(void* b) : base(reinterpret_cast<char*>(b)) {}
  void* take(size_t bytes) {
    void* p = base ? base + off : nullptr;
    off += (bytes + 255) / 256 * 256;
    return p;
  }
};

struct Ctx {
  etp_planner* pl; hipStream_t st; int dt; size_t es;  // element size of T
  int H, I, nh;
  hipStream_t sw;                                      // stream of the weight-gradient launches (== st when no aux stream)
  hipStream_t s3;                                      // stream of the d_txt accumulate chain (== st when no aux2 stream)
  // Deferred side-stream launches: leaves (weight gradients, the d_txt contribution) are collected and forked once per
  // layer by flush_side() instead of once per GEMM -- a quarter of the event-record marker packets in the main queue
  // and of the host calls.  (Under rocprofv3 the kernel behind each marker started ~13 us late, tools/timeline.py;
  // unprofiled the step time is unchanged within noise, 5.21 -> 5.20 ms.)
  std::vector<std::function<int()>>* pend;
  // Weight-gradient products collected since the last flush: launched as ONE grouped grid per flush (gemm_group_kernel) --
  // a layer's four to seven small TN products fill the chip together, which none of them does alone.
  std::vector<GemmArgs>* wq;
};
// dropout sites: (entry point, layer, slot) -> independent mask streams
enum { SITE_EMBED = 0, SITE_ATT_P = 1, SITE_ATT_O = 2, SITE_FFN_O = 3, SITE_FFN_I = 4, SITE_X_P = 5, SITE_X_O = 6, SITE_HEAD = 7,
       SITE_ENV = 8 };
enum { MODE_TXT = 1, MODE_PANO = 2, MODE_NAV = 3 };
static inline Drop site(const Ctx& c, float p, int mode, int layer, int slot) {
  return p > 0.f ? drop_site(p, c.pl->drop_seed, (uint32_t)(mode << 16 | layer << 4 | slot)) : drop_none();
}
static inline Drop hid(const Ctx& c, int mode, int layer, int slot) { return site(c, c.pl->p_hidden, mode, layer, slot); }
static inline Drop att(const Ctx& c, int mode, int layer, int slot) { return site(c, c.pl->p_attn, mode, layer, slot); }

static Ctx make_ctx(etp_planner* pl, etp_stream_t s) {
  Ctx c;
  c.pl = pl; c.st = reinterpret_cast<hipStream_t>(s); c.dt = pl->cfg.dtype; c.es = dtype_size(c.dt);
  c.sw = (pl->aux != nullptr && pl->aux != c.st) ? pl->aux : c.st;
  c.s3 = (pl->aux2 != nullptr && pl->aux2 != c.st) ? pl->aux2 : c.st;
  c.H = pl->cfg.hidden; c.I = pl->cfg.inter; c.nh = pl->cfg.heads;
  c.pend = nullptr;
  c.wq = nullptr;
  return c;
}

// order `to` after everything enqueued so far on `from` (capturable: becomes a graph edge)
static int stream_after(etp_planner* pl, hipStream_t from, hipStream_t to) {
  if (from == to) return ETP_OK;
  hipEvent_t e = pl->next_event();
  ETP_CHECK_HIP(event_record(e, from));
  ETP_CHECK_HIP(stream_wait_event(to, e));
  return ETP_OK;
}
static int flush_side(const Ctx& c) {
  const bool have_w = c.wq && !c.wq->empty(), have_p = c.pend && !c.pend->empty();
  if (!have_w && !have_p) return ETP_OK;
  ETP_TRY(stream_after(c.pl, c.st, c.sw));            // one fork for everything collected since the last flush
  if (have_w) {
    for (size_t i = 0; i < c.wq->size(); i += ETP_GEMM_GROUP_MAX) {
      const int n = (int)std::min<size_t>(ETP_GEMM_GROUP_MAX, c.wq->size() - i);
      ETP_TRY(launch_gemm_group(c.dt, ETP_F32, 1, 1, c.wq->data() + i, n, c.sw));
    }
    c.wq->clear();
  }
  if (have_p) {
    for (auto& f : *c.pend) ETP_TRY(f());
    c.pend->clear();
  }
  return ETP_OK;
}
// run `f` on the side stream after everything enqueued on the main stream so far (now, or at the next flush_side)
static int on_side(const Ctx& c, std::function<int()> f) {
  if (c.pend) { c.pend->push_back(std::move(f)); return ETP_OK; }
  ETP_TRY(stream_after(c.pl, c.st, c.sw));
  return f();
}
static int join_wgrads(const Ctx& c) {
  ETP_TRY(flush_side(c));
  return stream_after(c.pl, c.sw, c.st);
}
// end of a backward entry point whose weight gradients nobody downstream of it reads (navigation, panorama)
static int finish_wgrads(const Ctx& c) {
  if (c.pl->lazy_join) return flush_side(c);
  return join_wgrads(c);
}

static GemmArgs base_args() {
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.nb_inner = 1; g.ksplit = 1; g.alpha = 1.f;
  return g;
}

// Y[M,N] = act(X[M,K] . W[N,K]^T + b) (+R)
static int linear_fwd(const Ctx& c, const void* X, long ldx, int wi, int bi, void* Y, long ldy, int M, int N, int K, int act,
                      void* Z, const void* R, long ldr, Drop drop = drop_none()) {
  GemmArgs g = base_args();
  g.drop = drop;
  g.A = X; g.lda = ldx; g.B = c.pl->pw(wi); g.ldb = K; g.C = Y; g.ldc = ldy;
  g.M = M; g.N = N; g.K = K;
  g.bias = bi >= 0 ? c.pl->pf(bi) : nullptr;
  g.act = act; g.Z = Z; g.ldz = ldy; g.R = R; g.ldr = ldr;
  return launch_gemm(c.dt, c.dt, 0, 0, g, 1, c.st);
}
// dX[M,K] = act_bwd(dY[M,N] . W[N,K]) (+R)      (B operand = W stored [N (reduction)][K])
static int linear_dgrad(const Ctx& c, const void* dY, long ldy, int wi, void* dX, long ldx, int M, int N, int K, int act,
                        void* Z, long ldz, const void* R, long ldr, int out_mode = 0, Drop drop = drop_none()) {
  GemmArgs g = base_args();
  g.drop = drop;
  g.A = dY; g.lda = ldy; g.B = c.pl->pw(wi); g.ldb = K; g.C = dX; g.ldc = ldx;
  g.M = M; g.N = K; g.K = N;
  g.act = act; g.Z = Z; g.ldz = ldz; g.R = R; g.ldr = ldr; g.out_mode = out_mode;
  return launch_gemm(c.dt, c.dt, 0, 1, g, 1, c.st);
}
// stream-output variants: C (and the residual R) are fp32 whatever the operand dtype -- the residual stream never
// drops to bf16 (mirrors autocast: half-precision GEMMs, fp32 residual adds and LayerNorms)
static int linear_fwd_s(const Ctx& c, const void* X, long ldx, int wi, int bi, float* Y, int M, int N, int K, const float* R,
                        Drop drop = drop_none()) {
  GemmArgs g = base_args();
  g.drop = drop;
  g.A = X; g.lda = ldx; g.B = c.pl->pw(wi); g.ldb = K; g.C = Y; g.ldc = N;
  g.M = M; g.N = N; g.K = K;
  g.bias = bi >= 0 ? c.pl->pf(bi) : nullptr;
  g.R = R; g.ldr = N;
  return launch_gemm(c.dt, ETP_F32, 0, 0, g, 1, c.st);
}
static int linear_dgrad_s(const Ctx& c, const void* dY, long ldy, int wi, float* dX, int M, int N, int K, const float* R,
                          int out_mode = 0, Drop drop = drop_none()) {
  GemmArgs g = base_args();
  g.drop = drop;
  g.A = dY; g.lda = ldy; g.B = c.pl->pw(wi); g.ldb = K; g.C = dX; g.ldc = K;
  g.M = M; g.N = K; g.K = N;
  g.R = R; g.ldr = K; g.out_mode = out_mode;
  return launch_gemm(c.dt, ETP_F32, 0, 1, g, 1, c.st);
}
// dW[N,K] += dY[M,N]^T . X[M,K] ; db[N] += colsum(dY)
static int linear_wgrad(const Ctx& c, const void* dY, long ldy, const void* X, long ldx, int wi, int bi, int M, int N, int K) {
  // MEASUREMENT ONLY (tools/r03_call4.sh): ETP_SKIP_WGRAD=1 drops every weight-gradient product, i.e. leaves the dependent chain
  // alone on the GPU -- the step time then shows what the leaf work costs the chain (the gradients are wrong in that mode)
#ifdef ETP_EXPERIMENTS       // measurement builds only (tools/build_variant.sh ... -DETP_EXPERIMENTS): never in the shipped library
  const bool skip = opt_int(OPT_SKIP_WGRAD, 0) == 1;
  if (skip) return ETP_OK;
#endif
  GemmArgs g = base_args();
  g.A = dY; g.lda = ldy; g.B = X; g.ldb = ldx; g.C = c.pl->gf(wi); g.ldc = K;
  g.M = N; g.N = K; g.K = M;
  // matrix-region gradients may be first-touch stores (etp_planner_set_grad_overwrite); tables (the tied MLM decoder adds
  // into the word-embedding gradient) always accumulate
  const bool store = c.pl->grad_overwrite && c.pl->params[wi].region == 0;
  const bool group_on = opt_on(OPT_WGRAD_GROUP, true);
  if (c.wq && group_on && gemm_uses_dma(c.dt, M, 1)) {
    // grouped path: whole token reduction in one tile pass (no split-K: the group as a whole fills the chip), bias
    // gradient fused as column sums of the dY tile
    g.out_mode = store ? 0 : 1;
    if (bi >= 0) g.a_colsum = c.pl->gf(bi);
    c.wq->push_back(g);
    return ETP_OK;
  }
  // split the token reduction (atomic fp32 accumulate) only when the weight alone badly under-fills 256 CUs and the
  // reduction is long; measured on MI355X (tools/gemm_bench.py): dW[768,768] over 2560 tokens 21 -> 17 us with 4 splits,
  // every other planner shape is fastest with a single read-modify-write pass.
  const int bk = c.dt == ETP_BF16 ? 64 : 32;
  const long tiles = (long)((N + 63) / 64) * ((K + 63) / 64);
  int ks = (tiles <= 144 && M >= 2048 && M % (4 * bk) == 0) ? 4 : 1;
  if (store) ks = 1;
  g.ksplit = ks;
  g.out_mode = ks > 1 ? 2 : (store ? 0 : 1);
  const bool fuse_bias = bi >= 0 && gemm_uses_dma(c.dt, M, ks);     // bias gradient rides in the wgrad kernel
  if (fuse_bias) g.a_colsum = c.pl->gf(bi);
  // weight gradients are leaves of the backward graph: issue them on the side stream, after dY's producer
  const int dt = c.dt;
  hipStream_t sw = c.sw;
  float* db = (bi >= 0 && !fuse_bias) ? c.pl->gf(bi) : nullptr;
  return on_side(c, [=]() -> int {
    ETP_TRY(launch_gemm(dt, ETP_F32, 1, 1, g, 1, sw));
    if (db) ETP_TRY(colsum(dt, dY, ldy, db, M, N, sw));
    return ETP_OK;
  });
}

// ---- attention (head dim 64, heads interleaved in the row) ------------------------------
static inline const void* offs(const void* p, long elems, size_t es) { return reinterpret_cast<const char*>(p) + elems * es; }
static inline void* offs(void* p, long elems, size_t es) { return reinterpret_cast<char*>(p) + elems * es; }

int attn_fwd_impl(int dt, int nh, const AttnBuf& a, void* P, void* ctx, long ldc, float alpha, hipStream_t st, Drop drop) {
  const int dh = 64;
  if (attn_rows_ok(dt, a, ldc)) return attn_rows_fwd(nh, a, P, ctx, ldc, alpha, st, drop);
  ETP_REQUIRE(a.kv_mod == 0, "per-episode K/V indirection (AttnBuf::kv_mod) needs the register-resident attention kernels (bf16, axes <= 128)");
  if (attn_fused_ok(dt, a, ldc)) return attn_fused_fwd(dt, nh, a, P, ctx, ldc, alpha, st, drop);
  if (attn_flash_ok(dt, a, ldc)) return attn_flash_fwd(nh, a, P, ctx, ldc, alpha, st, drop);
  ETP_REQUIRE(drop.p == 0.f || a.Pd, "attention dropout on the unfused path needs the second probability buffer (AttnBuf::Pd)");
  GemmArgs g = base_args();
  // S = alpha * Q K^T
  g.A = a.Q; g.lda = a.ldq; g.sAo = (long)a.Lq * a.ldq; g.sAi = dh;
  g.B = a.K; g.ldb = a.ldk; g.sBo = (long)a.Lk * a.ldk; g.sBi = dh;
  g.C = P; g.ldc = a.ldS; g.sCo = (long)nh * a.Lq * a.ldS; g.sCi = (long)a.Lq * a.ldS;
  g.M = a.Lq; g.N = a.Lk; g.K = dh; g.nb_inner = nh; g.alpha = alpha;
  ETP_TRY(launch_gemm(dt, dt, 0, 0, g, a.B * nh, st));
  ETP_TRY(softmax_fwd(dt, P, a.keymask, a.dist, a.sp_w, a.sp_b, a.B, nh, a.Lq, a.Lk, a.ldS, a.mask_mode, st));
  // ctx = dropout(P) V: P stays undropped for the softmax backward, the dropped copy feeds P.V here and dV in backward
  const void* Pv = P;
  if (drop.p > 0.f) {
    ETP_TRY(drop_rows(dt, P, a.Pd, (long)a.B * nh * a.Lq, a.Lk, a.ldS, drop, st));
    Pv = a.Pd;
  }
  GemmArgs h = base_args();
  h.A = Pv; h.lda = a.ldS; h.sAo = g.sCo; h.sAi = g.sCi;
  h.B = a.V; h.ldb = a.ldv; h.sBo = (long)a.Lk * a.ldv; h.sBi = dh;
  h.C = ctx; h.ldc = ldc; h.sCo = (long)a.Lq * ldc; h.sCi = dh;
  h.M = a.Lq; h.N = dh; h.K = a.Lk; h.nb_inner = nh;
  return launch_gemm(dt, dt, 0, 1, h, a.B * nh, st);
}

int attn_bwd_impl(int dt, int nh, const AttnBuf& a, const void* P, const void* dctx, long ldd, void* dP, void* dQ, long lddq,
                  void* dK, long lddk, void* dV, long lddv, float alpha, float* d_sp_w, float* d_sp_b, hipStream_t st,
                  Drop drop) {
  const int dh = 64;
  {
    const int epc = dt == ETP_BF16 ? 8 : 4;
    // Which kernel family ran the FORWARD of this shape decides what the P buffer holds: the register-resident and the
    // streaming kernels leave only lse (+ D) there, the LDS-tile kernel and the batched-GEMM path leave probabilities.  The
    // backward must therefore take the same family or fail loudly -- falling through to the batched-GEMM path would read lse
    // as probabilities and return garbage without an error.  The forward's choice is re-derived from the same
    // predicate with the forward's ctx leading dimension (= ldo of the saved output when the caller passes it, else ldd).
    const long ldc_f = a.O != nullptr ? a.ldo : ldd;
    const bool dal = lddq % epc == 0 && lddk % epc == 0 && lddv % epc == 0;
    if (attn_rows_ok(dt, a, ldc_f)) {
      ETP_REQUIRE(ldd % 8 == 0 && dal, "the forward of this shape kept lse only (register-resident kernels): the backward needs "
                                       "16-byte-aligned dctx / dQ / dK / dV rows");
      return attn_rows_bwd(nh, a, P, dctx, ldd, dQ, lddq, dK, lddk, dV, lddv, alpha, d_sp_w, d_sp_b, st, drop);
    }
    ETP_REQUIRE(a.kv_mod == 0, "per-episode K/V indirection (AttnBuf::kv_mod) needs the register-resident attention kernels (bf16, axes <= 128)");
    if (attn_fused_ok(dt, a, ldc_f)) {
      if (attn_fused_ok(dt, a, ldd) && dal)
        return attn_fused_bwd(dt, nh, a, P, dctx, ldd, dQ, lddq, dK, lddk, dV, lddv, alpha, d_sp_w, d_sp_b, st, drop);
      // (the LDS-tile forward saved probabilities: the batched-GEMM backward below can read them)
    } else if (attn_flash_ok(dt, a, ldc_f)) {
      ETP_REQUIRE(a.O != nullptr && a.ldo % epc == 0 && ldd % epc == 0 && dal,
                  "the forward of this shape kept lse only (streaming kernels): the backward needs the forward output (ctx) and "
                  "16-byte-aligned dctx / dQ / dK / dV rows");
      return attn_flash_bwd(nh, a, P, dctx, ldd, dQ, lddq, dK, lddk, dV, lddv, alpha, st, drop);
    }
  }
  ETP_REQUIRE(drop.p == 0.f || a.Pd, "attention dropout on the unfused path needs the second probability buffer (AttnBuf::Pd)");
  const long sPo = (long)nh * a.Lq * a.ldS, sPi = (long)a.Lq * a.ldS;
  // dP = dctx V^T
  GemmArgs g = base_args();
  g.A = dctx; g.lda = ldd; g.sAo = (long)a.Lq * ldd; g.sAi = dh;
  g.B = a.V; g.ldb = a.ldv; g.sBo = (long)a.Lk * a.ldv; g.sBi = dh;
  g.C = dP; g.ldc = a.ldS; g.sCo = sPo; g.sCi = sPi;
  g.M = a.Lq; g.N = a.Lk; g.K = dh; g.nb_inner = nh;
  ETP_TRY(launch_gemm(dt, dt, 0, 0, g, a.B * nh, st));
  if (drop.p > 0.f) ETP_TRY(drop_rows(dt, dP, dP, (long)a.B * nh * a.Lq, a.Lk, a.ldS, drop, st));   // dP = dP_dropped * mask/(1-p)
  // dV = dropout(P)^T dctx
  GemmArgs v = base_args();
  v.A = drop.p > 0.f ? a.Pd : P; v.lda = a.ldS; v.sAo = sPo; v.sAi = sPi;
  v.B = dctx; v.ldb = ldd; v.sBo = (long)a.Lq * ldd; v.sBi = dh;
  v.C = dV; v.ldc = lddv; v.sCo = (long)a.Lk * lddv; v.sCi = dh;
  v.M = a.Lk; v.N = dh; v.K = a.Lq; v.nb_inner = nh;
  ETP_TRY(launch_gemm(dt, dt, 1, 1, v, a.B * nh, st));
  // dS = P * (dP - rowsum(dP*P))
  ETP_TRY(softmax_bwd(dt, P, dP, a.dist, d_sp_w, d_sp_b, a.B, nh, a.Lq, a.Lk, a.ldS, st));
  // dQ = alpha dS K
  GemmArgs q = base_args();
  q.A = dP; q.lda = a.ldS; q.sAo = sPo; q.sAi = sPi;
  q.B = a.K; q.ldb = a.ldk; q.sBo = (long)a.Lk * a.ldk; q.sBi = dh;
  q.C = dQ; q.ldc = lddq; q.sCo = (long)a.Lq * lddq; q.sCi = dh;
  q.M = a.Lq; q.N = dh; q.K = a.Lk; q.nb_inner = nh; q.alpha = alpha;
  ETP_TRY(launch_gemm(dt, dt, 0, 1, q, a.B * nh, st));
  // dK = alpha dS^T Q
  GemmArgs k = base_args();
  k.A = dP; k.lda = a.ldS; k.sAo = sPo; k.sAi = sPi;
  k.B = a.Q; k.ldb = a.ldq; k.sBo = (long)a.Lq * a.ldq; k.sBi = dh;
  k.C = dK; k.ldc = lddk; k.sCo = (long)a.Lk * lddk; k.sCi = dh;
  k.M = a.Lk; k.N = dh; k.K = a.Lq; k.nb_inner = nh; k.alpha = alpha;
  return launch_gemm(dt, dt, 1, 1, k, a.B * nh, st);
}

// When to fold a projection into the register-resident attention kernels (switches ATTN_PROJ / ATTN_QKV: 0 = never, 1 = always, unset =
// the rule below).  Measured (profiles/r06_attn_fusion.txt): a fused prologue works on a smaller tile than the GEMM it replaces
// (16*n x 64 rows x columns per workgroup against 128 x 64 / 128 x 128), i.e. it pulls more operand bytes per CU through the ~35 B/clk
// L2 -> CU feed.
//   * out-projection dgrad in the backward (96 KB of weights + the block's dY rows per workgroup): with ONE workgroup per CU
//     (batch * heads <= CUs: configs 4 and 5, the SAP unit, rollout steps of <= 21 episodes) it wins about the launch it removes
//     (config 5: -1.0 ... -1.4 % of the step); with two per CU (config 2: 384 workgroups of five wavefronts at 158 registers) the
//     doubled feed costs more than the launch saved (text 80 x 80: 32.8 us against 22.2 us for the pair; step +1.7 %).  Rule: by grid.
//   * QKV projection in the forward (three times the weights: 288 KB + the block's rows per workgroup): slower than the launch pair at
//     every grid size measured (B = 8: 17.4 against 14.2 us, B = 32: 33.6 against 24.0 us; config 5 step +1.4 %).  Rule: off.
static bool fold_projection(Opt o, int workgroups, bool by_grid) {
  const char* s = opt_str(o);
  if (s) return s[0] != '0';
  return by_grid && workgroups <= cu_count();
}

// Input gradient of an attention block's out-projection + the attention backward.  Where the register-resident kernels run (bf16, both
// axes <= 128: every R2R-CE shape) AND the grid leaves every workgroup a CU of its own this is ONE launch -- the workgroup of a (batch,
// head) computes its own dctx tile from dY and the projection's weight (attn_rows.hip, PROJ) -- otherwise the GEMM into `dctx` and the
// attention backward reading it.
static int attn_bwd_proj(const Ctx& c, const AttnBuf& a, const void* P, const void* dy, int wi, void* dctx, int M, void* dP, void* dQ,
                         long lddq, void* dK, long lddk, void* dV, long lddv, float* d_sp_w, float* d_sp_b, Drop drop) {
  const int H = c.H;
  const void* W = c.pl->pw(wi);
  const int epc = 8;
  // ... or four-wavefront workgroups (both axes <= 64: panorama 36 x 36, graph self-attention 16 x 16), which co-reside two per CU without
  // the register-file lottery of the five-wavefront shapes: 15.8 against 16.9 us and 11.7 against 13.5 us at 384 workgroups
  const bool four_waves = a.Lq <= 64 && a.Lk <= 64;
  if (c.dt == ETP_BF16 && (fold_projection(OPT_ATTN_PROJ, a.B * c.nh, true) || (four_waves && fold_projection(OPT_ATTN_PROJ, 0, true))) &&
      attn_rows_ok(c.dt, a, H) && attn_rows_proj_ok(H, W, H) &&
      lddq % epc == 0 && lddk % epc == 0 && lddv % epc == 0)
    return attn_rows_bwd(c.nh, a, P, dy, H, dQ, lddq, dK, lddk, dV, lddv, 0.125f, d_sp_w, d_sp_b, c.st, drop, W, H, H);
  ETP_TRY(linear_dgrad(c, dy, H, wi, dctx, H, M, H, H, ETP_ACT_NONE, nullptr, 0, nullptr, 0));
  return attn_bwd_impl(c.dt, c.nh, a, P, dctx, H, dP, dQ, lddq, dK, lddk, dV, lddv, 0.125f, d_sp_w, d_sp_b, c.st, drop);
}

// QKV projection + attention of a SELF-attention block: the GEMM into `qkv` and the attention reading it.  With ATTN_QKV=1 (bf16, axis
// <= 128, hidden 768) ONE launch instead -- the workgroup of a (batch, head) projects its own Q / K / V rows from the block's input and
// writes them to the stash (attn_rows.hip, QKV) -- built and parity-tested in round 6, measured slower than the pair, off by default.
static int attn_fwd_qkv(const Ctx& c, const AttnBuf& a, void* P, void* ctx, const void* x, int wi, int bi, void* qkv, int M, Drop drop) {
  const int H = c.H;
  const void* W = c.pl->pw(wi);
  if (c.dt == ETP_BF16 && fold_projection(OPT_ATTN_QKV, a.B * c.nh, false) && attn_rows_ok(c.dt, a, H) && attn_rows_qkv_ok(c.nh, a, x, H, W, H))
    return attn_rows_fwd(c.nh, a, P, ctx, H, 0.125f, c.st, drop, x, H, W, H, bi >= 0 ? c.pl->pf(bi) : nullptr);
  ETP_TRY(linear_fwd(c, x, H, wi, bi, qkv, 3 * H, M, 3 * H, H, ETP_ACT_NONE, nullptr, nullptr, 0));
  return attn_fwd_impl(c.dt, c.nh, a, P, ctx, H, 0.125f, c.st, drop);
}

// ---- activations on the residual stream: fp32 tensor + (bf16 mode) a copy in the GEMM operand dtype -----------------
struct Act { float* f; void* t; };
static Act take_act(Bump& b, int dt, long n) {
  Act a;
  a.f = (float*)b.take((size_t)n * 4);
  a.t = dt == ETP_BF16 ? b.take((size_t)n * 2) : (void*)a.f;
  return a;
}
static inline void* lp(const Act& a, int dt) { return dt == ETP_BF16 ? a.t : nullptr; }   // second kernel output (or none)
// backward scratch: the operand copy is ALWAYS a separate buffer (under dropout it differs from the fp32 gradient)
static Act take_act2(Bump& b, int dt, long n) {
  Act a;
  a.f = (float*)b.take((size_t)n * 4);
  a.t = b.take((size_t)n * dtype_size(dt));
  return a;
}
// pointer the producing kernel writes the operand copy to (NULL: fp32 mode without dropout -> the fp32 tensor is the operand)
static inline void* lp2(const Ctx& c, const Act& a, const Drop& d) { return (c.dt == ETP_BF16 || d.p > 0.f) ? a.t : nullptr; }
// pointer the consuming GEMMs read
static inline const void* op2(const Ctx& c, const Act& a, const Drop& d) { return (c.dt == ETP_BF16 || d.p > 0.f) ? a.t : (void*)a.f; }

// ---- post-LN sub-blocks (BertAttention / BertXAttention / BertIntermediate+BertOutput) ---------
struct SelfAttStash { void *qkv, *P, *ctx; float* s; float* st; Act y; void* Pd; };
struct FfnStash { void *z, *h; float* s; float* st; Act y; };      // z: gelu'(pre-activation) (round 5: the forward saves the derivative)
struct CrossStash { void *q, *kv, *P, *ctx; float* s; float* st; Act y; void* Pd; };

static SelfAttStash plan_self(Bump& b, int dt, long M, int Bn, int nh, int L, int ldS, int H) {
  const size_t es = dtype_size(dt);
  SelfAttStash s;
  s.qkv = b.take(M * 3 * H * es);
  s.P = b.take((size_t)Bn * nh * L * ldS * es);
  s.ctx = b.take(M * H * es);
  s.s = (float*)b.take(M * H * 4);
  s.st = (float*)b.take(M * 2 * sizeof(float));
  s.y = take_act(b, dt, M * H);
  s.Pd = attn_needs_unfused(dt, L, L) ? b.take((size_t)Bn * nh * L * ldS * es) : nullptr;   // dropped P (training, unfused path)
  return s;
}
static FfnStash plan_ffn(Bump& b, int dt, long M, int H, int I) {
  const size_t es = dtype_size(dt);
  FfnStash f;
  f.z = b.take(M * I * es);
  f.h = b.take(M * I * es);
  f.s = (float*)b.take(M * H * 4);
  f.st = (float*)b.take(M * 2 * sizeof(float));
  f.y = take_act(b, dt, M * H);
  return f;
}

// Scratch of ONE backward sub-block.  Every sub-block gets a fresh set (HBM is plentiful) so that weight-gradient
// GEMMs still running on the side stream never see their dY operand overwritten by a later layer.
struct BwdWs { Act t1; void *t2, *dI, *dqkv, *dP; float* lnp; };   // lnp: slabs of the two-stage LayerNorm dgamma/dbeta reduction
static BwdWs plan_ws(Bump& b, int dt, long M, int Bn, int nh, int Lq, int ldS, int H, int I) {
  const size_t es = dtype_size(dt);
  BwdWs w;
  w.t1 = take_act2(b, dt, M * H);
  w.t2 = b.take(M * H * es);
  w.dI = b.take(M * I * es);
  w.dqkv = b.take(M * 3 * H * es);
  w.dP = b.take((size_t)Bn * nh * Lq * ldS * es);
  w.lnp = (float*)b.take(ln_bwd_part_bytes((int)M, H));
  return w;
}

// LayerNorm backward on the dependent chain; the dgamma/dbeta reduction over the per-block slabs is a leaf and goes to the
// side stream with the layer's weight gradients
static int ln_bwd_chain(const Ctx& c, const float* dy, const float* x, const float* stats, int gi, int bi, const float* add,
                        float* dx, void* dxt, int M, const Drop& d, float* part) {
  etp_planner* pl = c.pl;
  const bool two_stage = opt_on(OPT_LNBWD_TWO_STAGE, true);
  if (!two_stage) part = nullptr;
  ETP_TRY(ln_bwd_s(c.dt, dy, x, stats, pl->pf(gi), add, dx, dxt, pl->gf(gi), pl->gf(bi), M, c.H, c.st, d, part));
  if (!part) return ETP_OK;
  float* dg = pl->gf(gi); float* db = pl->gf(bi);
  const int H = c.H;
  hipStream_t sw = c.sw;
  return on_side(c, [=]() -> int { return ln_part_reduce(part, M, H, dg, db, sw); });
}

// y = LN(dropout(dense(attn(x))) + x)
static int self_att_fwd(const Ctx& c, const AttnP& p, const Act& x, SelfAttStash& s, int Bn, int L, const uint8_t* keymask,
                        const float* dist, const float* sp_w, const float* sp_b, float eps, int mode, int layer) {
  const int H = c.H, M = Bn * L;
  AttnBuf a{s.qkv, 3L * H, offs(s.qkv, H, c.es), 3L * H, offs(s.qkv, 2 * H, c.es), 3L * H, Bn, L, L, (int)round_up(L, 8),
            keymask, 0, dist, sp_w, sp_b};
  a.Pd = s.Pd;
  ETP_TRY(attn_fwd_qkv(c, a, s.P, s.ctx, x.t, p.qkv_w, p.qkv_b, s.qkv, M, att(c, mode, layer, SITE_ATT_P)));
  ETP_TRY(linear_fwd_s(c, s.ctx, H, p.o_w, p.o_b, s.s, M, H, H, x.f, hid(c, mode, layer, SITE_ATT_O)));
  return ln_fwd_s(c.dt, s.s, c.pl->pf(p.ln_g), c.pl->pf(p.ln_b), s.y.f, lp(s.y, c.dt), s.st, M, H, eps, c.st);
}
// g (fp32): in = dL/dy, out = dL/dx (same buffer)
// tail_flush (round 6): the caller's LAST sub-block of the step -- nothing runs behind it that could hide its weight gradients, so each of
// the two goes out the moment its operands exist (out-projection: behind the LayerNorm backward; QKV: behind the attention backward)
// instead of together at the end of the block
static int self_att_bwd(const Ctx& c, const AttnP& p, const Act& x, const SelfAttStash& s, int Bn, int L,
                        const uint8_t* keymask, const float* dist, const float* sp_w, const float* sp_b, float* d_sp_w,
                        float* d_sp_b, float* g, const BwdWs& w, int mode, int layer, bool tail_flush = false) {
  const int H = c.H, M = Bn * L;
  etp_planner* pl = c.pl;
  const Drop dh = hid(c, mode, layer, SITE_ATT_O);
  ETP_TRY(ln_bwd_chain(c, g, s.s, s.st, p.ln_g, p.ln_b, nullptr, w.t1.f, lp2(c, w.t1, dh), M, dh, w.lnp));   // t1.f = ds, operand copy = ds * mask
  const void* ds = op2(c, w.t1, dh);
  ETP_TRY(linear_wgrad(c, ds, H, s.ctx, H, p.o_w, p.o_b, M, H, H));
  if (tail_flush) ETP_TRY(flush_side(c));
  AttnBuf a{s.qkv, 3L * H, offs(s.qkv, H, c.es), 3L * H, offs(s.qkv, 2 * H, c.es), 3L * H, Bn, L, L, (int)round_up(L, 8),
            keymask, 0, dist, sp_w, sp_b};
  a.Pd = s.Pd;
  a.O = s.ctx; a.ldo = H;
  ETP_TRY(attn_bwd_proj(c, a, s.P, ds, p.o_w, w.t2, M, w.dP, w.dqkv, 3L * H, offs(w.dqkv, H, c.es), 3L * H,                    // t2 = dctx
                        offs(w.dqkv, 2 * H, c.es), 3L * H, d_sp_w, d_sp_b, att(c, mode, layer, SITE_ATT_P)));
  ETP_TRY(linear_wgrad(c, w.dqkv, 3 * H, x.t, H, p.qkv_w, p.qkv_b, M, 3 * H, H));
  if (tail_flush) ETP_TRY(flush_side(c));
  return linear_dgrad_s(c, w.dqkv, 3 * H, p.qkv_w, g, M, 3 * H, H, w.t1.f);                                  // g = dx
}

static int ffn_fwd(const Ctx& c, const FfnP& p, const Act& x, FfnStash& f, int M, float eps, int mode, int layer) {
  const int H = c.H, I = c.I;
  ETP_TRY(linear_fwd(c, x.t, H, p.i_w, p.i_b, f.h, I, M, I, H, ETP_ACT_GELU_SAVEGRAD, f.z, nullptr, 0));    // f.z = gelu'(pre-activation)
  ETP_TRY(linear_fwd_s(c, f.h, I, p.o_w, p.o_b, f.s, M, H, I, x.f, hid(c, mode, layer, SITE_FFN_O)));
  return ln_fwd_s(c.dt, f.s, c.pl->pf(p.ln_g), c.pl->pf(p.ln_b), f.y.f, lp(f.y, c.dt), f.st, M, H, eps, c.st);
}
static int ffn_bwd(const Ctx& c, const FfnP& p, const Act& x, const FfnStash& f, int M, float* g, const BwdWs& w, int mode,
                   int layer, const float* g_in = nullptr, int flush_behind = 0) {
  const int H = c.H, I = c.I;
  etp_planner* pl = c.pl;
  const Drop dh = hid(c, mode, layer, SITE_FFN_O);
  ETP_TRY(ln_bwd_chain(c, g_in ? g_in : g, f.s, f.st, p.ln_g, p.ln_b, nullptr, w.t1.f, lp2(c, w.t1, dh), M, dh, w.lnp));
  const void* ds = op2(c, w.t1, dh);
  ETP_TRY(linear_wgrad(c, ds, H, f.h, I, p.o_w, p.o_b, M, H, I));
  ETP_TRY(linear_dgrad(c, ds, H, p.o_w, w.dI, I, M, H, I, ETP_ACT_MUL_Z, f.z, I, nullptr, 0));               // dI = dz
  // the PREVIOUS layer's weight gradients (held back at the end of that layer) + this layer's FFN-down one start behind this launch:
  // the 480-workgroup, two-per-CU dgrad above then finds the chip free instead of queueing behind 432 resident leaf workgroups
  if (flush_behind == 1) ETP_TRY(flush_side(c));
  ETP_TRY(linear_wgrad(c, w.dI, I, x.t, H, p.i_w, p.i_b, M, I, H));
  ETP_TRY(linear_dgrad_s(c, w.dI, I, p.i_w, g, M, I, H, w.t1.f));                                             // g = dx
  if (flush_behind == 2) ETP_TRY(flush_side(c));      // variant: behind the whole FFN half (this layer's FFN pair rides along)
  return ETP_OK;
}

// ======================================================================================
// forward_txt
// ======================================================================================
struct TxtStash {
  Act x0; float* st0;
  std::vector<SelfAttStash> att; std::vector<FfnStash> ffn;
};
static TxtStash plan_txt(const etp_planner* pl, Bump& b, int Bn, int L) {
  const int dt = pl->cfg.dtype, H = pl->cfg.hidden;
  const long M = (long)Bn * L;
  TxtStash t;
  t.x0 = take_act(b, dt, M * H);
  t.st0 = (float*)b.take(M * 2 * sizeof(float));
  for (int l = 0; l < pl->cfg.n_l; ++l) {
    t.att.push_back(plan_self(b, dt, M, Bn, pl->cfg.heads, L, (int)round_up(L, 8), H));
    t.ffn.push_back(plan_ffn(b, dt, M, H, pl->cfg.inter));
  }
  return t;
}

}  // namespace etp

using namespace etp;

extern "C" {

etp_planner* etp_planner_create(const etp_config* cfg) {
  if (!cfg) { set_error("etp_planner_create: null config"); return nullptr; }
  // hidden 256 / 512 / 768 (the reference's planners are BERT-base / XLM-R-base, 768): the row kernels keep a row's 4 * hidden / 256
  // values per lane in registers, and at 1024 the panorama / node embedding kernels no longer fit the 256 VGPRs a kernel without
  // MFMAs may use (tools/kernel_resources.py)
  if (cfg->hidden % 256 != 0 || cfg->hidden > 768 || cfg->heads * 64 != cfg->hidden || cfg->inter % 8 != 0 ||
      cfg->img_feat % 8 != 0 || (cfg->use_depth && cfg->dep_feat % 8 != 0) || cfg->ang_feat != 4 ||
      (cfg->dtype != ETP_F32 && cfg->dtype != ETP_BF16) || cfg->n_l < 0 || cfg->n_p < 0 || cfg->n_x < 0) {
    set_error("etp_planner_create: unsupported config (hidden must be 256, 512 or 768 with 64-wide heads, "
              "feature sizes multiples of 8, angle_feat_size 4)");
    return nullptr;
  }
  etp_planner* pl = new etp_planner();
  pl->cfg = *cfg;
  build_layout(pl);
  return pl;
}
void etp_planner_destroy(etp_planner* p) { delete p; }
int etp_planner_param_count(const etp_planner* p) { return p ? (int)p->params.size() : 0; }
int etp_planner_param_info(const etp_planner* p, int i, etp_param_info* out) {
  ETP_REQUIRE(p && out && i >= 0 && i < (int)p->params.size(), "bad index");
  const PInfo& q = p->params[i];
  memset(out, 0, sizeof(*out));
  strncpy(out->name, q.name.c_str(), sizeof(out->name) - 1);
  out->ndim = q.ndim; out->shape[0] = q.shape[0]; out->shape[1] = q.shape[1]; out->offset = q.offset;
  return ETP_OK;
}
int64_t etp_planner_arena_elems(const etp_planner* p) { return p ? p->total : 0; }
int64_t etp_planner_matrix_elems(const etp_planner* p) { return p ? p->n_matrix : 0; }
int etp_planner_bind(etp_planner* p, float* params, void* shadow, float* grads) {
  ETP_REQUIRE(p && params, "null planner/params");
  ETP_REQUIRE(p->cfg.dtype == ETP_F32 || shadow != nullptr, "bf16 mode needs a shadow arena");
  ETP_REQUIRE(((uintptr_t)params % 256 == 0) && ((uintptr_t)shadow % 256 == 0) && ((uintptr_t)grads % 256 == 0),
              "arenas must be 256-byte aligned");
  p->P = params; p->S = shadow; p->G = grads;
  return ETP_OK;
}
int etp_planner_set_aux_stream(etp_planner* p, etp_stream_t aux) {
  ETP_REQUIRE(p, "null planner");
  p->aux = reinterpret_cast<hipStream_t>(aux);
  return ETP_OK;
}
int etp_planner_set_aux2_stream(etp_planner* p, etp_stream_t aux2) {
  ETP_REQUIRE(p, "null planner");
  p->aux2 = reinterpret_cast<hipStream_t>(aux2);
  return ETP_OK;
}
int etp_planner_set_lazy_join(etp_planner* p, int lazy) {
  ETP_REQUIRE(p, "null planner");
  p->lazy_join = lazy < 0 ? 0 : (lazy > 2 ? 2 : lazy);
  return ETP_OK;
}
int etp_planner_set_grad_overwrite(etp_planner* p, int on) {
  ETP_REQUIRE(p, "null planner");
  p->grad_overwrite = on != 0;
  return ETP_OK;
}
int etp_planner_join_aux(etp_planner* p, etp_stream_t stream) {
  ETP_REQUIRE(p, "null planner");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (p->dtxt_owed(st)) ETP_CHECK_HIP(stream_wait_event(st, p->dtxt_ready));
  if (p->aux == nullptr || p->aux == st) return ETP_OK;
  return stream_after(p, p->aux, st);
}
int etp_planner_set_dropout(etp_planner* p, float p_hidden, float p_attn, float p_head, float p_env, uint64_t seed) {
  ETP_REQUIRE(p && p_hidden >= 0.f && p_hidden < 1.f && p_attn >= 0.f && p_attn < 1.f && p_env >= 0.f && p_env < 1.f &&
                  p_head >= 0.f && p_head < 1.f,
              "dropout rates must be in [0, 1)");
  p->p_hidden = p_hidden; p->p_attn = p_attn; p->p_env = p_env; p->p_head = p_head; p->drop_seed = seed;
  return ETP_OK;
}
int etp_dropout_multipliers(float p, uint64_t seed, int mode, int layer, int slot, int64_t n, float* out_host) {
  ETP_REQUIRE(out_host && n >= 0 && p >= 0.f && p < 1.f && mode >= 0 && layer >= 0 && slot >= 0 && slot < 16, "bad arguments");
  const Drop d = drop_site(p, seed, (uint32_t)(mode << 16 | layer << 4 | slot));
  for (int64_t i = 0; i < n; ++i) out_host[i] = p > 0.f ? drop_mult(d.seed, (uint32_t)i, d.p, d.inv_keep) : 1.f;
  return ETP_OK;
}
int etp_planner_refresh_weights(etp_planner* p, etp_stream_t stream) {
  ETP_REQUIRE(p && p->P, "planner not bound");
  if (p->cfg.dtype != ETP_BF16) return ETP_OK;
  return cast_f32_to_bf16(p->P, p->S, p->n_matrix, (hipStream_t)stream);
}
int etp_planner_refresh_part(etp_planner* p, int part, etp_stream_t stream) {
  ETP_REQUIRE(p && p->P && part >= 0 && part <= 2, "planner not bound / part must be 0 (text), 1 (panorama) or 2 (navigation)");
  if (p->cfg.dtype != ETP_BF16) return ETP_OK;
  // matrix region order (build_layout): text layers | view projections + panorama layers | x-layers + SAP head
  const long txt_end = p->off(p->img_w);
  const long pano_end = p->cfg.n_x > 0 ? p->off(p->xl[0].self.qkv_w) : p->off(p->sap0_w);
  const long lo = part == 0 ? 0 : part == 1 ? txt_end : pano_end;
  const long hi = part == 0 ? txt_end : part == 1 ? pano_end : p->n_matrix;
  if (hi <= lo) return ETP_OK;
  return cast_f32_to_bf16(p->P + lo, reinterpret_cast<uint16_t*>(p->S) + lo, hi - lo, (hipStream_t)stream);
}

// bf16 shadow of the text encoder with only layer 0 on the dependent chain: layer 0's matrices are cast on `main`, layers
// 1.. on `side` (bandwidth-bound, ~40 us for BERT-base, which used to sit in front of the first text GEMM of every step);
// the next etp_txt_fwd on `main` waits for the side cast right after it has enqueued layer 0.
int etp_planner_refresh_text_split(etp_planner* p, etp_stream_t main, etp_stream_t side) {
  ETP_REQUIRE(p && p->P, "planner not bound");
  if (p->cfg.dtype != ETP_BF16) return ETP_OK;
  const long txt_end = p->off(p->img_w);
  hipStream_t sm = (hipStream_t)main, ss = (hipStream_t)side;
  if (p->cfg.n_l < 2 || ss == nullptr || ss == sm) return cast_f32_to_bf16(p->P, p->S, txt_end, sm);
  const long l1 = p->off(p->txt[1].att.qkv_w);
  ETP_TRY(cast_f32_to_bf16(p->P, p->S, l1, sm));
  ETP_TRY(stream_after(p, sm, ss));                     // the side stream starts no earlier than this step (ordering with the optimizer)
  // (round 6: layer 0's cast on the side stream too, beside the embedding kernel: 3.949 against 3.931 ms, no gain -- r06_ab_runs.json r6c11)
  ETP_TRY(cast_f32_to_bf16(p->P + l1, reinterpret_cast<uint16_t*>(p->S) + l1, txt_end - l1, ss));
  if (!p->txt_w_ready) ETP_CHECK_HIP(hipEventCreateWithFlags(&p->txt_w_ready, hipEventDisableTiming));
  ETP_CHECK_HIP(event_record(p->txt_w_ready, ss));
  p->txt_w_pending = true;
  return ETP_OK;
}

// ---------------------------------------------------------------------------------------
int64_t etp_txt_stash_bytes(const etp_planner* p, int B, int L) {
  if (!p) return 0;
  Bump b(nullptr);
  plan_txt(p, b, B, L);
  return (int64_t)b.off + 256;
}
int64_t etp_txt_ws_bytes(const etp_planner* p, int B, int L) {
  if (!p) return 0;
  Bump b(nullptr);
  b.take((size_t)B * L * p->cfg.hidden * 4);
  for (int l = 0; l < 2 * p->cfg.n_l; ++l)
    plan_ws(b, p->cfg.dtype, (long)B * L, B, p->cfg.heads, L, (int)round_up(L, 8), p->cfg.hidden, p->cfg.inter);
  return (int64_t)b.off + 256;
}

int etp_txt_fwd(etp_planner* p, const int64_t* ids, const uint8_t* mask, int B, int L, float* out, void* stash,
                etp_stream_t stream) {
  ETP_REQUIRE(p && p->P && ids && mask && out && stash && B > 0 && L > 0 && L <= p->cfg.max_pos, "bad arguments");
  Ctx c = make_ctx(p, stream);
  Bump b(stash);
  TxtStash t = plan_txt(p, b, B, L);
  const int H = c.H, M = B * L;
  const float eps = p->cfg.ln_eps;
  ETP_TRY(text_embed_fwd(c.dt, ids, p->pf(p->word), p->pf(p->pos), p->pf(p->type), p->pf(p->emb_g), p->pf(p->emb_b), t.x0.f,
                         lp(t.x0, c.dt), t.st0, B, L, H, eps, c.st, hid(c, MODE_TXT, 0, SITE_EMBED)));
  Act x = t.x0;
  for (int l = 0; l < p->cfg.n_l; ++l) {
    stamp_mark(c.st, 1000 + l);
    ETP_TRY(self_att_fwd(c, p->txt[l].att, x, t.att[l], B, L, mask, nullptr, nullptr, nullptr, eps, MODE_TXT, l));
    if (l == p->cfg.n_l - 1) t.ffn[l].y.f = out;      // the last LayerNorm writes the API tensor itself (backward never reads it)
    ETP_TRY(ffn_fwd(c, p->txt[l].ffn, t.att[l].y, t.ffn[l], M, eps, MODE_TXT, l));
    x = t.ffn[l].y;
    if (l == 0 && p->txt_w_pending) {                    // layers >= 1 read weights cast on the side stream
      ETP_CHECK_HIP(stream_wait_event(c.st, p->txt_w_ready));
      p->txt_w_pending = false;
    }
  }
  if (p->cfg.n_l == 0) ETP_TRY(copy_f32(x.f, out, (long)M * H, c.st));
  stamp_mark(c.st, 1099);
  return ETP_OK;
}

// Backward of text layers [layer_lo, layer_hi) (processed from layer_hi-1 down).  The running gradient lives at the front of
// `ws`, so consecutive calls with the same ws continue where the previous one stopped; a caller can therefore interleave
// gradient all-reduces of the finished layers with the rest of the backward (data-parallel overlap).
int etp_txt_bwd_range(etp_planner* p, const float* dout, const int64_t* ids, const uint8_t* mask, int B, int L, void* stash,
                      void* ws, int layer_lo, int layer_hi, etp_stream_t stream) {
  ETP_REQUIRE(p && p->P && p->G && dout && ids && mask && stash && ws && B > 0 && L > 0, "bad arguments");
  ETP_REQUIRE(layer_lo >= 0 && layer_lo <= layer_hi && layer_hi <= p->cfg.n_l, "bad layer range");
  Ctx c = make_ctx(p, stream);
  // `dout` may be the d txt_embeds a lazily joined etp_nav_bwd ON THIS STREAM left running on the aux2 stream: wait for it once.  The debt
  // is kept per main stream (MicroBatchedStep: the navigation backward of every micro-batch on its own stream, then their text backwards;
  // the one event is re-recorded by every deferral and aux2 is in-order, so the latest record covers the earlier ones), and a stream that
  // never deferred never waits -- a stale event must not be waited for inside a stream capture
  if (p->dtxt_owed(c.st)) ETP_CHECK_HIP(stream_wait_event(c.st, p->dtxt_ready));
  std::vector<std::function<int()>> pend;
  std::vector<GemmArgs> wq;
  if (c.sw != c.st) c.pend = &pend;
  c.wq = &wq;
  Bump b(stash);
  TxtStash t = plan_txt(p, b, B, L);
  Bump wb(ws);
  const int H = c.H, M = B * L;
  float* g = (float*)wb.take((size_t)M * H * 4);
  if (layer_hi == p->cfg.n_l && p->cfg.n_l == 0) ETP_TRY(copy_f32(dout, g, (long)M * H, c.st));
  for (int l = p->cfg.n_l - 1; l >= 0; --l) {
    const Act x = l == 0 ? t.x0 : t.ffn[l - 1].y;
    BwdWs wf = plan_ws(wb, c.dt, M, B, c.nh, L, (int)round_up(L, 8), H, c.I);   // same carving in every call
    BwdWs wa = plan_ws(wb, c.dt, M, B, c.nh, L, (int)round_up(L, 8), H, c.I);
    if (l >= layer_hi || l < layer_lo) continue;
    stamp_mark(c.st, 2200 + 10 * l);
    // the top layer reads the incoming gradient in place (dout) and leaves dL/dx in the running buffer g
    // Round 5: a layer's four weight gradients are NOT forked at the end of the layer.  There the 432 leaf workgroups took
    // every slot of the chip just before the next layer's FFN dgrad -- a 480-workgroup, two-per-CU grid -- which then ran 64.6 us in the
    // step against 27.9 us alone (profiles/r05_chain_vs_isolated.txt), 9 x 37 us.  They are held back and go out right BEHIND that dgrad
    // (1, the default): the leaf work then shares the chip with the FFN-up dgrad, LayerNorm, out-projection and attention backward, whose
    // one-per-CU 74-KB workgroups and small kernels leave room beside it.  Same-box A/B (profiles/r05_ab_runs.json): 4.12 / 4.13 ->
    // 4.06 / 4.06 ms; behind the whole FFN half (2): 4.14 against 4.10 (1) and 4.18 (0) on a second box.  Only where that dgrad IS one
    // resident round (at most 512 tiles of 128 x 128: configs 2 and 5): at config 4's 8192 rows it is three rounds anyway and holding
    // the leaf work back costs 1.4 % (10.77 against 10.62 ms).  ETP_FLUSH_DELAY=0 / 1 / 2 forces a mode.
    const int delay_env = opt_int(OPT_FLUSH_DELAY, -1);
    const int delay = delay_env >= 0 ? delay_env : (((long)(M / 128) * (c.I / 128) <= 512) ? 1 : 0);
    ETP_TRY(ffn_bwd(c, p->txt[l].ffn, t.att[l].y, t.ffn[l], M, g, wf, MODE_TXT, l, l == p->cfg.n_l - 1 ? dout : nullptr, delay));
    stamp_mark(c.st, 2200 + 10 * l + 1);
    // the LAST layer of the backward: its two FFN weight gradients go out now instead of with the attention ones at the end of
    // the layer -- nothing runs behind this layer that could hide them (the step's end waits for the weight-gradient stream)
    // (round 5, measured again with the mm32 kernels: a second fork per layer -- FFN pair here, attention pair at the end -- and a
    // 256-tile budget that leaves every CU room for a chain workgroup both cost +1..2 %: profiles/r05_ab_runs.json; the patch is
    // tools/experiments/r05_ln_fold_and_flush_split.patch)
    if (l == 0 && layer_lo == 0) ETP_TRY(flush_side(c));
    // TXT_LAST_SPLIT (round 6): layer 0's two attention weight gradients each start as soon as their operands exist; held back to the end
    // of the layer (as every other layer's are) they begin when the chain has 40 us left and set a ~66-us tail behind it.  Measured
    // (r06_ab_runs.json r6c13): config 2 -- tail 66 -> 44 us, but the attention half they now share the chip with 142 -> 154 us: +0.3 %;
    // config 5 +0.5 %; config 4 (8192 rows: long products, the tail is theirs) 10.516 against 10.574 ms, -0.55 %.  Rule: where the
    // weight gradients are NOT held back behind the next layer's dgrad either (delay == 0: multi-round grids); 0 / 1 force it.
    const int split_env = opt_int(OPT_TXT_LAST_SPLIT, -1);
    const bool last_split = l == 0 && layer_lo == 0 && (split_env >= 0 ? split_env != 0 : delay == 0);
    ETP_TRY(self_att_bwd(c, p->txt[l].att, x, t.att[l], B, L, mask, nullptr, nullptr, nullptr, nullptr, nullptr, g, wa, MODE_TXT, l,
                         last_split));
    // one fork per layer: this layer's four weight gradients as one grouped launch.  ETP_FLUSH_EVERY=n (measurement knob,
    // tools/r03_call15.sh) forks every n layers instead: 4n products per launch, fewer launch tails, later start of the leaf work
    const int every = std::max(1, opt_int(OPT_FLUSH_EVERY, 1));
    if (delay && l != layer_lo) continue;             // held back: goes out behind the next layer's FFN dgrad
    if (every == 1 || (layer_hi - 1 - l) % every == every - 1 || l == layer_lo) ETP_TRY(flush_side(c));
  }
  if (layer_lo == 0) {
    // TXT_TAIL=1 (round 6 experiment, default OFF): the embedding backward produces parameter gradients only -- a LEAF, and the last kernel
    // of the step's chain, which then still waits ~66 us for the weight-gradient backlog (profiles/r06_chain_waits.txt).  On a third stream
    // (aux2: idle by now) it runs BESIDE that backlog instead of in front of the wait.  Measured (r06_ab_runs.json r6c10): the chain ends
    // 35 us earlier and the backlog it now shares the chip with takes 28 us longer: -0.26 % alone, and WORSE than without it once the
    // navigation tail (NAV_TAIL) is off the chain (3.991 against 3.977 ms).  Kept as a switch.
    hipStream_t se = (opt_on(OPT_TXT_TAIL, false) && c.s3 != c.st && c.s3 != c.sw) ? c.s3 : c.st;
    if (se != c.st) ETP_TRY(stream_after(p, c.st, se));
    ETP_TRY(text_embed_bwd(c.dt, g, ids, p->pf(p->word), p->pf(p->pos), p->pf(p->type), p->pf(p->emb_g), t.st0, p->gf(p->word),
                           p->gf(p->pos), p->gf(p->type), p->gf(p->emb_g), p->gf(p->emb_b), B, L, H, se,
                           hid(c, MODE_TXT, 0, SITE_EMBED)));
    if (se != c.st) {
      stamp_mark(c.st, 2299);
      ETP_TRY(join_wgrads(c));
      return stream_after(p, se, c.st);
    }
  }
  stamp_mark(c.st, 2299);
  // a range that stops above layer 0 is followed by another one: with lazy level 2 its weight gradients keep running on the
  // side stream (the final range, or etp_planner_join_aux, joins them)
  if (p->lazy_join >= 2 && layer_lo > 0) return flush_side(c);
  return join_wgrads(c);
}
int etp_txt_bwd(etp_planner* p, const float* dout, const int64_t* ids, const uint8_t* mask, int B, int L, void* stash, void* ws,
                etp_stream_t stream) {
  ETP_REQUIRE(p, "null planner");
  return etp_txt_bwd_range(p, dout, ids, mask, B, L, stash, ws, 0, p->cfg.n_l, stream);
}

// ======================================================================================
// forward_panorama
// ======================================================================================
namespace {
struct PanoLayerStash { void* a; float* st1; void *qkv, *P, *ctx; float* x1; void* f; float* st2; void *z, *h; float* x2; void* Pd; };
struct PanoStash {
  void *rgbT, *depT, *a, *d; float* est; float* x0; uint8_t* mask;
  std::vector<PanoLayerStash> layers;
  float* stn;
};
PanoStash plan_pano(const etp_planner* pl, Bump& b, int Bn, int V) {
  const etp_config& c = pl->cfg;
  const size_t es = dtype_size(c.dtype);
  const long M = (long)Bn * V;
  const int H = c.hidden, I = c.inter, ldS = (int)round_up(V, 8);
  PanoStash s;
  s.rgbT = b.take(M * c.img_feat * es);      // operand copy of the RGB features (bf16 cast and/or drop_env mask)
  s.depT = (c.dtype == ETP_BF16 && c.use_depth) ? b.take(M * c.dep_feat * es) : nullptr;
  s.a = b.take(M * H * es);
  s.d = c.use_depth ? b.take(M * H * es) : nullptr;
  s.est = (float*)b.take(M * 8 * sizeof(float));
  s.x0 = (float*)b.take(M * H * 4);
  s.mask = (uint8_t*)b.take(M);
  for (int l = 0; l < c.n_p; ++l) {
    PanoLayerStash q;
    q.a = b.take(M * H * es);
    q.st1 = (float*)b.take(M * 2 * sizeof(float));
    q.qkv = b.take(M * 3 * H * es);
    q.P = b.take((size_t)Bn * c.heads * V * ldS * es);
    q.ctx = b.take(M * H * es);
    q.x1 = (float*)b.take(M * H * 4);
    q.f = b.take(M * H * es);
    q.st2 = (float*)b.take(M * 2 * sizeof(float));
    q.z = b.take(M * I * es);
    q.h = b.take(M * I * es);
    q.x2 = (float*)b.take(M * H * 4);
    q.Pd = attn_needs_unfused(c.dtype, V, V) ? b.take((size_t)Bn * c.heads * V * ldS * es) : nullptr;
    s.layers.push_back(q);
  }
  s.stn = (float*)b.take(M * 2 * sizeof(float));
  return s;
}
__global__ void seq_mask_kernel(const int64_t* __restrict__ lens, uint8_t* __restrict__ m1, uint8_t* __restrict__ m2, int Bn,
                                int V) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < Bn * V) {
    const uint8_t v = (i % V) < lens[i / V] ? 1 : 0;      // gen_seq_masks common/ops.py:36-44
    m1[i] = v;
    if (m2) m2[i] = v;
  }
}
PanoEmbedParams pano_params(const etp_planner* p) {
  PanoEmbedParams q;
  q.g_img = p->pf(p->img_g); q.b_img = p->pf(p->img_bb);
  q.g_dep = p->cfg.use_depth ? p->pf(p->dep_g) : nullptr; q.b_dep = p->cfg.use_depth ? p->pf(p->dep_bb) : nullptr;
  q.w_loc = p->pf(p->loc_w); q.bias_loc = p->pf(p->loc_b); q.g_loc = p->pf(p->loc_g); q.b_loc = p->pf(p->loc_bb);
  q.nav_emb = p->pf(p->nav_emb); q.type1 = p->pf(p->type) + p->cfg.hidden;
  q.g_out = p->pf(p->pe_g); q.b_out = p->pf(p->pe_b);
  return q;
}
PanoEmbedGrads pano_grads(const etp_planner* p) {
  PanoEmbedGrads q;
  q.g_img = p->gf(p->img_g); q.b_img = p->gf(p->img_bb);
  q.g_dep = p->cfg.use_depth ? p->gf(p->dep_g) : nullptr; q.b_dep = p->cfg.use_depth ? p->gf(p->dep_bb) : nullptr;
  q.w_loc = p->gf(p->loc_w); q.bias_loc = p->gf(p->loc_b); q.g_loc = p->gf(p->loc_g); q.b_loc = p->gf(p->loc_bb);
  q.nav_emb = p->gf(p->nav_emb); q.type1 = p->gf(p->type) + p->cfg.hidden;
  q.g_out = p->gf(p->pe_g); q.b_out = p->gf(p->pe_b);
  return q;
}
// per-layer backward scratch of the pre-LN panorama layer
struct PanoWs { Act g; float* t1f; Act t2; void *dI, *t1, *dqkv, *dP; float *lnp, *lnp2; };
PanoWs plan_pano_ws(Bump& b, int dt, long M, int Bn, int nh, int V, int ldS, int H, int I) {
  const size_t es = dtype_size(dt);
  PanoWs w;
  w.g = take_act2(b, dt, M * H);         // dx of this layer (fp32 stream + operand copy: it feeds a weight gradient)
  w.t1f = (float*)b.take(M * H * 4);
  w.t2 = take_act2(b, dt, M * H);
  w.dI = b.take(M * I * es);
  w.t1 = b.take(M * H * es);
  w.dqkv = b.take(M * 3 * H * es);
  w.dP = b.take((size_t)Bn * nh * V * ldS * es);
  w.lnp = (float*)b.take(ln_bwd_part_bytes((int)M, H));
  w.lnp2 = (float*)b.take(ln_bwd_part_bytes((int)M, H));
  return w;
}
}  // namespace

int64_t etp_pano_stash_bytes(const etp_planner* p, int B, int V) {
  if (!p) return 0;
  Bump b(nullptr);
  plan_pano(p, b, B, V);
  return (int64_t)b.off + 256;
}
int64_t etp_pano_ws_bytes(const etp_planner* p, int B, int V) {
  if (!p) return 0;
  Bump b(nullptr);
  for (int l = 0; l < p->cfg.n_p + 2; ++l)
    plan_pano_ws(b, p->cfg.dtype, (long)B * V, B, p->cfg.heads, V, (int)round_up(V, 8), p->cfg.hidden, p->cfg.inter);
  return (int64_t)b.off + 256;
}

int etp_pano_fwd(etp_planner* p, const float* rgb, const float* dep, const float* loc, const int64_t* nav,
                 const int64_t* view_lens, int B, int V, float* out, uint8_t* out_mask, void* stash, etp_stream_t stream) {
  ETP_REQUIRE(p && p->P && rgb && loc && nav && view_lens && out && stash && B > 0 && V > 0, "bad arguments");
  ETP_REQUIRE(!p->cfg.use_depth || dep, "depth features required");
  Ctx c = make_ctx(p, stream);
  Bump b(stash);
  PanoStash s = plan_pano(p, b, B, V);
  const etp_config& cf = p->cfg;
  const int H = c.H, I = c.I, M = B * V, ldS = (int)round_up(V, 8);
  stamp_mark(c.st, 1100);
  ETP_LAUNCH(seq_mask_kernel, dim3((M + 255) / 256), dim3(256), 0, c.st, view_lens, s.mask, out_mask, B, V);
  ETP_CHECK_LAUNCH("seq_mask");
  const void* rgbT = rgb; const void* depT = dep;
  const Drop denv = site(c, p->p_env, MODE_PANO, 0, SITE_ENV);     // Policy_ViewSelection_ETP.py:102,345 (drop_env on the RGB features)
  if (c.dt == ETP_BF16 || denv.p > 0.f) {
    ETP_TRY(cast_drop(c.dt, rgb, s.rgbT, (long)M * cf.img_feat, denv, c.st));
    rgbT = s.rgbT;
  }
  if (c.dt == ETP_BF16 && cf.use_depth) { ETP_TRY(cast_f32_to_bf16(dep, s.depT, (long)M * cf.dep_feat, c.st)); depT = s.depT; }
  ETP_TRY(linear_fwd(c, rgbT, cf.img_feat, p->img_w, p->img_b, s.a, H, M, H, cf.img_feat, ETP_ACT_NONE, nullptr, nullptr, 0));
  if (cf.use_depth)
    ETP_TRY(linear_fwd(c, depT, cf.dep_feat, p->dep_w, p->dep_b, s.d, H, M, H, cf.dep_feat, ETP_ACT_NONE, nullptr, nullptr, 0));
  float* x0 = cf.n_p == 0 ? out : s.x0;
  ETP_TRY(pano_embed_fwd(c.dt, s.a, s.d, loc, nav, pano_params(p), x0, s.est, M, H, c.st, hid(c, MODE_PANO, 0, SITE_EMBED)));
  const float* x = x0;
  for (int l = 0; l < cf.n_p; ++l) {   // TransformerEncoderLayer.forward_pre common/transformer.py:170-182
    const PanoLayerP& q = p->pano[l];
    PanoLayerStash& t = s.layers[l];
    ETP_TRY(ln_fwd_s(c.dt, x, p->pf(q.n1_g), p->pf(q.n1_b), c.dt == ETP_BF16 ? nullptr : (float*)t.a,
                     c.dt == ETP_BF16 ? t.a : nullptr, t.st1, M, H, 1e-5f, c.st));
    AttnBuf a{t.qkv, 3L * H, offs(t.qkv, H, c.es), 3L * H, offs(t.qkv, 2 * H, c.es), 3L * H, B, V, V, ldS, s.mask, 1, nullptr,
              nullptr, nullptr};
    a.Pd = t.Pd;
    ETP_TRY(attn_fwd_qkv(c, a, t.P, t.ctx, t.a, q.in_w, q.in_b, t.qkv, M, hid(c, MODE_PANO, l, SITE_ATT_P)));   // MHA dropout = hidden rate
    ETP_TRY(linear_fwd_s(c, t.ctx, H, q.out_w, q.out_b, t.x1, M, H, H, x, hid(c, MODE_PANO, l, SITE_ATT_O)));
    ETP_TRY(ln_fwd_s(c.dt, t.x1, p->pf(q.n2_g), p->pf(q.n2_b), c.dt == ETP_BF16 ? nullptr : (float*)t.f,
                     c.dt == ETP_BF16 ? t.f : nullptr, t.st2, M, H, 1e-5f, c.st));
    ETP_TRY(linear_fwd(c, t.f, H, q.l1_w, q.l1_b, t.h, I, M, I, H, ETP_ACT_GELU_SAVEGRAD, t.z, nullptr, 0, hid(c, MODE_PANO, l, SITE_FFN_I)));
    ETP_TRY(linear_fwd_s(c, t.h, I, q.l2_w, q.l2_b, t.x2, M, H, I, t.x1, hid(c, MODE_PANO, l, SITE_FFN_O)));
    x = t.x2;
  }
  if (cf.n_p > 0) ETP_TRY(ln_fwd_s(c.dt, x, p->pf(p->pn_g), p->pf(p->pn_b), out, nullptr, s.stn, M, H, 1e-12f, c.st));
  stamp_mark(c.st, 1199);
  return ETP_OK;
}

int etp_pano_bwd(etp_planner* p, const float* dout, const float* rgb, const float* dep, const float* loc, const int64_t* nav,
                 int B, int V, float* d_rgb, void* stash, void* ws, etp_stream_t stream) {
  ETP_REQUIRE(p && p->P && p->G && dout && rgb && loc && nav && stash && ws && B > 0 && V > 0, "bad arguments");
  Ctx c = make_ctx(p, stream);
  std::vector<std::function<int()>> pend;
  std::vector<GemmArgs> wq;
  if (c.sw != c.st) c.pend = &pend;
  c.wq = &wq;
  Bump b(stash);
  PanoStash s = plan_pano(p, b, B, V);
  const etp_config& cf = p->cfg;
  const int H = c.H, I = c.I, M = B * V, ldS = (int)round_up(V, 8);
  Bump wb(ws);
  PanoWs w0 = plan_pano_ws(wb, c.dt, M, B, c.nh, V, ldS, H, I);
  stamp_mark(c.st, 2100);
  Act g = w0.g;
  Drop gd = drop_none();
  if (cf.n_p > 0) {
    const float* xin = s.layers[cf.n_p - 1].x2;
    gd = hid(c, MODE_PANO, cf.n_p - 1, SITE_FFN_O);      // g's operand copy carries the mask of the layer that consumes it
    ETP_TRY(ln_bwd_chain(c, dout, xin, s.stn, p->pn_g, p->pn_b, nullptr, g.f, lp2(c, g, gd), M, gd, w0.lnp));
  } else {
    ETP_TRY(copy_f32(dout, g.f, (long)M * H, c.st));
  }
  for (int l = cf.n_p - 1; l >= 0; --l) {
    const PanoLayerP& q = p->pano[l];
    const PanoLayerStash& t = s.layers[l];
    const float* x = l == 0 ? s.x0 : s.layers[l - 1].x2;
    PanoWs w = plan_pano_ws(wb, c.dt, M, B, c.nh, V, ldS, H, I);
    // FFN: x2 = x1 + W2 gelu(W1 LN2(x1))
    const void* gop = op2(c, g, gd);                                                                              // dx2 * mask(dropout2)
    ETP_TRY(linear_wgrad(c, gop, H, t.h, I, q.l2_w, q.l2_b, M, H, I));
    ETP_TRY(linear_dgrad(c, gop, H, q.l2_w, w.dI, I, M, H, I, ETP_ACT_MUL_Z, t.z, I, nullptr, 0, 0,
                         hid(c, MODE_PANO, l, SITE_FFN_I)));
    ETP_TRY(linear_wgrad(c, w.dI, I, t.f, H, q.l1_w, q.l1_b, M, I, H));
    ETP_TRY(linear_dgrad_s(c, w.dI, I, q.l1_w, w.t1f, M, I, H, nullptr));                                         // t1f = df
    const Drop d1 = hid(c, MODE_PANO, l, SITE_ATT_O);
    ETP_TRY(ln_bwd_chain(c, w.t1f, t.x1, t.st2, q.n2_g, q.n2_b, g.f, w.t2.f, lp2(c, w.t2, d1), M, d1, w.lnp));     // t2 = dx1
    // attention: x1 = x + dropout1(Wo attn(LN1(x)))
    const void* t2op = op2(c, w.t2, d1);
    ETP_TRY(linear_wgrad(c, t2op, H, t.ctx, H, q.out_w, q.out_b, M, H, H));
    AttnBuf a{t.qkv, 3L * H, offs(t.qkv, H, c.es), 3L * H, offs(t.qkv, 2 * H, c.es), 3L * H, B, V, V, ldS, s.mask, 1, nullptr,
              nullptr, nullptr};
    a.Pd = t.Pd;
    a.O = t.ctx; a.ldo = H;
    ETP_TRY(attn_bwd_proj(c, a, t.P, t2op, q.out_w, w.t1, M, w.dP, w.dqkv, 3L * H, offs(w.dqkv, H, c.es), 3L * H,               // t1 = dctx
                          offs(w.dqkv, 2 * H, c.es), 3L * H, nullptr, nullptr, hid(c, MODE_PANO, l, SITE_ATT_P)));
    ETP_TRY(linear_wgrad(c, w.dqkv, 3 * H, t.a, H, q.in_w, q.in_b, M, 3 * H, H));
    ETP_TRY(linear_dgrad_s(c, w.dqkv, 3 * H, q.in_w, w.t1f, M, 3 * H, H, nullptr));                               // t1f = da
    gd = l > 0 ? hid(c, MODE_PANO, l - 1, SITE_FFN_O) : drop_none();
    ETP_TRY(ln_bwd_chain(c, w.t1f, x, t.st1, q.n1_g, q.n1_b, w.t2.f, w.g.f, l > 0 ? lp2(c, w.g, gd) : nullptr, M, gd,
                         w.lnp2));                                                                                  // dx
    ETP_TRY(flush_side(c));
    g = w.g;
  }
  // embedding fuse backward -> da (t1), dd (dI reused as [M,H])
  PanoWs w = plan_pano_ws(wb, c.dt, M, B, c.nh, V, ldS, H, I);
  ETP_TRY(pano_embed_bwd(c.dt, g.f, s.a, s.d, loc, nav, s.est, pano_params(p), pano_grads(p), w.t1, w.dI, M, H, c.st,
                         hid(c, MODE_PANO, 0, SITE_EMBED)));
  const Drop denv = site(c, p->p_env, MODE_PANO, 0, SITE_ENV);
  const void* rgbT = (c.dt == ETP_BF16 || denv.p > 0.f) ? s.rgbT : (const void*)rgb;
  const void* depT = c.dt == ETP_BF16 ? s.depT : (const void*)dep;
  ETP_TRY(linear_wgrad(c, w.t1, H, rgbT, cf.img_feat, p->img_w, p->img_b, M, H, cf.img_feat));
  if (cf.use_depth) ETP_TRY(linear_wgrad(c, w.dI, H, depT, cf.dep_feat, p->dep_w, p->dep_b, M, H, cf.dep_feat));
  if (d_rgb) ETP_TRY(linear_dgrad_s(c, w.t1, H, p->img_w, d_rgb, M, H, cf.img_feat, nullptr, 0, denv));
  stamp_mark(c.st, 2199);
  return finish_wgrads(c);
}

// ======================================================================================
// forward_navigation
// ======================================================================================
namespace {
struct XStash { CrossStash cross; SelfAttStash self; FfnStash ffn; };
struct NavStash {
  void* txtT;                  // text embeddings in the operand dtype (K/V projections)
  Act x0; float* st0;
  std::vector<XStash> layers;
  void* r; float* str;
};
NavStash plan_nav(const etp_planner* pl, Bump& b, int Bn, int L, int G) {
  const etp_config& c = pl->cfg;
  const int dt = c.dtype;
  const size_t es = dtype_size(dt);
  const long Mg = (long)Bn * G, Mt = (long)Bn * L;
  const int H = c.hidden, ldL = (int)round_up(L, 8), ldG = (int)round_up(G, 8);
  NavStash s;
  s.txtT = dt == ETP_BF16 ? b.take(Mt * H * es) : nullptr;
  s.x0 = take_act(b, dt, Mg * H);
  s.st0 = (float*)b.take(Mg * 2 * sizeof(float));
  for (int l = 0; l < c.n_x; ++l) {
    XStash x;
    x.cross.q = b.take(Mg * H * es);
    x.cross.kv = b.take(Mt * 2 * H * es);
    x.cross.P = b.take((size_t)Bn * c.heads * G * ldL * es);
    x.cross.ctx = b.take(Mg * H * es);
    x.cross.s = (float*)b.take(Mg * H * 4);
    x.cross.st = (float*)b.take(Mg * 2 * sizeof(float));
    x.cross.y = take_act(b, dt, Mg * H);
    x.cross.Pd = attn_needs_unfused(dt, G, L) ? b.take((size_t)Bn * c.heads * G * ldL * es) : nullptr;
    x.self = plan_self(b, dt, Mg, Bn, c.heads, G, ldG, H);
    x.ffn = plan_ffn(b, dt, Mg, H, c.inter);
    s.layers.push_back(x);
  }
  s.r = b.take(Mg * H * es);
  s.str = (float*)b.take(Mg * 2 * sizeof(float));
  return s;
}
struct NavCrossWs { BwdWs w; void *dq, *dkv, *dPx; };
struct NavWs { float* g; BwdWs head; std::vector<BwdWs> ffn, self; std::vector<NavCrossWs> cross; };
NavWs plan_nav_ws(const etp_planner* pl, Bump& b, int Bn, int L, int G) {
  const etp_config& c = pl->cfg;
  const int dt = c.dtype;
  const size_t es = dtype_size(dt);
  const long Mg = (long)Bn * G, Mt = (long)Bn * L;
  const int ldG = (int)round_up(G, 8);
  NavWs n;
  n.g = (float*)b.take(Mg * c.hidden * 4);
  n.head = plan_ws(b, dt, Mg, Bn, c.heads, G, ldG, c.hidden, c.inter);
  for (int l = 0; l < c.n_x; ++l) {   // fresh scratch per sub-block (see plan_ws)
    n.ffn.push_back(plan_ws(b, dt, Mg, Bn, c.heads, G, ldG, c.hidden, c.inter));
    n.self.push_back(plan_ws(b, dt, Mg, Bn, c.heads, G, ldG, c.hidden, c.inter));
    NavCrossWs x;
    x.w = plan_ws(b, dt, Mg, Bn, c.heads, G, ldG, c.hidden, c.inter);
    x.dq = b.take(Mg * c.hidden * es);
    x.dkv = b.take(Mt * 2 * c.hidden * es);
    x.dPx = b.take((size_t)Bn * c.heads * G * round_up(L, 8) * es);
    n.cross.push_back(x);
  }
  return n;
}
}  // namespace

int64_t etp_nav_stash_bytes(const etp_planner* p, int B, int L, int G) {
  if (!p) return 0;
  Bump b(nullptr);
  plan_nav(p, b, B, L, G);
  return (int64_t)b.off + 256;
}
int64_t etp_nav_ws_bytes(const etp_planner* p, int B, int L, int G) {
  if (!p) return 0;
  Bump b(nullptr);
  plan_nav_ws(p, b, B, L, G);
  return (int64_t)b.off + 256;
}

}  // extern "C"
namespace {
// Text K/V cache: the instruction is fixed for a whole episode, yet BertOutAttention re-projects it to
// keys/values in each of the 4 x-layers at every rollout step (vilmodel_cmt.py:326-328,387-389; ss_trainer_ETP.py:819-822).
// Layout of the caller-owned cache: [ text in the operand dtype (bf16 mode only) | K|V of x-layer 0 | ... | x-layer n_x-1 ],
// each K|V block [B*L, 2H] in the operand dtype -- exactly what the cross-attention kernels consume.
struct KvCache { void* txtT; std::vector<void*> kv; };
KvCache plan_kv(const etp_planner* pl, void* buf, int Bn, int L) {
  const etp_config& c = pl->cfg;
  const size_t es = dtype_size(c.dtype);
  const long Mt = (long)Bn * L;
  Bump b(buf);
  KvCache k;
  k.txtT = c.dtype == ETP_BF16 ? b.take(Mt * c.hidden * es) : nullptr;
  for (int l = 0; l < c.n_x; ++l) k.kv.push_back(b.take(Mt * 2 * c.hidden * es));
  return k;
}
int64_t kv_bytes(const etp_planner* pl, int Bn, int L) {
  const etp_config& c = pl->cfg;
  const size_t es = dtype_size(c.dtype);
  const long Mt = (long)Bn * L;
  Bump b(nullptr);
  if (c.dtype == ETP_BF16) b.take(Mt * c.hidden * es);
  for (int l = 0; l < c.n_x; ++l) b.take(Mt * 2 * c.hidden * es);
  return (int64_t)b.off + 256;
}
}  // namespace
extern "C" {

int64_t etp_nav_kv_bytes(const etp_planner* p, int B, int L) { return p ? kv_bytes(p, B, L) : 0; }
int64_t etp_nav_kv_offset(const etp_planner* p, int B, int L) {   // byte offset of the first K|V block inside the cache
  if (!p) return 0;
  KvCache k = plan_kv(p, reinterpret_cast<void*>(0x1000), B, L);
  return p->cfg.n_x > 0 ? (int64_t)(reinterpret_cast<char*>(k.kv[0]) - reinterpret_cast<char*>(0x1000)) : 0;
}
int64_t etp_nav_kv_grad_elems(const etp_planner* p, int B, int L) {
  return p ? (int64_t)p->cfg.n_x * B * L * 2 * p->cfg.hidden : 0;
}

int etp_nav_kv_fwd(etp_planner* p, const float* txt, int B, int L, void* kvbuf, etp_stream_t stream) {
  ETP_REQUIRE(p && p->P && txt && kvbuf && B > 0 && L > 0, "bad arguments");
  Ctx c = make_ctx(p, stream);
  KvCache kc = plan_kv(p, kvbuf, B, L);
  const int H = c.H, Mt = B * L;
  const void* txtT = txt;
  if (c.dt == ETP_BF16) { ETP_TRY(cast_f32_to_bf16(txt, kc.txtT, (long)Mt * H, c.st)); txtT = kc.txtT; }
  for (int l = 0; l < p->cfg.n_x; ++l)
    ETP_TRY(linear_fwd(c, txtT, H, p->xl[l].kv_w, p->xl[l].kv_b, kc.kv[l], 2 * H, Mt, 2 * H, H, ETP_ACT_NONE, nullptr, nullptr, 0));
  return ETP_OK;
}
// Batched rollout on the cache (forward_navigation_steps stacks T steps along the batch axis, episode t*Bt + b reads
// instruction b): replicate every K|V block of the Bt-instruction cache T times into a cache laid out for T*Bt episodes --
// a copy instead of T re-projections of the same text rows (the text block of the destination is not written: the cached entry
// points never read it) -- and the reduction of the stacked call's d_kv over the T steps for etp_nav_kv_bwd.
int etp_nav_kv_repeat(etp_planner* p, const void* kvbuf, int Bt, int L, int T, void* kvbuf_steps, etp_stream_t stream) {
  ETP_REQUIRE(p && kvbuf && kvbuf_steps && Bt > 0 && L > 0 && T > 0, "bad arguments");
  Ctx c = make_ctx(p, stream);
  KvCache src = plan_kv(p, const_cast<void*>(kvbuf), Bt, L), dst = plan_kv(p, kvbuf_steps, T * Bt, L);
  const long blk = (long)Bt * L * 2 * c.H * (long)c.es;
  for (int l = 0; l < p->cfg.n_x; ++l) ETP_TRY(repeat_block(src.kv[l], dst.kv[l], blk, T, c.st));
  return ETP_OK;
}
int etp_nav_kv_sum_steps(etp_planner* p, const void* d_kv_steps, int Bt, int L, int T, void* d_kv, etp_stream_t stream) {
  ETP_REQUIRE(p && d_kv_steps && d_kv && Bt > 0 && L > 0 && T > 0, "bad arguments");
  Ctx c = make_ctx(p, stream);
  const long n = (long)Bt * L * 2 * c.H;
  for (int l = 0; l < p->cfg.n_x; ++l)
    ETP_TRY(sum_steps(c.dt, offs(d_kv_steps, (long)l * T * n, c.es), offs(d_kv, (long)l * n, c.es), n, T, c.st));
  return ETP_OK;
}
// d_kv: [n_x][B*L][2H] in the operand dtype = the SUM over the rollout's steps of what etp_nav_bwd_kv returned
int etp_nav_kv_bwd(etp_planner* p, const float* txt, const void* d_kv, int B, int L, const void* kvbuf, float* d_txt,
                   etp_stream_t stream) {
  ETP_REQUIRE(p && p->P && p->G && txt && d_kv && kvbuf && d_txt && B > 0 && L > 0, "bad arguments");
  Ctx c = make_ctx(p, stream);
  std::vector<std::function<int()>> pend;
  std::vector<GemmArgs> wq;
  if (c.sw != c.st) c.pend = &pend;
  c.wq = &wq;
  KvCache kc = plan_kv(p, const_cast<void*>(kvbuf), B, L);
  const int H = c.H, Mt = B * L;
  const void* txtT = c.dt == ETP_BF16 ? kc.txtT : (const void*)txt;
  if (p->cfg.n_x == 0) ETP_CHECK_HIP(memset_async(d_txt, 0, (size_t)Mt * H * 4, c.st));
  for (int l = 0; l < p->cfg.n_x; ++l) {
    const void* d = offs(d_kv, (long)l * Mt * 2 * H, c.es);
    ETP_TRY(linear_wgrad(c, d, 2 * H, txtT, H, p->xl[l].kv_w, p->xl[l].kv_b, Mt, 2 * H, H));
    ETP_TRY(linear_dgrad_s(c, d, 2 * H, p->xl[l].kv_w, d_txt, Mt, 2 * H, H, nullptr, l == 0 ? 0 : 1));
  }
  return join_wgrads(c);
}

}  // extern "C"
namespace {
int nav_fwd_impl(etp_planner* p, const float* txt, void* kvbuf, const uint8_t* txt_mask, const int64_t* step_ids, const float* img,
                 const float* pos, const uint8_t* gmask, const uint8_t* visited, const float* dists, int B, int L, int G,
                 float* out_embeds, float* out_logits, void* stash, etp_stream_t stream, int kv_mod = 0) {
  const bool cached = kvbuf != nullptr;
  ETP_REQUIRE(p && p->P && (txt || cached) && txt_mask && step_ids && img && pos && gmask && visited && out_embeds && out_logits &&
                  stash && B > 0 && L > 0 && G > 0,
              "bad arguments");
  ETP_REQUIRE(!p->cfg.use_sprels || dists, "gmap_pair_dists required when graph_sprels is on");
  Ctx c = make_ctx(p, stream);
  Bump b(stash);
  NavStash s = plan_nav(p, b, B, L, G);
  const etp_config& cf = p->cfg;
  const int H = c.H, Mg = B * G, Mt = B * L, ldL = (int)round_up(L, 8);
  const float eps = cf.ln_eps;
  KvCache kc;
  if (cached) kc = plan_kv(p, kvbuf, kv_mod > 0 ? kv_mod : B, L);     // kv_mod: the cache holds kv_mod instructions, episode b reads b % kv_mod
  const void* txtT = txt;
  // The text K/V projections of ALL x-layers depend only on the text (M = B*L rows, the largest GEMMs of this entry
  // point), not on the node chain: with a side stream they are issued up front and each layer waits for its own.
  const bool kv_side = !cached && c.sw != c.st;
  // their bf16 operand copy of the text is read by nothing else in this entry point (the backward's K/V weight gradients read it on
  // the same side stream): with a side stream the cast goes there too (round 6), off the chain
  if (!cached && c.dt == ETP_BF16) {
    if (kv_side) ETP_TRY(stream_after(p, c.st, c.sw));
    ETP_TRY(cast_f32_to_bf16(txt, s.txtT, (long)Mt * H, kv_side ? c.sw : c.st));
    txtT = s.txtT;
  }
  ETP_TRY(gmap_embed_fwd(c.dt, img, step_ids, pos, p->pf(p->step_emb), p->pf(p->gpos_w), p->pf(p->gpos_b), p->pf(p->gpos_g),
                         p->pf(p->gpos_bb), s.x0.f, lp(s.x0, c.dt), s.st0, Mg, H, cf.ang_feat + 3, c.st));
  const float* spw = cf.use_sprels ? p->pf(p->sp_w) : nullptr;
  const float* spb = cf.use_sprels ? p->pf(p->sp_b) : nullptr;
  Act x = s.x0;
  std::vector<hipEvent_t> kv_ready(cf.n_x);
  if (kv_side) {
    if (c.dt != ETP_BF16) ETP_TRY(stream_after(p, c.st, c.sw));       // (bf16: forked above, in front of the operand cast)
    Ctx cs = c;
    cs.st = c.sw;
    // (round 6: the n_x projections as ONE grouped launch -- on this stream or on the chain itself -- measured +0.9 % on config 2 and +0.8 %
    // on config 5, profiles/r06_ab_runs.json r6c9: the grouped class is the older 16x16x32 family, and x-layer 0 then waits for all four)
    for (int l = 0; l < cf.n_x; ++l) {
      ETP_TRY(linear_fwd(cs, txtT, H, p->xl[l].kv_w, p->xl[l].kv_b, s.layers[l].cross.kv, 2 * H, Mt, 2 * H, H, ETP_ACT_NONE,
                         nullptr, nullptr, 0));
      kv_ready[l] = p->next_event();
      ETP_CHECK_HIP(event_record(kv_ready[l], c.sw));
    }
  }
  for (int l = 0; l < cf.n_x; ++l) {   // GraphLXRTXLayer.forward vilmodel_cmt.py:383-398
    const XLayerP& q = p->xl[l];
    XStash& t = s.layers[l];
    stamp_mark(c.st, 1200 + l);
    // cross attention nodes -> text (BertXAttention :360-363)
    ETP_TRY(linear_fwd(c, x.t, H, q.q_w, q.q_b, t.cross.q, H, Mg, H, H, ETP_ACT_NONE, nullptr, nullptr, 0));
    if (kv_side) ETP_CHECK_HIP(stream_wait_event(c.st, kv_ready[l]));
    else if (!cached)
      ETP_TRY(linear_fwd(c, txtT, H, q.kv_w, q.kv_b, t.cross.kv, 2 * H, Mt, 2 * H, H, ETP_ACT_NONE, nullptr, nullptr, 0));
    void* kv = cached ? kc.kv[l] : t.cross.kv;
    AttnBuf a{t.cross.q, (long)H, kv, 2L * H, offs(kv, H, c.es), 2L * H, B, G, L, ldL, txt_mask, 0, nullptr, nullptr, nullptr};
    a.Pd = t.cross.Pd;
    a.kv_mod = cached ? kv_mod : 0;
    ETP_TRY(attn_fwd_impl(c.dt, c.nh, a, t.cross.P, t.cross.ctx, H, 0.125f, c.st, att(c, MODE_NAV, l, SITE_X_P)));
    ETP_TRY(linear_fwd_s(c, t.cross.ctx, H, q.xo_w, q.xo_b, t.cross.s, Mg, H, H, x.f, hid(c, MODE_NAV, l, SITE_X_O)));
    ETP_TRY(ln_fwd_s(c.dt, t.cross.s, p->pf(q.xln_g), p->pf(q.xln_b), t.cross.y.f, lp(t.cross.y, c.dt), t.cross.st, Mg, H, eps,
                     c.st));
    // graph self attention with the pairwise-distance bias (:391-393)
    ETP_TRY(self_att_fwd(c, q.self, t.cross.y, t.self, B, G, gmask, cf.use_sprels ? dists : nullptr, spw, spb, eps, MODE_NAV, l));
    // bf16 mode: gmap_embeds is the last LayerNorm's own fp32 output (the head and the backward read the bf16 copy y.t);
    // fp32 mode keeps the stash copy because there y.t IS y.f
    const bool direct = l == cf.n_x - 1 && c.dt == ETP_BF16;
    if (direct) t.ffn.y.f = out_embeds;
    ETP_TRY(ffn_fwd(c, q.ffn, t.self.y, t.ffn, Mg, eps, MODE_NAV, l));
    x = t.ffn.y;
  }
  if (cf.n_x == 0 || c.dt != ETP_BF16) ETP_TRY(copy_f32(x.f, out_embeds, (long)Mg * H, c.st));
  // SAP head: Linear -> ReLU (GEMM epilogue) -> LN -> Linear(H->1) -> masks
  ETP_TRY(linear_fwd(c, x.t, H, p->sap0_w, p->sap0_b, s.r, H, Mg, H, H, ETP_ACT_RELU, nullptr, nullptr, 0));
  ETP_TRY(sap_tail_fwd(c.dt, s.r, p->pf(p->sap2_g), p->pf(p->sap2_b), p->pf(p->sap4_w), p->pf(p->sap4_b), visited, gmask,
                       out_logits, s.str, Mg, H, c.st, site(c, p->p_head, MODE_NAV, 0, SITE_HEAD)));
  stamp_mark(c.st, 1299);
  return ETP_OK;
}
}  // namespace
extern "C" {
int etp_nav_fwd(etp_planner* p, const float* txt, const uint8_t* txt_mask, const int64_t* step_ids, const float* img,
                const float* pos, const uint8_t* gmask, const uint8_t* visited, const float* dists, int B, int L, int G,
                float* out_embeds, float* out_logits, void* stash, etp_stream_t stream) {
  ETP_REQUIRE(txt, "txt_embeds required");
  return nav_fwd_impl(p, txt, nullptr, txt_mask, step_ids, img, pos, gmask, visited, dists, B, L, G, out_embeds, out_logits, stash,
                      stream);
}
int etp_nav_fwd_kv(etp_planner* p, const void* kvbuf, const uint8_t* txt_mask, const int64_t* step_ids, const float* img,
                   const float* pos, const uint8_t* gmask, const uint8_t* visited, const float* dists, int B, int L, int G,
                   float* out_embeds, float* out_logits, void* stash, etp_stream_t stream) {
  ETP_REQUIRE(kvbuf, "K/V cache required");
  return nav_fwd_impl(p, nullptr, const_cast<void*>(kvbuf), txt_mask, step_ids, img, pos, gmask, visited, dists, B, L, G, out_embeds,
                      out_logits, stash, stream);
}
}  // extern "C"
namespace {

int nav_bwd_impl(etp_planner* p, const float* d_embeds, const float* d_logits, const float* txt, const void* kvbuf,
                 const uint8_t* txt_mask, const int64_t* step_ids, const float* pos, const uint8_t* gmask, const uint8_t* visited,
                 const float* dists, int B, int L, int G, float* d_txt, void* d_kv, float* d_img, void* stash, void* ws,
                 etp_stream_t stream, int kv_mod = 0) {
  const bool cached = kvbuf != nullptr;
  ETP_REQUIRE(p && p->P && p->G && (cached ? d_kv != nullptr : (txt && d_txt)) && txt_mask && step_ids && pos && gmask && visited &&
                  d_img && stash && ws && B > 0 && L > 0 && G > 0 && (d_embeds || d_logits),
              "bad arguments");
  Ctx c = make_ctx(p, stream);
  std::vector<std::function<int()>> pend;
  std::vector<GemmArgs> wq;
  if (c.sw != c.st) c.pend = &pend;
  c.wq = &wq;
  Bump b(stash);
  NavStash s = plan_nav(p, b, B, L, G);
  Bump wb(ws);
  NavWs n = plan_nav_ws(p, wb, B, L, G);
  const etp_config& cf = p->cfg;
  const int H = c.H, Mg = B * G, Mt = B * L, ldL = (int)round_up(L, 8);
  const float* spw = cf.use_sprels ? p->pf(p->sp_w) : nullptr;
  const float* spb = cf.use_sprels ? p->pf(p->sp_b) : nullptr;
  float* dspw = cf.use_sprels ? p->gf(p->sp_w) : nullptr;
  float* dspb = cf.use_sprels ? p->gf(p->sp_b) : nullptr;
  const void* txtT = c.dt == ETP_BF16 ? s.txtT : (const void*)txt;
  KvCache kc;
  if (cached) kc = plan_kv(p, const_cast<void*>(kvbuf), kv_mod > 0 ? kv_mod : B, L);
  const Act xlast = cf.n_x == 0 ? s.x0 : s.layers[cf.n_x - 1].ffn.y;
  float* g = n.g;
  stamp_mark(c.st, 2000);
  if (d_logits) {
    ETP_TRY(sap_tail_bwd(c.dt, d_logits, s.r, p->pf(p->sap2_g), p->pf(p->sap2_b), p->pf(p->sap4_w), s.str, visited, gmask,
                         n.head.t2, p->gf(p->sap2_g), p->gf(p->sap2_b), p->gf(p->sap4_w), p->gf(p->sap4_b), Mg, H, c.st,
                         site(c, p->p_head, MODE_NAV, 0, SITE_HEAD)));
    ETP_TRY(linear_wgrad(c, n.head.t2, H, xlast.t, H, p->sap0_w, p->sap0_b, Mg, H, H));
    ETP_TRY(linear_dgrad_s(c, n.head.t2, H, p->sap0_w, g, Mg, H, H, d_embeds));
  } else {
    ETP_TRY(copy_f32(d_embeds, g, (long)Mg * H, c.st));
  }
  if (cf.n_x == 0 && !cached) ETP_CHECK_HIP(memset_async(d_txt, 0, (size_t)Mt * H * 4, c.st));
  for (int l = cf.n_x - 1; l >= 0; --l) {
    const XLayerP& q = p->xl[l];
    const XStash& t = s.layers[l];
    const Act x = l == 0 ? s.x0 : s.layers[l - 1].ffn.y;
    const NavCrossWs& xc = n.cross[l];
    const BwdWs& w = xc.w;
    stamp_mark(c.st, 2010 + 10 * l);
    ETP_TRY(ffn_bwd(c, q.ffn, t.self.y, t.ffn, Mg, g, n.ffn[l], MODE_NAV, l));
    ETP_TRY(self_att_bwd(c, q.self, t.cross.y, t.self, B, G, gmask, cf.use_sprels ? dists : nullptr, spw, spb, dspw, dspb, g,
                         n.self[l], MODE_NAV, l));
    // cross attention backward
    const Drop dx = hid(c, MODE_NAV, l, SITE_X_O);
    ETP_TRY(ln_bwd_chain(c, g, t.cross.s, t.cross.st, q.xln_g, q.xln_b, nullptr, w.t1.f, lp2(c, w.t1, dx), Mg, dx, w.lnp));
    const void* dso = op2(c, w.t1, dx);
    ETP_TRY(linear_wgrad(c, dso, H, t.cross.ctx, H, q.xo_w, q.xo_b, Mg, H, H));
    void* kv = cached ? kc.kv[l] : t.cross.kv;
    // with the cache, dK|dV of this step go straight to the caller's d_kv block of this layer (summed over the rollout's
    // steps by the caller, projected back to the text once by etp_nav_kv_bwd)
    void* dkv_out = cached ? offs(d_kv, (long)l * Mt * 2 * H, c.es) : xc.dkv;
    AttnBuf a{t.cross.q, (long)H, kv, 2L * H, offs(kv, H, c.es), 2L * H, B, G, L, ldL, txt_mask, 0, nullptr, nullptr, nullptr};
    a.Pd = t.cross.Pd;
    a.O = t.cross.ctx; a.ldo = H;
    a.kv_mod = cached ? kv_mod : 0;
    ETP_TRY(attn_bwd_proj(c, a, t.cross.P, dso, q.xo_w, w.t2, Mg, xc.dPx, xc.dq, H, dkv_out, 2L * H, offs(dkv_out, H, c.es), 2L * H,
                          nullptr, nullptr, att(c, MODE_NAV, l, SITE_X_P)));
    ETP_TRY(linear_wgrad(c, xc.dq, H, x.t, H, q.q_w, q.q_b, Mg, H, H));
    // layer 0 leaves dL/d(node embeddings) = dL/d(gmap_img_fts) in the caller's d_img directly
    ETP_TRY(linear_dgrad_s(c, xc.dq, H, q.q_w, l == 0 ? d_img : g, Mg, H, H, w.t1.f));
    if (cached) { ETP_TRY(flush_side(c)); continue; }
    ETP_TRY(linear_wgrad(c, xc.dkv, 2 * H, txtT, H, q.kv_w, q.kv_b, Mt, 2 * H, H));
    // d_txt (consumed by the text backward right after this entry point) must not queue behind this entry point's weight
    // gradients on the weight-gradient stream; with a dedicated stream its serial accumulate chain leaves the node chain
    // (joined back at the end of this entry point), without one it stays on the main stream
    {
      Ctx c3 = c;
      if (c.s3 != c.st) { ETP_TRY(stream_after(p, c.st, c.s3)); c3.st = c.s3; }
      ETP_TRY(linear_dgrad_s(c3, xc.dkv, 2 * H, q.kv_w, d_txt, Mt, 2 * H, H, nullptr, l == cf.n_x - 1 ? 0 : 1));
    }
    ETP_TRY(flush_side(c));          // this layer's weight gradients: one fork
  }
  if (cf.n_x == 0) ETP_TRY(copy_f32(g, d_img, (long)Mg * H, c.st));
  stamp_mark(c.st, 2090);
  // NAV_TAIL (round 6; bit 0, default on): the node-embedding backward produces parameter gradients only -- a LEAF like the weight gradients:
  // it rides on their stream instead of holding the chain for ~38 us.  Bit 1 (default on, lazy joins only = PlannerStep): the chain does
  // not wait for the d txt_embeds accumulate chain on the aux2 stream here; its consumers do (etp_txt_bwd_range, etp_planner_join_aux), so
  // the node-assembly backward and the panorama fork overlap it.  Same-box A/B (r06_ab_runs.json r6c10, three rounds): 3.977 against
  // 4.010 ms (-0.8 %); bit 0 alone 3.992; device-side stamps: `nav_bwd x-layer 0 done -> text backward begins` 99 -> 50 us.
  const int tail = opt_int(OPT_NAV_TAIL, 3);
  if (!cached && c.s3 != c.st) {
    if ((tail & 2) && p->lazy_join >= 1) {
      if (!p->dtxt_ready) ETP_CHECK_HIP(hipEventCreateWithFlags(&p->dtxt_ready, hipEventDisableTiming));
      ETP_CHECK_HIP(event_record(p->dtxt_ready, c.s3));
      (void)p->dtxt_owed(c.st);
      p->dtxt_waiters.push_back(c.st);
    } else {
      ETP_TRY(stream_after(p, c.s3, c.st));                                // d_txt complete in `stream` order
      (void)p->dtxt_owed(c.st);
    }
  }
  stamp_mark(c.st, 2091);
  {
    const int dt = c.dt, PK = cf.ang_feat + 3;
    const float* st0 = s.st0;
    auto embed_bwd = [=](hipStream_t st) -> int {
      return gmap_embed_bwd(dt, d_img, step_ids, pos, p->pf(p->gpos_w), p->pf(p->gpos_b), p->pf(p->gpos_g), st0, p->gf(p->step_emb),
                            p->gf(p->gpos_w), p->gf(p->gpos_b), p->gf(p->gpos_g), p->gf(p->gpos_bb), Mg, H, PK, st);
    };
    if ((tail & 1) && c.sw != c.st) {
      hipStream_t sw = c.sw;
      ETP_TRY(on_side(c, [=]() -> int { return embed_bwd(sw); }));
    } else {
      ETP_TRY(embed_bwd(c.st));
    }
  }
  return finish_wgrads(c);
}
}  // namespace

extern "C" {
int etp_nav_bwd(etp_planner* p, const float* d_embeds, const float* d_logits, const float* txt, const uint8_t* txt_mask,
                const int64_t* step_ids, const float* pos, const uint8_t* gmask, const uint8_t* visited, const float* dists, int B,
                int L, int G, float* d_txt, float* d_img, void* stash, void* ws, etp_stream_t stream) {
  ETP_REQUIRE(txt && d_txt, "txt_embeds and d_txt_embeds required");
  return nav_bwd_impl(p, d_embeds, d_logits, txt, nullptr, txt_mask, step_ids, pos, gmask, visited, dists, B, L, G, d_txt, nullptr,
                      d_img, stash, ws, stream);
}
// Batched rollout on the cache WITHOUT the replicated copy (round 6, VERDICT r5 missing #4 / N1): B = T * Bt stacked episodes, the cache
// and the key masks hold the Bt instructions, episode e reads instruction e % Bt inside the cross-attention kernels.
static int steps_ok(const etp_planner* p, int B, int L, int G, int Bt) {
  ETP_REQUIRE(p && Bt > 0 && B % Bt == 0, "B must be a multiple of Bt");
  ETP_REQUIRE(p->cfg.dtype == ETP_BF16 && L <= 128 && G <= 128,
              "per-episode K/V indirection needs the register-resident attention kernels (bf16, L and G <= 128): use etp_nav_kv_repeat");
  return ETP_OK;
}
int etp_nav_fwd_kv_steps(etp_planner* p, const void* kvbuf, const uint8_t* txt_mask, const int64_t* step_ids, const float* img,
                         const float* pos, const uint8_t* gmask, const uint8_t* visited, const float* dists, int B, int L, int G, int Bt,
                         float* out_embeds, float* out_logits, void* stash, etp_stream_t stream) {
  ETP_REQUIRE(kvbuf, "K/V cache required");
  ETP_TRY(steps_ok(p, B, L, G, Bt));
  return nav_fwd_impl(p, nullptr, const_cast<void*>(kvbuf), txt_mask, step_ids, img, pos, gmask, visited, dists, B, L, G, out_embeds,
                      out_logits, stash, stream, Bt);
}
int etp_nav_bwd_kv_steps(etp_planner* p, const float* d_embeds, const float* d_logits, const void* kvbuf, const uint8_t* txt_mask,
                         const int64_t* step_ids, const float* pos, const uint8_t* gmask, const uint8_t* visited, const float* dists,
                         int B, int L, int G, int Bt, void* d_kv, float* d_img, void* stash, void* ws, etp_stream_t stream) {
  ETP_REQUIRE(kvbuf && d_kv, "K/V cache and d_kv required");
  ETP_TRY(steps_ok(p, B, L, G, Bt));
  return nav_bwd_impl(p, d_embeds, d_logits, nullptr, kvbuf, txt_mask, step_ids, pos, gmask, visited, dists, B, L, G, nullptr, d_kv,
                      d_img, stash, ws, stream, Bt);
}
int etp_nav_bwd_kv(etp_planner* p, const float* d_embeds, const float* d_logits, const void* kvbuf, const uint8_t* txt_mask,
                   const int64_t* step_ids, const float* pos, const uint8_t* gmask, const uint8_t* visited, const float* dists,
                   int B, int L, int G, void* d_kv, float* d_img, void* stash, void* ws, etp_stream_t stream) {
  ETP_REQUIRE(kvbuf && d_kv, "K/V cache and d_kv required");
  return nav_bwd_impl(p, d_embeds, d_logits, nullptr, kvbuf, txt_mask, step_ids, pos, gmask, visited, dists, B, L, G, nullptr, d_kv,
                      d_img, stash, ws, stream);
}
}  // extern "C"

// ======================================================================================
// Pre-training MLM task: GlocalTextPathCMT.forward_mlm pretrain vilmodel.py:708-754 ->
// GraphLXRTXLayer.forward_lang2visn :400-411 in every x-layer (the text attends to the UNCHANGING graph-node inputs) ->
// BertOnlyMLMHead :258-299 on the masked positions (pretrain_cmt.py:141-163) -> cross-entropy over the vocabulary.
// The decoder is tied to the word-embedding table: its gradient accumulates into that table's gradient.
// ======================================================================================
namespace {
constexpr int MODE_MLM = 4;
struct MlmLayerStash { void *q, *kv, *P, *ctx; float* s; float* st; Act y; void* Pd; SelfAttStash self; FfnStash ffn; };
struct MlmStash {
  void* langT; Act nodes; float* st0;
  std::vector<MlmLayerStash> layers;
  void *wordT, *hm, *tz, *tg, *hn, *dl; float* stn; float* logits;
};
MlmStash plan_mlm(const etp_planner* pl, Bump& b, int Bn, int L, int G, int Nm) {
  const etp_config& c = pl->cfg;
  const int dt = c.dtype;
  const size_t es = dtype_size(dt);
  const long Mt = (long)Bn * L, Mg = (long)Bn * G;
  const int H = c.hidden, ldG = (int)round_up(G, 8), ldL = (int)round_up(L, 8);
  const long ldv = round_up(c.vocab, 8);
  MlmStash s;
  s.langT = dt == ETP_BF16 ? b.take(Mt * H * es) : nullptr;
  s.nodes = take_act(b, dt, Mg * H);
  s.st0 = (float*)b.take(Mg * 2 * sizeof(float));
  for (int l = 0; l < c.n_x; ++l) {
    MlmLayerStash x;
    x.q = b.take(Mt * H * es);
    x.kv = b.take(Mg * 2 * H * es);
    x.P = b.take((size_t)Bn * c.heads * L * ldG * es);
    x.ctx = b.take(Mt * H * es);
    x.s = (float*)b.take(Mt * H * 4);
    x.st = (float*)b.take(Mt * 2 * sizeof(float));
    x.y = take_act(b, dt, Mt * H);
    x.Pd = attn_needs_unfused(dt, L, G) ? b.take((size_t)Bn * c.heads * L * ldG * es) : nullptr;
    x.self = plan_self(b, dt, Mt, Bn, c.heads, L, ldL, H);
    x.ffn = plan_ffn(b, dt, Mt, H, c.inter);
    s.layers.push_back(x);
  }
  s.wordT = dt == ETP_BF16 ? b.take((size_t)c.vocab * H * es) : nullptr;
  s.hm = b.take((size_t)Nm * H * es);
  s.tz = b.take((size_t)Nm * H * es);
  s.tg = b.take((size_t)Nm * H * es);
  s.hn = b.take((size_t)Nm * H * es);
  s.stn = (float*)b.take((size_t)Nm * 2 * sizeof(float));
  s.dl = b.take((size_t)Nm * ldv * es);
  s.logits = (float*)b.take((size_t)Nm * ldv * 4);
  return s;
}
struct MlmWs { float* g; float* d_nodes; void *d_hn, *d_tg; float* d_hm; std::vector<BwdWs> ffn, self, cross; std::vector<void*> dq, dkv, dPx; };
MlmWs plan_mlm_ws(const etp_planner* pl, Bump& b, int Bn, int L, int G, int Nm) {
  const etp_config& c = pl->cfg;
  const int dt = c.dtype;
  const size_t es = dtype_size(dt);
  const long Mt = (long)Bn * L, Mg = (long)Bn * G;
  const int H = c.hidden, ldL = (int)round_up(L, 8), ldG = (int)round_up(G, 8);
  MlmWs w;
  w.g = (float*)b.take(Mt * H * 4);
  w.d_nodes = (float*)b.take(Mg * H * 4);
  w.d_hn = b.take((size_t)Nm * H * es);
  w.d_tg = b.take((size_t)Nm * H * es);
  w.d_hm = (float*)b.take((size_t)Nm * H * 4);
  for (int l = 0; l < c.n_x; ++l) {
    w.ffn.push_back(plan_ws(b, dt, Mt, Bn, c.heads, L, ldL, H, c.inter));
    w.self.push_back(plan_ws(b, dt, Mt, Bn, c.heads, L, ldL, H, c.inter));
    w.cross.push_back(plan_ws(b, dt, Mt, Bn, c.heads, L, ldL, H, c.inter));
    w.dq.push_back(b.take(Mt * H * es));
    w.dkv.push_back(b.take(Mg * 2 * H * es));
    w.dPx.push_back(b.take((size_t)Bn * c.heads * L * ldG * es));
  }
  return w;
}
}  // namespace

extern "C" {
int64_t etp_mlm_stash_bytes(const etp_planner* p, int B, int L, int G, int Nm) {
  if (!p || !p->cfg.use_lang2visn) return 0;
  Bump b(nullptr);
  plan_mlm(p, b, B, L, G, Nm);
  return (int64_t)b.off + 256;
}
int64_t etp_mlm_ws_bytes(const etp_planner* p, int B, int L, int G, int Nm) {
  if (!p || !p->cfg.use_lang2visn) return 0;
  Bump b(nullptr);
  plan_mlm_ws(p, b, B, L, G, Nm);
  return (int64_t)b.off + 256;
}

int etp_mlm_fwd(etp_planner* p, const float* txt, const uint8_t* txt_mask, const int64_t* step_ids, const float* img,
                const float* pos, const uint8_t* gmask, const int32_t* sel_ptr, const int32_t* sel_idx, const float* sel_w,
                const int64_t* labels, int B, int L, int G, int Nm, float scale, float* loss, void* stash, etp_stream_t stream) {
  ETP_REQUIRE(p && p->P && p->cfg.use_lang2visn, "planner was not built with use_lang2visn (pre-training variant)");
  ETP_REQUIRE(txt && txt_mask && step_ids && img && pos && gmask && sel_ptr && sel_idx && sel_w && labels && loss && stash &&
                  B > 0 && L > 0 && G > 0 && Nm > 0,
              "bad arguments");
  Ctx c = make_ctx(p, stream);
  Bump b(stash);
  MlmStash s = plan_mlm(p, b, B, L, G, Nm);
  const etp_config& cf = p->cfg;
  const int H = c.H, Mt = B * L, Mg = B * G, ldG = (int)round_up(G, 8);
  const float eps = cf.ln_eps;
  const long ldv = round_up(cf.vocab, 8);
  Act x;
  x.f = const_cast<float*>(txt);
  x.t = x.f;
  if (c.dt == ETP_BF16) { ETP_TRY(cast_f32_to_bf16(txt, s.langT, (long)Mt * H, c.st)); x.t = s.langT; }
  ETP_TRY(gmap_embed_fwd(c.dt, img, step_ids, pos, p->pf(p->step_emb), p->pf(p->gpos_w), p->pf(p->gpos_b), p->pf(p->gpos_g),
                         p->pf(p->gpos_bb), s.nodes.f, lp(s.nodes, c.dt), s.st0, Mg, H, cf.ang_feat + 3, c.st));
  for (int l = 0; l < cf.n_x; ++l) {
    const XLayerP& q = p->xl[l];
    MlmLayerStash& t = s.layers[l];
    // visual_attention with the roles swapped: queries from the text, keys/values from the node inputs
    ETP_TRY(linear_fwd(c, x.t, H, q.q_w, q.q_b, t.q, H, Mt, H, H, ETP_ACT_NONE, nullptr, nullptr, 0));
    ETP_TRY(linear_fwd(c, s.nodes.t, H, q.kv_w, q.kv_b, t.kv, 2 * H, Mg, 2 * H, H, ETP_ACT_NONE, nullptr, nullptr, 0));
    AttnBuf a{t.q, (long)H, t.kv, 2L * H, offs(t.kv, H, c.es), 2L * H, B, L, G, ldG, gmask, 0, nullptr, nullptr, nullptr};
    a.Pd = t.Pd;
    ETP_TRY(attn_fwd_impl(c.dt, c.nh, a, t.P, t.ctx, H, 0.125f, c.st, att(c, MODE_MLM, l, SITE_X_P)));
    ETP_TRY(linear_fwd_s(c, t.ctx, H, q.xo_w, q.xo_b, t.s, Mt, H, H, x.f, hid(c, MODE_MLM, l, SITE_X_O)));
    ETP_TRY(ln_fwd_s(c.dt, t.s, p->pf(q.xln_g), p->pf(q.xln_b), t.y.f, lp(t.y, c.dt), t.st, Mt, H, eps, c.st));
    ETP_TRY(self_att_fwd(c, q.lself, t.y, t.self, B, L, txt_mask, nullptr, nullptr, nullptr, eps, MODE_MLM, l));
    ETP_TRY(ffn_fwd(c, q.lffn, t.self.y, t.ffn, Mt, eps, MODE_MLM, l));
    x = t.ffn.y;
  }
  // masked positions only (pretrain_cmt.py:148-149), then the MLM head
  ETP_TRY(gather_sum(c.dt, x.t, sel_ptr, sel_idx, sel_w, s.hm, Nm, H, 0, c.st));
  ETP_TRY(linear_fwd(c, s.hm, H, p->mlm_w, p->mlm_b, s.tg, H, Nm, H, H, ETP_ACT_GELU, s.tz, nullptr, 0));
  ETP_TRY(ln_fwd(c.dt, s.tg, p->pf(p->mlm_g), p->pf(p->mlm_bb), s.hn, s.stn, Nm, H, eps, c.st));
  const void* wordT = p->pf(p->word);
  if (c.dt == ETP_BF16) { ETP_TRY(cast_f32_to_bf16(p->pf(p->word), s.wordT, (long)cf.vocab * H, c.st)); wordT = s.wordT; }
  {
    GemmArgs g = base_args();      // logits[Nm, V] = hn . word^T + bias   (decoder tied to the embedding table)
    g.A = s.hn; g.lda = H; g.B = wordT; g.ldb = H; g.C = s.logits; g.ldc = ldv;
    g.M = Nm; g.N = cf.vocab; g.K = H; g.bias = p->pf(p->mlm_vb);
    ETP_TRY(launch_gemm(c.dt, ETP_F32, 0, 0, g, 1, c.st));
  }
  return vocab_ce(c.dt, s.logits, labels, loss, s.dl, Nm, cf.vocab, (int)ldv, scale, c.st);
}

int etp_mlm_bwd(etp_planner* p, const float* txt, const uint8_t* txt_mask, const int64_t* step_ids, const float* pos,
                const uint8_t* gmask, const int32_t* selT_ptr, const int32_t* selT_idx, const float* selT_w, int B, int L, int G,
                int Nm, float* d_txt, float* d_img, void* stash, void* ws, etp_stream_t stream) {
  ETP_REQUIRE(p && p->P && p->G && p->cfg.use_lang2visn, "planner was not built with use_lang2visn (pre-training variant)");
  ETP_REQUIRE(txt && txt_mask && step_ids && pos && gmask && selT_ptr && selT_idx && selT_w && d_txt && d_img && stash && ws &&
                  B > 0 && L > 0 && G > 0 && Nm > 0,
              "bad arguments");
  Ctx c = make_ctx(p, stream);
  std::vector<std::function<int()>> pend;
  std::vector<GemmArgs> wq;
  if (c.sw != c.st) c.pend = &pend;
  c.wq = &wq;
  Bump b(stash);
  MlmStash s = plan_mlm(p, b, B, L, G, Nm);
  Bump wb(ws);
  MlmWs w = plan_mlm_ws(p, wb, B, L, G, Nm);
  const etp_config& cf = p->cfg;
  const int H = c.H, Mt = B * L, Mg = B * G, ldG = (int)round_up(G, 8);
  const long ldv = round_up(cf.vocab, 8);
  const void* wordT = c.dt == ETP_BF16 ? s.wordT : (const void*)p->pf(p->word);
  // head: dlogits (saved by the forward CE) -> tied decoder / bias gradients, d hn
  ETP_TRY(linear_wgrad(c, s.dl, ldv, s.hn, H, p->word, -1, Nm, cf.vocab, H));
  {
    const int dt = c.dt;
    hipStream_t sw = c.sw;
    const void* dlp = s.dl;
    float* dvb = p->gf(p->mlm_vb);                 // the bias slot is padded to ldv entries in the arena (64-element alignment)
    ETP_TRY(on_side(c, [=]() -> int { return colsum(dt, dlp, ldv, dvb, Nm, (int)ldv, sw); }));
  }
  {
    GemmArgs g = base_args();      // d hn[Nm, H] = dlogits[Nm, V] . word[V, H]
    g.A = s.dl; g.lda = ldv; g.B = wordT; g.ldb = H; g.C = w.d_hn; g.ldc = H;
    g.M = Nm; g.N = H; g.K = cf.vocab;
    ETP_TRY(launch_gemm(c.dt, c.dt, 0, 1, g, 1, c.st));
  }
  ETP_TRY(ln_bwd(c.dt, w.d_hn, s.tg, s.stn, p->pf(p->mlm_g), nullptr, w.d_tg, p->gf(p->mlm_g), p->gf(p->mlm_bb), Nm, H, c.st));
  ETP_TRY(gelu_bwd_inplace(c.dt, w.d_tg, s.tz, (long)Nm * H, c.st));
  ETP_TRY(linear_wgrad(c, w.d_tg, H, s.hm, H, p->mlm_w, p->mlm_b, Nm, H, H));
  ETP_TRY(linear_dgrad_s(c, w.d_tg, H, p->mlm_w, w.d_hm, Nm, H, H, nullptr));
  // scatter the masked rows back into the text gradient (zero elsewhere)
  float* g = w.g;
  ETP_TRY(gather_sum(ETP_F32, w.d_hm, selT_ptr, selT_idx, selT_w, g, Mt, H, 0, c.st));
  for (int l = cf.n_x - 1; l >= 0; --l) {
    const XLayerP& q = p->xl[l];
    const MlmLayerStash& t = s.layers[l];
    Act x;
    if (l == 0) { x.f = const_cast<float*>(txt); x.t = c.dt == ETP_BF16 ? s.langT : (void*)x.f; }
    else x = s.layers[l - 1].ffn.y;
    const BwdWs& wc = w.cross[l];
    ETP_TRY(ffn_bwd(c, q.lffn, t.self.y, t.ffn, Mt, g, w.ffn[l], MODE_MLM, l));
    ETP_TRY(self_att_bwd(c, q.lself, t.y, t.self, B, L, txt_mask, nullptr, nullptr, nullptr, nullptr, nullptr, g, w.self[l],
                         MODE_MLM, l));
    const Drop dx = hid(c, MODE_MLM, l, SITE_X_O);
    ETP_TRY(ln_bwd_chain(c, g, t.s, t.st, q.xln_g, q.xln_b, nullptr, wc.t1.f, lp2(c, wc.t1, dx), Mt, dx, wc.lnp));
    const void* dso = op2(c, wc.t1, dx);
    ETP_TRY(linear_wgrad(c, dso, H, t.ctx, H, q.xo_w, q.xo_b, Mt, H, H));
    AttnBuf a{t.q, (long)H, t.kv, 2L * H, offs(t.kv, H, c.es), 2L * H, B, L, G, ldG, gmask, 0, nullptr, nullptr, nullptr};
    a.Pd = t.Pd;
    a.O = t.ctx; a.ldo = H;
    void* dq = w.dq[l]; void* dkv = w.dkv[l];
    ETP_TRY(attn_bwd_proj(c, a, t.P, dso, q.xo_w, wc.t2, Mt, w.dPx[l], dq, H, dkv, 2L * H, offs(dkv, H, c.es), 2L * H, nullptr,
                          nullptr, att(c, MODE_MLM, l, SITE_X_P)));
    ETP_TRY(linear_wgrad(c, dq, H, x.t, H, q.q_w, q.q_b, Mt, H, H));
    ETP_TRY(linear_dgrad_s(c, dq, H, q.q_w, g, Mt, H, H, wc.t1.f));
    ETP_TRY(linear_wgrad(c, dkv, 2 * H, s.nodes.t, H, q.kv_w, q.kv_b, Mg, 2 * H, H));
    ETP_TRY(linear_dgrad_s(c, dkv, 2 * H, q.kv_w, w.d_nodes, Mg, 2 * H, H, nullptr, l == cf.n_x - 1 ? 0 : 1));
    ETP_TRY(flush_side(c));
  }
  ETP_TRY(copy_f32(g, d_txt, (long)Mt * H, c.st));
  if (cf.n_x == 0) ETP_CHECK_HIP(memset_async(w.d_nodes, 0, (size_t)Mg * H * 4, c.st));
  ETP_TRY(copy_f32(w.d_nodes, d_img, (long)Mg * H, c.st));
  ETP_TRY(gmap_embed_bwd(c.dt, d_img, step_ids, pos, p->pf(p->gpos_w), p->pf(p->gpos_b), p->pf(p->gpos_g), s.st0,
                         p->gf(p->step_emb), p->gf(p->gpos_w), p->gf(p->gpos_b), p->gf(p->gpos_g), p->gf(p->gpos_bb), Mg, H,
                         cf.ang_feat + 3, c.st));
  return join_wgrads(c);
}
}  // extern "C"
