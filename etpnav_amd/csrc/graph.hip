// Device-side graph-input assembly for forward_navigation.
//
// Replaces, per rollout step, the host work of RLTrainer._nav_gmap_variable (ss_trainer_ETP.py:344-417) and of the
// GraphMap queries it makes (vlnce_baselines/models/graph_utils.py): networkx all-pairs Dijkstra over the visited-node
// graph (update_graph :256-257), front_to_ghost_dist (:259-270), get_pos_fts (:278-322) and the O(G^2) Python loop that
// fills the pairwise distance matrix (ss_trainer_ETP.py:371-387), followed by H2D copies of every result.  Input: compact
// per-episode arrays (positions, edge weights, ghost fronts, current pose) -- etpnav_amd/graph_inputs.py packs them from
// the reference's GraphMap objects or from its own GraphMapLite.  Output: the padded tensors forward_navigation takes.
//
// One 256-thread workgroup per episode; the whole problem lives in LDS (<= 64 visited nodes, <= 192 ghosts):
//   1. Floyd-Warshall over the node graph, carrying the node count of each shortest path (len(nx path)),
//   2. nearest front node of every ghost,
//   3. 7-d position features, step ids, masks, pairwise distances -- written once, padded with zeros.
// Heading features avoid arcsin: calculate_vp_rel_pos_fts (:21-44) defines heading0 by sin = -dx/xz, cos = -+|dz|/xz, so
// sin/cos of (2*pi - (heading0 - base)) follow from the angle-difference identities at full fp32 accuracy.
#include "kernels.h"

namespace etp {

constexpr int GN = 64;      // max visited nodes per episode
constexpr int GM = 192;     // max ghost nodes per episode
constexpr float G_MAX_DIST = 30.f, G_MAX_STEP = 10.f;   // graph_utils.py:9-10

struct GmapArgs {
  const float* node_pos; const int32_t* node_step; const int32_t* n_nodes; const float* adj;
  const float* ghost_pos; const int32_t* n_ghost; const int32_t* front_ptr; const int32_t* front_idx;
  const int32_t* cur_node; const float* cur_pos; const float* cur_heading;
  int Nmax, Mmax, Fmax, G;
  int64_t* step_ids; uint8_t* gmask; uint8_t* visited; float* pos_fts; float* pair;
};

__device__ __forceinline__ void rel_fts(const float* a, const float* b, float sb, float cb, float* out /*[5]: sin h, cos h, sin e, cos e, dist*/) {
  const float dx = b[0] - a[0], dy = b[1] - a[1], dz = b[2] - a[2];
  const float xz_raw = sqrtf(dx * dx + dz * dz), xyz_raw = sqrtf(dx * dx + dy * dy + dz * dz);
  const float xz = fmaxf(xz_raw, 1e-8f), xyz = fmaxf(xyz_raw, 1e-8f);
  const float s0 = -dx / xz;                                   // sin(heading0), heading0 = arcsin(-dx/xz) or pi - arcsin
  const float ca = xz_raw > 1e-8f ? fabsf(dz) / xz : 1.f;      // cos(arcsin(.)) >= 0
  const float c0 = dz > 0.f ? -ca : ca;                        // "if b[2] > a[2]: heading = pi - heading"
  const float sd = s0 * cb - c0 * sb, cd = c0 * cb + s0 * sb;  // heading0 - base
  out[0] = -sd; out[1] = cd;                                   // to_clock: 2*pi - (.)
  const float se = dz / xyz;                                   // elevation = arcsin(dz / xyz)  (the reference's own axis choice)
  out[2] = se;
  out[3] = xyz_raw > 1e-8f ? sqrtf(dx * dx + dy * dy) / xyz : 1.f;
  out[4] = xyz;
}

__global__ __launch_bounds__(256) void gmap_assemble_kernel(const GmapArgs a) {
  __shared__ float D[GN][GN + 1];
  __shared__ int C[GN][GN + 1];
  __shared__ float npos[GN][3], gpos[GM][3], fd[GM];
  __shared__ int fv[GM];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = a.n_nodes[b], m = a.n_ghost[b], G = a.G, L = 1 + n + m;
  const float* adj = a.adj + (long)b * a.Nmax * a.Nmax;
  for (int e = tid; e < GN * GN; e += 256) {
    const int i = e / GN, j = e % GN;
    float d = INFINITY; int c = 0;
    if (i < n && j < n) {
      if (i == j) { d = 0.f; c = 1; }
      else { const float w = adj[i * a.Nmax + j]; if (w >= 0.f) { d = w; c = 2; } }
    }
    D[i][j] = d; C[i][j] = c;
  }
  for (int e = tid; e < n * 3; e += 256) npos[e / 3][e % 3] = a.node_pos[((long)b * a.Nmax) * 3 + e];
  for (int e = tid; e < m * 3; e += 256) gpos[e / 3][e % 3] = a.ghost_pos[((long)b * a.Mmax) * 3 + e];
  __syncthreads();
  // Floyd-Warshall; row k and column k are fixed points of iteration k, so in-place relaxation is race-free
  for (int k = 0; k < n; ++k) {
    for (int e = tid; e < n * n; e += 256) {
      const int i = e / n, j = e % n;
      const float via = D[i][k] + D[k][j];
      if (via < D[i][j]) { D[i][j] = via; C[i][j] = C[i][k] + C[k][j] - 1; }
    }
    __syncthreads();
  }
  // nearest front of every ghost (first minimum in list order, graph_utils.py:259-270)
  for (int g = tid; g < m; g += 256) {
    const int32_t* fp = a.front_ptr + (long)b * (a.Mmax + 1);
    float best = 10000.f; int bv = 0;
    for (int q = fp[g]; q < fp[g + 1]; ++q) {
      const int f = a.front_idx[(long)b * a.Fmax + q];
      const float dx = npos[f][0] - gpos[g][0], dy = npos[f][1] - gpos[g][1], dz = npos[f][2] - gpos[g][2];
      const float d = sqrtf(dx * dx + dy * dy + dz * dz);
      if (d < best) { best = d; bv = f; }
    }
    fd[g] = best; fv[g] = bv;
  }
  __syncthreads();
  const int cur = a.cur_node[b];
  const float cp[3] = {a.cur_pos[b * 3], a.cur_pos[b * 3 + 1], a.cur_pos[b * 3 + 2]};
  float sb, cb;
  sincosf(a.cur_heading[b], &sb, &cb);
  for (int t = tid; t < G; t += 256) {
    float f[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int64_t sid = 0; uint8_t vis = 0;
    if (t == 0) { f[1] = 1.f; f[3] = 1.f; }                       // vp None: angles (0, 0)
    else if (t < L) {
      float r[5];
      float sd; int ss;
      if (t <= n) {
        const int v = t - 1;
        rel_fts(cp, npos[v], sb, cb, r);
        sd = D[cur][v]; ss = C[cur][v];
        sid = a.node_step[(long)b * a.Nmax + v]; vis = 1;
      } else {
        const int g = t - 1 - n;
        rel_fts(cp, gpos[g], sb, cb, r);
        sd = D[cur][fv[g]] + fd[g]; ss = C[cur][fv[g]] + 1;
      }
      f[0] = r[0]; f[1] = r[1]; f[2] = r[2]; f[3] = r[3];
      f[4] = r[4] / G_MAX_DIST; f[5] = sd / G_MAX_DIST; f[6] = (float)ss / G_MAX_STEP;
    }
    float* o = a.pos_fts + ((long)b * G + t) * 7;
#pragma unroll
    for (int e = 0; e < 7; ++e) o[e] = f[e];
    a.step_ids[(long)b * G + t] = sid;
    a.visited[(long)b * G + t] = vis;
    a.gmask[(long)b * G + t] = t < L ? 1 : 0;
  }
  // pairwise distances (ss_trainer_ETP.py:371-387): anchor node + extra distance of every entry
  for (int e = tid; e < G * G; e += 256) {
    const int j = min(e / G, e % G), k = max(e / G, e % G);        // evaluate (j<k) once: the matrix is exactly symmetric,
    float v = 0.f;                                                 // as the reference's pair[j,k] = pair[k,j] = dist
    if (j >= 1 && k < L && j != k) {
      const int aj = j <= n ? j - 1 : fv[j - 1 - n], ak = k <= n ? k - 1 : fv[k - 1 - n];
      const float dj = j <= n ? 0.f : fd[j - 1 - n], dk = k <= n ? 0.f : fd[k - 1 - n];
      v = (dj + D[aj][ak] + dk) / G_MAX_DIST;
    }
    a.pair[(long)b * G * G + e] = v;
  }
}

// RLTrainer._vp_feature_variable (ss_trainer_ETP.py:308-342): per episode, the candidate-view features first, then the
// panorama views that are not a candidate's image in index order; zero-padded to the batch maximum; nav_types 1 / 0.
// One workgroup per output row.  cand: packed [sum K, F] with cand_ptr [B+1]; pano: [B, P, F] (pano_bstride = P*F) or one
// shared [P, F] table (pano_bstride = 0, the pano_angle_fts case); cand_mask [B, P].
__global__ __launch_bounds__(256) void vp_gather_kernel(const float* __restrict__ cand, const int32_t* __restrict__ cand_ptr,
                                                        const float* __restrict__ pano, long pano_bstride,
                                                        const uint8_t* __restrict__ cand_mask, int P, int F, int V,
                                                        float* __restrict__ out, int64_t* __restrict__ nav_types,
                                                        int64_t* __restrict__ view_lens) {
  const int b = blockIdx.x / V, v = blockIdx.x % V;
  const int K = cand_ptr[b + 1] - cand_ptr[b];
  const uint8_t* m = cand_mask + (long)b * P;
  int free_views = 0;
  for (int j = 0; j < P; ++j) free_views += m[j] ? 0 : 1;
  const int len = K + free_views;
  const float* src = nullptr;
  if (v < K) src = cand + ((long)cand_ptr[b] + v) * F;
  else if (v < len) {
    int want = v - K, j = 0;
    for (; j < P; ++j)
      if (!m[j] && want-- == 0) break;
    src = pano + b * pano_bstride + (long)j * F;
  }
  float* dst = out + ((long)b * V + v) * F;
  for (int c = threadIdx.x; c < F; c += 256) dst[c] = src ? src[c] : 0.f;
  if (threadIdx.x == 0) {
    if (nav_types) nav_types[(long)b * V + v] = v < K ? 1 : 0;
    if (view_lens && v == 0) view_lens[b] = len;
  }
}

int gmap_assemble(const GmapArgs& a, int B, hipStream_t st) {
  ETP_REQUIRE(B > 0 && a.Nmax >= 1 && a.Nmax <= GN && a.Mmax >= 0 && a.Mmax <= GM && a.G >= 1 && a.Fmax >= 0,
              "graph limits: <= 64 visited nodes and <= 192 ghosts per episode");
  ETP_LAUNCH(gmap_assemble_kernel, dim3(B), dim3(256), 0, st, a);
  ETP_CHECK_LAUNCH("gmap_assemble");
  return ETP_OK;
}

}  // namespace etp

extern "C" int etp_gmap_assemble(const float* node_pos, const int32_t* node_step, const int32_t* n_nodes, const float* adj,
                                 const float* ghost_pos, const int32_t* n_ghost, const int32_t* front_ptr,
                                 const int32_t* front_idx, const int32_t* cur_node, const float* cur_pos,
                                 const float* cur_heading, int B, int Nmax, int Mmax, int Fmax, int G, int64_t* gmap_step_ids,
                                 uint8_t* gmap_masks, uint8_t* gmap_visited_masks, float* gmap_pos_fts, float* gmap_pair_dists,
                                 etp_stream_t stream) {
  using namespace etp;
  ETP_REQUIRE(node_pos && node_step && n_nodes && adj && n_ghost && cur_node && cur_pos && cur_heading && gmap_step_ids &&
                  gmap_masks && gmap_visited_masks && gmap_pos_fts && gmap_pair_dists,
              "null pointer");
  ETP_REQUIRE(Mmax == 0 || (ghost_pos && front_ptr && front_idx), "ghost arrays required when Mmax > 0");
  GmapArgs a;
  a.node_pos = node_pos; a.node_step = node_step; a.n_nodes = n_nodes; a.adj = adj; a.ghost_pos = ghost_pos; a.n_ghost = n_ghost;
  a.front_ptr = front_ptr; a.front_idx = front_idx; a.cur_node = cur_node; a.cur_pos = cur_pos; a.cur_heading = cur_heading;
  a.Nmax = Nmax; a.Mmax = Mmax; a.Fmax = Fmax; a.G = G;
  a.step_ids = gmap_step_ids; a.gmask = gmap_masks; a.visited = gmap_visited_masks; a.pos_fts = gmap_pos_fts; a.pair = gmap_pair_dists;
  return gmap_assemble(a, B, (hipStream_t)stream);
}

extern "C" int etp_vp_gather(const float* cand_fts, const int32_t* cand_ptr, const float* pano_fts, int64_t pano_batch_stride,
                             const uint8_t* cand_mask, int B, int P, int F, int V, float* out_fts, int64_t* nav_types,
                             int64_t* view_lens, etp_stream_t stream) {
  using namespace etp;
  ETP_REQUIRE(cand_ptr && pano_fts && cand_mask && out_fts && B > 0 && P > 0 && F > 0 && V > 0, "bad arguments");
  ETP_LAUNCH(vp_gather_kernel, dim3(B * V), dim3(256), 0, (hipStream_t)stream, cand_fts, cand_ptr, pano_fts,
                     (long)pano_batch_stride, cand_mask, P, F, V, out_fts, nav_types, view_lens);
  ETP_CHECK_LAUNCH("vp_gather");
  return ETP_OK;
}
