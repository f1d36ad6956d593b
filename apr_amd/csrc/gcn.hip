// The overlap-attention module of one scan pair as ONE library call (GCN.forward, Predator_APR/models/gcn.py:171-205;
// SelfAttention :38-77, MultiHeadedAttention :101-116, AttentionalPropagation :119-128).
//
// On the ~1.4 k points of KPFCNN's coarsest level every kernel of this module is a few workgroups; the Python modules
// issued them as 62 library calls and 6 torch.cat per pair, and KPFCNN's single scheduler thread -- not the GPU -- was the
// limit of the stacked pipeline (scripts/predator_host_split.py: 2.0-2.7 ms of host time per pair against 2.6-2.8 ms of
// wall time; the same rate with 2 hardware queues as with 4).  Here the chain leaves from C over one scratch arena: the
// kernels, their arguments and their order are the modules' own (same bits, tests/test_predator_gpu.py), the
// concatenations are column slices of one buffer.  Pure host code.
#include "common.h"

namespace {
inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Gcn {
  const apr_gcn_desc& D;
  bool dry;
  char* base;
  size_t off = 0;
  hipStream_t st;
  int rc = APR_OK;
  // shared by the two clouds (one stream: they run one behind the other)
  int32_t* knn;
  float *e, *y, *ss, *cat4, *x3, *q, *k, *v, *att, *cat2, *h, *h2, *delta;
  void* stat;
  size_t stat_bytes;
  float* cur[2];
  float* nxt[2];

  Gcn(const apr_gcn_desc& d, void* scratch, hipStream_t s)
      : D(d), dry(scratch == nullptr), base((char*)(((uintptr_t)scratch + 255) & ~(uintptr_t)255)), st(s) {}
  void* take(size_t bytes) {
    void* p = dry ? nullptr : (void*)(base + off);
    off += al256(bytes);
    return p;
  }
  void run(int r) {
    if (rc == APR_OK && r != APR_OK) rc = r;
  }
  bool ok() const { return rc == APR_OK; }

  void carve(int32_t n0, int32_t n1) {
    const int64_t n = n0 > n1 ? n0 : n1;
    const int c = D.c;
    int kmax = 1;
    for (int i = 0; i < D.n_layers; ++i)
      if (D.layer[i].kind == 0 && D.layer[i].k > kmax) kmax = D.layer[i].k;
    const int64_t nk = n * kmax;
    knn = (int32_t*)take((size_t)nk * 4);
    e = (float*)take((size_t)nk * 2 * c * 4);
    y = (float*)take((size_t)nk * 2 * c * 4);
    ss = (float*)take((size_t)4 * c * 4);
    cat4 = (float*)take((size_t)n * 4 * c * 4);
    x3 = (float*)take((size_t)n * c * 4);
    q = (float*)take((size_t)n * c * 4);
    k = (float*)take((size_t)n * c * 4);
    v = (float*)take((size_t)n * c * 4);
    att = (float*)take((size_t)n * c * 4);
    cat2 = (float*)take((size_t)n * 2 * c * 4);
    h = (float*)take((size_t)n * 2 * c * 4);
    h2 = (float*)take((size_t)n * 2 * c * 4);
    delta = (float*)take((size_t)n * c * 4);
    stat_bytes = apr_bn_stats_scratch_bytes(nk > n ? nk : n, 2 * c);
    stat = take(stat_bytes);
    cur[0] = (float*)take((size_t)n0 * c * 4);
    cur[1] = (float*)take((size_t)n1 * c * 4);
    nxt[0] = (float*)take((size_t)n0 * c * 4);
    nxt[1] = (float*)take((size_t)n1 * c * 4);
  }

  int gemm(const float* in, int64_t ldi, int64_t m, int cin, int cout, const void* w, const float* shift, float* out, int64_t ldo) {
    return apr_dense_gemm_bf3(in, ldi, m, cin, cout, w, nullptr, shift, nullptr, 0, 0, out, ldo, st);
  }

  // SelfAttention.forward on one cloud: x [n, c] (ldx) -> out [n, c] (ldo)
  void self_layer(const apr_gcn_layer& L, const float* pts, int32_t n, const float* x, int64_t ldx, float* out, int64_t ldo) {
    const int c = D.c, kk = L.k;
    const int64_t nk = (int64_t)n * kk;
    run(apr_knn(pts, n, kk, 1, knn, st));
    // edge conv 1: [x_i, x_j - x_i] -> conv1 -> InstanceNorm2d over n*k -> LeakyReLU(0.2) -> max over the k neighbours
    if (ok()) run(apr_edge_features(x, ldx, n, c, knn, kk, e, st));
    if (ok()) run(gemm(e, 2 * c, nk, 2 * c, c, L.w1, nullptr, y, c));
    if (ok()) run(apr_norm_params(y, c, nk, c, L.eps1, ss, ss + c, stat, stat_bytes, st));
    if (ok()) run(apr_group_max(y, c, n, kk, c, ss, ss + c, 0.2f, cat4 + c, 4 * c, st));          // x1 -> columns [c, 2c) of [x0 | x1 | x2]
    // edge conv 2 on x1
    if (ok()) run(apr_edge_features(cat4 + c, 4 * c, n, c, knn, kk, e, st));
    if (ok()) run(gemm(e, 2 * c, nk, 2 * c, 2 * c, L.w2, nullptr, y, 2 * c));
    if (ok()) run(apr_norm_params(y, 2 * c, nk, 2 * c, L.eps2, ss, ss + 2 * c, stat, stat_bytes, st));
    if (ok()) run(apr_group_max(y, 2 * c, n, kk, 2 * c, ss, ss + 2 * c, 0.2f, cat4 + 2 * c, 4 * c, st));
    if (ok()) run(apr_affine_act(x, ldx, n, c, nullptr, nullptr, nullptr, 0, 0, 0.f, cat4, 4 * c, st));   // x0 -> columns [0, c)
    if (ok()) run(gemm(cat4, 4 * c, n, 4 * c, c, L.w3, nullptr, x3, c));
    if (ok()) run(apr_instance_norm_act(x3, c, n, c, L.eps3, nullptr, 0, 2, 0.2f, out, ldo, stat, stat_bytes, st));
  }

  // x + AttentionalPropagation(x, source): x [n, c], source [m, c] -> out [n, c]
  void cross_layer(const apr_gcn_layer& L, const float* x, int64_t ldx, int32_t n, const float* src, int64_t lds, int32_t m,
                   float* out, int64_t ldo) {
    const int c = D.c;
    run(gemm(x, ldx, n, c, c, L.wq, L.bq, q, c));
    if (ok()) run(gemm(src, lds, m, c, c, L.wk, L.bk, k, c));
    if (ok()) run(gemm(src, lds, m, c, c, L.wv, L.bv, v, c));
    if (ok()) run(apr_mha_headmajor(q, k, v, n, m, c / L.heads, L.heads, att, st));
    if (ok()) run(gemm(att, c, n, c, c, L.wm, L.bm, cat2 + c, 2 * c));                                  // message -> [x | message]
    if (ok()) run(apr_affine_act(x, ldx, n, c, nullptr, nullptr, nullptr, 0, 0, 0.f, cat2, 2 * c, st));
    if (ok()) run(gemm(cat2, 2 * c, n, 2 * c, 2 * c, L.w1, L.b1, h, 2 * c));
    if (ok()) run(apr_instance_norm_act(h, 2 * c, n, 2 * c, L.eps1, nullptr, 0, 1, 0.f, h2, 2 * c, stat, stat_bytes, st));
    if (ok()) run(gemm(h2, 2 * c, n, 2 * c, c, L.w2, L.b2, delta, c));
    if (ok()) run(apr_affine_act(delta, c, n, c, nullptr, nullptr, x, ldx, 0, 0.f, out, ldo, st));
  }
};

bool gcn_ok(const apr_gcn_desc* d, int32_t n0, int32_t n1) {
  if (!d || d->n_layers < 1 || d->n_layers > APR_GCN_MAX_LAYERS || d->c <= 0 || d->c % 64 != 0 || n0 <= 0 || n1 <= 0) return false;
  for (int i = 0; i < d->n_layers; ++i) {
    const apr_gcn_layer& L = d->layer[i];
    if (L.kind == 0) {
      if (L.k < 1 || L.k + 1 > 16 || !L.w1 || !L.w2 || !L.w3 || L.k >= n0 || L.k >= n1) return false;
    } else if (L.kind == 1) {
      if (L.heads < 1 || d->c % L.heads != 0 || d->c / L.heads != 64 || !L.w1 || !L.w2 || !L.wq || !L.wk || !L.wv || !L.wm) return false;
    } else {
      return false;
    }
  }
  return true;
}
}  // namespace

APR_API size_t apr_gcn_scratch_bytes(const apr_gcn_desc* d, int32_t n0, int32_t n1) {
  if (!gcn_ok(d, n0, n1)) return 0;
  Gcn g(*d, nullptr, nullptr);
  g.carve(n0, n1);
  return g.off + 512;
}

APR_API int apr_gcn_forward(const apr_gcn_desc* d, const float* pts0, int32_t n0, const float* pts1, int32_t n1, const float* x0,
                            int64_t ldx0, const float* x1, int64_t ldx1, float* out0, int64_t ldo0, float* out1, int64_t ldo1,
                            void* scratch, size_t scratch_bytes, void* stream) {
  APR_CHECK_ARG(gcn_ok(d, n0, n1), "apr_gcn_forward: descriptor / sizes not covered (c %% 64 == 0, 64 channels per head, k < n)");
  APR_CHECK_ARG(pts0 && pts1 && x0 && x1 && out0 && out1 && scratch && ldx0 >= d->c && ldx1 >= d->c && ldo0 >= d->c && ldo1 >= d->c,
                "apr_gcn_forward: bad arguments");
  APR_CHECK_ARG(scratch_bytes >= apr_gcn_scratch_bytes(d, n0, n1), "apr_gcn_forward: scratch too small");
  APR_CHECK_ARG(ldx0 % 4 == 0 && ldx1 % 4 == 0 && ((uintptr_t)x0 | (uintptr_t)x1) % 16 == 0,
                "apr_gcn_forward: the descriptors must be 16-byte aligned rows");
  Gcn g(*d, scratch, (hipStream_t)stream);
  g.carve(n0, n1);
  const float* in[2] = {x0, x1};
  int64_t ldi[2] = {ldx0, ldx1};
  const float* pts[2] = {pts0, pts1};
  const int32_t n[2] = {n0, n1};
  float* const outp[2] = {out0, out1};
  const int64_t ldo[2] = {ldo0, ldo1};
  const int c = d->c;
  for (int li = 0; li < d->n_layers && g.ok(); ++li) {
    const apr_gcn_layer& L = d->layer[li];
    const bool last = li == d->n_layers - 1;
    // where this layer's results go: the caller's rows for the last layer, else the other pair of buffers
    float* dst[2];
    int64_t ldd[2];
    for (int s = 0; s < 2; ++s) {
      dst[s] = last ? outp[s] : (in[s] == g.cur[s] ? g.nxt[s] : g.cur[s]);
      ldd[s] = last ? ldo[s] : c;
    }
    if (L.kind == 0) {
      g.self_layer(L, pts[0], n[0], in[0], ldi[0], dst[0], ldd[0]);
      if (g.ok()) g.self_layer(L, pts[1], n[1], in[1], ldi[1], dst[1], ldd[1]);
    } else {
      // desc0 <- desc0 + layer(desc0, desc1); desc1 <- desc1 + layer(desc1, the NEW desc0)   (gcn.py:197-199)
      g.cross_layer(L, in[0], ldi[0], n[0], in[1], ldi[1], n[1], dst[0], ldd[0]);
      if (g.ok()) g.cross_layer(L, in[1], ldi[1], n[1], dst[0], ldd[0], n[0], dst[1], ldd[1]);
    }
    for (int s = 0; s < 2; ++s) {
      in[s] = dst[s];
      ldi[s] = ldd[s];
    }
  }
  return g.rc;
}
