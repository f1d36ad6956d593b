// One ResUNet encode (ResUNet2.forward in eval mode, FCGF_APR/model/resunet.py:142-191; blocks: model/residual_block.py:
// 23-53) enqueued by ONE library call: apr_resunet_encode.
//
// The module-by-module plan (apr_amd/fcgf/model/resunet.py forward_fused) issues conv1 on the occupancy bitmap, 7 kernel
// maps + 3 transposed ones, the pair lists of the routed layers and 23 conv launches through ~35 ctypes calls, ~80
// allocations and a few hundred attribute lookups: 0.8-1.1 ms of Python per encode whatever the batch holds.  With one
// pair per call (the reference loop's shape, FCGF_APR/scripts/test_apr.py:111-163) that is more than the GPU needs for
// the same encode: the host set the pace (scripts/one_pair_split.py: 1.13 ms of host time in the encode + match phase, the
// GPU waited for).  Here the same launches -- same kernels, same arguments, same order, hence the same bits -- leave from
// C over one scratch arena.  Pure host code: every launch goes through the library's own entry points.
//
// The walk is written ONCE (struct Enc) and run twice: dry (sizes the arena) and live (carves it and launches), so the two
// cannot disagree.
#include "common.h"

namespace {

constexpr int kLevels = 4, kStages = 7, kMaps = 10, kMaxLists = 24;
inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

inline bool ws_shape_ok(int K, int cin, int cout) { return K <= 27 && cin % 64 == 0 && cin <= 512 && cout % 64 == 0; }

struct ListSlot {     // a pair list of one kernel map: kind 0 per-offset, 1 triples, 2 output-stationary tiles
  int map, kind, rows;
  void* blob;
  size_t bytes;
  int32_t* counters;
  bool built;
};

struct Enc {
  const apr_resunet_plan& P;
  const apr_level_map* lv;
  const int32_t* bbox;
  bool dry;
  char* base;
  size_t off = 0;
  hipStream_t st;
  int rc = APR_OK;

  int32_t* kmap[kMaps] = {};
  ListSlot lists[kMaxLists];
  int nlists = 0, ncounter = 0;
  int32_t* counters = nullptr;
  int n_counter_slots = 0;
  void* occ = nullptr;
  size_t occ_bytes = 0;
  int32_t* tpool[3] = {};
  bool tpool_ready = false;
  float* prod = nullptr;
  size_t prod_bytes = 0;

  apr_spconv_desc descs[4];
  int ndesc = 0;

  Enc(const apr_resunet_plan& p, const apr_level_map* l, const int32_t* b, void* scratch, hipStream_t s)
      : P(p), lv(l), bbox(b), dry(scratch == nullptr), base((char*)(((uintptr_t)scratch + 255) & ~(uintptr_t)255)), st(s) {}

  void* take(size_t bytes) {
    void* q = dry ? nullptr : (void*)(base + off);
    off += al256(bytes);
    return q;
  }
  float* rows(int64_t n, int c) { return (float*)take((size_t)n * c * 4); }
  bool ok() const { return rc == APR_OK; }
  void run(int r) {
    if (rc == APR_OK && r != APR_OK) rc = r;
  }

  // map ids: same-level l -> l; strided l -> l + 1: 4 + l; transposed l + 1 -> l: 7 + l
  int64_t map_rows(int m) const { return m < 4 ? lv[m].n : m < 7 ? lv[m - 4 + 1].n : lv[m - 7].n; }
  int64_t map_in_rows(int m) const { return m < 4 ? lv[m].n : m < 7 ? lv[m - 4].n : lv[m - 7 + 1].n; }

  int32_t* kernel_map(int m) {
    if (kmap[m] || (dry && kmap_sized[m])) return kmap[m];
    kmap_sized[m] = true;
    const int64_t n_out = map_rows(m);
    if (m >= 7) {
      // coarse -> fine: the transpose of the strided fine -> coarse table (the encoder built it): a scatter into a table
      // that one fill cleared together with the other levels'
      const int l = m - 7;
      int32_t* fwd = kernel_map(4 + l);
      if (!tpool_ready) {
        size_t ints = 0;
        for (int a = 0; a < 3; ++a) ints += (size_t)lv[a].n * 27;
        ints = (ints + 63) / 64 * 64 + 64;
        int32_t* buf = (int32_t*)take(ints * 4);
        size_t pos = 0;
        for (int a = 0; a < 3; ++a) {
          tpool[a] = dry ? nullptr : buf + pos;
          pos += (size_t)lv[a].n * 27;
        }
        if (!dry) run(apr_fill_bytes(buf, 0xFF, ints * 4, st));
        tpool_ready = true;
      }
      kmap[m] = tpool[l];
      if (!dry) run(apr_kernel_map_transpose_prefilled(fwd, lv[l + 1].n, 27, lv[l].n, kmap[m], st));
      return kmap[m];
    }
    kmap[m] = (int32_t*)take((size_t)n_out * 27 * 4);
    if (dry) return kmap[m];
    const int lin = m < 4 ? m : m - 4, lout = m < 4 ? m : m - 4 + 1;
    const apr_level_map& in = lv[lin];
    const apr_level_map& outm = lv[lout];
    if (lin == 0 && P.occ_kernel_map && occ)
      run(apr_kernel_map_occ(outm.coords, outm.n, nullptr, in.keys, in.vals, in.cap, 3, 1, bbox, P.conv1_ks, occ, kmap[m], st));
    else
      run(apr_kernel_map(outm.coords, outm.n, nullptr, in.keys, in.vals, in.cap, 3, 1 << lin, kmap[m], st));
    return kmap[m];
  }
  bool kmap_sized[kMaps] = {};

  ListSlot* find_list(int m, int kind, int rws) {
    for (int i = 0; i < nlists; ++i)
      if (lists[i].map == m && lists[i].kind == kind && lists[i].rows == rws) return &lists[i];
    return nullptr;
  }

  ListSlot* ws_list(const apr_resunet_layer& L, int m, bool same_level) {
    bool tri = P.ws3 && L.w_bf3 && apr_spconv_ws3_supported(L.K, L.cin, L.cout);
    // 128 input channels on the largest same-level maps: the zero-padded MFMA share costs more than the halved product
    // rows save (resunet.py: WS3_MAX_ROWS_128)
    if (tri && L.cin == 128 && ((same_level && map_rows(m) > P.ws3_max_rows_128) || !P.ws3_cin128)) tri = false;
    const int kind = tri ? 1 : 0;
    ListSlot* s = find_list(m, kind, 0);
    if (s) return s;
    if (nlists >= kMaxLists) {
      apr_set_error("apr_resunet_encode: more than %d pair lists", kMaxLists);
      run(APR_EINVAL);
      return nullptr;
    }
    s = &lists[nlists++];
    s->map = m;
    s->kind = kind;
    s->rows = 0;
    s->bytes = tri ? apr_pairlist3_bytes(map_rows(m)) : apr_pairlist_bytes(map_rows(m), 27);
    s->blob = take(s->bytes);
    s->built = false;
    s->counters = nullptr;
    if (!dry) {
      if (ncounter >= n_counter_slots) {
        apr_set_error("apr_resunet_encode: more than %d pair lists need a counter block", n_counter_slots);
        run(APR_EINVAL);
        return nullptr;
      }
      s->counters = counters + (size_t)ncounter * apr_pairlist_counter_ints();
    }
    ++ncounter;
    return s;
  }

  ListSlot* os_list(const apr_resunet_layer& L, int m) {
    const int64_t n_out = map_rows(m), n_in = map_in_rows(m);
    const int rws = n_in < (1ll << 23) ? apr_spconv_os_tile_rows(n_out, L.cin, L.cout) : 0;
    if (rws <= 0) return nullptr;
    ListSlot* s = find_list(m, 2, rws);
    if (s) return s;
    if (nlists >= kMaxLists) {
      apr_set_error("apr_resunet_encode: more than %d pair lists", kMaxLists);
      run(APR_EINVAL);
      return nullptr;
    }
    s = &lists[nlists++];
    s->map = m;
    s->kind = 2;
    s->rows = rws;
    s->bytes = apr_spconv_os_pairs_bytes(n_out, 27, rws);
    s->blob = take(s->bytes);
    s->built = false;
    s->counters = nullptr;
    return s;
  }

  float* prod_rows(size_t floats) {      // launches run in order on one stream: one product buffer, as large as the largest need
    if (dry) {
      if (floats * 4 > prod_bytes) prod_bytes = floats * 4;
      return nullptr;
    }
    return prod;
  }

  // one conv launch (SpconvBatch.add in apr_amd/ops.py, the same routing)
  void conv(const apr_resunet_layer& L, const float* in, int64_t ldi, int m, int64_t n_out, ListSlot* list, const float* residual,
            int64_t ldr, float* out, int64_t ldo, bool l2norm) {
    if (!ok()) return;
    int32_t* nbr = m >= 0 ? kernel_map(m) : nullptr;
    apr_spconv_desc d;
    memset(&d, 0, sizeof(d));
    d.in = in; d.ldi = ldi; d.nbr = nbr; d.n_out = n_out;
    d.K = L.K; d.cin = L.cin; d.cout = L.cout; d.relu = L.relu;
    d.w_packed = L.w_packed; d.scale = L.scale; d.shift = L.shift;
    d.residual = residual; d.ldr = residual ? ldr : 0;
    d.out = out; d.ldo = ldo;
    d.l2norm = l2norm ? 1 : 0;
    if (list && list->kind == 2 && L.w_bf3 && m >= 0) {
      d.os_pairs = list->blob; d.os_rows = list->rows; d.os_n_in = map_in_rows(m); d.w_bf3 = L.w_bf3;
      if (!list->built) {
        d.os_build_bytes = (int64_t)list->bytes;
        list->built = true;
      }
    } else if (list && list->kind != 2 && m >= 0 && ws_shape_ok(L.K, L.cin, L.cout)) {
      const bool is3 = list->kind == 1;
      d.prod_scratch = prod_rows((size_t)n_out * (is3 ? 9 : 27) * L.cout);
      d.counters = list->counters; d.plist = list->blob; d.ws3 = is3 ? 1 : 0;
      d.w_bf3 = L.w_bf3;
      if (!list->built) {
        d.plist_bytes = (int64_t)list->bytes;
        list->built = true;
      }
    } else if (m < 0 && L.K == 1 && L.w_bf3 && ((L.cin % 64 == 0 && L.cout % 64 == 0) || apr_dense_rows_bf3_ok(L.cin, L.cout)) &&
               ldi % 4 == 0 && ldo % 4 == 0 &&
               (dry || (((uintptr_t)in | (uintptr_t)out) % 16 == 0 && (!residual || (ldr % 4 == 0 && (uintptr_t)residual % 16 == 0)) &&
                        (!L.scale || (uintptr_t)L.scale % 16 == 0) && (!L.shift || (uintptr_t)L.shift % 16 == 0)))) {
      d.w_bf3 = L.w_bf3;          // identity map: the dense GEMMs on the bf16 split (dense.hip / dense_rows.hip)
    }
    if (!dry) descs[ndesc++] = d;
  }
  void flush() {      // a stage's launches leave together (forward_fused: batch.launch() per stage)
    if (!dry && ok() && ndesc > 0) run(apr_spconv_fwd_batch(descs, ndesc, st));
    ndesc = 0;
  }

  void walk(float* out, int64_t ldo) {
    const apr_resunet_layer* L = P.layer;
    const int64_t n[4] = {lv[0].n, lv[1].n, lv[2].n, lv[3].n};
    // arena head: the pieces whose size does not depend on the routing
    occ_bytes = apr_occ_conv_scratch_bytes(bbox, P.conv1_ks);
    occ = take(occ_bytes + 256);
    if (!counters) {
      n_counter_slots = 16;
      counters = (int32_t*)take((size_t)n_counter_slots * apr_pairlist_counter_ints() * 4);
      if (!dry) run(apr_fill_bytes(counters, 0, (size_t)n_counter_slots * apr_pairlist_counter_ints() * 4, st));
    }
    if (!dry) prod = (float*)take(prod_bytes_live);

    // feature rows.  Widths from the layers: CH[l] = cout of stage l's conv, TR = cout of the transposed stages
    const int ch1 = L[0].cout, ch2 = L[3].cout, ch3 = L[6].cout, ch4 = L[9].cout;
    const int tr4 = L[12].cout, tr3 = L[15].cout, tr2 = L[18].cout;
    const int w1 = tr2 + ch1, w2 = tr3 + ch2, w3 = tr4 + ch3;      // concat buffers: [decoder output | encoder skip]
    float* cat1 = rows(n[0], w1);
    float* cat2 = rows(n[1], w2);
    float* cat3 = rows(n[2], w3);
    float* s8 = rows(n[3], ch4);
    // per stage: the conv's output (the block's residual) and the block's intermediate
    int64_t amax = 0;
    {
      const int64_t c[kStages] = {n[0] * ch1, n[1] * ch2, n[2] * ch3, n[3] * ch4, n[2] * tr4, n[1] * tr3, n[0] * tr2};
      for (int s = 0; s < kStages; ++s) amax = c[s] > amax ? c[s] : amax;
    }
    float* A = (float*)take((size_t)amax * 4);
    float* H = (float*)take((size_t)amax * 4);
    float* h1 = rows(n[0], L[21].cout);

    struct StageIo {
      const float* in; int64_t ldi; int cmap, bmap, lout; float* out; int64_t ldo;
    };
    const StageIo io[kStages] = {
        {nullptr, 0, -1, 0, 0, cat1 + tr2, w1},        // "1": conv1 on occupancy, block1 -> skip columns of cat1
        {cat1 + tr2, w1, 4, 1, 1, cat2 + tr3, w2},     // "2"
        {cat2 + tr3, w2, 5, 2, 2, cat3 + tr4, w3},     // "3"
        {cat3 + tr4, w3, 6, 3, 3, s8, ch4},            // "4"
        {s8, ch4, 9, 2, 2, cat3, w3},                  // "4_tr" -> decoder columns of cat3
        {cat3, w3, 8, 1, 1, cat2, w2},                 // "3_tr"
        {cat2, w2, 7, 0, 0, cat1, w1},                 // "2_tr"
    };
    for (int s = 0; s < kStages && ok(); ++s) {
      const apr_resunet_layer &C = L[3 * s], &B1 = L[3 * s + 1], &B2 = L[3 * s + 2];
      const StageIo& q = io[s];
      const int64_t n_out = n[q.lout];
      if (s == 0) {
        if (!dry)
          run(apr_occ_conv(lv[0].coords, n_out, bbox, P.conv1_ks, P.conv1_w, C.cout, C.scale, C.shift, nullptr, 0, C.relu, A,
                           C.cout, occ, occ_bytes + 256, st));
      } else {
        ListSlot* cl = ((P.ws_conv >> s) & 1u) && ws_shape_ok(C.K, C.cin, C.cout) ? ws_list(C, q.cmap, false) : nullptr;
        conv(C, q.in, q.ldi, q.cmap, n_out, cl, nullptr, 0, A, C.cout, false);
      }
      ListSlot* bl = nullptr;
      if (((P.os_block >> s) & 1u) && B1.cin == 64 && B1.w_bf3 && n_out >= P.os_min_rows) bl = os_list(B1, q.bmap);
      if (!bl && ((P.ws_block >> s) & 1u) && ws_shape_ok(B1.K, B1.cin, B1.cout)) bl = ws_list(B1, q.bmap, true);
      conv(B1, A, B1.cin, q.bmap, n_out, bl, nullptr, 0, H, B1.cout, false);
      conv(B2, H, B2.cin, q.bmap, n_out, bl, A, B1.cin, q.out, q.ldo, false);
      flush();
    }
    conv(L[21], cat1, w1, -1, n[0], nullptr, nullptr, 0, h1, L[21].cout, false);
    conv(L[22], h1, L[21].cout, -1, n[0], nullptr, nullptr, 0, out, ldo, P.normalize != 0);
    flush();
  }
  size_t prod_bytes_live = 0;
};

bool plan_ok(const apr_resunet_plan* p, const apr_level_map* lv, const int32_t* bbox) {
  if (!p || !lv || !bbox || !p->conv1_w) return false;
  for (int l = 0; l < kLevels; ++l)
    if (lv[l].n <= 0 || !lv[l].coords || !lv[l].keys || !lv[l].vals || lv[l].cap <= 0 || (lv[l].cap & (lv[l].cap - 1))) return false;
  for (int i = 0; i < 23; ++i)
    if (!p->layer[i].w_packed || p->layer[i].cin <= 0 || p->layer[i].cout <= 0) return false;
  for (int i = 1; i < 21; ++i)
    if (p->layer[i].K != 27) return false;
  if (p->layer[21].K != 1 || p->layer[22].K != 1) return false;
  // the chain's widths must fit together (a variant with other skip wiring takes the module-by-module plan)
  const apr_resunet_layer* L = p->layer;
  for (int s = 0; s < kStages; ++s)
    if (L[3 * s + 1].cin != L[3 * s].cout || L[3 * s + 1].cout != L[3 * s].cout || L[3 * s + 2].cin != L[3 * s].cout ||
        L[3 * s + 2].cout != L[3 * s].cout)
      return false;
  if (L[3].cin != L[0].cout || L[6].cin != L[3].cout || L[9].cin != L[6].cout || L[12].cin != L[9].cout ||
      L[15].cin != L[12].cout + L[6].cout || L[18].cin != L[15].cout + L[3].cout || L[21].cin != L[18].cout + L[0].cout ||
      L[22].cin != L[21].cout)
    return false;
  if (L[0].cin != 1 || L[0].K != p->conv1_ks * p->conv1_ks * p->conv1_ks || L[0].cout % 8 != 0) return false;
  if (p->conv1_ks != 3 && p->conv1_ks != 5 && p->conv1_ks != 7) return false;
  return true;
}

}  // namespace

APR_API int apr_resunet_encode_supported(const apr_resunet_plan* plan, const apr_level_map* lv, const int32_t* bbox_host) {
  if (!plan_ok(plan, lv, bbox_host)) return 0;
  return apr_occ_conv_pays(bbox_host, plan->conv1_ks, lv[0].n) ? 1 : 0;
}

APR_API size_t apr_resunet_encode_scratch_bytes(const apr_resunet_plan* plan, const apr_level_map* lv, const int32_t* bbox_host) {
  if (!apr_resunet_encode_supported(plan, lv, bbox_host)) return 0;
  Enc e(*plan, lv, bbox_host, nullptr, nullptr);
  e.walk(nullptr, plan->layer[22].cout);
  return e.off + al256(e.prod_bytes) + 512;
}

APR_API int apr_resunet_encode(const apr_resunet_plan* plan, const apr_level_map* lv, const int32_t* bbox_host, int32_t* counters,
                               int32_t n_counter_slots, void* scratch, size_t scratch_bytes, float* out, int64_t ldo,
                               void* stream) {
  APR_CHECK_ARG(apr_resunet_encode_supported(plan, lv, bbox_host),
                "apr_resunet_encode: plan / maps / box not covered (apr_resunet_encode_supported)");
  APR_CHECK_ARG(scratch && out && ldo >= plan->layer[22].cout && (!counters || n_counter_slots > 0),
                "apr_resunet_encode: bad arguments");
  Enc dry(*plan, lv, bbox_host, nullptr, nullptr);
  dry.walk(nullptr, ldo);
  APR_CHECK_ARG(scratch_bytes >= dry.off + al256(dry.prod_bytes) + 512, "apr_resunet_encode: scratch too small");
  Enc e(*plan, lv, bbox_host, scratch, (hipStream_t)stream);
  e.counters = counters;
  e.n_counter_slots = counters ? n_counter_slots : 0;
  e.prod_bytes_live = dry.prod_bytes;
  e.walk(out, ldo);
  return e.rc;
}

// ---------------------------------------------------------------------------------------------------------------
// The front end of a step as one call (see include/apr_hip.h): the same kernels in the same order as
// PairRegistration.voxelize_batch issued them tensor by tensor (apr_amd/fcgf/pipeline.py).
// ---------------------------------------------------------------------------------------------------------------
namespace {

struct Front {
  bool dry;
  char* base;
  size_t off = 0;
  Front(void* arena) : dry(arena == nullptr), base((char*)(((uintptr_t)arena + 255) & ~(uintptr_t)255)) {}
  void* take(size_t bytes) {
    void* q = dry ? nullptr : (void*)(base + off);
    off += al256(bytes);
    return q;
  }
};

struct MapBufs {
  uint64_t* keys; int32_t* vals; int64_t cap; int32_t* coords; int64_t* first; int32_t *n_dev, *status; void* scratch;
  size_t scratch_bytes; int64_t rows;
};

// the builds run one behind the other on one stream: they share ONE scratch (sized for the largest, the first)
MapBufs take_map(Front& f, int64_t rows, bool want_first, int32_t* hdr_slot, void* shared_scratch, size_t shared_bytes) {
  MapBufs m;
  m.rows = rows;
  m.cap = apr_hash_capacity(rows);
  m.keys = (uint64_t*)f.take((size_t)m.cap * 8);
  m.vals = (int32_t*)f.take((size_t)m.cap * 4);
  m.coords = (int32_t*)f.take((size_t)rows * 16);
  m.first = want_first ? (int64_t*)f.take((size_t)rows * 8) : nullptr;
  m.scratch_bytes = shared_bytes;
  m.scratch = shared_scratch;
  m.n_dev = hdr_slot;                   // the map writes its row count and status straight into the header
  m.status = hdr_slot ? hdr_slot + 1 : nullptr;
  return m;
}

// compact table: a scan puts >= ~8 points into a voxel, so the de-duplicating table (sized for the RAW points) is ~10x
// larger than the voxels it ends up holding and every kernel-map probe into it would miss L2 (CoordinateManager.__init__)
inline int64_t compact_rows_for(int64_t n_points) { return n_points > 4 * 65536 ? (n_points / 4 > 65536 ? n_points / 4 : 65536) : 0; }

int front_walk(Front& f, const float* const* frames, const int64_t* offs, int32_t nseg, float vs, apr_pyramid* out,
               hipStream_t st) {
  const int64_t N = offs ? offs[nseg] : 0;
  const int64_t R = compact_rows_for(N);
  const int hints = 10 + nseg + 8;
  int32_t* hdr = (int32_t*)f.take((size_t)hints * 4);
  int32_t* counters = (int32_t*)f.take((size_t)16 * apr_pairlist_counter_ints() * 4);
  int32_t* raw = (int32_t*)f.take((size_t)N * 16);
  int64_t* offs_dev = (int64_t*)f.take((size_t)(nseg + 1) * 8);
  float* pts = (float*)f.take((size_t)N * 12);
  // level 0 = the compact table if there is one, else the de-duplicating table itself
  const size_t sbytes = apr_map_scratch_bytes(N);
  void* const sc = f.take(sbytes);
  MapBufs dedup = take_map(f, N, true, f.dry ? nullptr : hdr + (R ? 8 : 0), sc, sbytes);
  MapBufs lvl[4];
  if (R) lvl[0] = take_map(f, R, false, f.dry ? nullptr : hdr, sc, sbytes);
  else lvl[0] = dedup;
  const int64_t rows = R ? R : N;
  for (int l = 1; l < 4; ++l) lvl[l] = take_map(f, rows, false, f.dry ? nullptr : hdr + 2 * l, sc, sbytes);
  if (f.dry) return APR_OK;

  int rc;
  // the header and the pair-list counters behind it (adjacent in the arena) cleared by one small launch (k_pack_i32's
  // zero leg).  A first version -- hipMemsetAsync here and a scratch of its own for each of the five builds -- cost the
  // three-steps-in-flight loop 1.7 %; the fill alone turned out neutral later, so it was the larger footprint
  const size_t clear = (size_t)((char*)counters - (char*)hdr) + (size_t)16 * apr_pairlist_counter_ints() * 4;
  if ((rc = apr_pack_i32(nullptr, nullptr, 0, nullptr, hdr, (int64_t)(clear / 4), st)) != APR_OK) return rc;
  if ((rc = apr_voxelize_frames(frames, offs, nseg, vs, raw, offs_dev, st)) != APR_OK) return rc;
  if ((rc = apr_map_build(raw, N, nullptr, 0, dedup.keys, dedup.vals, dedup.cap, dedup.coords, dedup.first, dedup.n_dev,
                          dedup.status, dedup.scratch, dedup.scratch_bytes, st)) != APR_OK)
    return rc;
  if ((rc = apr_segment_counts(dedup.first, dedup.n_dev, offs_dev, nseg, hdr + 10, st)) != APR_OK) return rc;
  if ((rc = apr_coords_bbox(raw, N, hdr + 10 + nseg, st)) != APR_OK) return rc;
  if ((rc = apr_gather_frame_points(frames, offs, nseg, dedup.first, dedup.n_dev, N, pts, st)) != APR_OK) return rc;
  if (R &&
      (rc = apr_map_build(dedup.coords, R, dedup.n_dev, 0, lvl[0].keys, lvl[0].vals, lvl[0].cap, lvl[0].coords, nullptr,
                          lvl[0].n_dev, lvl[0].status, lvl[0].scratch, lvl[0].scratch_bytes, st)) != APR_OK)
    return rc;
  for (int l = 1; l < 4; ++l)
    if ((rc = apr_map_build(lvl[l - 1].coords, rows, lvl[l - 1].n_dev, 1 << l, lvl[l].keys, lvl[l].vals, lvl[l].cap, lvl[l].coords,
                            nullptr, lvl[l].n_dev, lvl[l].status, lvl[l].scratch, lvl[l].scratch_bytes, st)) != APR_OK)
      return rc;
  for (int l = 0; l < 4; ++l) {
    out->lv[l].coords = lvl[l].coords; out->lv[l].keys = lvl[l].keys; out->lv[l].vals = lvl[l].vals;
    out->lv[l].cap = lvl[l].cap; out->lv[l].n = lvl[l].rows;
  }
  out->first = dedup.first;
  out->pts = pts;
  out->header = hdr;
  out->header_ints = hints;
  out->compact = R ? 1 : 0;
  out->compact_rows = R;
  out->counters = counters;
  out->n_counter_slots = 16;
  return APR_OK;
}

}  // namespace

APR_API size_t apr_voxel_pyramid_scratch_bytes(int64_t n_points, int32_t nseg) {
  if (n_points <= 0 || nseg <= 0 || nseg > APR_MAX_FRAMES) return 0;
  Front f(nullptr);
  int64_t offs[APR_MAX_FRAMES + 1] = {};
  offs[nseg] = n_points;
  (void)front_walk(f, nullptr, offs, nseg, 1.f, nullptr, nullptr);
  return f.off + 512;
}

APR_API int apr_voxel_pyramid(const float* const* frames_host, const int64_t* offsets_host, int32_t nseg, float voxel_size,
                              void* arena, size_t arena_bytes, apr_pyramid* out, void* stream) {
  APR_CHECK_ARG(frames_host && offsets_host && nseg >= 1 && nseg <= APR_MAX_FRAMES && arena && out && voxel_size > 0.f,
                "apr_voxel_pyramid: bad arguments (1 .. %d frames)", APR_MAX_FRAMES);
  APR_CHECK_ARG(offsets_host[0] == 0 && offsets_host[nseg] > 0, "apr_voxel_pyramid: offsets must run 0 .. total points > 0");
  APR_CHECK_ARG(arena_bytes >= apr_voxel_pyramid_scratch_bytes(offsets_host[nseg], nseg), "apr_voxel_pyramid: arena too small");
  Front f(arena);
  return front_walk(f, frames_host, offsets_host, nseg, voxel_size, out, (hipStream_t)stream);
}
