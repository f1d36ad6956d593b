// One KPConv ResnetBottleneckBlock (Predator_APR/models/blocks.py:596-681, eval mode) as ONE library call.
//
// The block is a fixed chain of 8-12 launches -- unary1 (Linear + InstanceNorm + LeakyReLU), KPConv (row sums, kernel-point
// correlation, [N, 15 mid] x [15 mid, mid] GEMM) + InstanceNorm + LeakyReLU, the shortcut (max-pool over the pooling table
// if strided, Linear + InstanceNorm if the widths differ), unary2 (Linear) + InstanceNorm + shortcut + LeakyReLU -- and the
// Python modules issued it as 8-12 ctypes calls with a dozen temporaries: 136 library calls and 2.6 ms of host time per
// scan pair for the whole network, against 2.8 ms of wall time (scripts/predator_host_split.py): the scheduler thread was
// the limit.  Here the chain is enqueued from C over one scratch arena; the kernels, their order and their arguments are
// the modules' own, so the result is the same bits (tests/test_predator_gpu.py).
#include "common.h"

namespace {
inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

struct KpCarve {
  float *u1, *rs, *wf, *kp, *pool, *sc, *y;
  void* stat;
  size_t stat_bytes, total;
};

KpCarve kp_carve(const apr_kp_resnet_desc& d, void* scratch) {
  KpCarve c;
  char* p = (char*)(((uintptr_t)scratch + 255) & ~(uintptr_t)255);
  char* const p0 = p;
  auto take = [&](size_t bytes) {
    char* q = p;
    p += al256(bytes);
    return q;
  };
  const int64_t ni = d.n_in, no = d.n_out;
  c.u1 = d.w_unary1 ? (float*)take((size_t)ni * d.mid * 4) : nullptr;
  c.rs = (float*)take((size_t)ni * 4);
  c.wf = (float*)take((size_t)no * d.n_kp * d.mid * 4);
  c.kp = (float*)take((size_t)no * d.mid * 4);
  c.pool = d.strided ? (float*)take((size_t)no * d.in_dim * 4) : nullptr;
  c.sc = d.w_shortcut ? (float*)take((size_t)no * d.out_dim * 4) : nullptr;
  c.y = (float*)take((size_t)no * d.out_dim * 4);
  const int64_t nmax = ni > no ? ni : no;
  const int32_t cmax = d.out_dim > d.in_dim ? d.out_dim : d.in_dim;
  c.stat_bytes = apr_bn_stats_scratch_bytes(nmax + 256 * (int64_t)(d.nseg > 0 ? d.nseg : 1), cmax);
  const size_t fused = apr_dense_gemm_bf3_norm_scratch_bytes(nmax, cmax, d.nseg > 0 ? d.nseg : 1);
  if (fused > c.stat_bytes) c.stat_bytes = fused;
  c.stat = take(c.stat_bytes);
  c.total = (size_t)(p - p0) + 256;
  return c;
}

int kp_norm(const apr_kp_resnet_desc& d, const KpCarve& c, const float* x, int64_t n, int32_t ch, const float* residual,
            int64_t ldr, bool act, float* y, int64_t ldy, const int64_t* seg, void* stream) {
  const int mode = act ? 2 : 0;
  if (d.nseg > 1)
    return apr_instance_norm_act_seg(x, ch, n, ch, d.eps, residual, ldr, mode, d.slope, y, ldy, seg, d.nseg, c.stat, c.stat_bytes,
                                     stream);
  return apr_instance_norm_act(x, ch, n, ch, d.eps, residual, ldr, mode, d.slope, y, ldy, c.stat, c.stat_bytes, stream);
}

// Linear + InstanceNorm (+ shortcut) (+ LeakyReLU): two launches -- the GEMM leaves the column sums of its tiles behind
// (apr_dense_gemm_bf3_norm_act) -- instead of GEMM, statistics, apply.  APR_KP_FUSED_NORM=0: the three launches (A/B).
int kp_linear_norm(const apr_kp_resnet_desc& d, const KpCarve& c, const float* x, int64_t ldx, int64_t n, int32_t cin,
                   int32_t cout, const void* w, const float* residual, int64_t ldr, bool act, float* y, int64_t ldy,
                   const int64_t* seg, void* stream) {
  static const int fused = env_int("APR_KP_FUSED_NORM", 1);
  if (fused)
    return apr_dense_gemm_bf3_norm_act(x, ldx, n, cin, cout, w, d.eps, residual, ldr, act ? 2 : 0, d.slope, y, ldy, seg,
                                       d.nseg > 1 ? d.nseg : 0, c.stat, c.stat_bytes, stream);
  int rc = apr_dense_gemm_bf3(x, ldx, n, cin, cout, w, nullptr, nullptr, nullptr, 0, 0, y, ldy, stream);
  if (rc != APR_OK) return rc;
  return kp_norm(d, c, y, n, cout, residual, ldr, act, y, ldy, seg, stream);
}
}  // namespace

APR_API size_t apr_kp_resnet_scratch_bytes(const apr_kp_resnet_desc* d) {
  if (!d || d->n_in <= 0 || d->n_out <= 0 || d->mid <= 0 || d->in_dim <= 0 || d->out_dim <= 0 || d->n_kp <= 0) return 0;
  return kp_carve(*d, nullptr).total;
}

APR_API int apr_kp_resnet_block(const apr_kp_resnet_desc* dp, void* stream) {
  APR_CHECK_ARG(dp != nullptr, "apr_kp_resnet_block: null descriptor");
  const apr_kp_resnet_desc& d = *dp;
  APR_CHECK_ARG(d.x && d.q_pts && d.s_pts && d.nbr && d.w_kpconv && d.w_unary2 && d.kernel_points && d.out && d.scratch,
                "apr_kp_resnet_block: null argument");
  APR_CHECK_ARG(d.n_in > 0 && d.n_out > 0 && d.H > 0 && d.n_kp > 0, "apr_kp_resnet_block: empty level");
  APR_CHECK_ARG(d.mid % 64 == 0 && d.in_dim % 64 == 0 && d.out_dim % 64 == 0 && d.ldx >= d.in_dim && d.ldo >= d.out_dim,
                "apr_kp_resnet_block: widths must be multiples of 64 (the bf16-split dense GEMM)");
  APR_CHECK_ARG((d.w_unary1 != nullptr) == (d.in_dim != d.mid), "apr_kp_resnet_block: unary1 exists iff in_dim != mid");
  APR_CHECK_ARG((d.w_shortcut != nullptr) == (d.in_dim != d.out_dim), "apr_kp_resnet_block: shortcut Linear exists iff in_dim != out_dim");
  APR_CHECK_ARG(d.strided || d.n_in == d.n_out, "apr_kp_resnet_block: a same-level block maps n_in rows to n_in rows");
  APR_CHECK_ARG(d.nseg <= 1 || (d.seg_in && d.seg_out), "apr_kp_resnet_block: segment offsets missing");
  const KpCarve c = kp_carve(d, d.scratch);
  APR_CHECK_ARG(d.scratch_bytes >= c.total, "apr_kp_resnet_block: scratch too small");
  int rc;
#define KP_TRY(call)            \
  do {                          \
    rc = (call);                \
    if (rc != APR_OK) return rc; \
  } while (0)
  // 1. unary1: Linear(in_dim -> mid) + InstanceNorm + LeakyReLU (Identity when the widths agree)
  const float* x1 = d.x;
  int64_t ldx1 = d.ldx;
  if (d.w_unary1) {
    KP_TRY(kp_linear_norm(d, c, d.x, d.ldx, d.n_in, d.in_dim, d.mid, d.w_unary1, nullptr, 0, true, c.u1, d.mid, d.seg_in, stream));
    x1 = c.u1;
    ldx1 = d.mid;
  }
  // 2. KPConv: kernel-point correlation, then [n_out, 15 mid] x [15 mid, mid]
  const int32_t kk = d.n_kp * d.mid;
  KP_TRY(apr_row_sums(x1, ldx1, d.n_in, d.mid, c.rs, stream));
  KP_TRY(apr_kpconv_weighted(d.q_pts, d.n_out, d.s_pts, d.n_in, d.nbr, d.H, x1, ldx1, d.mid, d.kernel_points, d.n_kp, d.extent,
                             c.rs, c.wf, kk, stream));
  // 3. ... + InstanceNorm + LeakyReLU
  KP_TRY(kp_linear_norm(d, c, c.wf, kk, d.n_out, kk, d.mid, d.w_kpconv, nullptr, 0, true, c.kp, d.mid, d.seg_out, stream));
  // 4. shortcut: max-pool over the pooling table when strided, Linear + InstanceNorm when the widths differ
  const float* sc = d.x;
  int64_t ldsc = d.ldx;
  if (d.strided) {
    KP_TRY(apr_gather_pool(d.x, d.ldx, d.n_in, d.in_dim, d.nbr, d.H, d.n_out, 0, c.pool, d.in_dim, stream));
    sc = c.pool;
    ldsc = d.in_dim;
  }
  if (d.w_shortcut) {
    KP_TRY(kp_linear_norm(d, c, sc, ldsc, d.n_out, d.in_dim, d.out_dim, d.w_shortcut, nullptr, 0, false, c.sc, d.out_dim, d.seg_out,
                          stream));
    sc = c.sc;
    ldsc = d.out_dim;
  }
  // 5. unary2 (no ReLU) + InstanceNorm + shortcut + LeakyReLU
  KP_TRY(kp_linear_norm(d, c, c.kp, d.mid, d.n_out, d.mid, d.out_dim, d.w_unary2, sc, ldsc, true, d.out, d.ldo, d.seg_out, stream));
#undef KP_TRY
  return APR_OK;
}
