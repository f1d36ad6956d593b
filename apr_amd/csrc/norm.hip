// Row-feature normalisation / elementwise kernels.
// All HBM-bound: 16-B vector loads where the row stride allows, fp64 accumulation
// for the batch statistics so training-mode BN matches torch's to ~1e-7.
#include "common.h"

namespace {

constexpr int kRowsPerBlock = 256;
constexpr int kMaxSeg = 32;
constexpr int kApplyRows = 256;   // rows per k_norm_apply workgroup (the statistics prelude is paid once per workgroup)

// Row segments normalised independently in ONE launch (blockIdx.z = segment): the scan pairs stacked into one KPFCNN
// forward.  row0[s] .. row0[s + 1]: rows of segment s; blk0[s]: first 256-row partial block of segment s.
struct Segs {
  long long row0[kMaxSeg + 1];
  int blk0[kMaxSeg + 1];
};

Segs one_segment(int64_t n) {
  Segs g;
  g.row0[0] = 0;
  g.row0[1] = n;
  g.blk0[0] = 0;
  g.blk0[1] = (int)cdiv64(n, kRowsPerBlock);
  return g;
}

// partial[blk][0][c] = sum x, partial[blk][1][c] = sum x^2 over this block's rows.
// 1024 threads = 64 columns x 16 row lanes: a thread sums 16 rows (two batches of 8 independent loads), the 16 row
// lanes meet in LDS in fixed order.  (256 threads x 64 rows each was a chain of 8 dependent L2 round trips per
// thread: 18 us per call whatever the size.)
__global__ __launch_bounds__(1024) void k_bn_partial(const float* __restrict__ x, int64_t ld, int c, Segs sg,
                                                     double* __restrict__ partial) {
  __shared__ double s_sum[16][64], s_sq[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = blockIdx.y * 64 + tx;
  const int seg = blockIdx.z;
  if ((int)blockIdx.x >= sg.blk0[seg + 1] - sg.blk0[seg]) return;        // workgroup-uniform: shorter segment
  x += sg.row0[seg] * ld;
  partial += (int64_t)sg.blk0[seg] * 2 * c;
  const int64_t n = sg.row0[seg + 1] - sg.row0[seg];
  const int64_t r0 = (int64_t)blockIdx.x * kRowsPerBlock;
  const int64_t r1 = min((long long)(r0 + kRowsPerBlock), (long long)n);
  // 16-B form (rows 16-byte aligned, c % 4 == 0): thread = 4 columns x one row of every 64 -- a quarter of the load
  // instructions of the 4-B form; the 64 row lanes of a column quad meet in LDS in fixed order
  const bool vec = (c & 3) == 0 && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  if (vec) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    __shared__ double s_v[64][16][8];                      // [row lane][column quad][4 sums, 4 sums of squares]
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int colv = blockIdx.y * 64 + cq * 4;
    double a4[4] = {0.0, 0.0, 0.0, 0.0}, q4[4] = {0.0, 0.0, 0.0, 0.0};
    if (colv < c) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t r = r0 + rl + 64 * u;
        v[u] = r < r1 ? *reinterpret_cast<const f32x4*>(x + r * ld + colv) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const double d = (double)v[u][e];
          a4[e] += d;
          q4[e] += d * d;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s_v[rl][cq][e] = a4[e];
      s_v[rl][cq][4 + e] = q4[e];
    }
    __syncthreads();
    if (threadIdx.x < 128) {                               // thread = (column 0..63, sum | sum of squares)
      const int cc = threadIdx.x & 63, which = threadIdx.x >> 6;
      const int colw = blockIdx.y * 64 + cc;
      if (colw < c) {
        double t = 0.0;
        for (int g = 0; g < 64; ++g) t += s_v[g][cc >> 2][which * 4 + (cc & 3)];
        partial[((int64_t)blockIdx.x * 2 + which) * c + colw] = t;
      }
    }
    return;
  }
  double s = 0.0, s2 = 0.0;
  if (col < c) {
    for (int64_t rb = r0 + ty; rb < r1; rb += 128) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t r = rb + 16 * u;
        v[u] = r < r1 ? x[r * ld + col] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const double d = (double)v[u];
        s += d;
        s2 += d * d;
      }
    }
  }
  s_sum[ty][tx] = s;
  s_sq[ty][tx] = s2;
  __syncthreads();
  if (ty == 0 && col < c) {
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      a += s_sum[g][tx];
      b += s_sq[g][tx];
    }
    partial[((int64_t)blockIdx.x * 2 + 0) * c + col] = a;
    partial[((int64_t)blockIdx.x * 2 + 1) * c + col] = b;
  }
}

// One workgroup per 64 columns: 4 groups of threads each sum a quarter of the block partials (8 loads in flight),
// the groups meet in LDS in fixed order.
__global__ __launch_bounds__(256) void k_bn_finish(const double* __restrict__ partial, int nblk, int64_t n, int c,
                                                   float* __restrict__ mean, float* __restrict__ var, float eps,
                                                   float* __restrict__ scale, float* __restrict__ shift) {
  __shared__ double s_a[4][64], s_q[4][64];
  const int tx = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + tx;
  double s = 0.0, s2 = 0.0;
  if (col < c) {
    for (int b0 = grp; b0 < nblk; b0 += 32) {
      double a[8], q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int b = b0 + 4 * u;
        const bool ok = b < nblk;
        a[u] = ok ? partial[((int64_t)b * 2 + 0) * c + col] : 0.0;
        q[u] = ok ? partial[((int64_t)b * 2 + 1) * c + col] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        s += a[u];
        s2 += q[u];
      }
    }
  }
  s_a[grp][tx] = s;
  s_q[grp][tx] = s2;
  __syncthreads();
  if (grp != 0 || col >= c) return;
  s = ((s_a[0][tx] + s_a[1][tx]) + s_a[2][tx]) + s_a[3][tx];
  s2 = ((s_q[0][tx] + s_q[1][tx]) + s_q[2][tx]) + s_q[3][tx];
  double m = s / (double)n;
  double v = s2 / (double)n - m * m;
  const float mf = (float)m, vf = (float)(v > 0.0 ? v : 0.0);
  if (mean) mean[col] = mf;
  if (var) var[col] = vf;
  if (scale) {   // no-affine normalisation parameters: y = x * scale + shift
    const float sc = 1.0f / sqrtf(vf + eps);
    scale[col] = sc;
    shift[col] = -mf * sc;
  }
}

__global__ void k_affine_act(const float* __restrict__ x, int64_t ldx, int64_t n, int c,
                             const float* __restrict__ scale, const float* __restrict__ shift,
                             const float* __restrict__ residual, int64_t ldr, int relu, float slope,
                             float* __restrict__ y, int64_t ldy) {
  const int64_t total = n * c;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = t / c;
    int col = (int)(t - r * c);
    float v = x[r * ldx + col];
    if (scale) v *= scale[col];
    if (shift) v += shift[col];
    if (residual) v += residual[r * ldr + col];
    if (relu == 1) v = fmaxf(v, 0.f);
    else if (relu == 2) v = v > 0.f ? v : v * slope;
    y[r * ldy + col] = v;
  }
}

// dz = dy * act'(z) read off the activation's OUTPUT y (ReLU: y > 0; LeakyReLU with slope > 0: sign(y) = sign(z))
__global__ void k_act_backward(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ y, int64_t ldy,
                               int64_t n, int c, int mode, float slope, float* __restrict__ dz, int64_t lddz) {
  const int64_t total = n * c;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / c;
    const int col = (int)(t - r * c);
    const float g = dy[r * lddy + col];
    const float o = y[r * ldy + col];
    dz[r * lddz + col] = (mode == 0 || o > 0.f) ? g : (mode == 2 ? g * slope : 0.f);
  }
}

// k_bn_finish + k_affine_act in one: a workgroup owns 64 columns x 64 rows; it first rebuilds the columns'
// (scale, shift) from the block partials (every workgroup of a column repeats the same fixed-order sum: a few
// hundred fp64 adds against a launch, a 4 KB round trip and a host-side call saved per normalisation), then
// y = act(x * scale + shift (+ residual)).
__global__ __launch_bounds__(256) void k_norm_apply(const float* __restrict__ x, int64_t ldx, int c, Segs sg,
                                                    const double* __restrict__ partial, float eps,
                                                    const float* __restrict__ residual, int64_t ldr, int relu,
                                                    float slope, float* __restrict__ y, int64_t ldy) {
  __shared__ double s_a[4][64], s_q[4][64];
  __shared__ __attribute__((aligned(16))) float s_scale[64];
  __shared__ __attribute__((aligned(16))) float s_shift[64];
  const int tx = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int col = blockIdx.y * 64 + tx;
  const int seg = blockIdx.z;
  const int64_t n = sg.row0[seg + 1] - sg.row0[seg];
  if ((int64_t)blockIdx.x * kApplyRows >= n) return;                       // workgroup-uniform: shorter segment
  const int nblk = sg.blk0[seg + 1] - sg.blk0[seg];
  x += sg.row0[seg] * ldx;
  y += sg.row0[seg] * ldy;
  if (residual) residual += sg.row0[seg] * ldr;
  partial += (int64_t)sg.blk0[seg] * 2 * c;
  double s = 0.0, s2 = 0.0;
  if (col < c) {
    for (int b0 = grp; b0 < nblk; b0 += 32) {      // same order as k_bn_finish: bit-identical statistics
      double a[8], q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int b = b0 + 4 * u;
        const bool ok = b < nblk;
        a[u] = ok ? partial[((int64_t)b * 2 + 0) * c + col] : 0.0;
        q[u] = ok ? partial[((int64_t)b * 2 + 1) * c + col] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        s += a[u];
        s2 += q[u];
      }
    }
  }
  s_a[grp][tx] = s;
  s_q[grp][tx] = s2;
  __syncthreads();
  if (grp == 0 && col < c) {
    s = ((s_a[0][tx] + s_a[1][tx]) + s_a[2][tx]) + s_a[3][tx];
    s2 = ((s_q[0][tx] + s_q[1][tx]) + s_q[2][tx]) + s_q[3][tx];
    const double m = s / (double)n;
    const double v = s2 / (double)n - m * m;
    const float mf = (float)m, vf = (float)(v > 0.0 ? v : 0.0);
    const float sc = 1.0f / sqrtf(vf + eps);
    s_scale[tx] = sc;
    s_shift[tx] = -mf * sc;
  }
  __syncthreads();
  // apply: the workgroup's kApplyRows x 64 slab, 16 B per lane where the rows allow it (thread = 4 columns x one row
  // of every 16), else 4 B per lane
  const int64_t r0 = (int64_t)blockIdx.x * kApplyRows;
  const int64_t r1 = min((long long)(r0 + kApplyRows), (long long)n);
  const bool vec = (c & 3) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0 && (!residual || (ldr & 3) == 0) &&
                   ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) |
                     reinterpret_cast<uintptr_t>(residual)) & 15) == 0;
  if (vec) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int c4 = (threadIdx.x & 15) * 4, rr = threadIdx.x >> 4;          // 16 lanes x 4 columns, 16 rows per pass
    const int colv = blockIdx.y * 64 + c4;
    if (colv >= c) return;
    const f32x4 scv = *reinterpret_cast<const f32x4*>(&s_scale[c4]), shv = *reinterpret_cast<const f32x4*>(&s_shift[c4]);
    for (int64_t r = r0 + rr; r < r1; r += 16) {
      f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + colv) * scv + shv;
      if (residual) v += *reinterpret_cast<const f32x4*>(residual + r * ldr + colv);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (relu == 1) v[e] = fmaxf(v[e], 0.f);
        else if (relu == 2) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
      }
      *reinterpret_cast<f32x4*>(y + r * ldy + colv) = v;
    }
    return;
  }
  if (col >= c) return;
  const float sc = s_scale[tx], sh = s_shift[tx];
  for (int64_t r = r0 + grp; r < r1; r += 4) {
    float v = x[r * ldx + col] * sc + sh;
    if (residual) v += residual[r * ldr + col];
    if (relu == 1) v = fmaxf(v, 0.f);
    else if (relu == 2) v = v > 0.f ? v : v * slope;
    y[r * ldy + col] = v;
  }
}

// one wave per row; c <= 64 * 8
__global__ void k_l2_normalize(const float* __restrict__ x, int64_t ldx, int64_t n, int c,
                               float* __restrict__ y, int64_t ldy) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= n) return;
  float ss = 0.f;
  for (int col = lane; col < c; col += 64) {
    float v = x[row * ldx + col];
    ss = fmaf(v, v, ss);
  }
  for (int d = 32; d >= 1; d >>= 1) ss += __shfl_xor(ss, d);
  float nrm = sqrtf(ss);
  for (int col = lane; col < c; col += 64) y[row * ldy + col] = x[row * ldx + col] / nrm;
}

// backward of y = x / |x|_2 per row: dx = (dy - y (y . dy)) / |x|; one wave per row
__global__ void k_l2_normalize_bwd(const float* __restrict__ x, int64_t ldx, const float* __restrict__ dy, int64_t lddy,
                                   int64_t n, int c, float* __restrict__ dx, int64_t lddx) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= n) return;
  float ss = 0.f, dot = 0.f;
  for (int col = lane; col < c; col += 64) {
    const float v = x[row * ldx + col];
    ss = fmaf(v, v, ss);
    dot = fmaf(v, dy[row * lddy + col], dot);
  }
  for (int d = 32; d >= 1; d >>= 1) {
    ss += __shfl_xor(ss, d);
    dot += __shfl_xor(dot, d);
  }
  const float nrm = sqrtf(ss);
  const float k = dot / (nrm * nrm);          // (y . dy) / |x| = (x . dy) / |x|^2
  for (int col = lane; col < c; col += 64) dx[row * lddx + col] = (dy[row * lddy + col] - x[row * ldx + col] * k) / nrm;
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of y = (x - mean) * rstd * gamma + beta over the rows of one segment (training-mode MinkowskiBatchNorm,
// FCGF_APR/model/common.py:6, lib/trainer.py:454-527; gamma = NULL: the affine-free InstanceNorm1d of KPFCNN's blocks,
// Predator_APR/models/blocks.py:459-468):
//   dbeta = sum dy,  dgamma = sum dy * xhat,  dx = gamma * rstd * (dy - dbeta / n - xhat * dgamma / n)
// Two launches of reductions (256-row partial sums in fp64, combined in fixed order: deterministic) + one apply pass.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_norm_bwd_partial(const float* __restrict__ x, int64_t ldx,
                                                          const float* __restrict__ dy, int64_t lddy, int64_t n, int c,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          double* __restrict__ partial) {
  __shared__ double s_a[4][64], s_b[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = blockIdx.y * 64 + tx;
  const int64_t r0 = (int64_t)blockIdx.x * kRowsPerBlock;
  const int64_t r1 = min((long long)(r0 + kRowsPerBlock), (long long)n);
  double a = 0.0, b = 0.0;
  if (col < c) {
    const float m = mean[col], rs = rstd[col];
    for (int64_t rb = r0 + ty; rb < r1; rb += 32) {
      float xv[8], gv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t r = rb + 4 * u;
        xv[u] = r < r1 ? x[r * ldx + col] : m;
        gv[u] = r < r1 ? dy[r * lddy + col] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a += (double)gv[u];
        b += (double)gv[u] * (double)((xv[u] - m) * rs);
      }
    }
  }
  s_a[ty][tx] = a;
  s_b[ty][tx] = b;
  __syncthreads();
  if (ty == 0 && col < c) {
    partial[((int64_t)blockIdx.x * 2 + 0) * c + col] = ((s_a[0][tx] + s_a[1][tx]) + s_a[2][tx]) + s_a[3][tx];
    partial[((int64_t)blockIdx.x * 2 + 1) * c + col] = ((s_b[0][tx] + s_b[1][tx]) + s_b[2][tx]) + s_b[3][tx];
  }
}

__global__ __launch_bounds__(256) void k_norm_bwd_finish(const double* __restrict__ partial, int nblk, int64_t n, int c,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                         float* __restrict__ k12) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= c) return;
  double a = 0.0, b = 0.0;
  for (int blk = 0; blk < nblk; ++blk) {      // fixed order
    a += partial[((int64_t)blk * 2 + 0) * c + col];
    b += partial[((int64_t)blk * 2 + 1) * c + col];
  }
  if (dbeta) dbeta[col] = (float)a;
  if (dgamma) dgamma[col] = (float)b;
  k12[col] = (float)(a / (double)n);
  k12[c + col] = (float)(b / (double)n);
}

__global__ void k_norm_bwd_apply(const float* __restrict__ x, int64_t ldx, const float* __restrict__ dy, int64_t lddy,
                                 int64_t n, int c, const float* __restrict__ mean, const float* __restrict__ rstd,
                                 const float* __restrict__ gamma, const float* __restrict__ k12, float* __restrict__ dx,
                                 int64_t lddx) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * c) return;
  const int64_t r = t / c;
  const int col = (int)(t - r * c);
  const float rs = rstd[col];
  const float xh = (x[r * ldx + col] - mean[col]) * rs;
  const float g = gamma ? gamma[col] : 1.f;
  dx[r * lddx + col] = g * rs * (dy[r * lddy + col] - k12[col] - xh * k12[c + col]);
}

// ---------------------------------------------------------------------------------------------------------------
// Training-mode BatchNorm fused with the activation and the residual add, forward and backward (round 5; the unit the
// APR training step repeats 21 times per encode: conv -> MinkowskiBatchNorm -> (+ x) -> ReLU, FCGF_APR/model/resunet.py:
// 142-193, model/residual_block.py:37-53 under lib/complement_trainer.py:350-512).  Both directions are the partial-sum
// launch + ONE apply launch whose workgroups rebuild the per-column constants from the block partials in a fixed order
// (as k_norm_apply does): 2 launches each, no torch op, deterministic.
// ---------------------------------------------------------------------------------------------------------------
__device__ inline void column_sums(const double* __restrict__ partial, int nblk, int c, int col, bool live, int tx, int grp,
                                   double (*s_a)[64], double (*s_q)[64], double& s, double& s2) {
  s = 0.0;
  s2 = 0.0;
  if (live) {
    for (int b0 = grp; b0 < nblk; b0 += 32) {      // the order of k_bn_finish / k_norm_apply
      double a[8], q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int b = b0 + 4 * u;
        const bool ok = b < nblk;
        a[u] = ok ? partial[((int64_t)b * 2 + 0) * c + col] : 0.0;
        q[u] = ok ? partial[((int64_t)b * 2 + 1) * c + col] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        s += a[u];
        s2 += q[u];
      }
    }
  }
  s_a[grp][tx] = s;
  s_q[grp][tx] = s2;
  __syncthreads();
  s = ((s_a[0][tx] + s_a[1][tx]) + s_a[2][tx]) + s_a[3][tx];
  s2 = ((s_q[0][tx] + s_q[1][tx]) + s_q[2][tx]) + s_q[3][tx];
}

// y = act((z - mean) * (rstd * gamma) + beta (+ residual)) per row segment (blockIdx.z; a segment = one forward call of the
// reference: the two frames of a pair stacked into one launch keep their own statistics); the workgroups of a segment's
// row block 0 publish its mean / rstd, and those of segment 0 update the running statistics segment after segment in
// order (momentum, unbiased variance: what torch.nn.BatchNorm1d does over consecutive calls)
__global__ __launch_bounds__(256) void k_bn_train_apply(const float* __restrict__ z, int64_t ldz, int c, Segs sg, int nseg,
                                                        const double* __restrict__ partial,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, float momentum, float* __restrict__ running_mean,
                                                        float* __restrict__ running_var, const float* __restrict__ residual,
                                                        int64_t ldr, int relu, float* __restrict__ y, int64_t ldy,
                                                        float* __restrict__ save_mean, float* __restrict__ save_rstd,
                                                        long long* __restrict__ num_batches_tracked) {
  __shared__ double s_a[4][64], s_q[4][64];
  __shared__ __attribute__((aligned(16))) float s_mean[64];
  __shared__ __attribute__((aligned(16))) float s_scale[64];
  __shared__ __attribute__((aligned(16))) float s_shift[64];
  const int seg = blockIdx.z;
  const int64_t n = sg.row0[seg + 1] - sg.row0[seg];
  if ((int64_t)blockIdx.x * kApplyRows >= n) return;                       // workgroup-uniform: shorter segment
  if (num_batches_tracked && blockIdx.x == 0 && blockIdx.y == 0 && seg == 0 && threadIdx.x == 0) *num_batches_tracked += nseg;
  const int tx = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int col = blockIdx.y * 64 + tx;
  double s, s2;
  if (running_mean && blockIdx.x == 0 && seg == 0) {                       // workgroup-uniform
    float rm = 0.f, rv = 0.f;
    if (grp == 0 && col < c) rm = running_mean[col], rv = running_var[col];
    for (int sgi = 0; sgi < nseg; ++sgi) {
      const int64_t ns = sg.row0[sgi + 1] - sg.row0[sgi];
      column_sums(partial + (int64_t)sg.blk0[sgi] * 2 * c, sg.blk0[sgi + 1] - sg.blk0[sgi], c, col, col < c, tx, grp, s_a, s_q, s,
                  s2);
      if (grp == 0 && col < c) {
        const double m = s / (double)ns;
        const double v = s2 / (double)ns - m * m;
        const float mf = (float)m, vf = (float)(v > 0.0 ? v : 0.0);
        const float unb = ns > 1 ? vf * ((float)ns / (float)(ns - 1)) : vf;
        rm = rm * (1.f - momentum) + mf * momentum;
        rv = rv * (1.f - momentum) + unb * momentum;
      }
      __syncthreads();                                                     // s_a / s_q are reused by the next segment
    }
    if (grp == 0 && col < c) running_mean[col] = rm, running_var[col] = rv;
  }
  const int nblk = sg.blk0[seg + 1] - sg.blk0[seg];
  column_sums(partial + (int64_t)sg.blk0[seg] * 2 * c, nblk, c, col, col < c, tx, grp, s_a, s_q, s, s2);
  if (grp == 0 && col < c) {
    const double m = s / (double)n;
    const double v = s2 / (double)n - m * m;
    const float mf = (float)m, vf = (float)(v > 0.0 ? v : 0.0);
    const float rs = 1.0f / sqrtf(vf + eps);
    s_mean[tx] = mf;
    s_scale[tx] = gamma ? rs * gamma[col] : rs;
    s_shift[tx] = beta ? beta[col] : 0.f;
    if (blockIdx.x == 0) {
      save_mean[(int64_t)seg * c + col] = mf;
      save_rstd[(int64_t)seg * c + col] = rs;
    }
  }
  __syncthreads();
  z += sg.row0[seg] * ldz;
  y += sg.row0[seg] * ldy;
  if (residual) residual += sg.row0[seg] * ldr;
  const int64_t r0 = (int64_t)blockIdx.x * kApplyRows;
  const int64_t r1 = min((long long)(r0 + kApplyRows), (long long)n);
  const bool vec = (c & 3) == 0 && (ldz & 3) == 0 && (ldy & 3) == 0 && (!residual || (ldr & 3) == 0) &&
                   ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(y) |
                     reinterpret_cast<uintptr_t>(residual)) & 15) == 0;
  if (vec) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int c4 = (threadIdx.x & 15) * 4, rr = threadIdx.x >> 4;
    const int colv = blockIdx.y * 64 + c4;
    if (colv >= c) return;
    const f32x4 mv = *reinterpret_cast<const f32x4*>(&s_mean[c4]), scv = *reinterpret_cast<const f32x4*>(&s_scale[c4]),
                shv = *reinterpret_cast<const f32x4*>(&s_shift[c4]);
    for (int64_t r = r0 + rr; r < r1; r += 16) {
      f32x4 v = (*reinterpret_cast<const f32x4*>(z + r * ldz + colv) - mv) * scv + shv;
      if (residual) v += *reinterpret_cast<const f32x4*>(residual + r * ldr + colv);
      if (relu)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      *reinterpret_cast<f32x4*>(y + r * ldy + colv) = v;
    }
    return;
  }
  if (col >= c) return;
  const float m = s_mean[tx], sc = s_scale[tx], sh = s_shift[tx];
  for (int64_t r = r0 + grp; r < r1; r += 4) {
    float v = (z[r * ldz + col] - m) * sc + sh;
    if (residual) v += residual[r * ldr + col];
    if (relu) v = fmaxf(v, 0.f);
    y[r * ldy + col] = v;
  }
}

// partial sums of g = act'(y) dy and of g * xhat per 256-row block (fp64); thread = 4 columns x one row of every 16
__global__ __launch_bounds__(256) void k_bn_train_bwd_partial(const float* __restrict__ z, int64_t ldz,
                                                              const float* __restrict__ y, int64_t ldy,
                                                              const float* __restrict__ dy, int64_t lddy, Segs sg, int c,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              int relu, double* __restrict__ partial) {
  __shared__ double s_v[16][16][8];
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int seg = blockIdx.z;
  if ((int)blockIdx.x >= sg.blk0[seg + 1] - sg.blk0[seg]) return;        // workgroup-uniform: shorter segment
  const int64_t n = sg.row0[seg + 1] - sg.row0[seg];
  z += sg.row0[seg] * ldz;
  dy += sg.row0[seg] * lddy;
  if (relu) y += sg.row0[seg] * ldy;
  mean += (int64_t)seg * c;
  rstd += (int64_t)seg * c;
  partial += (int64_t)sg.blk0[seg] * 2 * c;
  const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int colv = blockIdx.y * 64 + cq * 4;
  const int64_t r0 = (int64_t)blockIdx.x * kRowsPerBlock;
  const int64_t r1 = min((long long)(r0 + kRowsPerBlock), (long long)n);
  double a4[4] = {0.0, 0.0, 0.0, 0.0}, b4[4] = {0.0, 0.0, 0.0, 0.0};
  const bool vec = (c & 3) == 0 && (ldz & 3) == 0 && (lddy & 3) == 0 && (!relu || (ldy & 3) == 0) &&
                   ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(dy) |
                     (relu ? reinterpret_cast<uintptr_t>(y) : (uintptr_t)0)) & 15) == 0;
  if (colv < c) {
    float mv[4], rv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mv[e] = colv + e < c ? mean[colv + e] : 0.f;
      rv[e] = colv + e < c ? rstd[colv + e] : 0.f;
    }
    for (int64_t rb = r0 + rl; rb < r1; rb += 64) {
      f32x4 zv[4], gv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t r = rb + 16 * u;
        if (r < r1) {
          if (vec) {
            zv[u] = *reinterpret_cast<const f32x4*>(z + r * ldz + colv);
            gv[u] = *reinterpret_cast<const f32x4*>(dy + r * lddy + colv);
            if (relu) {
              const f32x4 yv = *reinterpret_cast<const f32x4*>(y + r * ldy + colv);
#pragma unroll
              for (int e = 0; e < 4; ++e) gv[u][e] = yv[e] > 0.f ? gv[u][e] : 0.f;
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const bool in = colv + e < c;
              zv[u][e] = in ? z[r * ldz + colv + e] : 0.f;
              float g = in ? dy[r * lddy + colv + e] : 0.f;
              if (relu && in && !(y[r * ldy + colv + e] > 0.f)) g = 0.f;
              gv[u][e] = g;
            }
          }
        } else {
          zv[u] = (f32x4){mv[0], mv[1], mv[2], mv[3]};
          gv[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a4[e] += (double)gv[u][e];
          b4[e] += (double)gv[u][e] * (double)((zv[u][e] - mv[e]) * rv[e]);
        }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    s_v[rl][cq][e] = a4[e];
    s_v[rl][cq][4 + e] = b4[e];
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    const int cc = threadIdx.x & 63, which = threadIdx.x >> 6;
    const int colw = blockIdx.y * 64 + cc;
    if (colw < c) {
      double t = 0.0;
      for (int g = 0; g < 16; ++g) t += s_v[g][cc >> 2][which * 4 + (cc & 3)];
      partial[((int64_t)blockIdx.x * 2 + which) * c + colw] = t;
    }
  }
}

// dx = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat)) with the segment's own means, g = act'(y) dy; dres = g;
// dgamma / dbeta (sums over ALL segments, in segment order) by the workgroups of segment 0's row block 0
__global__ __launch_bounds__(256) void k_bn_train_bwd_apply(const float* __restrict__ z, int64_t ldz,
                                                            const float* __restrict__ y, int64_t ldy,
                                                            const float* __restrict__ dy, int64_t lddy, Segs sg, int nseg, int c,
                                                            const double* __restrict__ partial,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, int relu,
                                                            float* __restrict__ dx, int64_t lddx, float* __restrict__ dres,
                                                            int64_t lddres, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta) {
  __shared__ double s_a[4][64], s_q[4][64];
  __shared__ __attribute__((aligned(16))) float s_mean[64];
  __shared__ __attribute__((aligned(16))) float s_rstd[64];
  __shared__ __attribute__((aligned(16))) float s_gs[64];
  __shared__ __attribute__((aligned(16))) float s_k1[64];
  __shared__ __attribute__((aligned(16))) float s_k2[64];
  const int seg = blockIdx.z;
  const int64_t n = sg.row0[seg + 1] - sg.row0[seg];
  if ((int64_t)blockIdx.x * kApplyRows >= n) return;                       // workgroup-uniform: shorter segment
  const int tx = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int col = blockIdx.y * 64 + tx;
  double s, s2;
  if ((dgamma || dbeta) && blockIdx.x == 0 && seg == 0) {                  // workgroup-uniform
    double ta = 0.0, tb = 0.0;
    for (int sgi = 0; sgi < nseg; ++sgi) {
      column_sums(partial + (int64_t)sg.blk0[sgi] * 2 * c, sg.blk0[sgi + 1] - sg.blk0[sgi], c, col, col < c, tx, grp, s_a, s_q, s,
                  s2);
      ta += s;
      tb += s2;
      __syncthreads();
    }
    if (grp == 0 && col < c) {
      if (dbeta) dbeta[col] = (float)ta;
      if (dgamma) dgamma[col] = (float)tb;
    }
  }
  column_sums(partial + (int64_t)sg.blk0[seg] * 2 * c, sg.blk0[seg + 1] - sg.blk0[seg], c, col, col < c, tx, grp, s_a, s_q, s, s2);
  if (grp == 0) {
    const bool live = col < c;
    const float rs = live ? rstd[(int64_t)seg * c + col] : 0.f;
    s_mean[tx] = live ? mean[(int64_t)seg * c + col] : 0.f;
    s_rstd[tx] = rs;
    s_gs[tx] = live ? (gamma ? gamma[col] * rs : rs) : 0.f;
    s_k1[tx] = (float)(s / (double)n);
    s_k2[tx] = (float)(s2 / (double)n);
  }
  __syncthreads();
  z += sg.row0[seg] * ldz;
  dy += sg.row0[seg] * lddy;
  dx += sg.row0[seg] * lddx;
  if (relu) y += sg.row0[seg] * ldy;
  if (dres) dres += sg.row0[seg] * lddres;
  const int64_t r0 = (int64_t)blockIdx.x * kApplyRows;
  const int64_t r1 = min((long long)(r0 + kApplyRows), (long long)n);
  const bool vec = (c & 3) == 0 && (ldz & 3) == 0 && (lddy & 3) == 0 && (lddx & 3) == 0 && (!relu || (ldy & 3) == 0) &&
                   (!dres || (lddres & 3) == 0) &&
                   ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx) |
                     reinterpret_cast<uintptr_t>(dres) | (relu ? reinterpret_cast<uintptr_t>(y) : (uintptr_t)0)) & 15) == 0;
  if (vec) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int c4 = (threadIdx.x & 15) * 4, rr = threadIdx.x >> 4;
    const int colv = blockIdx.y * 64 + c4;
    if (colv >= c) return;
    const f32x4 mv = *reinterpret_cast<const f32x4*>(&s_mean[c4]), rv = *reinterpret_cast<const f32x4*>(&s_rstd[c4]),
                gs = *reinterpret_cast<const f32x4*>(&s_gs[c4]), k1 = *reinterpret_cast<const f32x4*>(&s_k1[c4]),
                k2 = *reinterpret_cast<const f32x4*>(&s_k2[c4]);
    for (int64_t r = r0 + rr; r < r1; r += 16) {
      const f32x4 zv = *reinterpret_cast<const f32x4*>(z + r * ldz + colv);
      f32x4 g = *reinterpret_cast<const f32x4*>(dy + r * lddy + colv);
      if (relu) {
        const f32x4 yv = *reinterpret_cast<const f32x4*>(y + r * ldy + colv);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = yv[e] > 0.f ? g[e] : 0.f;
      }
      if (dres) *reinterpret_cast<f32x4*>(dres + r * lddres + colv) = g;
      const f32x4 xh = (zv - mv) * rv;
      *reinterpret_cast<f32x4*>(dx + r * lddx + colv) = gs * (g - k1 - xh * k2);
    }
    return;
  }
  if (col >= c) return;
  for (int64_t r = r0 + grp; r < r1; r += 4) {
    float g = dy[r * lddy + col];
    if (relu && !(y[r * ldy + col] > 0.f)) g = 0.f;
    if (dres) dres[r * lddres + col] = g;
    const float xh = (z[r * ldz + col] - s_mean[tx]) * s_rstd[tx];
    dx[r * lddx + col] = s_gs[tx] * (g - s_k1[tx] - xh * s_k2[tx]);
  }
}

}  // namespace

int apr_internal_norm_apply_partials(const float* x, int64_t ldx, int32_t c, const int64_t* seg_row0, const int* seg_blk0,
                                     int32_t nseg, const double* partial, float eps, const float* residual, int64_t ldr,
                                     int32_t act_mode, float negative_slope, float* y, int64_t ldy, hipStream_t st) {
  APR_CHECK_ARG(nseg >= 1 && nseg <= kMaxSeg, "norm_apply_partials: 1 .. %d segments", kMaxSeg);
  Segs sg;
  int64_t max_rows = 0;
  for (int i = 0; i <= nseg; ++i) {
    sg.row0[i] = seg_row0[i];
    sg.blk0[i] = seg_blk0[i];
    if (i && seg_row0[i] - seg_row0[i - 1] > max_rows) max_rows = seg_row0[i] - seg_row0[i - 1];
  }
  hipLaunchKernelGGL(k_norm_apply, dim3((unsigned)cdiv64(max_rows, kApplyRows), (c + 63) / 64, nseg), dim3(256), 0, st, x, ldx, c,
                     sg, partial, eps, residual, ldr, act_mode, negative_slope, y, ldy);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API size_t apr_norm_backward_scratch_bytes(int64_t n, int32_t c) {
  return (size_t)cdiv64(n > 0 ? n : 1, kRowsPerBlock) * 2 * (size_t)c * sizeof(double) + (size_t)2 * c * sizeof(float) + 512;
}

APR_API int apr_norm_backward(const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t n, int32_t c,
                              const float* mean, const float* rstd, const float* gamma, float* dx, int64_t lddx,
                              float* dgamma, float* dbeta, void* scratch, size_t scratch_bytes, void* stream) {
  APR_CHECK_ARG(x && dy && mean && rstd && dx && n > 0 && c > 0 && ldx >= c && lddy >= c && lddx >= c,
                "apr_norm_backward: bad arguments");
  APR_CHECK_ARG(scratch && scratch_bytes >= apr_norm_backward_scratch_bytes(n, c), "apr_norm_backward: scratch too small");
  hipStream_t st = (hipStream_t)stream;
  const int nblk = (int)cdiv64(n, kRowsPerBlock);
  double* partial = (double*)(((uintptr_t)scratch + 255) & ~(uintptr_t)255);
  float* k12 = (float*)(partial + (size_t)nblk * 2 * c);
  hipLaunchKernelGGL(k_norm_bwd_partial, dim3(nblk, (unsigned)cdiv64(c, 64)), dim3(256), 0, st, x, ldx, dy, lddy, n, c, mean,
                     rstd, partial);
  hipLaunchKernelGGL(k_norm_bwd_finish, dim3((unsigned)cdiv64(c, 256)), dim3(256), 0, st, partial, nblk, n, c, dgamma, dbeta,
                     k12);
  hipLaunchKernelGGL(k_norm_bwd_apply, dim3((unsigned)cdiv64(n * c, 256)), dim3(256), 0, st, x, ldx, dy, lddy, n, c, mean,
                     rstd, gamma, k12, dx, lddx);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API size_t apr_bn_stats_scratch_bytes(int64_t n, int32_t c) {
  return (size_t)cdiv64(n > 0 ? n : 1, kRowsPerBlock) * 2 * (size_t)c * sizeof(double);
}

APR_API int apr_bn_stats(const float* x, int64_t ld, int64_t n, int32_t c, float* mean, float* var,
                         void* scratch, size_t scratch_bytes, void* stream) {
  APR_CHECK_ARG(n > 0 && c > 0 && ld >= c, "apr_bn_stats: bad shape n=%lld c=%d", (long long)n, c);
  APR_CHECK_ARG(scratch_bytes >= apr_bn_stats_scratch_bytes(n, c), "apr_bn_stats: scratch too small");
  const int nblk = (int)cdiv64(n, kRowsPerBlock);
  hipLaunchKernelGGL(k_bn_partial, dim3(nblk, (c + 63) / 64), dim3(1024), 0, (hipStream_t)stream, x, ld, c,
                     one_segment(n), (double*)scratch);
  hipLaunchKernelGGL(k_bn_finish, dim3((c + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                     (const double*)scratch, nblk, n, c, mean, var, 0.f, (float*)nullptr, (float*)nullptr);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

// sums[c] = sum over the rows of x[:, c]: the bias gradient of a Linear / 1x1 convolution in the training path (fp64 partial
// sums per row block combined in fixed order, as apr_bn_stats: the same bits every run)
APR_API int apr_col_sums(const float* x, int64_t ld, int64_t n, int32_t c, float* sums, void* scratch, size_t scratch_bytes,
                         void* stream) {
  APR_CHECK_ARG(n > 0 && c > 0 && ld >= c && sums, "apr_col_sums: bad shape n=%lld c=%d", (long long)n, c);
  APR_CHECK_ARG(scratch_bytes >= apr_bn_stats_scratch_bytes(n, c), "apr_col_sums: scratch too small");
  const int nblk = (int)cdiv64(n, kRowsPerBlock);
  hipLaunchKernelGGL(k_bn_partial, dim3(nblk, (c + 63) / 64), dim3(1024), 0, (hipStream_t)stream, x, ld, c,
                     one_segment(n), (double*)scratch);
  hipLaunchKernelGGL(k_bn_finish, dim3((c + 63) / 64), dim3(256), 0, (hipStream_t)stream,      // "mean" over ONE row = the sum
                     (const double*)scratch, nblk, (int64_t)1, c, sums, (float*)nullptr, 0.f, (float*)nullptr, (float*)nullptr);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API int apr_norm_params(const float* x, int64_t ld, int64_t n, int32_t c, float eps, float* scale, float* shift,
                            void* scratch, size_t scratch_bytes, void* stream) {
  APR_CHECK_ARG(n > 0 && c > 0 && ld >= c && eps >= 0.f, "apr_norm_params: bad arguments");
  APR_CHECK_ARG(scratch_bytes >= apr_bn_stats_scratch_bytes(n, c), "apr_norm_params: scratch too small");
  const int nblk = (int)cdiv64(n, kRowsPerBlock);
  hipLaunchKernelGGL(k_bn_partial, dim3(nblk, (c + 63) / 64), dim3(1024), 0, (hipStream_t)stream, x, ld, c,
                     one_segment(n), (double*)scratch);
  hipLaunchKernelGGL(k_bn_finish, dim3((c + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                     (const double*)scratch, nblk, n, c, (float*)nullptr, (float*)nullptr, eps, scale, shift);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API int apr_instance_norm_act(const float* x, int64_t ldx, int64_t n, int32_t c, float eps, const float* residual,
                                  int64_t ldr, int32_t relu, float negative_slope, float* y, int64_t ldy, void* scratch,
                                  size_t scratch_bytes, void* stream) {
  APR_CHECK_ARG(n > 0 && c > 0 && ldx >= c && ldy >= c && eps >= 0.f && x && y, "apr_instance_norm_act: bad arguments");
  APR_CHECK_ARG(!residual || ldr >= c, "apr_instance_norm_act: ldr < c");
  APR_CHECK_ARG(scratch_bytes >= apr_bn_stats_scratch_bytes(n, c), "apr_instance_norm_act: scratch too small");
  const int nblk = (int)cdiv64(n, kRowsPerBlock);
  hipLaunchKernelGGL(k_bn_partial, dim3(nblk, (c + 63) / 64), dim3(1024), 0, (hipStream_t)stream, x, ldx, c,
                     one_segment(n), (double*)scratch);
  hipLaunchKernelGGL(k_norm_apply, dim3((unsigned)cdiv64(n, kApplyRows), (c + 63) / 64), dim3(256), 0, (hipStream_t)stream, x,
                     ldx, c, one_segment(n), (const double*)scratch, eps, residual, ldr, relu, negative_slope, y, ldy);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

// The same per row segment: seg_offsets_host[s] .. seg_offsets_host[s + 1] are the rows of segment s (a scan PAIR when
// several pairs are stacked into one KPFCNN forward: the reference normalises over the stacked points of ONE pair,
// batch size 1, Predator_APR/configs/test/kitti.yaml).  scratch: apr_bn_stats_scratch_bytes(n + 256 * nseg, c).
APR_API int apr_instance_norm_act_seg(const float* x, int64_t ldx, int64_t n, int32_t c, float eps, const float* residual,
                                      int64_t ldr, int32_t relu, float negative_slope, float* y, int64_t ldy,
                                      const int64_t* seg_offsets_host, int32_t nseg, void* scratch, size_t scratch_bytes,
                                      void* stream) {
  APR_CHECK_ARG(n > 0 && c > 0 && ldx >= c && ldy >= c && eps >= 0.f && x && y && seg_offsets_host && nseg >= 1,
                "apr_instance_norm_act_seg: bad arguments");
  APR_CHECK_ARG(!residual || ldr >= c, "apr_instance_norm_act_seg: ldr < c");
  APR_CHECK_ARG(seg_offsets_host[0] == 0 && seg_offsets_host[nseg] == n, "apr_instance_norm_act_seg: offsets must run 0 .. n");
  APR_CHECK_ARG(scratch_bytes >= apr_bn_stats_scratch_bytes(n + 256 * (int64_t)nseg, c),
                "apr_instance_norm_act_seg: scratch too small");
  APR_CHECK_ARG(nseg <= kMaxSeg, "apr_instance_norm_act_seg: at most %d segments", kMaxSeg);
  Segs sg;
  sg.row0[0] = 0;
  sg.blk0[0] = 0;
  int64_t max_rows = 0;
  for (int sgi = 0; sgi < nseg; ++sgi) {
    const int64_t rows = seg_offsets_host[sgi + 1] - seg_offsets_host[sgi];
    APR_CHECK_ARG(rows > 0, "apr_instance_norm_act_seg: empty segment %d", sgi);
    sg.row0[sgi + 1] = seg_offsets_host[sgi + 1];
    sg.blk0[sgi + 1] = sg.blk0[sgi] + (int)cdiv64(rows, kRowsPerBlock);
    if (rows > max_rows) max_rows = rows;
  }
  // ONE pair of launches for all segments (blockIdx.z): the per-pair statistics cost no extra launches
  hipLaunchKernelGGL(k_bn_partial, dim3((unsigned)cdiv64(max_rows, kRowsPerBlock), (c + 63) / 64, nseg), dim3(1024), 0,
                     (hipStream_t)stream, x, ldx, c, sg, (double*)scratch);
  hipLaunchKernelGGL(k_norm_apply, dim3((unsigned)cdiv64(max_rows, kApplyRows), (c + 63) / 64, nseg), dim3(256), 0,
                     (hipStream_t)stream, x, ldx, c, sg, (const double*)scratch, eps, residual, ldr, relu, negative_slope,
                     y, ldy);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API int apr_affine_act(const float* x, int64_t ldx, int64_t n, int32_t c, const float* scale,
                           const float* shift, const float* residual, int64_t ldr, int32_t relu,
                           float negative_slope, float* y, int64_t ldy, void* stream) {
  APR_CHECK_ARG(n >= 0 && c > 0 && ldx >= c && ldy >= c, "apr_affine_act: bad shape");
  if (n == 0) return APR_OK;
  int64_t nblk = cdiv64(n * c, 256);
  if (nblk > 8192) nblk = 8192;
  hipLaunchKernelGGL(k_affine_act, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, x, ldx, n, c,
                     scale, shift, residual, ldr, relu, negative_slope, y, ldy);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API int apr_l2_normalize(const float* x, int64_t ldx, int64_t n, int32_t c, float* y, int64_t ldy,
                             void* stream) {
  APR_CHECK_ARG(n >= 0 && c > 0 && ldx >= c && ldy >= c, "apr_l2_normalize: bad shape");
  if (n == 0) return APR_OK;
  hipLaunchKernelGGL(k_l2_normalize, dim3((unsigned)cdiv64(n, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                     n, c, y, ldy);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

// Backward of the activation behind a normalisation (training path of KPFCNN's blocks, Predator_APR/models/blocks.py:459-468,
// 489, 574 under lib/trainer.py:142-280): dz = dy where the forward's output y is positive, dy * negative_slope (mode 2,
// LeakyReLU) or 0 (mode 1, ReLU) elsewhere; mode 0 copies.
APR_API int apr_act_backward(const float* dy, int64_t lddy, const float* y, int64_t ldy, int64_t n, int32_t c, int32_t mode,
                             float negative_slope, float* dz, int64_t lddz, void* stream) {
  APR_CHECK_ARG(dy && y && dz && n >= 0 && c > 0 && lddy >= c && ldy >= c && lddz >= c && mode >= 0 && mode <= 2,
                "apr_act_backward: bad arguments");
  if (n == 0) return APR_OK;
  int64_t nblk = cdiv64(n * c, 256);
  if (nblk > 8192) nblk = 8192;
  hipLaunchKernelGGL(k_act_backward, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, dy, lddy, y, ldy, n, c, mode,
                     negative_slope, dz, lddz);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

// Training-mode BatchNorm + residual + ReLU in two launches (statistics, apply) -- see the kernels' header.
static int make_segs(const int64_t* seg_offsets_host, int32_t nseg, int64_t n, Segs& sg, int64_t& max_rows) {
  if (!seg_offsets_host || nseg <= 1) {
    sg = one_segment(n);
    max_rows = n;
    return 1;
  }
  if (nseg > kMaxSeg || seg_offsets_host[0] != 0 || seg_offsets_host[nseg] != n) return -1;
  sg.row0[0] = 0;
  sg.blk0[0] = 0;
  max_rows = 0;
  for (int i = 0; i < nseg; ++i) {
    const int64_t rows = seg_offsets_host[i + 1] - seg_offsets_host[i];
    if (rows < 2) return -1;
    sg.row0[i + 1] = seg_offsets_host[i + 1];
    sg.blk0[i + 1] = sg.blk0[i] + (int)cdiv64(rows, kRowsPerBlock);
    if (rows > max_rows) max_rows = rows;
  }
  return nseg;
}

APR_API int apr_bn_train_fwd(const float* z, int64_t ldz, int64_t n, int32_t c, const float* gamma, const float* beta,
                             float eps, float momentum, float* running_mean, float* running_var, const float* residual,
                             int64_t ldr, int32_t relu, float* y, int64_t ldy, float* save_mean, float* save_rstd,
                             int64_t* num_batches_tracked, const int64_t* seg_offsets_host, int32_t nseg, void* scratch,
                             size_t scratch_bytes, void* stream) {
  APR_CHECK_ARG(z && y && save_mean && save_rstd && n > 1 && c > 0 && ldz >= c && ldy >= c && eps >= 0.f,
                "apr_bn_train_fwd: bad arguments (n=%lld c=%d)", (long long)n, c);
  APR_CHECK_ARG(!residual || ldr >= c, "apr_bn_train_fwd: ldr < c");
  APR_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "apr_bn_train_fwd: running_mean / running_var go together");
  Segs sg;
  int64_t max_rows;
  const int ns = make_segs(seg_offsets_host, nseg, n, sg, max_rows);
  APR_CHECK_ARG(ns >= 1, "apr_bn_train_fwd: segment offsets must run 0 .. n in at most %d segments of >= 2 rows", kMaxSeg);
  APR_CHECK_ARG(scratch && scratch_bytes >= apr_bn_stats_scratch_bytes(n + 256 * (int64_t)ns, c), "apr_bn_train_fwd: scratch too small");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_bn_partial, dim3((unsigned)cdiv64(max_rows, kRowsPerBlock), (c + 63) / 64, ns), dim3(1024), 0, st, z, ldz,
                     c, sg, (double*)scratch);
  hipLaunchKernelGGL(k_bn_train_apply, dim3((unsigned)cdiv64(max_rows, kApplyRows), (c + 63) / 64, ns), dim3(256), 0, st, z, ldz,
                     c, sg, ns, (const double*)scratch, gamma, beta, eps, momentum, running_mean, running_var, residual, ldr,
                     relu, y, ldy, save_mean, save_rstd, (long long*)num_batches_tracked);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API int apr_bn_train_bwd(const float* z, int64_t ldz, const float* y, int64_t ldy, const float* dy, int64_t lddy,
                             int64_t n, int32_t c, const float* mean, const float* rstd, const float* gamma, int32_t relu,
                             float* dx, int64_t lddx, float* dres, int64_t lddres, float* dgamma, float* dbeta,
                             const int64_t* seg_offsets_host, int32_t nseg, void* scratch, size_t scratch_bytes, void* stream) {
  APR_CHECK_ARG(z && dy && mean && rstd && dx && n > 0 && c > 0 && ldz >= c && lddy >= c && lddx >= c,
                "apr_bn_train_bwd: bad arguments");
  APR_CHECK_ARG(!relu || (y && ldy >= c), "apr_bn_train_bwd: the ReLU mask needs the forward's output y");
  APR_CHECK_ARG(!dres || lddres >= c, "apr_bn_train_bwd: lddres < c");
  Segs sg;
  int64_t max_rows;
  const int ns = make_segs(seg_offsets_host, nseg, n, sg, max_rows);
  APR_CHECK_ARG(ns >= 1, "apr_bn_train_bwd: segment offsets must run 0 .. n in at most %d segments of >= 2 rows", kMaxSeg);
  APR_CHECK_ARG(scratch && scratch_bytes >= apr_bn_stats_scratch_bytes(n + 256 * (int64_t)ns, c), "apr_bn_train_bwd: scratch too small");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_bn_train_bwd_partial, dim3((unsigned)cdiv64(max_rows, kRowsPerBlock), (c + 63) / 64, ns), dim3(256), 0, st,
                     z, ldz, y, ldy, dy, lddy, sg, c, mean, rstd, relu, (double*)scratch);
  hipLaunchKernelGGL(k_bn_train_bwd_apply, dim3((unsigned)cdiv64(max_rows, kApplyRows), (c + 63) / 64, ns), dim3(256), 0, st, z,
                     ldz, y, ldy, dy, lddy, sg, ns, c, (const double*)scratch, mean, rstd, gamma, relu, dx, lddx, dres, lddres,
                     dgamma, dbeta);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API int apr_l2_normalize_backward(const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t n, int32_t c,
                                      float* dx, int64_t lddx, void* stream) {
  APR_CHECK_ARG(x && dy && dx && n >= 0 && c > 0 && ldx >= c && lddy >= c && lddx >= c, "apr_l2_normalize_backward: bad shape");
  if (n == 0) return APR_OK;
  hipLaunchKernelGGL(k_l2_normalize_bwd, dim3((unsigned)cdiv64(n, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, dy, lddy, n,
                     c, dx, lddx);
  APR_LAUNCH_CHECK();
  return APR_OK;
}
