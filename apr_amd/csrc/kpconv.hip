// KPConv encoder + overlap-attention kernels (K10, K11).
//
// KPConv (Predator_APR/models/blocks.py:229-374, rigid / linear influence / sum aggregation):
//   out[q] = ( sum_k ( sum_h w[q,k,h] * x[nbr[q,h]] ) @ W[k] ) / max(#{h : sum_c x[nbr[q,h],c] > 0}, 1)
//   w[q,k,h] = max(0, 1 - |s[nbr[q,h]] - q - kp[k]| / extent)
// The reference materialises [N,H,15,3] differences and two batched matmuls.  Here step 1 (the
// kernel-point correlation) is one wave per query on the fp32 MFMA: A = w[k][h] computed on the
// fly in registers (M = 16 >= 15 kernel points, K = neighbours), B = neighbour features gathered
// straight from HBM with one 16-B load per lane (channel permutation c = 4*r + cb makes every
// gathered row a single 256-B coalesced read AND every output row a 16-B store), already
// divided by the neighbour count.  Step 2 is the dense [Nq, 15*Cin] x [15*Cin, Cout] GEMM on the
// sparse-conv MFMA kernel (identity map).  The remaining kernels are the small gather / reduce
// ops of the blocks and of the overlap-attention module; none materialises an N x N tensor.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kKP = 15;
constexpr int kMaxH = 128;

__global__ void k_row_sums(const float* __restrict__ x, int64_t ld, int64_t n, int c, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= n) return;
  float s = 0.f;
  for (int j = lane; j < c; j += 64) s += x[row * ld + j];
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
  if (lane == 0) out[row] = s;
}

__device__ inline float kp_weight(float dx, float dy, float dz, float kx, float ky, float kz, float inv_extent) {
  const float ex = dx - kx, ey = dy - ky, ez = dz - kz;
  const float d2 = ex * ex + ey * ey + ez * ez;
  return fmaxf(1.f - sqrtf(d2) * inv_extent, 0.f);
}

// Step 1, MFMA form.  One wave per query; Cin % 64 == 0; NG = 64-channel groups per pass.
template <int NG>
__global__ __launch_bounds__(256) void k_kpconv_weighted_mfma(
    const float* __restrict__ q_pts, const float* __restrict__ s_pts, const int* __restrict__ nbr, int H,
    const float* __restrict__ x, int64_t ldx, int cin, const float* __restrict__ kp, float extent,
    const float* __restrict__ rowsum, float* __restrict__ wf, int64_t ldwf, int nq, int ns, int xcd_swz) {
  __shared__ int s_idx[4][kMaxH];
  __shared__ float s_diff[4][kMaxH][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, qd = lane >> 4;
  int blk = blockIdx.x;
  if (xcd_swz) {      // consecutive queries (they share neighbours) behind ONE XCD's L2
    const int nb = gridDim.x, xcd = blk & 7, loc = blk >> 3;
    blk = xcd * (nb >> 3) + min(xcd, nb & 7) + loc;
  }
  const int qi = blk * 4 + wave;
  if (qi >= nq) return;
  const float qx = q_pts[3 * (int64_t)qi], qy = q_pts[3 * (int64_t)qi + 1], qz = q_pts[3 * (int64_t)qi + 2];
  int cnt = 0;
  for (int h = lane; h < kMaxH; h += 64) {
    int idx = -1;
    float dx = 1e6f, dy = 1e6f, dz = 1e6f;
    if (h < H) {
      idx = nbr[(int64_t)qi * H + h];
      if (idx >= 0 && idx < ns) {
        dx = s_pts[3 * (int64_t)idx] - qx;
        dy = s_pts[3 * (int64_t)idx + 1] - qy;
        dz = s_pts[3 * (int64_t)idx + 2] - qz;
        cnt += rowsum[idx] > 0.f ? 1 : 0;
      } else {
        idx = -1;  // shadow neighbour: zero feature row, point at +1e6 => zero influence
      }
    }
    s_idx[wave][h] = idx;
    s_diff[wave][h][0] = dx; s_diff[wave][h][1] = dy; s_diff[wave][h][2] = dz;
  }
  for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
  const float inv_num = 1.f / (float)max(cnt, 1);
  const float inv_extent = 1.f / extent;
  const bool kvalid = r16 < kKP;
  const float kx = kvalid ? kp[3 * r16] : 0.f, ky = kvalid ? kp[3 * r16 + 1] : 0.f, kz = kvalid ? kp[3 * r16 + 2] : 0.f;
  const int nsteps = (H + 3) >> 2;
  for (int g0 = 0; g0 < cin / 64; g0 += NG) {
    f32x4 acc[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[g][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int st = 0; st < nsteps; ++st) {
      const int h = st * 4 + qd;
      const int idx = s_idx[wave][h];
      float w = 0.f;
      if (kvalid && idx >= 0)
        w = kp_weight(s_diff[wave][h][0], s_diff[wave][h][1], s_diff[wave][h][2], kx, ky, kz, inv_extent);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        f32x4 b = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (idx >= 0 && g0 + g < cin / 64)
          b = *reinterpret_cast<const f32x4*>(x + (int64_t)idx * ldx + (g0 + g) * 64 + r16 * 4);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[g][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, b[cb], acc[g][cb], 0, 0, 0);
      }
    }
    // D[k = 4*qd + i][col r16 of block cb] <-> channel (g0+g)*64 + 4*r16 + cb
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g0 + g >= cin / 64) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = qd * 4 + i;
        if (k < kKP) {
          f32x4 v = {acc[g][0][i] * inv_num, acc[g][1][i] * inv_num, acc[g][2][i] * inv_num, acc[g][3][i] * inv_num};
          *reinterpret_cast<f32x4*>(wf + (int64_t)qi * ldwf + (int64_t)k * cin + (g0 + g) * 64 + r16 * 4) = v;
        }
      }
    }
  }
}

// Step 1, VALU form for any Cin (the first layer has Cin = 1).  One thread per (query, kernel point).
__global__ void k_kpconv_weighted_generic(const float* __restrict__ q_pts, const float* __restrict__ s_pts,
                                          const int* __restrict__ nbr, int H, const float* __restrict__ x, int64_t ldx,
                                          int cin, const float* __restrict__ kp, float extent,
                                          const float* __restrict__ rowsum, float* __restrict__ wf, int64_t ldwf,
                                          int nq, int ns) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)nq * kKP) return;
  const int qi = (int)(t / kKP), k = (int)(t - (int64_t)qi * kKP);
  const float qx = q_pts[3 * (int64_t)qi], qy = q_pts[3 * (int64_t)qi + 1], qz = q_pts[3 * (int64_t)qi + 2];
  const float kx = kp[3 * k], ky = kp[3 * k + 1], kz = kp[3 * k + 2];
  const float inv_extent = 1.f / extent;
  int cnt = 0;
  for (int h = 0; h < H; ++h) {
    const int idx = nbr[(int64_t)qi * H + h];
    if (idx >= 0 && idx < ns && rowsum[idx] > 0.f) ++cnt;
  }
  const float inv_num = 1.f / (float)max(cnt, 1);
  for (int c0 = 0; c0 < cin; c0 += 8) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int nc = min(8, cin - c0);
    for (int h = 0; h < H; ++h) {
      const int idx = nbr[(int64_t)qi * H + h];
      if (idx < 0 || idx >= ns) continue;
      const float w = kp_weight(s_pts[3 * (int64_t)idx] - qx, s_pts[3 * (int64_t)idx + 1] - qy,
                                s_pts[3 * (int64_t)idx + 2] - qz, kx, ky, kz, inv_extent);
      if (w == 0.f) continue;
      for (int c = 0; c < nc; ++c) acc[c] = fmaf(w, x[(int64_t)idx * ldx + c0 + c], acc[c]);
    }
    for (int c = 0; c < nc; ++c) wf[(int64_t)qi * ldwf + (int64_t)k * cin + c0 + c] = acc[c] * inv_num;
  }
}

// Backward of step 1 with respect to the neighbour features (Predator side):
//   dx[nbr[q,h], c] += (1 / num_q) * sum_k w[q,k,h] * dwf[q, k*cin + c]
// (the neighbour count num_q and the influences w depend on the geometry and on the SIGN of the feature sums only:
// piecewise constant in x, no gradient through them - the same graph torch builds for blocks.py:326-374).
// One wave per query: the 15 x H influences go to LDS once, a lane keeps the 15 values dwf[q, :, c] of its channel
// in registers and walks the neighbours; the sums leave through float atomics (a support point is the neighbour of
// ~H queries: the transposed table is not available, and torch's own index backward adds atomically as well).
template <int CPL>   // channels per lane: cin <= 64 * CPL
__global__ __launch_bounds__(256) void k_kpconv_dfeat(
    const float* __restrict__ q_pts, const float* __restrict__ s_pts, const int* __restrict__ nbr, int H,
    const float* __restrict__ dwf, int64_t lddwf, int cin, const float* __restrict__ kp, float extent,
    const float* __restrict__ rowsum, float* __restrict__ dx, int64_t lddx, int nq, int ns,
    float* __restrict__ contrib) {      // non-NULL: the deterministic form -- row (q, h) of contrib instead of an atomic add
  __shared__ int s_idx[4][kMaxH];
  __shared__ float s_w[4][kMaxH][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qi = blockIdx.x * 4 + wave;
  if (qi >= nq) return;
  const float qx = q_pts[3 * (int64_t)qi], qy = q_pts[3 * (int64_t)qi + 1], qz = q_pts[3 * (int64_t)qi + 2];
  const float inv_extent = 1.f / extent;
  int cnt = 0;
  for (int h = lane; h < H; h += 64) {
    int idx = nbr[(int64_t)qi * H + h];
    if (idx < 0 || idx >= ns) idx = -1;
    s_idx[wave][h] = idx;
    if (idx >= 0) {
      cnt += rowsum[idx] > 0.f ? 1 : 0;
      const float dx_ = s_pts[3 * (int64_t)idx] - qx, dy_ = s_pts[3 * (int64_t)idx + 1] - qy,
                  dz_ = s_pts[3 * (int64_t)idx + 2] - qz;
#pragma unroll
      for (int k = 0; k < kKP; ++k)
        s_w[wave][h][k] = kp_weight(dx_, dy_, dz_, kp[3 * k], kp[3 * k + 1], kp[3 * k + 2], inv_extent);
    }
  }
  for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
  const float inv_num = 1.f / (float)max(cnt, 1);
  // (the wave's LDS writes are read back by other lanes of the same wave: in-order LDS queue, no barrier needed)
  float g[CPL][kKP];
#pragma unroll
  for (int u = 0; u < CPL; ++u) {
    const int c = lane + 64 * u;
#pragma unroll
    for (int k = 0; k < kKP; ++k)
      g[u][k] = c < cin ? dwf[(int64_t)qi * lddwf + (int64_t)k * cin + c] * inv_num : 0.f;
  }
  for (int h = 0; h < H; ++h) {
    const int idx = s_idx[wave][h];      // wave-uniform
    if (idx < 0) continue;
    float w[kKP];
#pragma unroll
    for (int k = 0; k < kKP; ++k) w[k] = s_w[wave][h][k];
#pragma unroll
    for (int u = 0; u < CPL; ++u) {
      const int c = lane + 64 * u;
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < kKP; ++k) acc = fmaf(w[k], g[u][k], acc);
      if (c < cin) {
        if (contrib) contrib[((int64_t)qi * H + h) * cin + c] = acc;
        else atomicAdd(dx + (int64_t)idx * lddx + c, acc);
      }
    }
  }
}

// out[q,:] = max_h x_pad[inds[q,h],:]  (x_pad = x plus a zero shadow row; blocks.py:86-102)
// mode 1: out[q,:] = x_pad[inds[q,0],:]  (closest_pool, blocks.py:71-83)
// 16-B form: thread = (query, 4 channels); rows of x / out 16-byte aligned, c % 4 == 0
__global__ void k_gather_pool4(const float* __restrict__ x, int64_t ldx, int ns, int c, const int* __restrict__ inds,
                               int H, int64_t nq, int mode, float* __restrict__ out, int64_t ldo) {
  const int c4 = c >> 2;
  const int64_t total = nq * c4;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const f32x4 ninf = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t qi = t / c4;
    const int col = (int)(t - qi * c4) * 4;
    f32x4 m;
    if (mode == 1) {
      const int idx = inds[qi * H];
      m = (idx >= 0 && idx < ns) ? *reinterpret_cast<const f32x4*>(x + (int64_t)idx * ldx + col) : zero;
    } else {
      m = ninf;
      for (int h0 = 0; h0 < H; h0 += 8) {
        int idx[8];
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) idx[u] = (h0 + u < H) ? inds[qi * H + h0 + u] : -2;
#pragma unroll
        for (int u = 0; u < 8; ++u)
          v[u] = (idx[u] >= 0 && idx[u] < ns) ? *reinterpret_cast<const f32x4*>(x + (int64_t)idx[u] * ldx + col)
                                              : (idx[u] == -2 ? ninf : zero);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[u][e]);
      }
    }
    *reinterpret_cast<f32x4*>(out + qi * ldo + col) = m;
  }
}

__global__ void k_gather_pool(const float* __restrict__ x, int64_t ldx, int ns, int c, const int* __restrict__ inds,
                              int H, int64_t nq, int mode, float* __restrict__ out, int64_t ldo) {
  const int64_t total = nq * c;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t qi = t / c;
    const int col = (int)(t - qi * c);
    float m;
    if (mode == 1) {
      const int idx = inds[qi * H];
      m = (idx >= 0 && idx < ns) ? x[(int64_t)idx * ldx + col] : 0.f;
    } else {
      m = -__builtin_inff();
      for (int h0 = 0; h0 < H; h0 += 8) {     // 8 index loads, then 8 gathers in flight (not 2 dependent trips per h)
        int idx[8];
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) idx[u] = (h0 + u < H) ? inds[qi * H + h0 + u] : -2;
#pragma unroll
        for (int u = 0; u < 8; ++u)
          v[u] = (idx[u] >= 0 && idx[u] < ns) ? x[(int64_t)idx[u] * ldx + col] : (idx[u] == -2 ? -__builtin_inff() : 0.f);
#pragma unroll
        for (int u = 0; u < 8; ++u) m = fmaxf(m, v[u]);
      }
    }
    out[qi * ldo + col] = m;
  }
}

// max_pool with the position of the maximum: amax[q, c] = the FIRST h whose x_pad[inds[q,h], c] equals the maximum (what
// the backward routes the gradient to, as torch.max(dim) does); thread = (query, channel)
__global__ void k_gather_pool_argmax(const float* __restrict__ x, int64_t ldx, int ns, int c, const int* __restrict__ inds,
                                     int H, int64_t nq, float* __restrict__ out, int64_t ldo,
                                     unsigned char* __restrict__ amax) {
  const int64_t total = nq * c;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t qi = t / c;
    const int col = (int)(t - qi * c);
    float m = -__builtin_inff();
    int best = 0;
    for (int h0 = 0; h0 < H; h0 += 8) {
      int idx[8];
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) idx[u] = (h0 + u < H) ? inds[qi * H + h0 + u] : -2;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        v[u] = (idx[u] >= 0 && idx[u] < ns) ? x[(int64_t)idx[u] * ldx + col] : (idx[u] == -2 ? -__builtin_inff() : 0.f);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (v[u] > m) {
          m = v[u];
          best = h0 + u;
        }
    }
    out[qi * ldo + col] = m;
    amax[qi * c + col] = (unsigned char)best;
  }
}

// Backward of max_pool / closest_pool as a GATHER over the reverse neighbour table (revtable.hip): dx[s, c] = sum of
// dout[q, c] over the entries (q, h) of row s's run with h == amax[q, c] (mode 0) or h == 0 (mode 1), in run order:
// deterministic, no float atomics, no contribution buffer.  thread = (support row, channel)
__global__ void k_gather_pool_bwd(const float* __restrict__ dout, int64_t lddo, int c, const int* __restrict__ rev_t,
                                  const int* __restrict__ start, int64_t ns, int H, const unsigned char* __restrict__ amax,
                                  int mode, float* __restrict__ dx, int64_t lddx) {
  const int64_t total = ns * c;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = t / c;
    const int col = (int)(t - s * c);
    float acc = 0.f;
    for (int e = start[s]; e < start[s + 1]; ++e) {
      const int pos = rev_t[e];
      const int q = pos / H, h = pos - q * H;
      const bool mine = mode == 1 ? h == 0 : (int)amax[(int64_t)q * c + col] == h;
      if (mine) acc += dout[(int64_t)q * lddo + col];
    }
    dx[s * lddx + col] = acc;
  }
}

// edge features of the DGCNN-style self attention: row (i,j) = [f_i, f_nbr(i,j) - f_i]   (gcn.py:9-35)
__global__ void k_edge_features(const float* __restrict__ f, int64_t ldf, int n, int c, const int* __restrict__ knn,
                                int k, float* __restrict__ out) {
  const int64_t total = (int64_t)n * k * c;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int col = (int)(t % c);
    const int64_t e = t / c;
    const int i = (int)(e / k);
    const int j = knn[e];
    const float fi = f[(int64_t)i * ldf + col];
    out[e * 2 * c + col] = fi;
    out[e * 2 * c + c + col] = f[(int64_t)j * ldf + col] - fi;
  }
}

// backward of k_edge_features w.r.t. f, first half: the centre's own terms d_center[i] = sum_j (de[(i,j), :c] - de[(i,j), c:])
// (ascending j: the same bits every run) and the neighbours' terms as contiguous contribution rows contrib[(i,j), :] =
// de[(i,j), c:], which a gather over the reverse table of knn then adds to the rows they point at (apr_reverse_gather_range).
__global__ void k_edge_features_bwd(const float* __restrict__ de, int n, int c, int k, float* __restrict__ d_center,
                                    int64_t ldd, float* __restrict__ contrib) {
  const int64_t total = (int64_t)n * c;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(t / c), col = (int)(t - (int64_t)i * c);
    float acc = 0.f;
    for (int j = 0; j < k; ++j) {
      const int64_t e = (int64_t)i * k + j;
      const float a = de[e * 2 * c + col], b = de[e * 2 * c + c + col];
      acc += a - b;
      contrib[e * c + col] = b;
    }
    d_center[(int64_t)i * ldd + col] = acc;
  }
}

// out[i,:] = max_j act(y[i*k + j,:] * scale + shift)   (InstanceNorm2d + LeakyReLU + max over k)
__global__ void k_group_max(const float* __restrict__ y, int64_t ldy, int n, int k, int c,
                            const float* __restrict__ scale, const float* __restrict__ shift, float slope,
                            float* __restrict__ out, int64_t ldo) {
  const int64_t total = (int64_t)n * c;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(t / c), col = (int)(t - (int64_t)i * c);
    const float sc = scale ? scale[col] : 1.f, sh = shift ? shift[col] : 0.f;
    float m = -__builtin_inff();
    for (int j = 0; j < k; ++j) {
      float v = y[((int64_t)i * k + j) * ldy + col] * sc + sh;
      v = v > 0.f ? v : v * slope;
      m = fmaxf(m, v);
    }
    out[(int64_t)i * ldo + col] = m;
  }
}

// Multi-head attention with channel layout c = d * heads + h (gcn.py:94-116):
//   out[n, d*heads+h] = sum_m softmax_m( sum_d q[n,d,h] k[m,d,h] / sqrt(dim) ) v[m,d,h]
// One workgroup per (kMhaQ = 8 queries, head): every key / value row fetched from global memory serves 8 queries
// (one workgroup per query re-read all of K and V for each of them: 2 MB x 4000 workgroups per call).  Scores of
// the 8 queries in LDS ([8][M], M <= ~1800), two-pass softmax per query in fixed order, fp32.
constexpr int kMhaQ = 8;

// Q values per thread reduced over the 256 threads of a workgroup at once: wave shuffles, then the 4 wave results in
// fixed order through LDS -- 2 barriers for all Q (a 256-wide LDS tree per query and quantity was ~17 barriers each:
// over a hundred per workgroup, most of the attention kernels' time).  s_part: [4][Q] floats.
template <int Q, bool MAX>
__device__ inline void block_reduce(float (&v)[Q], float* s_part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int qi = 0; qi < Q; ++qi) {
    float x = v[qi];
    for (int d = 32; d >= 1; d >>= 1) {
      const float o = __shfl_xor(x, d);
      x = MAX ? fmaxf(x, o) : x + o;
    }
    if (lane == 0) s_part[wave * Q + qi] = x;
  }
  __syncthreads();
#pragma unroll
  for (int qi = 0; qi < Q; ++qi) {
    const float a = s_part[qi], b = s_part[Q + qi], c = s_part[2 * Q + qi], d = s_part[3 * Q + qi];
    v[qi] = MAX ? fmaxf(fmaxf(a, b), fmaxf(c, d)) : (a + b) + (c + d);
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void k_mha(const float* __restrict__ q, const float* __restrict__ kk,
                                             const float* __restrict__ v, int n, int m, int dim, int heads,
                                             float* __restrict__ out) {
  extern __shared__ float s_sc[];            // [kMhaQ][m] scores, then [kMhaQ][dim] queries
  float* s_q = s_sc + kMhaQ * (m > 256 ? m : 256);
  __shared__ float s_red[256];
  __shared__ float s_inv[kMhaQ];
  const int nqb = (n + kMhaQ - 1) / kMhaQ;
  const int qb = blockIdx.x % nqb, h = blockIdx.x / nqb;
  const int n0 = qb * kMhaQ;
  const int c = dim * heads;
  for (int e = threadIdx.x; e < kMhaQ * dim; e += 256) {
    const int qi = e / dim, d = e - qi * dim;
    const int row = min(n0 + qi, n - 1);
    s_q[e] = q[(int64_t)row * c + d * heads + h];
  }
  __syncthreads();
  const float scl = 1.f / sqrtf((float)dim);
  float mx[kMhaQ];
#pragma unroll
  for (int qi = 0; qi < kMhaQ; ++qi) mx[qi] = -__builtin_inff();
  for (int j = threadIdx.x; j < m; j += 256) {
    float s[kMhaQ];
#pragma unroll
    for (int qi = 0; qi < kMhaQ; ++qi) s[qi] = 0.f;
    const float* krow = kk + (int64_t)j * c + h;
    for (int d = 0; d < dim; ++d) {
      const float kv = krow[d * heads];
#pragma unroll
      for (int qi = 0; qi < kMhaQ; ++qi) s[qi] = fmaf(s_q[qi * dim + d], kv, s[qi]);
    }
#pragma unroll
    for (int qi = 0; qi < kMhaQ; ++qi) {
      s[qi] *= scl;
      s_sc[qi * m + j] = s[qi];
      mx[qi] = fmaxf(mx[qi], s[qi]);
    }
  }
  // per-query max, then exp + sum, all kMhaQ queries at once
  block_reduce<kMhaQ, true>(mx, s_red);
  float sum[kMhaQ];
#pragma unroll
  for (int qi = 0; qi < kMhaQ; ++qi) sum[qi] = 0.f;
  for (int j = threadIdx.x; j < m; j += 256) {
#pragma unroll
    for (int qi = 0; qi < kMhaQ; ++qi) {
      const float e = expf(s_sc[qi * m + j] - mx[qi]);
      s_sc[qi * m + j] = e;
      sum[qi] += e;
    }
  }
  block_reduce<kMhaQ, false>(sum, s_red);
#pragma unroll
  for (int qi = 0; qi < kMhaQ; ++qi)
    if (threadIdx.x == qi) s_inv[qi] = 1.f / sum[qi];
  __syncthreads();
  // out[q][d] = sum_j p[q][j] v[j][d]: thread (d, g) owns channel d and the keys j = g (mod ngrp); 4 value loads
  // in flight per thread (a plain loop over all keys is one dependent L2 round trip per key: 250 us per call), the
  // ngrp partial sums meet in LDS in fixed order
  const int d = threadIdx.x % dim, grp = threadIdx.x / dim, ngrp = 256 / dim;   // dim divides 256 (checked on the host)
  float acc[kMhaQ];
#pragma unroll
  for (int qi = 0; qi < kMhaQ; ++qi) acc[qi] = 0.f;
  const float* vcol = v + d * heads + h;
  for (int j0 = grp; j0 < m; j0 += 4 * ngrp) {
    float vv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u * ngrp;
      vv[u] = j < m ? vcol[(int64_t)j * c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u * ngrp;
      if (j < m) {
#pragma unroll
        for (int qi = 0; qi < kMhaQ; ++qi) acc[qi] = fmaf(s_sc[qi * m + j], vv[u], acc[qi]);
      }
    }
  }
  __syncthreads();                       // all probabilities consumed: reuse the head of s_sc for the partial sums
  float* s_part = s_sc;                  // [ngrp][kMhaQ][dim] = 256 * kMhaQ floats <= the score area (>= kMhaQ * 256)
#pragma unroll
  for (int qi = 0; qi < kMhaQ; ++qi) s_part[(grp * kMhaQ + qi) * dim + d] = acc[qi];
  __syncthreads();
  for (int e = threadIdx.x; e < kMhaQ * dim; e += 256) {
    const int qi = e / dim, dd = e - qi * dim;
    float sum = 0.f;
    for (int g = 0; g < ngrp; ++g) sum += s_part[(g * kMhaQ + qi) * dim + dd];
    if (n0 + qi < n) out[(int64_t)(n0 + qi) * c + dd * heads + h] = sum * s_inv[qi];
  }
}

// s[n] = sum_m softmax_m( <a[n,:], b[m,:]> / temperature ) * w[m]    (architectures.py:176-181)
__global__ __launch_bounds__(256) void k_softmax_matvec(const float* __restrict__ a, const float* __restrict__ b,
                                                        const float* __restrict__ w, int n, int m, int c,
                                                        float temperature, float* __restrict__ out) {
  extern __shared__ float s_sc[];  // [m] + [c]
  float* s_a = s_sc + m;
  __shared__ float s_red[256];
  const int ni = blockIdx.x;
  for (int d = threadIdx.x; d < c; d += 256) s_a[d] = a[(int64_t)ni * c + d];
  __syncthreads();
  float mx = -__builtin_inff();
  for (int j = threadIdx.x; j < m; j += 256) {
    float s = 0.f;
    for (int d = 0; d < c; ++d) s = fmaf(s_a[d], b[(int64_t)j * c + d], s);
    s /= temperature;
    s_sc[j] = s;
    mx = fmaxf(mx, s);
  }
  s_red[threadIdx.x] = mx;
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if (threadIdx.x < st) s_red[threadIdx.x] = fmaxf(s_red[threadIdx.x], s_red[threadIdx.x + st]);
    __syncthreads();
  }
  mx = s_red[0];
  __syncthreads();
  float num = 0.f, den = 0.f;
  for (int j = threadIdx.x; j < m; j += 256) {
    float e = expf(s_sc[j] - mx);
    den += e;
    num = fmaf(e, w[j], num);
  }
  s_red[threadIdx.x] = den;
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if (threadIdx.x < st) s_red[threadIdx.x] += s_red[threadIdx.x + st];
    __syncthreads();
  }
  den = s_red[0];
  __syncthreads();
  s_red[threadIdx.x] = num;
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if (threadIdx.x < st) s_red[threadIdx.x] += s_red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[ni] = s_red[0] / den;
}

// The same with b given TRANSPOSED (bt [c, m]) and kSmvQ queries per workgroup: lane j reads bt[d * m + j] -- coalesced,
// and every element fetched serves kSmvQ queries -- where k_softmax_matvec's lanes each walk their own 1 KB row of b
// (700 workgroups x 717 KB through the L1: 110 us per call at 700 x 700 x 256); the softmax reductions of the
// kSmvQ queries run together (block_reduce).
constexpr int kSmvQ = 4;
__global__ __launch_bounds__(256) void k_softmax_matvec_t(const float* __restrict__ a, const float* __restrict__ bt,
                                                          const float* __restrict__ w, int n, int m, int c,
                                                          float temperature, float* __restrict__ out) {
  extern __shared__ float s_sc[];  // [kSmvQ][m] + [kSmvQ][c]
  float* s_a = s_sc + kSmvQ * m;
  __shared__ float s_red[256];
  const int n0 = blockIdx.x * kSmvQ;
  for (int e = threadIdx.x; e < kSmvQ * c; e += 256) {
    const int qi = e / c, d = e - qi * c;
    s_a[e] = a[(int64_t)min(n0 + qi, n - 1) * c + d];
  }
  __syncthreads();
  float mx[kSmvQ];
#pragma unroll
  for (int qi = 0; qi < kSmvQ; ++qi) mx[qi] = -__builtin_inff();
  for (int j = threadIdx.x; j < m; j += 256) {
    float sc[kSmvQ];
#pragma unroll
    for (int qi = 0; qi < kSmvQ; ++qi) sc[qi] = 0.f;
    for (int d = 0; d < c; ++d) {
      const float bv = bt[(int64_t)d * m + j];
#pragma unroll
      for (int qi = 0; qi < kSmvQ; ++qi) sc[qi] = fmaf(s_a[qi * c + d], bv, sc[qi]);
    }
#pragma unroll
    for (int qi = 0; qi < kSmvQ; ++qi) {
      sc[qi] /= temperature;
      s_sc[qi * m + j] = sc[qi];
      mx[qi] = fmaxf(mx[qi], sc[qi]);
    }
  }
  block_reduce<kSmvQ, true>(mx, s_red);
  float num[kSmvQ], den[kSmvQ];
#pragma unroll
  for (int qi = 0; qi < kSmvQ; ++qi) num[qi] = den[qi] = 0.f;
  for (int j = threadIdx.x; j < m; j += 256) {
    const float wj = w[j];
#pragma unroll
    for (int qi = 0; qi < kSmvQ; ++qi) {
      const float e = expf(s_sc[qi * m + j] - mx[qi]);
      den[qi] += e;
      num[qi] = fmaf(e, wj, num[qi]);
    }
  }
  block_reduce<kSmvQ, false>(den, s_red);
  block_reduce<kSmvQ, false>(num, s_red);
#pragma unroll
  for (int qi = 0; qi < kSmvQ; ++qi)
    if (threadIdx.x == qi && n0 + qi < n) out[n0 + qi] = num[qi] / den[qi];
}

// The same on the fp32 MFMA: a workgroup owns 16 queries, its 4 waves walk the 16-key tiles in turn.  Per tile the
// scores are ONE accumulator: rows = keys, columns = queries (A operand: key r16, B operand: query r16, both lanes of a
// k-slot q reading channel 16 s + 4 q + e -- a 16-B load serves 4 MFMAs), so lane (r16, q) ends with the scores of
// keys 4q .. 4q+3 against ITS query r16 and keeps a running (max, denominator, numerator) for it; the 16 partial
// triples per query (8 waves x 4 k-slots) are merged in fixed order; the next tile's key rows are in flight under the
// MFMAs of the current one.
constexpr int kSmvWaves = 8;
template <int C16>   // c = 16 * C16
__global__ __launch_bounds__(64 * kSmvWaves) void k_softmax_matvec_mfma(const float* __restrict__ a,
                                                                        const float* __restrict__ b,
                                                                        const float* __restrict__ w, int n, int m,
                                                                        float inv_temperature, float* __restrict__ out) {
  constexpr int NP = kSmvWaves * 4;                                    // partial triples per query
  __shared__ float s_mx[NP][16], s_den[NP][16], s_num[NP][16];        // [part = wave * 4 + q][query]
  constexpr int c = 16 * C16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  const int i0 = blockIdx.x * 16;
  const float* arow = a + (int64_t)min(i0 + r16, n - 1) * c + 4 * q;
  f32x4 qa[C16];
#pragma unroll
  for (int sidx = 0; sidx < C16; ++sidx) qa[sidx] = *reinterpret_cast<const f32x4*>(arow + 16 * sidx);
  float mx = -__builtin_inff(), den = 0.f, num = 0.f;
  const int ntile = (m + 15) >> 4;
  // the key rows of the NEXT tile are in flight while this tile's MFMAs run; two accumulator chains per tile
  f32x4 kb[C16], kn[C16];
  if (wave < ntile) {
    const float* krow = b + (int64_t)min(wave * 16 + r16, m - 1) * c + 4 * q;
#pragma unroll
    for (int sidx = 0; sidx < C16; ++sidx) kb[sidx] = *reinterpret_cast<const f32x4*>(krow + 16 * sidx);
  }
  for (int t = wave; t < ntile; t += kSmvWaves) {
    if (t + kSmvWaves < ntile) {
      const float* krow = b + (int64_t)min((t + kSmvWaves) * 16 + r16, m - 1) * c + 4 * q;
#pragma unroll
      for (int sidx = 0; sidx < C16; ++sidx) kn[sidx] = *reinterpret_cast<const f32x4*>(krow + 16 * sidx);
    }
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sidx = 0; sidx < C16; ++sidx) {
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kb[sidx][e], qa[sidx][e], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kb[sidx][e + 1], qa[sidx][e + 1], acc1, 0, 0, 0);
      }
    }
    const f32x4 acc = acc0 + acc1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = t * 16 + 4 * q + i;
      if (key < m) {
        const float sc = acc[i] * inv_temperature, wk = w[key];
        if (sc > mx) {
          const float f = expf(mx - sc);       // 0 on the first key (mx = -inf)
          den = den * f + 1.f;
          num = num * f + wk;
          mx = sc;
        } else {
          const float e = expf(sc - mx);
          den += e;
          num = fmaf(e, wk, num);
        }
      }
    }
#pragma unroll
    for (int sidx = 0; sidx < C16; ++sidx) kb[sidx] = kn[sidx];
  }
  s_mx[wave * 4 + q][r16] = mx;
  s_den[wave * 4 + q][r16] = den;
  s_num[wave * 4 + q][r16] = num;
  __syncthreads();
  if (threadIdx.x < 16 && i0 + (int)threadIdx.x < n) {
    const int qi = threadIdx.x;
    float M = -__builtin_inff();
    for (int p = 0; p < NP; ++p) M = fmaxf(M, s_mx[p][qi]);
    float D = 0.f, N = 0.f;
    for (int p = 0; p < NP; ++p) {
      if (s_den[p][qi] > 0.f) {                 // a part that saw no key (more parts than key tiles) has den = 0
        const float f = expf(s_mx[p][qi] - M);
        D = fmaf(s_den[p][qi], f, D);
        N = fmaf(s_num[p][qi], f, N);
      }
    }
    out[i0 + qi] = N / D;
  }
}

// Multi-head attention on the fp32 MFMA, inputs HEAD-MAJOR (channel h * 64 + d: the caller permutes the output channels
// of the three projections, see gcn.py), output in the reference's interleaved layout (channel d * heads + h) so the
// merge layer is untouched.  A workgroup owns 16 queries of ONE head; its 4 waves walk the 16-key tiles in turn with an
// online softmax.  Per tile: scores^T (rows = keys, columns = queries) as in k_softmax_matvec_mfma, so lane (r16, q)
// holds keys 4q .. 4q+3 against ITS query r16; the per-query tile maximum needs two cross-lane steps (the 4 q-lanes of
// a query); the probabilities are then already the B operand of O^T += V^T P (k-slot q of step i = key 4q + i), the A
// operand being V[key 4q+i][4 r16 + b] -- one 16-byte load per step serves the 4 row blocks b, and the accumulators end
// as O^T[d = 4 (4q+i) + b][query r16]: one query per lane, so the online rescale is a per-lane scalar.  The 4 waves'
// (max, denominator, numerator) triples are merged in fixed order through LDS.
__global__ __launch_bounds__(256) void k_mha_mfma(const float* __restrict__ qm, const float* __restrict__ km,
                                                  const float* __restrict__ vm, int n, int m, int heads,
                                                  float scl, float* __restrict__ out) {
  constexpr int DIM = 64;
  __shared__ float s_o[4][16][DIM + 1];      // [wave][query][d]
  __shared__ float s_mx[4][16], s_den[4][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  const int nqb = (n + 15) >> 4;
  const int qb = blockIdx.x % nqb, h = blockIdx.x / nqb;
  const int i0 = qb * 16, c = DIM * heads;
  const float* qrow = qm + (int64_t)min(i0 + r16, n - 1) * c + h * DIM + 4 * q;
  f32x4 qa[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) qa[s] = *reinterpret_cast<const f32x4*>(qrow + 16 * s);
  const float* kbase = km + h * DIM + 4 * q;     // + key * c + 16 s   (key = tile * 16 + r16)
  const float* vbase = vm + h * DIM + 4 * r16;   // + key * c          (key = tile * 16 + 4 q + i)
  const int ntile = (m + 15) >> 4;
  f32x4 kb[4], vv[4], kn[4], vn[4];
  auto load_tile = [&](int t, f32x4* kd, f32x4* vd) {
    const float* krow = kbase + (int64_t)min(t * 16 + r16, m - 1) * c;
#pragma unroll
    for (int s = 0; s < 4; ++s) kd[s] = *reinterpret_cast<const f32x4*>(krow + 16 * s);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      vd[i] = *reinterpret_cast<const f32x4*>(vbase + (int64_t)min(t * 16 + 4 * q + i, m - 1) * c);
  };
  if (wave < ntile) load_tile(wave, kb, vv);
  float mx = -__builtin_inff(), den = 0.f;
  f32x4 o[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) o[b] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int t = wave; t < ntile; t += 4) {
    if (t + 4 < ntile) load_tile(t + 4, kn, vn);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kb[s][e], qa[s][e], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kb[s][e + 1], qa[s][e + 1], acc1, 0, 0, 0);
      }
    }
    float sc[4], tmx = -__builtin_inff();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sc[i] = t * 16 + 4 * q + i < m ? (acc0[i] + acc1[i]) * scl : -__builtin_inff();
      tmx = fmaxf(tmx, sc[i]);
    }
    tmx = fmaxf(tmx, __shfl_xor(tmx, 16));
    tmx = fmaxf(tmx, __shfl_xor(tmx, 32));      // the tile holds at least one key: finite
    const float mnew = fmaxf(mx, tmx);
    const float f = expf(mx - mnew);            // 0 on the first tile
    mx = mnew;
    float p[4], ps = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      p[i] = expf(sc[i] - mnew);                // exp(-inf) = 0 for the keys past m
      ps += p[i];
    }
    den = fmaf(den, f, ps);
#pragma unroll
    for (int b = 0; b < 4; ++b) o[b] *= f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int b = 0; b < 4; ++b) o[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[i][b], p[i], o[b], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      kb[s] = kn[s];
      vv[s] = vn[s];
    }
  }
  den += __shfl_xor(den, 16);
  den += __shfl_xor(den, 32);
  if (q == 0) {
    s_mx[wave][r16] = mx;
    s_den[wave][r16] = den;
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) {
#pragma unroll
    for (int i = 0; i < 4; ++i) s_o[wave][r16][4 * (4 * q + i) + b] = o[b][i];
  }
  __syncthreads();
  // thread -> (query = tid / 16, d = tid % 16 + 16 j): waves that saw no tile carry den = 0
  const int qi = threadIdx.x >> 4, d0 = threadIdx.x & 15;
  if (i0 + qi < n) {
    float M = -__builtin_inff();
#pragma unroll
    for (int w = 0; w < 4; ++w) M = fmaxf(M, s_mx[w][qi]);
    float fw[4], D = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      fw[w] = s_den[w][qi] > 0.f ? expf(s_mx[w][qi] - M) : 0.f;
      D = fmaf(s_den[w][qi], fw[w], D);
    }
    const float inv = 1.f / D;
    float* orow = out + (int64_t)(i0 + qi) * c + h;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int d = d0 + 16 * j;
      float N = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) N = fmaf(s_o[w][qi][d], fw[w], N);
      orow[d * heads] = N * inv;
    }
  }
}

// y = clamp(sigmoid(x), 0, 1) with NaN / Inf -> 0   (architectures.py:131-134, 203-207)
__global__ void k_score_head(const float* __restrict__ x, int64_t ldx, int64_t n, float* __restrict__ y) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = 1.f / (1.f + expf(-x[i * ldx]));
  v = fminf(fmaxf(v, 0.f), 1.f);
  if (isnan(v) || isinf(v)) v = 0.f;
  y[i] = v;
}

}  // namespace

APR_API int apr_row_sums(const float* x, int64_t ld, int64_t n, int32_t c, float* out, void* stream) {
  APR_CHECK_ARG(n >= 0 && c > 0 && ld >= c, "apr_row_sums: bad shape");
  if (n == 0) return APR_OK;
  hipLaunchKernelGGL(k_row_sums, dim3((unsigned)cdiv64(n, 4)), dim3(256), 0, (hipStream_t)stream, x, ld, n, c, out);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API int apr_kpconv_weighted(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns, const int32_t* nbr,
                                int32_t H, const float* x, int64_t ldx, int32_t cin, const float* kernel_points,
                                int32_t n_kp, float extent, const float* rowsum, float* wf, int64_t ldwf, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APR_CHECK_ARG(n_kp == kKP, "apr_kpconv_weighted: built for %d kernel points, got %d", kKP, n_kp);
  APR_CHECK_ARG(nq >= 0 && ns > 0 && H > 0 && cin > 0 && extent > 0.f, "apr_kpconv_weighted: bad arguments");
  APR_CHECK_ARG(ldx >= cin && ldwf >= (int64_t)kKP * cin, "apr_kpconv_weighted: leading dimension too small");
  if (nq == 0) return APR_OK;
  const bool vec = cin % 64 == 0 && H <= kMaxH && ldx % 4 == 0 && ldwf % 4 == 0 && ((((uintptr_t)x) | ((uintptr_t)wf)) & 15) == 0;
  if (vec) {
    const unsigned grid = (unsigned)cdiv64(nq, 4);
    static const int s_xcd = env_int("APR_KPCONV_XCD", 1);      // A/B switch
    if (cin >= 256)
      hipLaunchKernelGGL(k_kpconv_weighted_mfma<4>, dim3(grid), dim3(256), 0, st, q_pts, s_pts, nbr, H, x, ldx, cin,
                         kernel_points, extent, rowsum, wf, ldwf, (int)nq, (int)ns, s_xcd);
    else if (cin == 128)
      hipLaunchKernelGGL(k_kpconv_weighted_mfma<2>, dim3(grid), dim3(256), 0, st, q_pts, s_pts, nbr, H, x, ldx, cin,
                         kernel_points, extent, rowsum, wf, ldwf, (int)nq, (int)ns, s_xcd);
    else
      hipLaunchKernelGGL(k_kpconv_weighted_mfma<1>, dim3(grid), dim3(256), 0, st, q_pts, s_pts, nbr, H, x, ldx, cin,
                         kernel_points, extent, rowsum, wf, ldwf, (int)nq, (int)ns, s_xcd);
  } else {
    hipLaunchKernelGGL(k_kpconv_weighted_generic, dim3((unsigned)cdiv64(nq * kKP, 256)), dim3(256), 0, st, q_pts, s_pts,
                       nbr, H, x, ldx, cin, kernel_points, extent, rowsum, wf, ldwf, (int)nq, (int)ns);
  }
  APR_LAUNCH_CHECK();
  return APR_OK;
}

static int kpconv_dfeat(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns, const int32_t* nbr, int32_t H,
                        const float* dwf, int64_t lddwf, int32_t cin, const float* kernel_points, int32_t n_kp,
                        float extent, const float* rowsum, float* dx, int64_t lddx, float* contrib, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APR_CHECK_ARG(n_kp == kKP, "apr_kpconv_dfeat: built for %d kernel points, got %d", kKP, n_kp);
  APR_CHECK_ARG(nq >= 0 && ns > 0 && H > 0 && H <= kMaxH && cin > 0 && cin <= 512 && extent > 0.f,
                "apr_kpconv_dfeat: bad arguments (H <= %d, cin <= 512)", kMaxH);
  APR_CHECK_ARG(lddx >= cin && lddwf >= (int64_t)kKP * cin, "apr_kpconv_dfeat: leading dimension too small");
  if (nq == 0) return APR_OK;
  const unsigned grid = (unsigned)cdiv64(nq, 4);
#define APR_DF(N)                                                                                                     \
  hipLaunchKernelGGL(k_kpconv_dfeat<N>, dim3(grid), dim3(256), 0, st, q_pts, s_pts, nbr, H, dwf, lddwf, cin, kernel_points, \
                     extent, rowsum, dx, lddx, (int)nq, (int)ns, contrib)
  if (cin <= 64) APR_DF(1);
  else if (cin <= 128) APR_DF(2);
  else if (cin <= 256) APR_DF(4);
  else APR_DF(8);
#undef APR_DF
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API int apr_kpconv_dfeat(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns, const int32_t* nbr, int32_t H,
                             const float* dwf, int64_t lddwf, int32_t cin, const float* kernel_points, int32_t n_kp,
                             float extent, const float* rowsum, float* dx, int64_t lddx, void* stream) {
  return kpconv_dfeat(q_pts, nq, s_pts, ns, nbr, H, dwf, lddwf, cin, kernel_points, n_kp, extent, rowsum, dx, lddx, nullptr,
                      stream);
}

// The deterministic form: the per-(query, neighbour) contributions go to contrib f32 [nq * H, cin] (rows of padding
// neighbours are left untouched and never read); apr_reverse_gather then sums every support point's rows in the order of
// the reverse table (apr_reverse_table_build): a fixed order, no float atomics.
APR_API int apr_kpconv_dfeat_contrib(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns, const int32_t* nbr,
                                     int32_t H, const float* dwf, int64_t lddwf, int32_t cin, const float* kernel_points,
                                     int32_t n_kp, float extent, const float* rowsum, float* contrib, void* stream) {
  APR_CHECK_ARG(contrib != nullptr, "apr_kpconv_dfeat_contrib: null contribution buffer");
  return kpconv_dfeat(q_pts, nq, s_pts, ns, nbr, H, dwf, lddwf, cin, kernel_points, n_kp, extent, rowsum, contrib, cin,
                      contrib, stream);
}

APR_API int apr_gather_pool(const float* x, int64_t ldx, int64_t ns, int32_t c, const int32_t* inds, int32_t H,
                            int64_t nq, int32_t mode, float* out, int64_t ldo, void* stream) {
  APR_CHECK_ARG(nq >= 0 && ns >= 0 && c > 0 && H > 0 && ldx >= c && ldo >= c && (mode == 0 || mode == 1),
                "apr_gather_pool: bad arguments");
  if (nq == 0) return APR_OK;
  const bool vec = (c & 3) == 0 && (ldx & 3) == 0 && (ldo & 3) == 0 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0;
  int64_t nblk = cdiv64(nq * (vec ? c / 4 : c), 256);
  if (nblk > 16384) nblk = 16384;
  if (vec)
    hipLaunchKernelGGL(k_gather_pool4, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, x, ldx, (int)ns, c, inds,
                       H, nq, mode, out, ldo);
  else
    hipLaunchKernelGGL(k_gather_pool, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, x, ldx, (int)ns, c, inds, H,
                       nq, mode, out, ldo);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

// max_pool (blocks.py:86-102) with the arg-max kept for the backward: amax u8 [nq, c] (H <= 255).
APR_API int apr_gather_pool_argmax(const float* x, int64_t ldx, int64_t ns, int32_t c, const int32_t* inds, int32_t H,
                                   int64_t nq, float* out, int64_t ldo, uint8_t* amax, void* stream) {
  APR_CHECK_ARG(nq >= 0 && ns >= 0 && c > 0 && H > 0 && H <= 255 && ldx >= c && ldo >= c && x && inds && out && amax,
                "apr_gather_pool_argmax: bad arguments (H <= 255)");
  if (nq == 0) return APR_OK;
  int64_t nblk = cdiv64(nq * c, 256);
  if (nblk > 16384) nblk = 16384;
  hipLaunchKernelGGL(k_gather_pool_argmax, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, x, ldx, (int)ns, c, inds, H,
                     nq, out, ldo, amax);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

// Input gradient of max_pool (mode 0, amax from apr_gather_pool_argmax) / closest_pool (mode 1, amax unused) over the
// reverse table of the SAME index tensor (apr_reverse_table_build(inds, nq, H, ns)): dx f32 [ns, c], every row written.
APR_API int apr_gather_pool_backward(const float* dout, int64_t lddo, int32_t c, const int32_t* rev_t, const int32_t* start,
                                     int64_t ns, int32_t H, const uint8_t* amax, int32_t mode, float* dx, int64_t lddx,
                                     void* stream) {
  APR_CHECK_ARG(dout && rev_t && start && dx && ns > 0 && c > 0 && H > 0 && lddo >= c && lddx >= c && (mode == 1 || amax) &&
                    (mode == 0 || mode == 1),
                "apr_gather_pool_backward: bad arguments");
  int64_t nblk = cdiv64(ns * c, 256);
  if (nblk > 16384) nblk = 16384;
  hipLaunchKernelGGL(k_gather_pool_bwd, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, dout, lddo, c, rev_t, start, ns,
                     H, amax, mode, dx, lddx);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API int apr_edge_features(const float* f, int64_t ldf, int32_t n, int32_t c, const int32_t* knn, int32_t k,
                              float* out, void* stream) {
  APR_CHECK_ARG(n > 0 && c > 0 && k > 0 && ldf >= c, "apr_edge_features: bad arguments");
  int64_t nblk = cdiv64((int64_t)n * k * c, 256);
  if (nblk > 16384) nblk = 16384;
  hipLaunchKernelGGL(k_edge_features, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, f, ldf, n, c, knn, k, out);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API int apr_edge_features_backward(const float* de, int32_t n, int32_t c, int32_t k, float* d_center, int64_t ldd,
                                       float* contrib, void* stream) {
  APR_CHECK_ARG(n > 0 && c > 0 && k > 0 && de && d_center && contrib && ldd >= c, "apr_edge_features_backward: bad arguments");
  int64_t nblk = cdiv64((int64_t)n * c, 256);
  if (nblk > 16384) nblk = 16384;
  hipLaunchKernelGGL(k_edge_features_bwd, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, de, n, c, k, d_center, ldd,
                     contrib);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API int apr_group_max(const float* y, int64_t ldy, int32_t n, int32_t k, int32_t c, const float* scale,
                          const float* shift, float slope, float* out, int64_t ldo, void* stream) {
  APR_CHECK_ARG(n > 0 && k > 0 && c > 0 && ldy >= c && ldo >= c, "apr_group_max: bad arguments");
  int64_t nblk = cdiv64((int64_t)n * c, 256);
  if (nblk > 16384) nblk = 16384;
  hipLaunchKernelGGL(k_group_max, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, y, ldy, n, k, c, scale, shift,
                     slope, out, ldo);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Multi-head attention for the TRAINING path (gcn.py:94-116 under lib/trainer.py:142-280): forward with the probabilities
// kept, backward from them.  Channel c = d * heads + h as the reference's view(-1, dim, heads).  The coarsest level holds
// ~1.4 k points per cloud: P [heads, n, m] is 31 MB, materialised once; every sum runs in a fixed order (no float atomics:
// the same bits every run).  Plain VALU kernels -- the inference path keeps k_mha_mfma.
//   k_mha_probs    block per (h, i): S_j = <q_i, k_j> / sqrt(dim) over the head's channels, P = softmax_j(S)
//   k_mha_rowmat   out[i, d*H + h] = scale * sum_j A[h, i, j] X[j, d*H + h]      (O = P V;  dQ = dS K / sqrt(dim))
//   k_mha_colmat   out[j, d*H + h] = scale * sum_i A[h, i, j] X[i, d*H + h]      (dV = P^T dO;  dK = dS^T Q / sqrt(dim))
//   k_mha_ds       block per (h, i): dP_j = <dO_i, v_j>,  D = sum_j P_j dP_j,  dS_j = P_j (dP_j - D)
// ---------------------------------------------------------------------------------------------------------------
__device__ inline float block_reduce_256(float v, bool is_max, float* s_red) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const float o = __shfl_xor(v, d);
    v = is_max ? fmaxf(v, o) : v + o;
  }
  const int wave = threadIdx.x >> 6;
  __syncthreads();                        // s_red may still be read from the previous reduction
  if ((threadIdx.x & 63) == 0) s_red[wave] = v;
  __syncthreads();
  float r = s_red[0];
  for (int w = 1; w < 4; ++w) r = is_max ? fmaxf(r, s_red[w]) : r + s_red[w];      // fixed order
  return r;
}

__global__ __launch_bounds__(256) void k_mha_probs(const float* __restrict__ q, const float* __restrict__ k, int n, int m, int dim,
                                                   int H, float* __restrict__ P) {
  extern __shared__ float s_q[];      // [dim] the query's head vector, then 4 reduction slots
  float* s_red = s_q + dim;
  const int h = blockIdx.x % H, i = blockIdx.x / H;
  const int c = dim * H;
  for (int d = threadIdx.x; d < dim; d += 256) s_q[d] = q[(int64_t)i * c + d * H + h];
  __syncthreads();
  const float scale = 1.0f / sqrtf((float)dim);
  float* row = P + ((int64_t)h * n + i) * m;
  float mx = -__builtin_inff();
  for (int j = threadIdx.x; j < m; j += 256) {
    const float* kj = k + (int64_t)j * c + h;
    float acc = 0.f;
    for (int d = 0; d < dim; ++d) acc += s_q[d] * kj[d * H];
    acc *= scale;
    row[j] = acc;
    mx = fmaxf(mx, acc);
  }
  mx = block_reduce_256(mx, true, s_red);
  float sum = 0.f;
  for (int j = threadIdx.x; j < m; j += 256) {
    const float e = __expf(row[j] - mx);
    row[j] = e;
    sum += e;
  }
  sum = block_reduce_256(sum, false, s_red);
  const float inv = 1.0f / sum;
  for (int j = threadIdx.x; j < m; j += 256) row[j] *= inv;
}

// thread per (i, channel): the head's row of A is walked in j order
__global__ void k_mha_rowmat(const float* __restrict__ A, const float* __restrict__ X, int n, int m, int dim, int H, float scale,
                             float* __restrict__ out) {
  const int c = dim * H;
  const int64_t total = (int64_t)n * c;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(t / c), ch = (int)(t - (int64_t)i * c), h = ch % H;
    const float* a = A + ((int64_t)h * n + i) * m;
    float acc = 0.f;
    for (int j = 0; j < m; ++j) acc += a[j] * X[(int64_t)j * c + ch];
    out[t] = acc * scale;
  }
}

// thread per (j, channel): the head's column of A is walked in i order
__global__ void k_mha_colmat(const float* __restrict__ A, const float* __restrict__ X, int n, int m, int dim, int H, float scale,
                             float* __restrict__ out) {
  const int c = dim * H;
  const int64_t total = (int64_t)m * c;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(t / c), ch = (int)(t - (int64_t)j * c), h = ch % H;
    const float* a = A + (int64_t)h * n * m + j;
    float acc = 0.f;
    for (int i = 0; i < n; ++i) acc += a[(int64_t)i * m] * X[(int64_t)i * c + ch];
    out[t] = acc * scale;
  }
}

__global__ __launch_bounds__(256) void k_mha_ds(const float* __restrict__ P, const float* __restrict__ dO, const float* __restrict__ v,
                                                int n, int m, int dim, int H, float* __restrict__ dS) {
  extern __shared__ float s_q[];      // [dim] dO_i's head vector, then 4 reduction slots
  float* s_red = s_q + dim;
  const int h = blockIdx.x % H, i = blockIdx.x / H;
  const int c = dim * H;
  for (int d = threadIdx.x; d < dim; d += 256) s_q[d] = dO[(int64_t)i * c + d * H + h];
  __syncthreads();
  const float* p = P + ((int64_t)h * n + i) * m;
  float* ds = dS + ((int64_t)h * n + i) * m;
  float dsum = 0.f;
  for (int j = threadIdx.x; j < m; j += 256) {
    const float* vj = v + (int64_t)j * c + h;
    float acc = 0.f;
    for (int d = 0; d < dim; ++d) acc += s_q[d] * vj[d * H];
    ds[j] = acc;                       // dP_j for now
    dsum += p[j] * acc;
  }
  const float D = block_reduce_256(dsum, false, s_red);
  for (int j = threadIdx.x; j < m; j += 256) ds[j] = p[j] * (ds[j] - D);
}

// vdim: channels per head of v / out / dout (0: the same as dim; 1 with heads = 1: the cross-saliency softmax(<a, b> / T) @ s
// of architectures.py:176-181, with q pre-scaled by sqrt(dim) / T)
APR_API int apr_mha_train_forward(const float* q, const float* k, const float* v, int32_t n, int32_t m, int32_t dim, int32_t heads,
                                  int32_t vdim, float* P, float* out, void* stream) {
  if (vdim <= 0) vdim = dim;
  APR_CHECK_ARG(n > 0 && m > 0 && dim > 0 && heads > 0 && dim <= 1024 && vdim <= 1024 && q && k && v && P && out,
                "apr_mha_train_forward: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_mha_probs, dim3((unsigned)(n * heads)), dim3(256), (size_t)(dim + 4) * 4, st, q, k, n, m, dim, heads, P);
  int64_t nblk = cdiv64((int64_t)n * vdim * heads, 256);
  hipLaunchKernelGGL(k_mha_rowmat, dim3((unsigned)(nblk > 16384 ? 16384 : nblk)), dim3(256), 0, st, P, v, n, m, vdim, heads, 1.0f,
                     out);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API int apr_mha_train_backward(const float* q, const float* k, const float* v, const float* P, const float* dout, int32_t n,
                                   int32_t m, int32_t dim, int32_t heads, int32_t vdim, float* dS, float* dq, float* dk, float* dv,
                                   void* stream) {
  if (vdim <= 0) vdim = dim;
  APR_CHECK_ARG(n > 0 && m > 0 && dim > 0 && heads > 0 && dim <= 1024 && vdim <= 1024 && q && k && v && P && dout && dS && dq && dk && dv,
                "apr_mha_train_backward: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const float scale = 1.0f / sqrtf((float)dim);
  auto blocks = [](int64_t rows, int ch) {
    int64_t b = cdiv64(rows * ch, 256);
    return (unsigned)(b > 16384 ? 16384 : b);
  };
  hipLaunchKernelGGL(k_mha_colmat, dim3(blocks(m, vdim * heads)), dim3(256), 0, st, P, dout, n, m, vdim, heads, 1.0f, dv);
  hipLaunchKernelGGL(k_mha_ds, dim3((unsigned)(n * heads)), dim3(256), (size_t)(vdim + 4) * 4, st, P, dout, v, n, m, vdim, heads, dS);
  hipLaunchKernelGGL(k_mha_rowmat, dim3(blocks(n, dim * heads)), dim3(256), 0, st, dS, k, n, m, dim, heads, scale, dq);
  hipLaunchKernelGGL(k_mha_colmat, dim3(blocks(m, dim * heads)), dim3(256), 0, st, dS, q, n, m, dim, heads, scale, dk);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API int apr_mha(const float* q, const float* k, const float* v, int32_t n, int32_t m, int32_t dim, int32_t heads,
                    float* out, void* stream) {
  APR_CHECK_ARG(n > 0 && m > 0 && dim > 0 && heads > 0, "apr_mha: bad arguments");
  APR_CHECK_ARG(dim <= 256 && 256 % dim == 0, "apr_mha: head dimension %d must divide 256", dim);
  // scores [kMhaQ][m] (reused for the [256 / dim][kMhaQ][dim] partial sums: needs m >= 256) + queries [kMhaQ][dim]
  const size_t lds = (size_t)kMhaQ * ((m > 256 ? m : 256) + dim) * 4;
  APR_CHECK_ARG(lds <= 60 * 1024, "apr_mha: at most %d keys supported", (int)((60 * 1024) / 4 / kMhaQ - dim));
  const unsigned nqb = (unsigned)((n + kMhaQ - 1) / kMhaQ);
  hipLaunchKernelGGL(k_mha, dim3(nqb * (unsigned)heads), dim3(256), lds, (hipStream_t)stream, q, k, v, n, m, dim, heads,
                     out);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

// apr_mha with q / k / v HEAD-MAJOR (channel h * dim + d) and out interleaved (channel d * heads + h) as apr_mha writes
// it: dim = 64 only (k_mha_mfma), rows 16-byte aligned; no limit on m.  Same value to fp32 summation order.
APR_API int apr_mha_headmajor(const float* q, const float* k, const float* v, int32_t n, int32_t m, int32_t dim,
                              int32_t heads, float* out, void* stream) {
  APR_CHECK_ARG(n > 0 && m > 0 && heads > 0 && q && k && v && out, "apr_mha_headmajor: bad arguments");
  APR_CHECK_ARG(dim == 64, "apr_mha_headmajor: head dimension %d not supported (64 only)", dim);
  APR_CHECK_ARG((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0, "apr_mha_headmajor: q, k, v must be 16-byte aligned");
  const unsigned nqb = (unsigned)((n + 15) / 16);
  hipLaunchKernelGGL(k_mha_mfma, dim3(nqb * (unsigned)heads), dim3(256), 0, (hipStream_t)stream, q, k, v, n, m, heads,
                     1.f / sqrtf((float)dim), out);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API int apr_softmax_matvec(const float* a, const float* b, const float* w, int32_t n, int32_t m, int32_t c,
                               float temperature, float* out, void* stream) {
  APR_CHECK_ARG(n > 0 && m > 0 && c > 0 && temperature > 0.f, "apr_softmax_matvec: bad arguments");
  APR_CHECK_ARG((size_t)(m + c) * 4 <= 60 * 1024, "apr_softmax_matvec: m + c too large for LDS");
  hipLaunchKernelGGL(k_softmax_matvec, dim3((unsigned)n), dim3(256), (size_t)(m + c) * 4, (hipStream_t)stream, a, b, w, n,
                     m, c, temperature, out);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

// b transposed by the caller (bt [c, m] row-major): coalesced reads, 4 queries per workgroup; same bits as the above
APR_API int apr_softmax_matvec_bt(const float* a, const float* bt, const float* w, int32_t n, int32_t m, int32_t c,
                                  float temperature, float* out, void* stream) {
  APR_CHECK_ARG(n > 0 && m > 0 && c > 0 && temperature > 0.f && a && bt && w && out, "apr_softmax_matvec_bt: bad arguments");
  const size_t lds = (size_t)kSmvQ * (m + c) * 4;
  APR_CHECK_ARG(lds <= 60 * 1024, "apr_softmax_matvec_bt: m + c too large for LDS");
  hipLaunchKernelGGL(k_softmax_matvec_t, dim3((unsigned)((n + kSmvQ - 1) / kSmvQ)), dim3(256), lds, (hipStream_t)stream,
                     a, bt, w, n, m, c, temperature, out);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

// The same on the fp32 MFMA (k_softmax_matvec_mfma): a and b row-major as in apr_softmax_matvec, 16-byte aligned,
// c in {32, 64, 128, 256}; no limit on m.  Same value to fp32 summation order.
APR_API int apr_softmax_matvec_mfma(const float* a, const float* b, const float* w, int32_t n, int32_t m, int32_t c,
                                    float temperature, float* out, void* stream) {
  APR_CHECK_ARG(n > 0 && m > 0 && temperature > 0.f && a && b && w && out, "apr_softmax_matvec_mfma: bad arguments");
  APR_CHECK_ARG((c == 32 || c == 64 || c == 128 || c == 256) && (((uintptr_t)a | (uintptr_t)b) & 15) == 0,
                "apr_softmax_matvec_mfma: c must be 32 / 64 / 128 / 256 and a, b 16-byte aligned");
  const dim3 grid((unsigned)((n + 15) / 16));
  const float it = 1.f / temperature;
  hipStream_t st = (hipStream_t)stream;
  if (c == 32) hipLaunchKernelGGL(k_softmax_matvec_mfma<2>, grid, dim3(64 * kSmvWaves), 0, st, a, b, w, n, m, it, out);
  else if (c == 64) hipLaunchKernelGGL(k_softmax_matvec_mfma<4>, grid, dim3(64 * kSmvWaves), 0, st, a, b, w, n, m, it, out);
  else if (c == 128) hipLaunchKernelGGL(k_softmax_matvec_mfma<8>, grid, dim3(64 * kSmvWaves), 0, st, a, b, w, n, m, it, out);
  else hipLaunchKernelGGL(k_softmax_matvec_mfma<16>, grid, dim3(64 * kSmvWaves), 0, st, a, b, w, n, m, it, out);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API int apr_score_head(const float* x, int64_t ldx, int64_t n, float* y, void* stream) {
  APR_CHECK_ARG(n >= 0 && ldx >= 1, "apr_score_head: bad arguments");
  if (n == 0) return APR_OK;
  hipLaunchKernelGGL(k_score_head, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, n, y);
  APR_LAUNCH_CHECK();
  return APR_OK;
}
