// Sparse convolution forward for gfx950: output-stationary gather -> LDS -> fp32 MFMA
// with a fused epilogue.
//
//   out[j,:] = act((sum_o in[nbr[j,o],:] @ W[o]) * scale + shift + residual[j,:])
//
// Design (MI355X-first, not a port of ME's gather-GEMM-scatter):
//   * output-stationary: a workgroup owns TM output rows x CN output channels and
//     walks the kernel offsets, so there is no scatter, no atomics and the result
//     is bitwise reproducible.  Offsets no row of the tile uses are skipped from a
//     27-bit occupancy mask built while the tile's slice of the neighbour table
//     is staged in LDS; 16-row MFMA blocks with no neighbour skip their MFMAs.
//   * the gathered rows (16-B loads, >=128 B contiguous per row) are staged in LDS
//     with a row stride of CK+8 floats, which makes the ds_read_b128 fragment reads
//     of v_mfma_f32_16x16x4_f32 conflict-free (slot = (2r+q) mod 16 per lane group).
//   * weights are pre-packed as Wp[k][cin/4][cout][4] so a B fragment is one
//     coalesced 16-B load per lane straight from L2 (all tiles share the weights).
//   * next step's gather + weight fragments are prefetched into registers while
//     the current step's MFMAs run.
//   * fp32 in / fp32 accumulate MFMA (exact f32; gfx950 has no xf32), so features
//     match the fp32 oracle to ~1e-6.
// Algorithmic bytes per launch: 4*P*(cin+cout) + 8*P, P = kernel-map pairs.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kPad = 8;

template <int TM, int CN, int WM, int WN, int CK>
__global__ __launch_bounds__(256) void k_spconv_mfma(
    const float* __restrict__ in, int64_t ldi, const int* __restrict__ nbr, int n_out, int K,
    int cin, int cout, const float* __restrict__ wp, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ residual, int64_t ldr, int relu,
    float* __restrict__ out, int64_t ldo) {
  constexpr int RBW = TM / WM / 16;       // 16-row blocks per wave
  constexpr int CBW = CN / WN / 16;       // 16-col blocks per wave
  constexpr int NRB = TM / 16;            // row blocks per tile
  constexpr int LDA = CK + kPad;          // LDS row stride (floats)
  constexpr int NV = TM * CK / 4 / 256;   // float4 gathers per thread per step
  constexpr int NJ = CK / 16;             // 16-wide k groups per chunk
  static_assert(NV >= 1, "tile too small");

  __shared__ int s_nbr[TM * 32];
  __shared__ __attribute__((aligned(16))) float s_A[TM * LDA];
  __shared__ unsigned s_rbmask[NRB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, q = lane >> 4;

  // XCD-aware block -> tile map: blocks that share an XCD (b % 8) get a contiguous
  // range of tiles, so neighbouring tiles (overlapping gathers) hit the same L2.
  const int nb = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, loc = bid >> 3;
  const int lin = xcd * (nb >> 3) + min(xcd, nb & 7) + loc;
  const int ncol = cout / CN;
  const int tile_m = lin / ncol, tile_n = lin - tile_m * ncol;
  const int row0 = tile_m * TM;

  if (tid < NRB) s_rbmask[tid] = 0u;
  __syncthreads();
  for (int t = tid; t < TM * K; t += 256) {
    int r = t / K, k = t - r * K;
    int row = row0 + r;
    int v = -1;
    if (row < n_out) v = nbr ? nbr[(int64_t)row0 * K + t] : row;
    s_nbr[r * 32 + k] = v;
    if (v >= 0) atomicOr(&s_rbmask[r >> 4], 1u << k);
  }
  __syncthreads();
  const int wm = wave / WN, wn = wave - wm * WN;
  const int wrow0 = wm * (TM / WM);
  unsigned kmask = 0u;
#pragma unroll
  for (int i = 0; i < NRB; ++i) kmask |= s_rbmask[i];
  unsigned rbm[RBW];  // occupancy of this wave's own 16-row blocks
#pragma unroll
  for (int i = 0; i < RBW; ++i) rbm[i] = s_rbmask[wrow0 / 16 + i];
  const int col0 = tile_n * CN + wn * (CN / WN);
  const int cinG = cin >> 2;
  const int nchunk = cin / CK;

  f32x4 acc[RBW][CBW];
#pragma unroll
  for (int a = 0; a < RBW; ++a)
#pragma unroll
    for (int b = 0; b < CBW; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  f32x4 pa[NV];            // prefetched gather
  f32x4 pb[CBW][NJ];       // prefetched weight fragments

  auto prefetch = [&](int k, int chunk) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      int e = v * 256 + tid;
      int r = e / (CK / 4), c4 = e - r * (CK / 4);
      int idx = s_nbr[r * 32 + k];
      if (idx >= 0)
        pa[v] = *reinterpret_cast<const f32x4*>(in + (int64_t)idx * ldi + chunk * CK + c4 * 4);
      else
        pa[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        int g = chunk * (CK / 4) + j * 4 + q;
        pb[cb][j] = *reinterpret_cast<const f32x4*>(
            wp + (((int64_t)k * cinG + g) * cout + col0 + cb * 16 + r16) * 4);
      }
  };

  unsigned rem = kmask;
  int k_cur = rem ? __builtin_ctz(rem) : -1;
  int ch_cur = 0;
  if (k_cur >= 0) prefetch(k_cur, 0);

  while (k_cur >= 0) {
    // next step
    int k_nxt = k_cur, ch_nxt = ch_cur + 1;
    if (ch_nxt == nchunk) {
      ch_nxt = 0;
      rem &= rem - 1;
      k_nxt = rem ? __builtin_ctz(rem) : -1;
    }
    __syncthreads();  // previous step's fragment reads are done
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      int e = v * 256 + tid;
      int r = e / (CK / 4), c4 = e - r * (CK / 4);
      *reinterpret_cast<f32x4*>(&s_A[r * LDA + c4 * 4]) = pa[v];
    }
    f32x4 b[CBW][NJ];
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
      for (int j = 0; j < NJ; ++j) b[cb][j] = pb[cb][j];
    __syncthreads();
    if (k_nxt >= 0) prefetch(k_nxt, ch_nxt);

#pragma unroll
    for (int rb = 0; rb < RBW; ++rb) {
      if (!((rbm[rb] >> k_cur) & 1u)) continue;  // wave-uniform
      const float* arow = &s_A[(wrow0 + rb * 16 + r16) * LDA + q * 4];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        f32x4 a = *reinterpret_cast<const f32x4*>(arow + j * 16);
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) {
          acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[cb][j][0], acc[rb][cb], 0, 0, 0);
          acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[cb][j][1], acc[rb][cb], 0, 0, 0);
          acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[cb][j][2], acc[rb][cb], 0, 0, 0);
          acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[cb][j][3], acc[rb][cb], 0, 0, 0);
        }
      }
    }
    k_cur = k_nxt;
    ch_cur = ch_nxt;
  }

  // epilogue: C/D map of 16x16 MFMA: col = lane&15, row = 4*(lane>>4) + reg
#pragma unroll
  for (int cb = 0; cb < CBW; ++cb) {
    const int col = col0 + cb * 16 + r16;
    const float sc = scale ? scale[col] : 1.f;
    const float sh = shift ? shift[col] : 0.f;
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int row = row0 + wrow0 + rb * 16 + q * 4 + i;
        if (row < n_out) {
          float v = acc[rb][cb][i] * sc + sh;
          if (residual) v += residual[(int64_t)row * ldr + col];
          if (relu) v = fmaxf(v, 0.f);
          out[(int64_t)row * ldo + col] = v;
        }
      }
    }
  }
}

// Generic VALU path: any cin/cout/K (conv1 with cin = 1 or 3, odd channel counts).
// One thread per (row, out channel); weights in the reference's [K,cin,cout] layout.
__global__ void k_spconv_generic(const float* __restrict__ in, int64_t ldi,
                                 const int* __restrict__ nbr, int64_t n_out, int K, int cin,
                                 int cout, const float* __restrict__ w,
                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                 const float* __restrict__ residual, int64_t ldr, int relu,
                                 float* __restrict__ out, int64_t ldo) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_out * cout) return;
  int64_t row = t / cout;
  int c = (int)(t - row * cout);
  float acc = 0.f;
  for (int k = 0; k < K; ++k) {
    int64_t idx = nbr ? nbr[row * K + k] : row;
    if (idx < 0) continue;
    const float* x = in + idx * ldi;
    const float* wk = w + (int64_t)k * cin * cout + c;
    for (int ci = 0; ci < cin; ++ci) acc = fmaf(x[ci], wk[(int64_t)ci * cout], acc);
  }
  float v = acc * (scale ? scale[c] : 1.f) + (shift ? shift[c] : 0.f);
  if (residual) v += residual[row * ldr + c];
  if (relu) v = fmaxf(v, 0.f);
  out[row * ldo + c] = v;
}

__global__ void k_pack_weights(const float* __restrict__ w, int K, int cin, int cout,
                               float* __restrict__ wp) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = (int64_t)K * cin * cout;
  if (t >= total) return;
  int co = (int)(t % cout);
  int ci = (int)((t / cout) % cin);
  int k = (int)(t / ((int64_t)cout * cin));
  // Wp[k][ci/4][co][ci%4]
  wp[(((int64_t)k * (cin >> 2) + (ci >> 2)) * cout + co) * 4 + (ci & 3)] = w[t];
}

inline bool use_mfma(int K, int cin, int cout) {
  return K <= 32 && cin % 32 == 0 && cout % 32 == 0;
}

template <int TM, int CN, int WM, int WN, int CK>
int launch_mfma(const float* in, int64_t ldi, const int* nbr, int64_t n_out, int K, int cin,
                int cout, const float* wp, const float* scale, const float* shift,
                const float* residual, int64_t ldr, int relu, float* out, int64_t ldo,
                hipStream_t st) {
  int64_t tiles = cdiv64(n_out, TM) * (cout / CN);
  hipLaunchKernelGGL((k_spconv_mfma<TM, CN, WM, WN, CK>), dim3((unsigned)tiles), dim3(256), 0, st, in,
                     ldi, nbr, (int)n_out, K, cin, cout, wp, scale, shift, residual, ldr, relu, out,
                     ldo);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

}  // namespace

APR_API int64_t apr_spconv_packed_size(int32_t K, int32_t cin, int32_t cout) {
  return (int64_t)K * cin * cout;
}

APR_API int apr_spconv_pack_weights(const float* w, int32_t K, int32_t cin, int32_t cout,
                                    float* w_packed, void* stream) {
  APR_CHECK_ARG(K > 0 && cin > 0 && cout > 0, "apr_spconv_pack_weights: bad shape");
  int64_t total = (int64_t)K * cin * cout;
  if (use_mfma(K, cin, cout)) {
    hipLaunchKernelGGL(k_pack_weights, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, w, K, cin, cout, w_packed);
    APR_LAUNCH_CHECK();
  } else {
    APR_HIP(hipMemcpyAsync(w_packed, w, total * sizeof(float), hipMemcpyDeviceToDevice,
                           (hipStream_t)stream));
  }
  return APR_OK;
}

APR_API int apr_spconv_fwd(const float* in, int64_t ldi, const int32_t* nbr, int64_t n_out, int32_t K,
                           int32_t cin, int32_t cout, const float* w_packed, const float* scale,
                           const float* shift, const float* residual, int64_t ldr, int32_t relu,
                           float* out, int64_t ldo, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APR_CHECK_ARG(n_out >= 0 && n_out < (1ll << 31), "apr_spconv_fwd: n_out=%lld", (long long)n_out);
  APR_CHECK_ARG(K >= 1 && cin >= 1 && cout >= 1, "apr_spconv_fwd: bad K/cin/cout");
  APR_CHECK_ARG(nbr != nullptr || K == 1, "apr_spconv_fwd: identity map needs K == 1");
  APR_CHECK_ARG(ldi >= cin && ldo >= cout, "apr_spconv_fwd: leading dimension smaller than channels");
  APR_CHECK_ARG(!residual || ldr >= cout, "apr_spconv_fwd: ldr < cout");
  if (n_out == 0) return APR_OK;
  if (use_mfma(K, cin, cout) && (ldi % 4) == 0 && (((uintptr_t)in) & 15) == 0) {
    const bool big = n_out >= 32768;
    if (cout % 64 == 0) {
      if (cin % 64 == 0) {
        if (big)
          return launch_mfma<64, 64, 1, 4, 64>(in, ldi, nbr, n_out, K, cin, cout, w_packed, scale, shift,
                                               residual, ldr, relu, out, ldo, st);
        return launch_mfma<32, 64, 1, 4, 64>(in, ldi, nbr, n_out, K, cin, cout, w_packed, scale, shift,
                                             residual, ldr, relu, out, ldo, st);
      }
      if (big)
        return launch_mfma<64, 64, 1, 4, 32>(in, ldi, nbr, n_out, K, cin, cout, w_packed, scale, shift,
                                             residual, ldr, relu, out, ldo, st);
      return launch_mfma<32, 64, 1, 4, 32>(in, ldi, nbr, n_out, K, cin, cout, w_packed, scale, shift,
                                           residual, ldr, relu, out, ldo, st);
    }
    if (cin % 64 == 0) {
      if (big)
        return launch_mfma<64, 32, 2, 2, 64>(in, ldi, nbr, n_out, K, cin, cout, w_packed, scale, shift,
                                             residual, ldr, relu, out, ldo, st);
      return launch_mfma<32, 32, 2, 2, 64>(in, ldi, nbr, n_out, K, cin, cout, w_packed, scale, shift,
                                           residual, ldr, relu, out, ldo, st);
    }
    if (big)
      return launch_mfma<64, 32, 2, 2, 32>(in, ldi, nbr, n_out, K, cin, cout, w_packed, scale, shift,
                                           residual, ldr, relu, out, ldo, st);
    return launch_mfma<32, 32, 2, 2, 32>(in, ldi, nbr, n_out, K, cin, cout, w_packed, scale, shift,
                                         residual, ldr, relu, out, ldo, st);
  }
  APR_CHECK_ARG(!use_mfma(K, cin, cout),
                "apr_spconv_fwd: MFMA-shaped layer needs 16-B aligned input rows (ldi %% 4 == 0)");
  int64_t total = n_out * cout;
  hipLaunchKernelGGL(k_spconv_generic, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, in, ldi, nbr,
                     n_out, K, cin, cout, w_packed, scale, shift, residual, ldr, relu, out, ldo);
  APR_LAUNCH_CHECK();
  return APR_OK;
}
