// Weight gradient of the sparse convolution (FCGF_APR/lib/trainer.py:454-527 calls
// loss.backward() through MinkowskiConvolution / MinkowskiConvolutionTranspose):
//     dW[k][ci][co] = sum_j [nbr[j,k] >= 0]  in[nbr[j,k]][ci] * dout[j][co]
// (the input gradient needs no new kernel: it IS a sparse conv of dout with transposed weights over the reverse
// map — the same table with mirrored offsets for same-level convs, the transposed-conv table for strided ones).
//
// Two deterministic phases, no float atomics:
//   k_wgrad_partial: workgroup = (row chunk of 512 output rows, offset k, 64x64 block of dW[k]); wave w owns input
//       channels 16w..16w+15 and the 4 column blocks: per step of 4 rows one v_mfma_f32_16x16x4_f32 per column block
//       with A[m = ci][kk = row] = gathered input value (0 for an empty table entry) and B[kk = row][n = co] = dout;
//       the 64x64 partial sum goes to scratch[chunk][k][ci][co];
//   k_wgrad_reduce: dW = sum over chunks in ascending order.
#include "common.h"
#include "bf3.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kChunkRows = 512;

__global__ __launch_bounds__(256) void k_wgrad_partial(const float* __restrict__ in, int64_t ldi,
                                                       const float* __restrict__ dout, int64_t ldo,
                                                       const int* __restrict__ nbr, int64_t n_out, int K, int cin,
                                                       int cout, float* __restrict__ part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, r = lane >> 4;
  const int chunk = blockIdx.x, k = blockIdx.y;
  const int nco = (cout + 63) / 64;
  const int ci0 = (blockIdx.z / nco) * 64 + wave * 16, co0 = (blockIdx.z % nco) * 64;
  const int64_t j0 = (int64_t)chunk * kChunkRows;
  const int64_t j1 = min((long long)(j0 + kChunkRows), (long long)n_out);
  const bool ci_ok = ci0 + m < cin;
  f32x4 acc[4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int64_t jb = j0; jb < j1; jb += 16) {       // 4 MFMA steps per iteration: 4 x 6 loads in flight
    float a[4], b[4][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int64_t j = jb + s * 4 + r;
      const bool row_ok = j < j1;
      const int idx = (nbr && row_ok) ? nbr[j * K + k] : (row_ok ? (int)j : -1);
      a[s] = (idx >= 0 && ci_ok) ? in[(int64_t)idx * ldi + ci0 + m] : 0.f;
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        const int co = co0 + cb * 16 + m;
        b[s][cb] = (row_ok && co < cout) ? dout[j * ldo + co] : 0.f;
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s][cb], acc[cb], 0, 0, 0);
  }
  // D[row = 4 r + reg][col = m]: row = input channel within the wave's 16, col = output channel within the block
  float* dst = part + (((int64_t)chunk * K + k) * cin) * cout;
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ci = ci0 + 4 * r + g, co = co0 + cb * 16 + m;
      if (ci < cin && co < cout) dst[(int64_t)ci * cout + co] = acc[cb][g];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the same gradient on the bf16 MFMA in the exact 3-way split (bf3.h: 6 of the 9 cross terms, fp32 accumulate --
// fp32-equivalent, 2.7x fewer MFMA cycles than v_mfma_f32_16x16x4_f32) with the gathered rows staged through LDS.
//   work item = (offset k, split s of S_k): the workgroup walks the 512-row blocks s, s + S_k, ... of the kernel map,
//   compacts the block's non-empty entries of column k into an LDS pair list (wave64 ballot + popcount, row order), and
//   contracts them 32 pairs per step: the 32 gathered input rows and the 32 dout rows of the step are split into three bf16
//   pieces and written as [pair][channel] images (XOR-swizzled 16-B chunks); both MFMA operands want 8 consecutive PAIRS
//   of one channel per lane, which is the transposed read ds_read_b64_tr_b16 of those images.  The 4 waves own the 2 x 2
//   quadrants of the CI x CO block of dW[k] in accumulator registers for the whole walk (no partial leaves the workgroup
//   until the end: S_k partials per offset instead of one per 512 rows); the rows of step t + 1 are in flight (registers)
//   under the MFMAs of step t, one barrier per step.  k_wgrad_reduce_work sums the S_k partials of an offset in ascending
//   order: no atomics, the same bits every run.
// ---------------------------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr int kRowBlock = 512;

template <int PITCH>
__device__ inline int img_off(int row, int ch16) {           // byte offset of 16-B chunk ch16 of a row of PITCH bytes
  constexpr int NCH = PITCH / 16;
  return PITCH * row + 16 * (ch16 ^ ((((row & 3) << 2) | ((row >> 2) & 3)) & (NCH - 1)));
}

__device__ inline void split3x4(const f32x4& v, u32x2& h, u32x2& m, u32x2& l) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float x0 = v[2 * i], x1 = v[2 * i + 1];
    h[i] = apr_pack_hi16(x1, x0);
    const float r0 = x0 - __uint_as_float(__float_as_uint(x0) & 0xFFFF0000u);
    const float r1 = x1 - __uint_as_float(__float_as_uint(x1) & 0xFFFF0000u);
    m[i] = apr_pack_hi16(r1, r0);
    const float t0 = r0 - __uint_as_float(__float_as_uint(r0) & 0xFFFF0000u);
    const float t1 = r1 - __uint_as_float(__float_as_uint(r1) & 0xFFFF0000u);
    l[i] = apr_pack_hi16(t1, t0);
  }
}

template <int PITCH>
__device__ inline bf16x8 tr_frag(const char* img, int c16, int lane) {
  // 16x16x32 operand: lane (g = lane >> 4, i = lane & 15) ends with channel 8 * c16 + i of pairs 8g .. 8g + 7
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  typedef s16x4 __attribute__((address_space(3))) * lds_p;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (lds_p)(img + img_off<PITCH>(8 * g + q, c16 + (p >> 1)) + 8 * (p & 1)));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (lds_p)(img + img_off<PITCH>(8 * g + 4 + q, c16 + (p >> 1)) + 8 * (p & 1)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <int CI, int CO>
__global__ __launch_bounds__(256) void k_wgrad_bf3(const float* __restrict__ in, int64_t ldi, const float* __restrict__ dout,
                                                   int64_t ldo, const int* __restrict__ nbr, int64_t n_out, int K, int cin,
                                                   int cout, int S_all, int S_c, int k_c, float* __restrict__ part) {
  constexpr int PA = CI * 2, PB = CO * 2;
  constexpr int IMG_A = 32 * PA, IMG_B = 32 * PB, STAGE = 3 * (IMG_A + IMG_B);
  constexpr int MT = CI / 32, NT = CO / 32;                 // 16 x 16 tiles of a wave's quadrant
  constexpr int LA = CI / 32, LB = CO / 32;                 // 16-B row chunks a thread stages per step and operand
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* s_pairs = (int*)(smem + 2 * STAGE);                  // [kRowBlock][2] = (input row, output row)
  __shared__ int s_wcnt[4][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // work item -> (offset, split): the offsets in order with S_all splits each; on a same-level map (k_c >= 0) the centre
  // offset -- it holds EVERY row, the others about a quarter -- is skipped there and comes last with S_c splits
  const int n_plain = (k_c >= 0 ? K - 1 : K) * S_all;
  const bool centre = (int)blockIdx.x >= n_plain;
  const int w = centre ? (int)blockIdx.x - n_plain : (int)blockIdx.x;
  const int kk = w / S_all;
  const int k = centre ? k_c : ((k_c >= 0 && kk >= k_c) ? kk + 1 : kk);
  const int s0 = centre ? w : w - kk * S_all;
  const int S = centre ? S_c : S_all;
  const int nco = cout / CO;
  const int ci0 = (blockIdx.y / nco) * CI, co0 = (blockIdx.y % nco) * CO;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int64_t nrb = (n_out + kRowBlock - 1) / kRowBlock;
  for (int64_t rb = s0; rb < nrb; rb += S) {
    // ---- compact the block's entries of column k (row order) ----
    const int64_t ja = rb * kRowBlock + tid, jb = ja + 256;
    const int ia = ja < n_out ? (nbr ? nbr[ja * K + k] : (int)ja) : -1;
    const int ib = jb < n_out ? (nbr ? nbr[jb * K + k] : (int)jb) : -1;
    const unsigned long long ba = __ballot(ia >= 0), bb = __ballot(ib >= 0);
    if (lane == 0) {
      s_wcnt[wave][0] = __popcll(ba);
      s_wcnt[wave][1] = __popcll(bb);
    }
    __syncthreads();
    int base_a = 0, base_b = 0, tot_a = 0, tot_b = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (w < wave) {
        base_a += s_wcnt[w][0];
        base_b += s_wcnt[w][1];
      }
      tot_a += s_wcnt[w][0];
      tot_b += s_wcnt[w][1];
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (ia >= 0) {
      const int pos = base_a + __popcll(ba & lt);
      s_pairs[2 * pos] = ia;
      s_pairs[2 * pos + 1] = (int)ja;
    }
    if (ib >= 0) {
      const int pos = tot_a + base_b + __popcll(bb & lt);
      s_pairs[2 * pos] = ib;
      s_pairs[2 * pos + 1] = (int)jb;
    }
    const int total = tot_a + tot_b;
    __syncthreads();
    const int nsteps = (total + 31) >> 5;
    if (nsteps == 0) continue;                               // workgroup-uniform
    f32x4 ra[LA], rbv[LB];
    auto issue = [&](int t) {
#pragma unroll
      for (int u = 0; u < LA; ++u) {
        const int e = tid + 256 * u, row = e / (CI / 4), c4 = e % (CI / 4);
        const int pr = t * 32 + row;
        const int idx = pr < total ? s_pairs[2 * pr] : -1;
        ra[u] = idx >= 0 ? *reinterpret_cast<const f32x4*>(in + (int64_t)idx * ldi + ci0 + 4 * c4) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < LB; ++u) {
        const int e = tid + 256 * u, row = e / (CO / 4), c4 = e % (CO / 4);
        const int pr = t * 32 + row;
        const int j = pr < total ? s_pairs[2 * pr + 1] : -1;
        rbv[u] = j >= 0 ? *reinterpret_cast<const f32x4*>(dout + (int64_t)j * ldo + co0 + 4 * c4) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    };
    auto stash = [&](int buf) {
      char* st = smem + buf * STAGE;
#pragma unroll
      for (int u = 0; u < LA; ++u) {
        const int e = tid + 256 * u, row = e / (CI / 4), c4 = e % (CI / 4);
        u32x2 h, m, l;
        split3x4(ra[u], h, m, l);
        const int off = img_off<PA>(row, c4 >> 1) + 8 * (c4 & 1);
        *reinterpret_cast<u32x2*>(st + off) = h;
        *reinterpret_cast<u32x2*>(st + IMG_A + off) = m;
        *reinterpret_cast<u32x2*>(st + 2 * IMG_A + off) = l;
      }
#pragma unroll
      for (int u = 0; u < LB; ++u) {
        const int e = tid + 256 * u, row = e / (CO / 4), c4 = e % (CO / 4);
        u32x2 h, m, l;
        split3x4(rbv[u], h, m, l);
        const int off = img_off<PB>(row, c4 >> 1) + 8 * (c4 & 1);
        *reinterpret_cast<u32x2*>(st + 3 * IMG_A + off) = h;
        *reinterpret_cast<u32x2*>(st + 3 * IMG_A + IMG_B + off) = m;
        *reinterpret_cast<u32x2*>(st + 3 * IMG_A + 2 * IMG_B + off) = l;
      }
    };
    issue(0);
    stash(0);
    __syncthreads();
    for (int t = 0; t < nsteps; ++t) {
      const bool more = t + 1 < nsteps;
      if (more) issue(t + 1);
      const char* st = smem + (t & 1) * STAGE;
      bf16x8 af[MT][3], bfr[NT][3];
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) af[a][pc] = tr_frag<PA>(st + pc * IMG_A, (wm * (CI / 2) + 16 * a) / 8, lane);
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
          bfr[b][pc] = tr_frag<PB>(st + 3 * IMG_A + pc * IMG_B, (wn * (CO / 2) + 16 * b) / 8, lane);
      // 6 of the 9 cross terms, small first; term-major over the tiles so that no MFMA waits on its predecessor
#define WG_TERM(PA_, PB_)                                                                                      \
  _Pragma("unroll") for (int a = 0; a < MT; ++a) _Pragma("unroll") for (int b = 0; b < NT; ++b) acc[a][b] =    \
      __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a][PA_], bfr[b][PB_], acc[a][b], 0, 0, 0);
      WG_TERM(2, 0)
      WG_TERM(0, 2)
      WG_TERM(1, 1)
      WG_TERM(1, 0)
      WG_TERM(0, 1)
      WG_TERM(0, 0)
#undef WG_TERM
      if (more) stash((t + 1) & 1);
      __syncthreads();
    }
  }
  // D[row = 4 (lane >> 4) + reg][col = lane & 15]: row = input channel, col = output channel
  float* dst = part + (int64_t)blockIdx.x * cin * cout;
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ci = ci0 + wm * (CI / 2) + 16 * a + 4 * (lane >> 4) + g, co = co0 + wn * (CO / 2) + 16 * b + (lane & 15);
        dst[(int64_t)ci * cout + co] = acc[a][b][g];
      }
}

// dW[k] = sum of the partials of offset k's work items in ascending order (the item numbering of k_wgrad_bf3)
__global__ void k_wgrad_reduce_work(const float* __restrict__ part, int K, int S_all, int S_c, int k_c, int64_t per_k,
                                    int64_t total, float* __restrict__ dw) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int k = (int)(t / per_k);
  const int64_t e = t - (int64_t)k * per_k;
  int w0, n;
  if (k_c >= 0 && k == k_c) {
    w0 = (K - 1) * S_all, n = S_c;
  } else {
    w0 = ((k_c >= 0 && k > k_c) ? k - 1 : k) * S_all, n = S_all;
  }
  float s = 0.f;
  for (int w = w0; w < w0 + n; ++w) s += part[(int64_t)w * per_k + e];
  dw[t] = s;
}

__global__ void k_wgrad_reduce(const float* __restrict__ part, int nchunk, int64_t per_chunk, float* __restrict__ dw) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= per_chunk) return;
  float s = 0.f;
  for (int c = 0; c < nchunk; ++c) s += part[(int64_t)c * per_chunk + t];
  dw[t] = s;
}

}  // namespace

static bool wgrad_bf3_ok(int32_t cin, int32_t cout) {
  static const int on = env_int("APR_WGRAD_BF3", 1);       // A/B switch: 0 = the fp32-MFMA kernel for every shape
  return on && cin % 32 == 0 && cout % 32 == 0;
}

// splits per offset: enough work items for ~2 workgroups per CU, never more than there are 512-row blocks
static void wgrad_splits(int64_t n_out, int32_t K, int32_t cin, int32_t cout, int same_level, int& CI, int& CO, int& S_all,
                         int& S_c, int& k_c) {
  CI = cin % 64 == 0 ? 64 : 32;
  CO = cout % 128 == 0 ? 128 : (cout % 64 == 0 ? 64 : 32);
  const int64_t tiles = (int64_t)(cin / CI) * (cout / CO);
  const int64_t nrb = cdiv64(n_out, kRowBlock);
  int64_t S = cdiv64(512, (int64_t)K * tiles);
  if (S > nrb) S = nrb;
  if (S < 1) S = 1;
  S_all = (int)S;
  k_c = (same_level && K > 1 && (K & 1)) ? K / 2 : -1;
  S_c = (int)(k_c >= 0 ? (4 * S > nrb ? nrb : 4 * S) : S);
}

static int64_t wgrad_items(int32_t K, int S_all, int S_c, int k_c) {
  return k_c >= 0 ? (int64_t)(K - 1) * S_all + S_c : (int64_t)K * S_all;
}

APR_API size_t apr_spconv_wgrad_scratch_bytes(int64_t n_out, int32_t K, int32_t cin, int32_t cout) {
  const int64_t nchunk = cdiv64(n_out > 0 ? n_out : 1, kChunkRows);
  size_t need = (size_t)nchunk * K * cin * cout * 4 + 256;
  if (wgrad_bf3_ok(cin, cout)) {
    int CI, CO, S_all, S_c, k_c;
    wgrad_splits(n_out > 0 ? n_out : 1, K, cin, cout, 1, CI, CO, S_all, S_c, k_c);      // same_level = 1: the larger of the two
    const size_t need3 = (size_t)wgrad_items(K, S_all, S_c, k_c) * cin * cout * 4 + 256;
    if (need3 > need) need = need3;                         // (unaligned rows fall back to the fp32 kernel at run time)
  }
  return need;
}

template <int CI, int CO>
static int launch_wgrad_bf3(const float* in, int64_t ldi, const float* dout, int64_t ldo, const int32_t* nbr, int64_t n_out,
                            int32_t K, int32_t cin, int32_t cout, int S_all, int S_c, int k_c, float* part, hipStream_t st) {
  constexpr int LDS = 2 * 3 * (32 * CI * 2 + 32 * CO * 2) + kRowBlock * 8;
  static bool attr[16] = {};
  int devi = 0;
  APR_HIP(hipGetDevice(&devi));
  if (devi < 16 && !attr[devi]) {
    APR_HIP(hipFuncSetAttribute((const void*)k_wgrad_bf3<CI, CO>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr[devi] = true;
  }
  const unsigned items = (unsigned)wgrad_items(K, S_all, S_c, k_c);
  hipLaunchKernelGGL((k_wgrad_bf3<CI, CO>), dim3(items, (unsigned)((cin / CI) * (cout / CO))), dim3(256), LDS, st, in, ldi, dout,
                     ldo, nbr, n_out, K, cin, cout, S_all, S_c, k_c, part);
  return APR_OK;
}

static int spconv_wgrad_impl(const float* in, int64_t ldi, const float* dout, int64_t ldo, const int32_t* nbr,
                             int64_t n_out, int32_t K, int32_t cin, int32_t cout, int same_level, float* dw, void* scratch,
                             size_t scratch_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APR_CHECK_ARG(n_out > 0 && K >= 1 && cin >= 1 && cout >= 1, "apr_spconv_wgrad: bad shape");
  APR_CHECK_ARG(nbr != nullptr || K == 1, "apr_spconv_wgrad: nbr may only be NULL for the identity map (K = 1)");
  APR_CHECK_ARG(scratch_bytes >= apr_spconv_wgrad_scratch_bytes(n_out, K, cin, cout), "apr_spconv_wgrad: scratch too small");
  float* part = (float*)(((uintptr_t)scratch + 255) & ~(uintptr_t)255);
  const int64_t per_chunk = (int64_t)K * cin * cout;
  if (wgrad_bf3_ok(cin, cout) && ldi % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)in & 15) == 0 && ((uintptr_t)dout & 15) == 0) {
    int CI, CO, S_all, S_c, k_c;
    wgrad_splits(n_out, K, cin, cout, same_level, CI, CO, S_all, S_c, k_c);
    int rc;
#define WG_CASE(A, B) \
  if (CI == A && CO == B) rc = launch_wgrad_bf3<A, B>(in, ldi, dout, ldo, nbr, n_out, K, cin, cout, S_all, S_c, k_c, part, st); else
    WG_CASE(32, 32) WG_CASE(32, 64) WG_CASE(32, 128) WG_CASE(64, 32) WG_CASE(64, 64) WG_CASE(64, 128) rc = APR_EINVAL;
#undef WG_CASE
    if (rc != APR_OK) return rc;
    hipLaunchKernelGGL(k_wgrad_reduce_work, dim3((unsigned)cdiv64(per_chunk, 256)), dim3(256), 0, st, part, K, S_all, S_c, k_c,
                       (int64_t)cin * cout, per_chunk, dw);
    APR_LAUNCH_CHECK();
    return APR_OK;
  }
  const int64_t nchunk = cdiv64(n_out, kChunkRows);
  APR_CHECK_ARG(nchunk < 65536 * 32 && K <= 65535, "apr_spconv_wgrad: too many rows / offsets for one launch");
  APR_CHECK_ARG(scratch_bytes >= (size_t)nchunk * K * cin * cout * 4 + 256, "apr_spconv_wgrad: scratch too small");
  const unsigned nz = (unsigned)(((cin + 63) / 64) * ((cout + 63) / 64));
  hipLaunchKernelGGL(k_wgrad_partial, dim3((unsigned)nchunk, (unsigned)K, nz), dim3(256), 0, st, in, ldi, dout, ldo,
                     nbr, n_out, K, cin, cout, part);
  hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)cdiv64(per_chunk, 256)), dim3(256), 0, st, part, (int)nchunk,
                     per_chunk, dw);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API int apr_spconv_wgrad(const float* in, int64_t ldi, const float* dout, int64_t ldo, const int32_t* nbr,
                             int64_t n_out, int32_t K, int32_t cin, int32_t cout, float* dw, void* scratch,
                             size_t scratch_bytes, void* stream) {
  return spconv_wgrad_impl(in, ldi, dout, ldo, nbr, n_out, K, cin, cout, 0, dw, scratch, scratch_bytes, stream);
}

// The same with the caller's knowledge that nbr is a same-level map (its centre column holds every row): the centre offset
// gets four times the splits of the others.  Same result up to the summation order of the partials.
APR_API int apr_spconv_wgrad_same_level(const float* in, int64_t ldi, const float* dout, int64_t ldo, const int32_t* nbr,
                                        int64_t n_out, int32_t K, int32_t cin, int32_t cout, float* dw, void* scratch,
                                        size_t scratch_bytes, void* stream) {
  return spconv_wgrad_impl(in, ldi, dout, ldo, nbr, n_out, K, cin, cout, 1, dw, scratch, scratch_bytes, stream);
}
