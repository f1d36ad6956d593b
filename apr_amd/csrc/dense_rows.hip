// K = 1 layers with FEW input channels over MANY rows: conv1_tr and `final` of the encoders
// (`FCGF_APR/model/resunet.py:126-140`: 96 -> 64 and 64 -> 32 in ResUNetBN2C, 160 -> 128 and 128 -> 128 in ResUNetFatBN, on
// the finest level's ~190 k rows per 12 frames) and the same operator in the training step.
//
// These are HBM streams (4 (cin + cout) bytes per row, 2 cin cout FLOP: 40-60 FLOP/B against a ridge of ~50 on the
// bf16-split pipe), and k_dense_gemm_bf3's tiling -- 64 G rows x 64 columns per workgroup, the weight slice staged per
// workgroup and per 64-channel chunk -- is built for long contractions: with two or three chunks per workgroup the prologue
// (first weight chunk, first rows) and the epilogue are most of its life, the rows are read once per 64-column block, and a
// width that is not a multiple of 64 (160, 96, 32 output columns) falls back to the sparse tile kernel (148 us for
// 160 -> 128 at 189 k rows, 5.4 x its HBM time).
//
// Here the WHOLE weight image lives in LDS for the life of the workgroup (cin x cout x 6 B: 12-147 KB; one workgroup of 8
// waves per CU) and the rows stream past it:
//   * a wave owns 32 rows (two 16-row groups) of a 256-row tile and ALL output columns: every row is read once, the row
//     normalisation of the last layer (`normalize_feature`, resunet.py:139-142) is a few adds in the epilogue;
//   * grid-stride over the tiles; the rows of the wave's NEXT tile are loaded into the very registers the current tile's
//     rows were split out of, step by step as they are consumed: a whole tile of MFMAs (~3-8 k cycles) hides the HBM
//     latency with ONE register set, no barrier anywhere after the weights have landed;
//   * per 32-channel step and 16-column block: 3 conflict-free ds_read_b128 (the h / m / l planes of the packed image,
//     apr_spconv_pack_weights_bf3) feed 12 MFMAs (6 products x 2 row groups) -- LDS at half of its rate when the matrix
//     pipe is full;
//   * same products in the same order as k_dense_gemm_bf3 (l.h, h.l, m.m, m.h, h.m, h.h per step, steps ascending): the
//     same bits where both kernels take the shape (tested).
#include "common.h"
#include "bf3.h"
#include <mutex>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRowsWaves = 8;                 // waves per workgroup
constexpr int kRowsTile = kRowsWaves * 32;    // rows per workgroup and tile

template <int NSTEP, int NC>                  // cin / 32 (2..6), cout / 16 (2, 4 or 8)
__global__ __launch_bounds__(64 * kRowsWaves, 1) void k_dense_rows_bf3(
    const float* __restrict__ in, int64_t ldi, int M, const unsigned char* __restrict__ wp3,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ residual, int64_t ldr,
    int relu, int l2norm, float* __restrict__ out, int64_t ldo, int ntile) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_w[];   // [64-col block][plane 3][step][col 64][quad 4][16 B]
  constexpr int G = 2;
  constexpr int NB64 = (NC + 3) / 4;
  constexpr int kPlane = NSTEP * 4096;
  constexpr int kImage = NB64 * 3 * kPlane;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, q = lane >> 4;
  const int frag_off = (r16 * 4 + ((r16 & 8) ? (q ^ 3) : q)) * 16;

  // this lane's rows of tile t (clamped: rows past M are computed on row M - 1 and not stored)
  int t = blockIdx.x;
  const float* arow[G];
#pragma unroll
  for (int gi = 0; gi < G; ++gi) {
    const int r = t * kRowsTile + wave * 32 + gi * 16 + r16;
    arow[gi] = in + (int64_t)(r < M ? r : M - 1) * ldi + q * 8;
  }
  f32x4 abuf[G][NSTEP][2];
#pragma unroll
  for (int s = 0; s < NSTEP; ++s)
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
      abuf[gi][s][0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(arow[gi] + s * 32));
      abuf[gi][s][1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(arow[gi] + s * 32 + 4));
    }
  // the weight image: a straight copy, 1 KB per wave and instruction (LDS-DMA: wave-uniform LDS base + lane * 16)
  for (int c = wave; c < kImage / 1024; c += kRowsWaves)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wp3 + (int64_t)c * 1024 + lane * 16),
                                     (__attribute__((address_space(3))) void*)(s_w + c * 1024), 16, 0, 0);
  // scale / shift behind the image (1 / 0 where absent: x * 1 + 0 = x): the epilogue reads them from LDS, so no global load
  // queues behind the next tile's rows in the in-order vmcnt
  float* s_aff = reinterpret_cast<float*>(s_w + kImage);      // [2][NC * 16]
  if (tid < NC * 16) {
    s_aff[tid] = scale ? scale[tid] : 1.f;
    s_aff[NC * 16 + tid] = shift ? shift[tid] : 0.f;
  }
  __syncthreads();

  for (; t < ntile; t += gridDim.x) {
    const int tn = t + gridDim.x;
    const bool more = tn < ntile;
    const float* nrow[G];
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
      const int64_t r = (int64_t)tn * kRowsTile + wave * 32 + gi * 16 + r16;
      nrow[gi] = in + (r < M ? r : (int64_t)M - 1) * ldi + q * 8;
    }
    f32x4 acc[G][NC];
#pragma unroll
    for (int gi = 0; gi < G; ++gi)
#pragma unroll
      for (int cb = 0; cb < NC; ++cb) acc[gi][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // Nothing writes the weight image, scale or shift inside this loop, so the compiler would hoist every fragment read and
    // every epilogue constant out of it (hundreds of registers, spilled).  The offsets go through an empty asm per tile:
    // the reads stay where they are used.
    int woff = frag_off, coff = q * 4;
    asm volatile("" : "+v"(woff), "+v"(coff));

#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      bf16x8 ah[G], am[G], al[G];
#pragma unroll
      for (int gi = 0; gi < G; ++gi) apr_split3(abuf[gi][s][0], abuf[gi][s][1], ah[gi], am[gi], al[gi]);
      if (more) {      // the next tile's rows of this step, into the registers just consumed
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
          abuf[gi][s][0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(nrow[gi] + s * 32));
          abuf[gi][s][1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(nrow[gi] + s * 32 + 4));
        }
      }
#pragma unroll
      for (int cb = 0; cb < NC; ++cb) {
        const unsigned char* wb = s_w + (cb >> 2) * (3 * kPlane) + s * 4096 + (cb & 3) * 1024 + woff;
        const bf16x8 wh = *reinterpret_cast<const bf16x8*>(wb);
        const bf16x8 wm = *reinterpret_cast<const bf16x8*>(wb + kPlane);
        const bf16x8 wl = *reinterpret_cast<const bf16x8*>(wb + 2 * kPlane);
        // the six products of a step in k_dense_gemm_bf3's order (small terms first), the two row groups alternating so that
        // no MFMA waits for its predecessor's result
#define APR_ROWS_MFMA(W, A)                                                                                          \
  _Pragma("unroll") for (int gi = 0; gi < G; ++gi)                                                                   \
    acc[gi][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W, A[gi], acc[gi][cb], 0, 0, 0);
        APR_ROWS_MFMA(wl, ah) APR_ROWS_MFMA(wh, al) APR_ROWS_MFMA(wm, am)
        APR_ROWS_MFMA(wm, ah) APR_ROWS_MFMA(wh, am) APR_ROWS_MFMA(wh, ah)
#undef APR_ROWS_MFMA
      }
    }

    // epilogue: lane (r16, q) holds columns cb * 16 + q * 4 .. + 3 of row r16 for every cb
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
      const int r = t * kRowsTile + wave * 32 + gi * 16 + r16;
      const bool live = r < M;
      f32x4 v[NC];
      if (residual) {      // wave-uniform; all loads of the row issued together (clamped row: always a valid address)
        const float* rr = residual + (int64_t)(live ? r : M - 1) * ldr + coff;
#pragma unroll
        for (int cb = 0; cb < NC; ++cb) v[cb] = *reinterpret_cast<const f32x4*>(rr + cb * 16);
      }
      {
#pragma clang fp contract(off)      // k_dense_gemm_bf3 multiplies, then adds (two roundings): keep its bits
#pragma unroll
        for (int cb = 0; cb < NC; ++cb) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(s_aff + cb * 16 + coff);
          const f32x4 sh = *reinterpret_cast<const f32x4*>(s_aff + NC * 16 + cb * 16 + coff);
          f32x4 a = acc[gi][cb] * sc;
          a = a + sh;
          if (residual) a = a + v[cb];
          if (relu) {
            a[0] = fmaxf(a[0], 0.f); a[1] = fmaxf(a[1], 0.f); a[2] = fmaxf(a[2], 0.f); a[3] = fmaxf(a[3], 0.f);
          }
          v[cb] = a;
        }
      }
      if (l2norm) {
        // row / |row|_2 with k_l2_normalize's summation tree (norm.hip: lane = channel mod 64, fmaf over the lane's
        // channels, then channel distances 32 .. 1): distance 64 is the fmaf chain (cout 128 only), 32 and 16 are column
        // blocks of this lane, 8 and 4 the lanes q ^ 2 and q ^ 1, 2 and 1 the four values of a lane -- the same bits as the
        // separate kernel
        constexpr int NS = NC >= 8 ? 4 : NC;
        f32x4 sq[NS];
#pragma unroll
        for (int cb = 0; cb < NS; ++cb)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            sq[cb][e] = fmaf(v[cb][e], v[cb][e], 0.f);
            if (NC >= 8) sq[cb][e] = fmaf(v[cb + 4][e], v[cb + 4][e], sq[cb][e]);
          }
#pragma unroll
        for (int d = NS / 2; d >= 1; d >>= 1)
#pragma unroll
          for (int cb = 0; cb < d; ++cb) sq[cb] += sq[cb + d];
        f32x4 s = sq[0];
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += __shfl_xor(s[e], 32);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += __shfl_xor(s[e], 16);
        const float nrm = sqrtf((s[0] + s[2]) + (s[1] + s[3]));
#pragma unroll
        for (int cb = 0; cb < NC; ++cb) {
          v[cb][0] = v[cb][0] / nrm; v[cb][1] = v[cb][1] / nrm; v[cb][2] = v[cb][2] / nrm; v[cb][3] = v[cb][3] / nrm;
        }
      }
      if (live) {
#pragma unroll
        for (int cb = 0; cb < NC; ++cb)
          *reinterpret_cast<f32x4*>(out + (int64_t)r * ldo + cb * 16 + q * 4) = v[cb];
      }
    }
  }
}

template <int NSTEP, int NC>
int launch_rows(const float* in, int64_t ldi, int64_t M, const void* w_bf3, const float* scale, const float* shift,
                const float* residual, int64_t ldr, int relu, int l2norm, float* out, int64_t ldo, hipStream_t st) {
  constexpr int kImage = ((NC + 3) / 4) * 3 * NSTEP * 4096;
  constexpr int kAffBytes = 2 * NC * 16 * 4;      // scale, shift
  static_assert(kImage + kAffBytes <= 160 * 1024, "weight image does not fit the LDS");
  auto kern = k_dense_rows_bf3<NSTEP, NC>;
  {      // > 64 KB of dynamic LDS needs the opt-in, once per kernel and device
    static std::mutex s_mu;
    static bool s_attr[64] = {};
    int dev = 0;
    APR_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(s_mu);
    if (dev >= 0 && dev < 64 && !s_attr[dev]) {
      APR_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kImage + kAffBytes));
      s_attr[dev] = true;
    }
  }
  const int ntile = (int)cdiv64(M, kRowsTile);
  static const int s_grid = env_int("APR_DENSE_ROWS_GRID", 256);      // one workgroup per CU; more only splits the tail finer
  const int grid = ntile < s_grid ? ntile : s_grid;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * kRowsWaves), kImage + kAffBytes, st, in, ldi, (int)M,
                     (const unsigned char*)w_bf3, scale, shift, residual, ldr, relu, l2norm, out, ldo, ntile);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

}  // namespace

// The shapes k_dense_rows_bf3 is instantiated for (the K = 1 layers of the encoders and their padded training widths).
APR_API int32_t apr_dense_rows_bf3_ok(int32_t cin, int32_t cout) {
  if (cin % 32 != 0 || cin < 64 || cin > 192) return 0;
  if (cout != 32 && cout != 64 && cout != 128) return 0;
  return (((cout + 63) / 64) * 3 * (cin / 32) * 4096 + 8 * cout <= 160 * 1024) ? 1 : 0;
}

// Whether the K = 1 dispatchers (apr_spconv_fwd_batch, apr_resunet_encode) send a layer here: an instantiated shape and enough
// rows for one 256-row tile per CU or so (below that k_dense_gemm_bf3's finer tiles fill the card better).
bool apr_internal_dense_rows_route(int64_t M, int32_t cin, int32_t cout) {
  static const int s_min = env_int("APR_DENSE_ROWS_MIN", 32768);      // 0: never
  return s_min > 0 && M >= s_min && apr_dense_rows_bf3_ok(cin, cout);
}

APR_API int32_t apr_dense_rows_bf3_route(int64_t M, int32_t cin, int32_t cout) {
  return apr_internal_dense_rows_route(M, cin, cout) ? 1 : 0;
}

// out = act((in @ W) * scale + shift + residual), optionally followed by the row normalisation out[j] /= |out[j]|_2, for an
// identity map with cin in {64, 96, 128, 160, 192} and cout in {32, 64, 128}.  w_bf3: apr_spconv_pack_weights_bf3(w, 1, cin, cout)
// (columns padded to a multiple of 64 inside the image).  Rows of in / out / residual 16-byte aligned.
APR_API int apr_dense_rows_bf3(const float* in, int64_t ldi, int64_t M, int32_t cin, int32_t cout, const void* w_bf3,
                               const float* scale, const float* shift, const float* residual, int64_t ldr, int32_t relu,
                               int32_t l2norm, float* out, int64_t ldo, void* stream) {
  APR_CHECK_ARG(in && out && w_bf3 && M > 0 && M < (1ll << 31) - kRowsTile && apr_dense_rows_bf3_ok(cin, cout),
                "apr_dense_rows_bf3: needs M > 0, cin in 64..192 step 32, cout 32 / 64 / 128 (got %d -> %d)", cin, cout);
  APR_CHECK_ARG(ldi >= cin && ldo >= cout && ldi % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)in & 15) == 0 &&
                    ((uintptr_t)out & 15) == 0,
                "apr_dense_rows_bf3: rows of in / out must be 16-byte aligned");
  APR_CHECK_ARG(!residual || (ldr >= cout && ldr % 4 == 0 && ((uintptr_t)residual & 15) == 0),
                "apr_dense_rows_bf3: residual rows must be 16-byte aligned");
  APR_CHECK_ARG((!scale || ((uintptr_t)scale & 15) == 0) && (!shift || ((uintptr_t)shift & 15) == 0),
                "apr_dense_rows_bf3: scale / shift must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
#define APR_ROWS_CASE(NS, NCB)                                                                                       \
  if (cin == NS * 32 && cout == NCB * 16)                                                                            \
    return launch_rows<NS, NCB>(in, ldi, M, w_bf3, scale, shift, residual, ldr, relu, l2norm, out, ldo, st);
  APR_ROWS_CASE(2, 2) APR_ROWS_CASE(2, 4) APR_ROWS_CASE(2, 8)
  APR_ROWS_CASE(3, 2) APR_ROWS_CASE(3, 4) APR_ROWS_CASE(3, 8)
  APR_ROWS_CASE(4, 2) APR_ROWS_CASE(4, 4) APR_ROWS_CASE(4, 8)
  APR_ROWS_CASE(5, 2) APR_ROWS_CASE(5, 4) APR_ROWS_CASE(5, 8)
  APR_ROWS_CASE(6, 2) APR_ROWS_CASE(6, 4) APR_ROWS_CASE(6, 8)
#undef APR_ROWS_CASE
  apr_set_error("apr_dense_rows_bf3: no instantiation for %d -> %d", cin, cout);
  return APR_EINVAL;
}
