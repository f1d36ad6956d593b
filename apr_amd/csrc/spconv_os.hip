// Output-stationary sparse convolution for the 64-channel levels (FCGF_APR/model/resunet.py:31-140,
// model/residual_block.py:23-33): the layers whose weight-stationary form (spconv_ws.hip) is bound by the round trip of
// its product rows through the fabric (round-2 PMC: 1.65x the algorithmic bytes).  Here NO product row ever leaves the CU:
//
//   * the kernel map is cut into TILES of R consecutive output rows (R <= 432, chosen on the host so that the tiles fill
//     the 256 CUs in whole rounds); k_os_build turns a tile's [R, K] slab of the table into K compact pair lists
//     (input row | local output row << 23), in row order, at fixed positions: no atomics, the same bytes every run;
//   * k_os_conv: one 512-thread workgroup per (tile, 64 output channels).  The tile's fp32 accumulators [R, 64] live in
//     LDS for the whole kernel (XOR-swizzled 16-B chunks), the bf16-split weight slice W[k][:, 64] (cin * 384 B) of the
//     CURRENT offset sits beside them and the NEXT offset's slice is in flight by LDS-DMA (global_load_lds, two
//     buffers).  The 8 waves deal the 16-pair groups of the offset among themselves: a wave gathers its 16 input rows
//     straight into MFMA operand registers (prefetched one group ahead, across offsets), splits them into three bf16
//     pieces (bf3.h), reads the 16 accumulator rows from LDS as the MFMA C operand, runs the 48 x cin/64
//     v_mfma_f32_16x16x32_bf16 of the group (D^T = W^T A^T: a lane ends with 4 consecutive channels of ONE pair) and
//     writes the rows back.  Rows of one offset are distinct, offsets are separated by ONE workgroup barrier, sums run
//     in ascending offset order: bitwise reproducible, no float atomics.
//   * the epilogue (scale / shift / residual / ReLU) streams the tile out of LDS with 256-B coalesced row stores.
// Traffic per launch: the gathered rows (L2 / Infinity-Cache resident), the weight slices (L2 resident, K x cin x 384 B per
// tile) and the output rows, once.
#include <mutex>

#include "common.h"
#include "bf3.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWaves = 8;
constexpr int kThreads = 64 * kWaves;
constexpr int kInBits = 23;                      // input row in the low 23 bits of a pair word, local output row above
constexpr unsigned kInMask = (1u << kInBits) - 1u;

struct OsViews {
  int* cnt;         // [ntiles][32]  pairs of (tile, offset)
  unsigned* pair;   // [ntiles][K][R]
};

__host__ __device__ inline size_t os_align256(size_t x) { return (x + 255) & ~(size_t)255; }

__host__ __device__ inline OsViews os_carve(void* blob, int64_t ntiles) {
  OsViews v;
  v.cnt = (int*)blob;
  v.pair = (unsigned*)((char*)blob + os_align256((size_t)ntiles * 32 * 4));
  return v;
}

// One workgroup per tile: the [rows, K] slab through LDS (coalesced), wave w ranks offsets w, w + 8, ... by ballot +
// popcount over 64-row chunks and writes the compact list of (tile, offset) in row order.
__global__ __launch_bounds__(kThreads) void k_os_build(const int* __restrict__ nbr, int n_out, int K, int R, OsViews v) {
  extern __shared__ int s_nbr[];   // [R][K]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x;
  const int row0 = tile * R;
  const int rows = min(R, n_out - row0);
  const int total = rows * K;
  const int* src = nbr + (int64_t)row0 * K;
  for (int e0 = threadIdx.x; e0 < total; e0 += 6 * kThreads) {
    int t[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int e = e0 + u * kThreads;
      t[u] = (e < total) ? src[e] : -1;
    }
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int e = e0 + u * kThreads;
      if (e < total) s_nbr[e] = t[u];
    }
  }
  __syncthreads();
  for (int k = wave; k < 32; k += kWaves) {
    int run = 0;
    if (k < K) {
      unsigned* dst = v.pair + ((int64_t)tile * K + k) * R;
      for (int c = 0; c < rows; c += 64) {
        const int r = c + lane;
        const int idx = (r < rows) ? s_nbr[r * K + k] : -1;
        const unsigned long long m = __ballot(idx >= 0);
        if (idx >= 0) dst[run + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned)idx | ((unsigned)r << kInBits);
        run += __popcll(m);
      }
    }
    if (lane == 0) v.cnt[tile * 32 + k] = run;
  }
}

// workgroup-uniform position in the item sequence: (offset k, slot g); k = 64: past the end
struct OsIt {
  int k, g;
};

template <int NCH>   // cin / 64: 1 or 2
__global__ __launch_bounds__(kThreads, 2) void k_os_conv(const float* __restrict__ in, int64_t ldi, OsViews v, int n_out,
                                                         int R, int K, int cout, const unsigned char* __restrict__ wp3,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ residual, int64_t ldr, int relu,
                                                         float* __restrict__ out, int64_t ldo) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  constexpr int nstep = NCH * 2;                      // 32-channel steps
  constexpr int plane_bytes = nstep * 4096;           // one split plane of a slice: [step][col 64][quad 4][8 bf16]
  constexpr int slice_bytes = 3 * plane_bytes;        // 24 KB x NCH
  constexpr int npiece = slice_bytes / 1024 / kWaves; // 1-KB LDS-DMA pieces per wave and slice
  unsigned char* const s_w = s_raw;                                  // two slices
  unsigned char* const s_acc = s_raw + 2 * slice_bytes;              // [R][16 chunks of 16 B], chunk c at c ^ (row & 15)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, q = lane >> 4;
  const int tile = blockIdx.x, cblk = blockIdx.y;
  const int row0 = tile * R;
  const int rows = min(R, n_out - row0);
  const int cnt_l = (lane < K) ? v.cnt[tile * 32 + lane] : 0;
  const unsigned long long live = __ballot(cnt_l > 0);
  const unsigned* const lists = v.pair + (int64_t)tile * K * R;
  const unsigned ldi32 = (unsigned)ldi;
  const int frag_off = (r16 * 4 + ((r16 & 8) ? (q ^ 3) : q)) * 16;

  for (int o = tid * 16; o < R * 256; o += kThreads * 16) *reinterpret_cast<f32x4*>(s_acc + o) = (f32x4){0.f, 0.f, 0.f, 0.f};

  // The item sequence is the SAME for all 8 waves: (offset k, slot s), s < ceil(groups(k) / 8); in slot s wave w owns group
  // w + 8 s of the offset, or nothing (it still issues the item's loads, from a clamped address: every path through the
  // loop then carries the same number of vector-memory operations and the waits are counted, never vmcnt(0)).
  auto cnt_of = [&](int k) { return __builtin_amdgcn_readlane(cnt_l, k); };
  auto next_live = [&](int k) {   // first live offset above k, or 64
    const unsigned long long m = (k >= 63) ? 0ull : (live & ~((2ull << k) - 1ull));
    return m ? (int)__builtin_ctzll(m) : 64;
  };
  auto advance = [&](OsIt it) {
    if (it.k >= 64) return it;
    const int ng = (cnt_of(it.k) + 15) >> 4;
    if ((it.g + 1) * kWaves < ng) return OsIt{it.k, it.g + 1};
    return OsIt{next_live(it.k), 0};
  };
  const int k_first = live ? (int)__builtin_ctzll(live) : 64;
  // pair word of item `it` for this lane; lanes past the list's end (and items past the end) read a valid word and are
  // masked at the accumulator write-back
  auto word_ptr = [&](OsIt it) {
    const int k = it.k < 64 ? it.k : k_first;
    const int slot = (wave + kWaves * it.g) * 16 + r16;
    return lists + (int64_t)k * R + ((it.k < 64 && slot < cnt_of(k)) ? slot : 0);
  };
  auto stage = [&](int k, int buf) {   // slice of offset k -> LDS buffer `buf`, this wave's pieces
    const unsigned char* src = wp3 + ((int64_t)k * (cout >> 6) + cblk) * slice_bytes + lane * 16;
#pragma unroll
    for (int u = 0; u < npiece; ++u) {
      const int piece = wave + kWaves * u;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 1024),
                                       (__attribute__((address_space(3))) void*)(s_w + buf * slice_bytes + piece * 1024),
                                       16, 0, 0);
    }
  };

  if (live) {
    stage(k_first, 0);
    OsIt it0 = OsIt{k_first, 0};
    OsIt it1 = advance(it0);
    unsigned pw0 = *word_ptr(it0);
    unsigned pw1 = *word_ptr(it1);
    f32x4 raw[4 * NCH];
    {
      const float* ab = in + (uint64_t)(pw0 & kInMask) * ldi32 + q * 8;
#pragma unroll
      for (int j = 0; j < 4 * NCH; ++j) raw[j] = *reinterpret_cast<const f32x4*>(ab + (j >> 2) * 64 + ((j >> 1) & 1) * 32 + (j & 1) * 4);
    }
    __syncthreads();   // accumulators zeroed, first slice landed (vmcnt(0))

    int buf = 0;
    while (it0.k < 64) {
      const int kcur = it0.k;
      const OsIt it2 = advance(it1);
      const bool last = it1.k != kcur;
      const int cnt_k = cnt_of(kcur);
      const bool real = (wave + kWaves * it0.g) * 16 < cnt_k;           // this wave has a group in this slot
      const unsigned char* const wbuf = s_w + buf * slice_bytes + frag_off;
      const int orow = (int)(pw0 >> kInBits);
      const bool valid = (wave + kWaves * it0.g) * 16 + r16 < cnt_k;
      unsigned char* const arow = s_acc + orow * 256;
      const int sw = orow & 15;
      bf16x8 ah[nstep], am[nstep], al[nstep];
      f32x4 acc[4];
      if (real) {
        // ---- operands of this item: split the gathered rows, accumulator rows as the MFMA C operand
#pragma unroll
        for (int s = 0; s < nstep; ++s) apr_split3(raw[2 * s], raw[2 * s + 1], ah[s], am[s], al[s]);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[cb] = *reinterpret_cast<const f32x4*>(arow + (((cb * 4 + q) ^ sw) << 4));
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- next offset's slice (first slot of the offset), next item's rows, the word of the item after next
      if (it0.g == 0) {
        const int knext = next_live(kcur);
        if (knext < 64) stage(knext, buf ^ 1);
      }
      {
        const float* ab = in + (uint64_t)(pw1 & kInMask) * ldi32 + q * 8;
#pragma unroll
        for (int j = 0; j < 4 * NCH; ++j)
          raw[j] = *reinterpret_cast<const f32x4*>(ab + (j >> 2) * 64 + ((j >> 1) & 1) * 32 + (j & 1) * 4);
      }
      const unsigned pw2 = *word_ptr(it2);
      __builtin_amdgcn_sched_barrier(0);
      if (real) {
        // ---- 48 x NCH MFMAs; W fragments of (step s, 16-column block cb) fetched two (s, cb) ahead
        bf16x8 wf[3][3];
#pragma unroll
        for (int i0 = 0; i0 < 2; ++i0)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            wf[i0][pl] = *reinterpret_cast<const bf16x8*>(wbuf + (i0 * 16) * 64 + pl * plane_bytes);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4 * nstep; ++i) {
          const int s = i >> 2, cb = i & 3;
          if (i + 2 < 4 * nstep) {
            const int s2 = (i + 2) >> 2, cb2 = (i + 2) & 3;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
              wf[(i + 2) % 3][pl] = *reinterpret_cast<const bf16x8*>(wbuf + (s2 * 64 + cb2 * 16) * 64 + pl * plane_bytes);
            __builtin_amdgcn_sched_barrier(0);
          }
          const bf16x8 wh = wf[i % 3][0], wm = wf[i % 3][1], wl = wf[i % 3][2];
          f32x4 t = acc[cb];
          t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, ah[s], t, 0, 0, 0);
          t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, al[s], t, 0, 0, 0);
          t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, am[s], t, 0, 0, 0);
          t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, ah[s], t, 0, 0, 0);
          t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, am[s], t, 0, 0, 0);
          t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ah[s], t, 0, 0, 0);
          acc[cb] = t;
          __builtin_amdgcn_sched_barrier(0);
        }
        if (valid) {
#pragma unroll
          for (int cb = 0; cb < 4; ++cb) *reinterpret_cast<f32x4*>(arow + (((cb * 4 + q) ^ sw) << 4)) = acc[cb];
        }
      }
      if (last) {
        // The wave's pieces of the next slice (issued in the offset's first slot, before that item's 4 * NCH + 1 prefetch
        // loads) must have landed before the barrier lets anyone read them; the prefetch loads may stay in flight.
        // Accumulator write-backs: lgkmcnt(0).  The barrier also orders this offset's updates before the next one's.
        if (NCH == 1) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        buf ^= 1;
      }
      it0 = it1;
      it1 = it2;
      pw0 = pw1;
      pw1 = pw2;
    }
  } else {
    __syncthreads();
  }

  // epilogue: lane -> logical chunk c of row r (stored at c ^ (r & 15)); 16 lanes = one 256-B output row segment
  const int col = cblk * 64 + (tid & 15) * 4;
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (scale) sc = *reinterpret_cast<const f32x4*>(scale + col);
  if (shift) sh = *reinterpret_cast<const f32x4*>(shift + col);
  for (int r = tid >> 4; r < rows; r += kThreads / 16) {
    f32x4 s = *reinterpret_cast<const f32x4*>(s_acc + r * 256 + ((((tid & 15)) ^ (r & 15)) << 4));
    s = s * sc + sh;
    const int64_t row = row0 + r;
    if (residual) s += *reinterpret_cast<const f32x4*>(residual + row * ldr + col);
    if (relu) {
      s[0] = fmaxf(s[0], 0.f); s[1] = fmaxf(s[1], 0.f); s[2] = fmaxf(s[2], 0.f); s[3] = fmaxf(s[3], 0.f);
    }
    *reinterpret_cast<f32x4*>(out + row * ldo + col) = s;
  }
}

inline int os_rmax(int cin) { return cin == 64 ? 432 : 240; }

}  // namespace

// Rows per tile for an [n_out]-row map feeding a cin -> cout layer: the smallest whole number m of rounds of 256
// workgroups (one per CU: the tile's accumulators + two weight slices take most of the 160 KB of LDS) whose tiles fit.
APR_API int32_t apr_spconv_os_tile_rows(int64_t n_out, int32_t cin, int32_t cout) {
  if (n_out <= 0 || (cin != 64 && cin != 128) || cout < 64 || cout % 64 != 0) return 0;
  static const int s_cus = env_int("APR_OS_CUS", 256);
  const int64_t ncol = cout / 64;
  const int rmax = os_rmax(cin);
  for (int64_t m = 1;; ++m) {
    int64_t r = cdiv64(n_out * ncol, s_cus * m);
    r = (r + 15) / 16 * 16;
    if (r < 64) r = 64;
    if (r <= rmax) return (int32_t)r;
  }
}

APR_API size_t apr_spconv_os_pairs_bytes(int64_t n_out, int32_t K, int32_t R) {
  if (n_out <= 0 || R <= 0 || K <= 0) return 0;
  const int64_t ntiles = cdiv64(n_out, R);
  return os_align256((size_t)ntiles * 32 * 4) + (size_t)ntiles * K * R * 4 + 256;
}

// nbr [n_out, K] (input rows < n_in < 2^23) -> per-tile pair lists in `blob` (apr_spconv_os_pairs_bytes)
APR_API int apr_spconv_os_pairs_build(const int32_t* nbr, int64_t n_out, int64_t n_in, int32_t K, int32_t R, void* blob,
                                      size_t blob_bytes, void* stream) {
  APR_CHECK_ARG(nbr && blob && n_out > 0 && n_out < (1ll << 31) / 32 && K >= 1 && K <= 27,
                "apr_spconv_os_pairs_build: needs 0 < n_out < 2^26 and 1 <= K <= 27");
  APR_CHECK_ARG(R >= 16 && R % 16 == 0 && R <= 448, "apr_spconv_os_pairs_build: tile rows must be a multiple of 16 in [16, 448]");
  APR_CHECK_ARG(n_in > 0 && n_in < (1ll << kInBits), "apr_spconv_os_pairs_build: needs n_in < 2^23 (pair words hold 23-bit input rows)");
  APR_CHECK_ARG(blob_bytes >= apr_spconv_os_pairs_bytes(n_out, K, R), "apr_spconv_os_pairs_build: blob too small");
  const int64_t ntiles = cdiv64(n_out, R);
  OsViews v = os_carve(blob, ntiles);
  hipLaunchKernelGGL(k_os_build, dim3((unsigned)ntiles), dim3(kThreads), (size_t)R * K * 4, (hipStream_t)stream, nbr,
                     (int)n_out, K, R, v);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

// out = act((sum_k in[nbr[., k]] @ W[k]) * scale + shift + residual) over the pair lists of apr_spconv_os_pairs_build;
// cin 64 or 128, cout % 64 == 0, w_bf3 from apr_spconv_pack_weights_bf3(w, K, cin, cout).
APR_API int apr_spconv_os_fwd(const float* in, int64_t ldi, const void* os_pairs, int64_t n_out, int32_t K, int32_t R,
                              int32_t cin, int32_t cout, const void* w_bf3, const float* scale, const float* shift,
                              const float* residual, int64_t ldr, int32_t relu, float* out, int64_t ldo, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APR_CHECK_ARG(in && os_pairs && w_bf3 && out && n_out > 0 && n_out < (1ll << 31) / 32 && K >= 1 && K <= 27,
                "apr_spconv_os_fwd: bad n_out / K / null argument");
  APR_CHECK_ARG((cin == 64 || cin == 128) && cout >= 64 && cout % 64 == 0, "apr_spconv_os_fwd: needs cin 64 or 128 and cout %% 64 == 0");
  APR_CHECK_ARG(R >= 16 && R % 16 == 0 && R <= os_rmax(cin), "apr_spconv_os_fwd: tile rows out of range for this cin");
  APR_CHECK_ARG(ldi > 0 && ldi < (1ll << 31) && ldi % 4 == 0 && ldo % 4 == 0 &&
                    ((((uintptr_t)in) | ((uintptr_t)out)) & 15) == 0,
                "apr_spconv_os_fwd: 16-byte aligned rows required");
  APR_CHECK_ARG(!residual || (ldr % 4 == 0 && (((uintptr_t)residual) & 15) == 0), "apr_spconv_os_fwd: residual alignment");
  APR_CHECK_ARG((!scale || ((uintptr_t)scale & 15) == 0) && (!shift || ((uintptr_t)shift & 15) == 0),
                "apr_spconv_os_fwd: scale / shift must be 16-byte aligned");
  {
    static std::mutex s_mu;
    static bool s_attr[64] = {};
    int dev = 0;
    APR_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(s_mu);
    if (dev >= 0 && dev < 64 && !s_attr[dev]) {
      APR_HIP(hipFuncSetAttribute((const void*)k_os_conv<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      APR_HIP(hipFuncSetAttribute((const void*)k_os_conv<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      s_attr[dev] = true;
    }
  }
  const int64_t ntiles = cdiv64(n_out, R);
  OsViews v = os_carve(const_cast<void*>(os_pairs), ntiles);
  const size_t lds = (size_t)2 * 3 * (cin / 32) * 4096 + (size_t)R * 256;
  const dim3 grid((unsigned)ntiles, (unsigned)(cout / 64));
  if (cin == 64)
    hipLaunchKernelGGL(k_os_conv<1>, grid, dim3(kThreads), lds, st, in, ldi, v, (int)n_out, R, K, cout,
                       (const unsigned char*)w_bf3, scale, shift, residual, ldr, relu, out, ldo);
  else
    hipLaunchKernelGGL(k_os_conv<2>, grid, dim3(kThreads), lds, st, in, ldi, v, (int)n_out, R, K, cout,
                       (const unsigned char*)w_bf3, scale, shift, residual, ldr, relu, out, ldo);
  APR_LAUNCH_CHECK();
  return APR_OK;
}
