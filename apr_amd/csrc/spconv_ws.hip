// Weight-stationary sparse convolution for maps with FEW pairs per output tile:
// strided / transposed convolutions and the deep, small levels of the UNet.
//
// The tile kernel (spconv.hip) re-reads W[k] for every (32-row tile, offset): with Cin*Cout >= 64*64 and
// only 2-8 pairs per (tile, offset) the weight stream from L2 (100-550 MB per launch) dominates.  Here the
// kernel map is first turned into per-offset pair lists (one-off per map, cached by the caller):
//     offset k owns the fixed region [k*n_out, k*n_out + cnt[k]) of pair_in[];  pair_id[row, k] = position or -1
// built by ONE kernel: a block's [256, K] slab of the table goes through LDS, wave64 ballot + popcount ranks the
// valid entries, one atomicAdd per (block, offset) reserves their range.  Pair positions therefore vary from
// run to run, the RESULT does not: each product row is computed independently of its position and summed per
// output row in fixed offset order.  Then
//   1. k_ws_gemm: a workgroup owns (offset k, 64*G consecutive pairs, 64 output channels); the weight slice
//      W[k][:, 64 cols] is staged ONCE per workgroup in LDS, each wave gathers its 16 pairs' input rows straight
//      into MFMA operand registers, accumulates over all Cin chunks in registers (v_mfma_f32_16x16x4_f32,
//      D^T form) and stores prod[p, :] with 16-B stores;
//   2. k_ws_reduce: out[j,:] = act((sum_k prod[pair_id[j,k],:]) * scale + shift + residual) in fixed k order.
// No float atomics, bitwise reproducible.  Extra HBM traffic: the product rows, written and read once.
#include <mutex>

#include "common.h"
#include "bf3.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kRows = 512;   // rows per block in the pair-list build (fewer, larger blocks: the K range counters are
                             // contended by every block, ~55 atomics per address on a 28 k-row map)
constexpr int kBuildWaves = 8;
constexpr int kMaxG = 64;  // <= 4096 pairs per unit (12 frames per call: 1.3 M pairs over 768 resident units need 27)

// The caller's counters (zero before apr_pairlist_build): the count of offset k lives at cnt[k * kCntStride], one
// counter per 256 B.  Packed into one 128-B line, the K x (#blocks) range reservations of a build all queue on ONE
// L2 channel (~10 ns each: 10 k atomics = the whole 103 us of a 189 k-row build); spread out, they proceed on K
// channels in parallel.  pairs of offset k sit at pair_in[k * n_out .. k * n_out + cnt)
constexpr int kCntStride = APR_PAIR_COUNTER_STRIDE;
struct PairHeader {
  int cnt[32 * kCntStride];
};

__host__ __device__ inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct PairViews {
  PairHeader* hdr;
  int* pair_in;   // [K, n_out] input row of each pair
  int* pair_id;   // [n_out, K] position of the pair (row, k) or -1
};

__host__ __device__ inline PairViews carve_pairs(int32_t* counters, void* blob, int64_t n_out, int K) {
  PairViews v;
  char* p = (char*)blob;
  v.hdr = (PairHeader*)counters;
  const size_t cap = (size_t)n_out * K;
  v.pair_in = (int*)p;
  p += align256(cap * 4);
  v.pair_id = (int*)p;
  return v;
}

// The block's [256, K] slab of the table goes through LDS once (coalesced); wave w then owns offsets
// w, w+4, ... : ballot + popcount over the 256 rows (row stride K ints: conflict-free for odd K), one atomicAdd
// reserves the block's range in the offset's region, ranks fill it; the id slab is written back coalesced.
__global__ __launch_bounds__(64 * kBuildWaves) void k_pairs_build(const int* __restrict__ nbr, int n_out, int K, PairViews v) {
  __shared__ int s_nbr[kRows * 27 + 64];   // K <= 27 on this path (larger K: see apr_pairlist_build)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = blockIdx.x * kRows;
  const int rows = min(kRows, n_out - row0);
  const int total = rows * K;
  const int* src = nbr + (int64_t)row0 * K;
  // 9 independent loads in flight per thread before the first LDS store (a plain copy loop serialises one L2
  // round trip per iteration)
  for (int e0 = threadIdx.x; e0 < total; e0 += 9 * 64 * kBuildWaves) {
    int t[9];
#pragma unroll
    for (int u = 0; u < 9; ++u) {
      const int e = e0 + u * 64 * kBuildWaves;
      t[u] = (e < total) ? src[e] : -1;
    }
#pragma unroll
    for (int u = 0; u < 9; ++u) {
      const int e = e0 + u * 64 * kBuildWaves;
      if (e < total) s_nbr[e] = t[u];
    }
  }
  __syncthreads();
  // phase 1: counts of this wave's offsets, all range reservations in flight together (one L2 round trip)
  int base[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int k = wave + kBuildWaves * s;
    base[s] = 0;
    if (k < K) {
      int cnt = 0;
#pragma unroll
      for (int c = 0; c < kRows / 64; ++c) {
        const int r = c * 64 + lane;
        cnt += __popcll(__ballot(r < rows && s_nbr[r * K + k] >= 0));
      }
      if (lane == 0 && cnt) base[s] = atomicAdd(&v.hdr->cnt[k * kCntStride], cnt);
    }
  }
  // phase 2: ranks -> positions
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int k = wave + kBuildWaves * s;
    if (k < K) {
      int run = __shfl(base[s], 0) + k * n_out;
#pragma unroll
      for (int c = 0; c < kRows / 64; ++c) {
        const int r = c * 64 + lane;
        const int idx = (r < rows) ? s_nbr[r * K + k] : -1;
        const unsigned long long m = __ballot(idx >= 0);
        if (r < rows) {
          int pos = -1;
          if (idx >= 0) {
            pos = run + __popcll(m & ((1ull << lane) - 1ull));
            // counters not cleared by the caller would push positions past the offset's region: never write there
            if (pos < (k + 1) * n_out) v.pair_in[pos] = idx; else pos = -1;
          }
          s_nbr[r * K + k] = pos;
        }
        run += __popcll(m);
      }
    }
  }
  __syncthreads();
  int* dst = v.pair_id + (int64_t)row0 * K;
  for (int e = threadIdx.x; e < total; e += 64 * kBuildWaves) dst[e] = s_nbr[e];
}

// Work unit = (offset k, block of 64*G consecutive pairs of k, 64 output channels); G is chosen ON THE DEVICE
// from the real pair count so that ~768 units exist, and a fixed-size grid strides over them (no host sync, no
// surplus workgroups).  The whole weight slice W[k][:, col0:col0+64] (cin*256 B, <= 128 KB) is staged in LDS
// once per unit (8 independent 16-B loads in flight per thread), then each wave walks 16-pair groups: gather
// rows -> registers (prefetched one 64-channel chunk ahead, across group boundaries), B fragments from LDS
// (conflict-free ds_read_b128), cin/64 * 64 MFMAs into 16 accumulator registers.
template <int NCH>   // cin / 64 when it is 1, 2 or 4; 0 = read it at run time
__global__ __launch_bounds__(256, 2) void k_ws_gemm(const float* __restrict__ in, int64_t ldi, PairViews v, int K,
                                                    int cin, int cout, const float* __restrict__ wp,
                                                    float* __restrict__ prod, int n_out, int target_units) {
  extern __shared__ __attribute__((aligned(16))) float s_w[];   // [g = cin/4][col 64][4]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the step loop below stays on the SALU
  const int r16 = lane & 15, q = lane >> 4;
  const int col0 = blockIdx.y * 64;
  const int cinG = cin >> 2;
  const int nchunk = NCH ? NCH : cin >> 6;   // cin % 64 == 0 on this path
  // unit table in registers: every wave redoes the 32-entry scan (no LDS, no barrier)
  const int cnt_l = (lane < K) ? v.hdr->cnt[lane * kCntStride] : 0;
  int P = cnt_l;
  for (int d = 16; d >= 1; d >>= 1) P += __shfl_xor(P, d);
  P = __shfl(P, 0);   // lanes >= 32 hold no offsets
  // pairs per unit: as few as possible (parallelism) while ALL units fit the chip in ONE round of workgroups.  The
  // estimate from the total ignores the partial last unit of every offset (up to K extra units): a grid that is a
  // few units short sends some workgroups round twice and doubles the kernel time, so G grows until the real count
  // fits (a handful of 5-step wave reductions).
  int G = (int)((((int64_t)(P + 63) >> 6) * gridDim.y + target_units - 1) / target_units);
  G = G < 1 ? 1 : (G > kMaxG ? kMaxG : G);
  int span, units, incl;
  for (;;) {
    span = 64 * G;
    units = (cnt_l + span - 1) / span;
    incl = units;
    for (int d = 1; d < 32; d <<= 1) {
      const int t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    if (__shfl(incl, 31) <= (int)gridDim.x || G >= kMaxG) break;
    ++G;
  }
  const int total_units = __shfl(incl, 31);

  for (int unit = blockIdx.x; unit < total_units; unit += gridDim.x) {
    const int k = __popcll(__ballot(lane < K && incl <= unit));
    const int excl = __builtin_amdgcn_readfirstlane(__shfl(incl - units, k));
    const int region = k * n_out;
    const int p_begin = region + (unit - excl) * span;
    const int p_end = min(p_begin + span, region + __builtin_amdgcn_readfirstlane(__shfl(cnt_l, k)));
    const int ngroups = (p_end - p_begin + 15) >> 4;

    // first group's rows (index load overlaps the weight staging)
    int g = wave;
    int my_p = p_begin + g * 16 + r16;
    // row indices are unsigned: the 64-bit row offset is then ONE v_mad_u64_u32 (a signed index costs a sign
    // extension that the compiler places right behind the index load, i.e. a wait for it inside the wrong step)
    const unsigned ldi32 = (unsigned)ldi;
    unsigned idx = (g < ngroups) ? (unsigned)v.pair_in[my_p < p_end ? my_p : p_begin] : 0u;
    unsigned idx_n = 0;   // row of the wave's second group: requested before the rows of the first, so that inside the step
    {                // loop "index older than rows older than stores" holds on every path (counted vmcnt waits)
      const int np = p_begin + (g + 4) * 16 + r16;
      if (g + 4 < ngroups) idx_n = (unsigned)v.pair_in[np < p_end ? np : p_begin];
    }
    {
      const float* src = wp + ((int64_t)k * cinG * cout + col0) * 4 + lane * 4;
      for (int g0 = wave; g0 < cinG; g0 += 32) {
        f32x4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (g0 + 4 * u < cinG) t[u] = *reinterpret_cast<const f32x4*>(src + (int64_t)(g0 + 4 * u) * cout * 4);
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (g0 + 4 * u < cinG) *reinterpret_cast<f32x4*>(&s_w[((g0 + 4 * u) * 64 + lane) * 4]) = t[u];
      }
    }
    // Flat (group, chunk) pipeline over this wave's groups g = wave, wave + 4, ...: two register buffers used in
    // turn (the loop is unrolled by two, no copies), the next step's rows are requested BEFORE the step's 64 MFMAs
    // and the row index of the group after next one step before its rows are — so every load has a full step
    // (>= 2048 MFMA cycles) to land.  Left to itself the compiler sinks a prefetch written as "a = an; an = load"
    // to the end of the step and waits for it there; the sched_barrier keeps issue order = program order.
    f32x4 bufA[4], bufB[4];
    const float* abase = in + (uint64_t)idx * ldi32 + q * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) bufA[j] = *reinterpret_cast<const f32x4*>(abase + j * 16);
    __syncthreads();

    const int nsteps = (g < ngroups) ? ((ngroups - g + 3) >> 2) * nchunk : 0;
    int c = 0;
    f32x4 acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};

#define APR_WS_STEP(cur, nxt)                                                                                     \
    {                                                                                                             \
      const bool last = (c + 1 == nchunk);                                                                        \
      const float* nb = last ? in + (uint64_t)idx_n * ldi32 + q * 4 : abase + (c + 1) * 64;                             \
      /* row of the group after next: always requested (clamped), so every path issues the same number of loads */ \
      const int np2 = p_begin + (g + 8) * 16 + r16;                                                               \
      const unsigned idx_nn = (unsigned)v.pair_in[np2 < p_end ? np2 : p_begin];                                                \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) nxt[j] = *reinterpret_cast<const f32x4*>(nb + j * 16);        \
      const float* wb = s_w + (c * 16 * 64) * 4;                                                                  \
      f32x4 bq[2][4];                                                                                             \
      _Pragma("unroll") for (int cb = 0; cb < 4; ++cb)                                                            \
        bq[0][cb] = *reinterpret_cast<const f32x4*>(wb + (q * 64 + cb * 16 + r16) * 4);                           \
      __builtin_amdgcn_sched_barrier(0);                                                                          \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                             \
        if (j < 3) {                                                                                              \
          _Pragma("unroll") for (int cb = 0; cb < 4; ++cb)                                                        \
            bq[(j + 1) & 1][cb] = *reinterpret_cast<const f32x4*>(wb + (((j + 1) * 4 + q) * 64 + cb * 16 + r16) * 4); \
          __builtin_amdgcn_sched_barrier(0); /* B fragments of j + 1 are in flight under the 16 MFMAs of j */    \
        }                                                                                                         \
        _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                             \
          _Pragma("unroll") for (int cb = 0; cb < 4; ++cb)                                                        \
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[j & 1][cb][t], cur[j][t], acc[cb], 0, 0, 0);        \
      }                                                                                                           \
      if (last) {                                                                                                 \
        /* D^T: lane (r16 = pair, q) holds channels cb*16 + 4q .. +3 of its pair */                               \
        /* lanes past the unit's end gathered the row of pair p_begin, so they hold exactly ITS product row (each   \
           MFMA output row depends on its own input row only): they store those same bits there, which keeps     \
           the stores unconditional (counted vmcnt waits, no exposed store acknowledgement) */                    \
        float* dst = prod + (int64_t)(my_p < p_end ? my_p : p_begin) * cout + col0 + q * 4;                       \
        _Pragma("unroll") for (int cb = 0; cb < 4; ++cb) /* written once, read once by k_ws_reduce */             \
          __builtin_nontemporal_store(acc[cb], reinterpret_cast<f32x4*>(dst + cb * 16));                          \
        _Pragma("unroll") for (int cb = 0; cb < 4; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};                   \
        g += 4;                                                                                                   \
        my_p += 64;                                                                                               \
        abase = nb;                                                                                               \
        idx_n = idx_nn;                                                                                           \
        c = 0;                                                                                                    \
      } else {                                                                                                    \
        ++c;                                                                                                      \
      }                                                                                                           \
    }

    int s = 0;
    for (; s + 2 <= nsteps; s += 2) {
      APR_WS_STEP(bufA, bufB)
      APR_WS_STEP(bufB, bufA)
    }
    if (s < nsteps) APR_WS_STEP(bufA, bufB)
#undef APR_WS_STEP
    __syncthreads();   // the slice is re-staged by the next unit
  }
}

// ---------------------------------------------------------------------------------------------------------------
// k_ws_gemm_bf3: the same unit structure, with the contraction on the bf16 MFMA in a 3-way split that keeps fp32
// accuracy.  Exact-fp32 MFMA runs at 1/16 of the bf16 rate on gfx950 and k_ws_gemm sits at 57-68 % of that roof: the
// remaining lever is fewer MFMA cycles, not better scheduling.  x = hi + mid + lo EXACTLY with three bf16 pieces (each
// the top 8 significant bits of what is left; the subtractions are exact in fp32: all 24 mantissa bits are kept); of the 9 cross terms of (a_h + a_m + a_l)(w_h + w_m + w_l) the three of order 2^-24 and below
// (a_m w_l, a_l w_m, a_l w_l) are dropped, the other six run as v_mfma_f32_16x16x32_bf16 (K = 32 per instruction, 16
// cycles), small terms first, fp32 accumulate.  Per 64-channel chunk and 16-pair group: 48 MFMAs x 16 cycles = 768
// cycles instead of 64 x 32 = 2048.  Error per product ~2 x 2^-24 relative: the same order as fp32 rounding (tests:
// features within 2e-5 rel-L2 of the fp32 oracle, measured ~1e-7).
// The weights are split once at pack time (apr_spconv_pack_weights_bf3) into the exact LDS image of a unit's slice:
// [k][64-column block][plane h/m/l][32-channel step][column 64][8-channel quad q'][8 bf16], q' = q ^ 3 for columns
// with bit 3 set: a straight copy stages the slice, and the fragment read (lane (column r16, quad q) -> 16 B) is a
// conflict-free ds_read_b128 (the 16 lanes of every hardware lane group land in 16 different 16-B slots).
// The input rows are split in registers after the gather (lane (pair r16, quad q) loads 8 consecutive channels per
// 32-channel step: two 16-B loads).
// w f32 -> wp3 (layout above).  One thread per 16-byte piece of the image (k, column block, step, column, quad): three
// coalesced 16-B stores (a thread per ELEMENT wrote 2 bytes at a 64-B stride: 15 us for a 27 x 128 x 128 kernel, and the
// training step re-packs ~40 kernels after every optimizer step).  `tr`: w is stored [K, cout, cin] and the image is that of
// its transpose -- with `flip` (offsets mirrored) the kernel of a convolution's INPUT GRADIENT straight from the parameter,
// no [K, cout, cin] copy in between.  Columns past cout (K = 1 images padded to 64) are zeros.
__global__ void k_pack_weights_bf3(const float* __restrict__ w, int K, int cin, int cout, int flip, int tr,
                                   __bf16* __restrict__ wp3) {
  const int nstep = cin >> 5, ncb = (cout + 63) >> 6;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)K * ncb * nstep * 64 * 4;
  if (t >= total) return;
  const int qs = (int)(t & 3), col = (int)((t >> 2) & 63);
  const int s = (int)((t >> 8) % nstep);
  const int cbk = (int)((t / ((int64_t)256 * nstep)) % ncb);
  const int k = (int)(t / ((int64_t)256 * nstep * ncb));
  const int q = (col & 8) ? (qs ^ 3) : qs;
  const int co = cbk * 64 + col, ci0 = s * 32 + q * 8;
  const int ks = flip ? K - 1 - k : k;
  typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
  u16x8 h = {0, 0, 0, 0, 0, 0, 0, 0}, m = h, l = h;
  if (co < cout) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = tr ? w[((int64_t)ks * cout + co) * cin + ci0 + e] : w[((int64_t)ks * cin + ci0 + e) * cout + co];
      const __bf16 hb = (__bf16)x;
      const float r = x - (float)hb;
      const __bf16 mb = (__bf16)r;
      const __bf16 lb = (__bf16)(r - (float)mb);
      h[e] = __builtin_bit_cast(unsigned short, hb);
      m[e] = __builtin_bit_cast(unsigned short, mb);
      l[e] = __builtin_bit_cast(unsigned short, lb);
    }
  }
  const int64_t slice = (int64_t)3 * nstep * 64 * 32;                  // bf16 elements of one (k, column block)
  const int64_t plane = (int64_t)nstep * 64 * 32;
  const int64_t base = ((int64_t)k * ncb + cbk) * slice + ((int64_t)s * 64 + col) * 32 + qs * 8;
  *reinterpret_cast<u16x8*>(wp3 + base) = h;
  *reinterpret_cast<u16x8*>(wp3 + base + plane) = m;
  *reinterpret_cast<u16x8*>(wp3 + base + 2 * plane) = l;
}

// NW waves per workgroup: 4, or 8 for the 256-channel slice (96 KB: ONE workgroup per CU whatever its size -- with 4
// waves that is one wave per SIMD and nothing to hide the gather -> split -> MFMA chain behind; 8 waves share the slice).
template <int NCH, int NW = 4>   // cin / 64: 1, 2 or 4
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void k_ws_gemm_bf3(const float* __restrict__ in, int64_t ldi, PairViews v, int K,
                                                        int cin, int cout, const __bf16* __restrict__ wp3,
                                                        float* __restrict__ prod, int n_out, int target_units) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];   // [plane 3][step][col 64][quad 4][8 bf16]
  constexpr int NT = 64 * NW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, q = lane >> 4;
  // 1-D grid of gx * ncb workgroups.  The ncb column-block workgroups of a unit gather the SAME input rows: they get
  // consecutive linear ids inside ONE XCD's share of the grid (workgroups are dealt round-robin over the 8 XCDs, each
  // with its own L2), so the second gather of a row finds it in that L2 instead of the Infinity Cache (12 frames per
  // launch: b4 256 -> 256 87 -> 75 us, FatBN's 128 -> 128 block at 189 k rows 440 -> 413 us; with ONE column block the
  // same re-ordering only moves units between XCDs and costs 12 % on c2tr, so it is not applied there).
  const int ncb = cout >> 6;
  const int nbk = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, loc = bid >> 3;
  const int lin = ncb > 1 ? xcd * (nbk >> 3) + min(xcd, nbk & 7) + loc : bid;   // one column block: nothing to share
  const int bx = lin / ncb, by = lin - bx * ncb, gx = nbk / ncb;
  const int col0 = by * 64;
  constexpr int nstep = NCH * 2;                       // 32-channel steps
  constexpr int plane_bytes = nstep * 64 * 64;         // one split plane of the slice
  constexpr int slice_bytes = 3 * plane_bytes;
  // unit table in registers (as k_ws_gemm)
  const int cnt_l = (lane < K) ? v.hdr->cnt[lane * kCntStride] : 0;
  int P = cnt_l;
  for (int d = 16; d >= 1; d >>= 1) P += __shfl_xor(P, d);
  P = __shfl(P, 0);
  int G = (int)((((int64_t)(P + 63) >> 6) * ncb + target_units - 1) / target_units);
  G = G < 1 ? 1 : (G > kMaxG ? kMaxG : G);
  int span, units, incl;
  for (;;) {
    span = 64 * G;
    units = (cnt_l + span - 1) / span;
    incl = units;
    for (int d = 1; d < 32; d <<= 1) {
      const int t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    if (__shfl(incl, 31) <= gx || G >= kMaxG) break;
    ++G;
  }
  const int total_units = __shfl(incl, 31);
  // fragment read offset of this lane inside a (step, 16-column block): column r16, swizzled quad
  const int frag_off = (r16 * 4 + ((r16 & 8) ? (q ^ 3) : q)) * 16;

  for (int unit = bx; unit < total_units; unit += gx) {
    const int k = __popcll(__ballot(lane < K && incl <= unit));
    const int excl = __builtin_amdgcn_readfirstlane(__shfl(incl - units, k));
    const int region = k * n_out;
    const int p_begin = region + (unit - excl) * span;
    const int p_end = min(p_begin + span, region + __builtin_amdgcn_readfirstlane(__shfl(cnt_l, k)));
    const int ngroups = (p_end - p_begin + 15) >> 4;

    int g = wave;
    int my_p = p_begin + g * 16 + r16;
    const unsigned ldi32 = (unsigned)ldi;
    unsigned idx = (g < ngroups) ? (unsigned)v.pair_in[my_p < p_end ? my_p : p_begin] : 0u;
    unsigned idx_n = 0;
    {
      const int np = p_begin + (g + NW) * 16 + r16;
      if (g + NW < ngroups) idx_n = (unsigned)v.pair_in[np < p_end ? np : p_begin];
    }
    {   // stage the slice: a straight copy (the global layout IS the LDS image)
      const unsigned char* src = reinterpret_cast<const unsigned char*>(wp3) +
                                 ((int64_t)k * (cout >> 6) + by) * slice_bytes;
      for (int o0 = tid * 16; o0 < slice_bytes; o0 += 8 * NT * 16) {
        f32x4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (o0 + u * NT * 16 < slice_bytes) t[u] = *reinterpret_cast<const f32x4*>(src + o0 + u * NT * 16);
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (o0 + u * NT * 16 < slice_bytes) *reinterpret_cast<f32x4*>(s_raw + o0 + u * NT * 16) = t[u];
      }
    }
    // rows: lane (pair r16, quad q) holds channels 32 s + 8 q .. + 7 of its pair for the chunk's two steps
    f32x4 bufA[4], bufB[4];
    const float* abase = in + (uint64_t)idx * ldi32 + q * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) bufA[j] = *reinterpret_cast<const f32x4*>(abase + (j >> 1) * 32 + (j & 1) * 4);
    __syncthreads();

    const int nsteps = (g < ngroups) ? ((ngroups - g + NW - 1) / NW) * NCH : 0;
    int c = 0;
    f32x4 acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};

#define APR_WS3_STEP(cur, nxt)                                                                                    \
    {                                                                                                             \
      const bool last = (c + 1 == NCH);                                                                           \
      const float* nb = last ? in + (uint64_t)idx_n * ldi32 + q * 8 : abase + (c + 1) * 64;                       \
      const int np2 = p_begin + (g + 2 * NW) * 16 + r16;                                                               \
      const unsigned idx_nn = (unsigned)v.pair_in[np2 < p_end ? np2 : p_begin];                                    \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                               \
        nxt[j] = *reinterpret_cast<const f32x4*>(nb + (j >> 1) * 32 + (j & 1) * 4);                               \
      bf16x8 ah[2], am[2], al[2];                                                                                 \
      apr_split3(cur[0], cur[1], ah[0], am[0], al[0]);                                                            \
      apr_split3(cur[2], cur[3], ah[1], am[1], al[1]);                                                            \
      /* W fragments of (step s, 16-column block cb): three 16-B reads, fetched TWO (s, cb) ahead of their MFMAs: six  \
         bf16 MFMAs are only 96 cycles, less than an LDS round trip (the fp32 form has 512 cycles per fragment set) */   \
      const unsigned char* wb = s_raw + (c * 2 * 64) * 64 + frag_off;                                             \
      bf16x8 wf[3][3];                                                                                            \
      _Pragma("unroll") for (int i0 = 0; i0 < 2; ++i0)                                                            \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                          \
          wf[i0][pl] = *reinterpret_cast<const bf16x8*>(wb + ((i0 >> 2) * 64 + (i0 & 3) * 16) * 64 + pl * plane_bytes); \
      __builtin_amdgcn_sched_barrier(0);                                                                          \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                             \
        const int s = i >> 2, cb = i & 3;                                                                         \
        if (i < 6) {                                                                                              \
          const int s2 = (i + 2) >> 2, cb2 = (i + 2) & 3;                                                         \
          _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                        \
            wf[(i + 2) % 3][pl] =                                                                                 \
                *reinterpret_cast<const bf16x8*>(wb + (s2 * 64 + cb2 * 16) * 64 + pl * plane_bytes);             \
          __builtin_amdgcn_sched_barrier(0);                                                                      \
        }                                                                                                         \
        const bf16x8 wh = wf[i % 3][0], wm = wf[i % 3][1], wl = wf[i % 3][2];                                     \
        f32x4 t = acc[cb];                                                                                        \
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, ah[s], t, 0, 0, 0);                                       \
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, al[s], t, 0, 0, 0);                                       \
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, am[s], t, 0, 0, 0);                                       \
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, ah[s], t, 0, 0, 0);                                       \
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, am[s], t, 0, 0, 0);                                       \
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ah[s], t, 0, 0, 0);                                       \
        acc[cb] = t;                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
      }                                                                                                           \
      if (last) {                                                                                                 \
        float* dst = prod + (int64_t)(my_p < p_end ? my_p : p_begin) * cout + col0 + q * 4;                       \
        _Pragma("unroll") for (int cb = 0; cb < 4; ++cb)                                                          \
          __builtin_nontemporal_store(acc[cb], reinterpret_cast<f32x4*>(dst + cb * 16));                         \
        _Pragma("unroll") for (int cb = 0; cb < 4; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};                   \
        g += NW;                                                                                                  \
        my_p += 16 * NW;                                                                                          \
        abase = nb;                                                                                               \
        idx_n = idx_nn;                                                                                           \
        c = 0;                                                                                                    \
      } else {                                                                                                    \
        ++c;                                                                                                      \
      }                                                                                                           \
    }

    int s_ = 0;
    for (; s_ + 2 <= nsteps; s_ += 2) {
      APR_WS3_STEP(bufA, bufB)
      APR_WS3_STEP(bufB, bufA)
    }
    if (s_ < nsteps) APR_WS3_STEP(bufA, bufB)
#undef APR_WS3_STEP
    __syncthreads();   // the slice is re-staged by the next unit
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Triple pair lists + k_ws3_gemm_bf3: HALF the product rows for 3^3 maps.
// The weight-stationary pair writes one product row per kernel-map pair and k_ws_reduce reads it back: 1.57x the layer's
// algorithmic bytes through the fabric (PMC), which is what bounds the family.  Offsets are enumerated x fastest, so
// k = 3t + j (j = 0, 1, 2) are the three x-neighbours of one (dy, dz): on surfaces they come together -- a row that has one
// of them usually has two or three.  A TRIPLE entry is (output row, its up to three input rows of triple t); the gemm unit
// is (triple t, 64 G consecutive entries, 64 output columns) with the three weight slices W[3t + j][:, 64] in LDS, a 16-entry
// group runs its 3 x cin/64 chunks into the SAME 16 accumulator registers (an absent neighbour gathers a zero row), and ONE
// product row leaves per entry: measured entries / pairs on the KITTI maps ~0.5.  The inner loop is the barrier-free step
// pipeline of k_ws_gemm_bf3; MFMA work grows by the zero-padded share (~1.6x of a pipe that was 30 % busy), the product
// traffic -- written by the gemm, read by the reduce -- halves.  The reduce is k_ws_reduce over 9 ids per row.
// cin 64 (two 72 KB workgroups per CU) and 128 (one 144 KB workgroup of 8 waves); K = 27 only.
constexpr int kRows3 = 256;
struct Pair3Views {
  PairHeader* hdr;   // counters of the 9 triples at cnt[t * kCntStride]
  int* ent;          // [9][n_out][3] input rows (-1: absent) of the entries of triple t, at its region's head
  int* pair_id;      // [n_out][9] position of the row's entry in triple t (t * n_out + rank) or -1
};

__host__ __device__ inline Pair3Views carve_pairs3(int32_t* counters, void* blob, int64_t n_out) {
  Pair3Views v;
  v.hdr = (PairHeader*)counters;
  v.ent = (int*)blob;
  v.pair_id = (int*)((char*)blob + align256((size_t)n_out * 9 * 12));
  return v;
}

__device__ float g_zero_row[512];      // zero-initialised: the row an absent neighbour of a triple is gathered from

__global__ __launch_bounds__(64 * kBuildWaves) void k_pairs3_build(const int* __restrict__ nbr, int n_out, Pair3Views v) {
  __shared__ int s_nbr[kRows3 * 27 + 64];
  __shared__ int s_id[kRows3 * 9];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = blockIdx.x * kRows3;
  const int rows = min(kRows3, n_out - row0);
  const int total = rows * 27;
  const int* src = nbr + (int64_t)row0 * 27;
  for (int e0 = threadIdx.x; e0 < total; e0 += 9 * 64 * kBuildWaves) {
    int t[9];
#pragma unroll
    for (int u = 0; u < 9; ++u) {
      const int e = e0 + u * 64 * kBuildWaves;
      t[u] = (e < total) ? src[e] : -1;
    }
#pragma unroll
    for (int u = 0; u < 9; ++u) {
      const int e = e0 + u * 64 * kBuildWaves;
      if (e < total) s_nbr[e] = t[u];
    }
  }
  __syncthreads();
  auto any3 = [&](int r, int t) {
    const int* p = &s_nbr[r * 27 + 3 * t];
    return (p[0] & p[1] & p[2]) >= 0;      // some index non-negative <=> the AND of the three has its sign bit clear
  };
  int base[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    const int t = wave + kBuildWaves * s2;
    base[s2] = 0;
    if (t < 9) {
      int cnt = 0;
#pragma unroll
      for (int c = 0; c < kRows3 / 64; ++c) {
        const int r = c * 64 + lane;
        cnt += __popcll(__ballot(r < rows && any3(r, t)));
      }
      if (lane == 0 && cnt) base[s2] = atomicAdd(&v.hdr->cnt[t * kCntStride], cnt);
    }
  }
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    const int t = wave + kBuildWaves * s2;
    if (t < 9) {
      int run = __shfl(base[s2], 0);
#pragma unroll
      for (int c = 0; c < kRows3 / 64; ++c) {
        const int r = c * 64 + lane;
        const bool ok = r < rows && any3(r, t);
        const unsigned long long m = __ballot(ok);
        if (r < rows) {
          int pos = -1;
          if (ok) {
            const int rank = run + __popcll(m & ((1ull << lane) - 1ull));
            if (rank < n_out) {      // counters not cleared by the caller would push past the region: never write there
              pos = t * n_out + rank;
              int* e = v.ent + (int64_t)pos * 3;
              e[0] = s_nbr[r * 27 + 3 * t];
              e[1] = s_nbr[r * 27 + 3 * t + 1];
              e[2] = s_nbr[r * 27 + 3 * t + 2];
            }
          }
          s_id[r * 9 + t] = pos;
        }
        run += __popcll(m);
      }
    }
  }
  __syncthreads();
  int* dst = v.pair_id + (int64_t)row0 * 9;
  for (int e = threadIdx.x; e < rows * 9; e += 64 * kBuildWaves) dst[e] = s_id[e];
}

template <int NCH, int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void k_ws3_gemm_bf3(const float* __restrict__ in, int64_t ldi, Pair3Views v,
                                                                          int cin, int cout, const __bf16* __restrict__ wp3,
                                                                          float* __restrict__ prod, int n_out, int target_units) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];   // [j 3][plane 3][step][col 64][quad 4][8 bf16]
  constexpr int NT = 64 * NW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, q = lane >> 4;
  const int ncb = cout >> 6;
  const int nbk = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, loc = bid >> 3;
  const int lin = ncb > 1 ? xcd * (nbk >> 3) + min(xcd, nbk & 7) + loc : bid;      // column blocks of a unit behind one L2
  const int bx = lin / ncb, by = lin - bx * ncb, gx = nbk / ncb;
  const int col0 = by * 64;
  constexpr int nstep = NCH * 2;
  constexpr int plane_bytes = nstep * 64 * 64;
  constexpr int slice_bytes = 3 * plane_bytes;
  const int cnt_l = (lane < 9) ? v.hdr->cnt[lane * kCntStride] : 0;
  int P = cnt_l;
  for (int d = 16; d >= 1; d >>= 1) P += __shfl_xor(P, d);
  P = __shfl(P, 0);
  int G = (int)((((int64_t)(P + 63) >> 6) * ncb + target_units - 1) / target_units);
  G = G < 1 ? 1 : (G > kMaxG ? kMaxG : G);
  int span, units, incl;
  for (;;) {
    span = 64 * G;
    units = (cnt_l + span - 1) / span;
    incl = units;
    for (int d = 1; d < 32; d <<= 1) {
      const int t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    if (__shfl(incl, 31) <= gx || G >= kMaxG) break;
    ++G;
  }
  const int total_units = __shfl(incl, 31);
  const int frag_off = (r16 * 4 + ((r16 & 8) ? (q ^ 3) : q)) * 16;
  const unsigned ldi32 = (unsigned)ldi;
  auto row_ptr = [&](int idx) { return (idx >= 0 ? in + (uint64_t)(unsigned)idx * ldi32 : g_zero_row) + q * 8; };

  for (int unit = bx; unit < total_units; unit += gx) {
    const int t3 = __popcll(__ballot(lane < 9 && incl <= unit));
    const int excl = __builtin_amdgcn_readfirstlane(__shfl(incl - units, t3));
    const int region = t3 * n_out;
    const int p_begin = region + (unit - excl) * span;
    const int p_end = min(p_begin + span, region + __builtin_amdgcn_readfirstlane(__shfl(cnt_l, t3)));
    const int ngroups = (p_end - p_begin + 15) >> 4;
    int g = wave;
    int my_p = p_begin + g * 16 + r16;
    // the three input rows of this lane's entry (lanes past the unit's end: its first entry -- they recompute and re-store
    // that entry's product row, the same bits), and those of the wave's next group
    auto entry = [&](int p, int& a, int& b, int& c2) {
      const int* e = v.ent + (int64_t)(p < p_end ? p : p_begin) * 3;
      a = e[0]; b = e[1]; c2 = e[2];
    };
    int i0 = -1, i1 = -1, i2 = -1, n0 = -1, n1 = -1, n2 = -1;
    if (g < ngroups) entry(my_p, i0, i1, i2);
    if (g + NW < ngroups) entry(my_p + 16 * NW, n0, n1, n2);
    {   // stage the three slices of the triple: straight copies (the global layout is the LDS image)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const unsigned char* src = reinterpret_cast<const unsigned char*>(wp3) +
                                   ((int64_t)(3 * t3 + j) * (cout >> 6) + by) * slice_bytes;
        unsigned char* dstl = s_raw + j * slice_bytes;
        for (int o0 = tid * 16; o0 < slice_bytes; o0 += 8 * NT * 16) {
          f32x4 tt[8];
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (o0 + u * NT * 16 < slice_bytes) tt[u] = *reinterpret_cast<const f32x4*>(src + o0 + u * NT * 16);
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (o0 + u * NT * 16 < slice_bytes) *reinterpret_cast<f32x4*>(dstl + o0 + u * NT * 16) = tt[u];
        }
      }
    }
    f32x4 bufA[4], bufB[4];
    {
      const float* a0 = row_ptr(i0);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) bufA[jj] = *reinterpret_cast<const f32x4*>(a0 + (jj >> 1) * 32 + (jj & 1) * 4);
    }
    __syncthreads();

    const int nsteps = (g < ngroups) ? ((ngroups - g + NW - 1) / NW) * 3 * NCH : 0;
    int j = 0, c = 0;
    f32x4 acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};

#define APR_WS33_STEP(cur, nxt)                                                                                   \
    {                                                                                                             \
      int c1 = c + 1, j1 = j;                                                                                     \
      bool newg = false;                                                                                          \
      if (c1 == NCH) {                                                                                            \
        c1 = 0;                                                                                                   \
        j1 = j + 1;                                                                                               \
        if (j1 == 3) { j1 = 0; newg = true; }                                                                     \
      }                                                                                                           \
      const int nidx = newg ? n0 : (j1 == 0 ? i0 : (j1 == 1 ? i1 : i2));                                          \
      const float* nb = row_ptr(nidx) + c1 * 64;                                                                  \
      int m0 = n0, m1 = n1, m2 = n2;                                                                              \
      if (newg) { /* the group after next: its entry is requested a whole group ahead */                          \
        m0 = m1 = m2 = -1;                                                                                        \
        if (g + 2 * NW < ngroups) entry(my_p + 32 * NW, m0, m1, m2);                                              \
      }                                                                                                           \
      _Pragma("unroll") for (int jj = 0; jj < 4; ++jj)                                                            \
        nxt[jj] = *reinterpret_cast<const f32x4*>(nb + (jj >> 1) * 32 + (jj & 1) * 4);                            \
      bf16x8 ah[2], am[2], al[2];                                                                                 \
      apr_split3(cur[0], cur[1], ah[0], am[0], al[0]);                                                            \
      apr_split3(cur[2], cur[3], ah[1], am[1], al[1]);                                                            \
      const unsigned char* wb = s_raw + j * slice_bytes + (c * 2 * 64) * 64 + frag_off;                           \
      bf16x8 wf[3][3];                                                                                            \
      _Pragma("unroll") for (int i0_ = 0; i0_ < 2; ++i0_)                                                         \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                          \
          wf[i0_][pl] = *reinterpret_cast<const bf16x8*>(wb + (i0_ * 16) * 64 + pl * plane_bytes);                \
      __builtin_amdgcn_sched_barrier(0);                                                                          \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                             \
        const int s = i >> 2, cb = i & 3;                                                                         \
        if (i < 6) {                                                                                              \
          const int s2 = (i + 2) >> 2, cb2 = (i + 2) & 3;                                                         \
          _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                        \
            wf[(i + 2) % 3][pl] =                                                                                 \
                *reinterpret_cast<const bf16x8*>(wb + (s2 * 64 + cb2 * 16) * 64 + pl * plane_bytes);             \
          __builtin_amdgcn_sched_barrier(0);                                                                      \
        }                                                                                                         \
        const bf16x8 wh = wf[i % 3][0], wm = wf[i % 3][1], wl = wf[i % 3][2];                                     \
        f32x4 t = acc[cb];                                                                                        \
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, ah[s], t, 0, 0, 0);                                       \
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, al[s], t, 0, 0, 0);                                       \
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, am[s], t, 0, 0, 0);                                       \
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, ah[s], t, 0, 0, 0);                                       \
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, am[s], t, 0, 0, 0);                                       \
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ah[s], t, 0, 0, 0);                                       \
        acc[cb] = t;                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
      }                                                                                                           \
      if (newg) {                                                                                                 \
        float* dst = prod + (int64_t)(my_p < p_end ? my_p : p_begin) * cout + col0 + q * 4;                       \
        _Pragma("unroll") for (int cb = 0; cb < 4; ++cb)                                                          \
          __builtin_nontemporal_store(acc[cb], reinterpret_cast<f32x4*>(dst + cb * 16));                         \
        _Pragma("unroll") for (int cb = 0; cb < 4; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};                   \
        g += NW;                                                                                                  \
        my_p += 16 * NW;                                                                                          \
        i0 = n0; i1 = n1; i2 = n2;                                                                                \
        n0 = m0; n1 = m1; n2 = m2;                                                                                \
      }                                                                                                           \
      j = j1;                                                                                                     \
      c = c1;                                                                                                     \
    }

    int s_ = 0;
    for (; s_ + 2 <= nsteps; s_ += 2) {
      APR_WS33_STEP(bufA, bufB)
      APR_WS33_STEP(bufB, bufA)
    }
    if (s_ < nsteps) APR_WS33_STEP(bufA, bufB)
#undef APR_WS33_STEP
    __syncthreads();   // the slices are re-staged by the next unit
  }
}

// All pair ids of the row first, then all product loads in flight at once (exec-masked), summed in offset order:
// the naive "load id -> branch -> load -> add" loop is a chain of K dependent L2 round trips (~25 us floor).
template <int KT>
__global__ __launch_bounds__(256) void k_ws_reduce(const float* __restrict__ prod, PairViews v, int64_t n_out, int K,
                                                   int cout, const float* __restrict__ scale,
                                                   const float* __restrict__ shift, const float* __restrict__ residual,
                                                   int64_t ldr, int relu, float* __restrict__ out, int64_t ldo) {
  const int c4n = cout >> 2;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_out * c4n) return;
  const int64_t row = t / c4n;
  const int col = (int)(t - row * c4n) * 4;
  const int* ids = v.pair_id + row * K;
  int pid[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) pid[k] = (k < K) ? ids[k] : -1;
  f32x4 p[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    p[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (pid[k] >= 0) p[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(prod + (int64_t)pid[k] * cout + col));
  }
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < KT; ++k) s += p[k];
  if (scale) s *= *reinterpret_cast<const f32x4*>(scale + col);
  if (shift) s += *reinterpret_cast<const f32x4*>(shift + col);
  if (residual) s += *reinterpret_cast<const f32x4*>(residual + row * ldr + col);
  if (relu) {
    s[0] = fmaxf(s[0], 0.f); s[1] = fmaxf(s[1], 0.f); s[2] = fmaxf(s[2], 0.f); s[3] = fmaxf(s[3], 0.f);
  }
  *reinterpret_cast<f32x4*>(out + row * ldo + col) = s;
}

}  // namespace

// K = 1 (dense layers, dense.hip / dense_rows.hip) also takes cin % 32 == 0 and cout % 16 == 0: the image is laid out per
// 32-channel step and 64-column block, the columns past cout are zeros.
static bool bf3_shape_ok(int32_t K, int32_t cin, int32_t cout) {
  if (K < 1 || cin < 32 || cout < 16) return false;
  if (K == 1) return cin % 32 == 0 && cout % 16 == 0;
  return cin % 64 == 0 && cout % 64 == 0;
}

APR_API int64_t apr_spconv_packed_bf3_bytes(int32_t K, int32_t cin, int32_t cout) {
  return bf3_shape_ok(K, cin, cout) ? (int64_t)K * cin * ((cout + 63) / 64 * 64) * 6 : 0;
}

// flip / transposed: the image of w'[k'][ci][co] = w[k][co][ci] (k' = K - 1 - k when flip) for w stored f32 [K, cout, cin] --
// the kernel of the convolution's input gradient (apr_weights_flip_transpose + apr_spconv_pack_weights_bf3 in one pass).
APR_API int apr_spconv_pack_weights_bf3_ex(const float* w, int32_t K, int32_t cin, int32_t cout, int32_t flip,
                                           int32_t transposed, void* w_bf3, void* stream) {
  APR_CHECK_ARG(w && w_bf3 && bf3_shape_ok(K, cin, cout),
                "apr_spconv_pack_weights_bf3: needs cin %% 64 == 0 and cout %% 64 == 0 (K = 1: cin %% 32 == 0, cout %% 16 == 0)");
  APR_CHECK_ARG(((uintptr_t)w_bf3 & 15) == 0, "apr_spconv_pack_weights_bf3: the image must be 16-byte aligned");
  const int64_t total = (int64_t)K * ((cout + 63) / 64) * (cin / 32) * 256;
  hipLaunchKernelGGL(k_pack_weights_bf3, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, w, K, cin,
                     cout, flip ? 1 : 0, transposed ? 1 : 0, (__bf16*)w_bf3);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API int apr_spconv_pack_weights_bf3(const float* w, int32_t K, int32_t cin, int32_t cout, void* w_bf3, void* stream) {
  return apr_spconv_pack_weights_bf3_ex(w, K, cin, cout, 0, 0, w_bf3, stream);
}

APR_API int32_t apr_pairlist_counter_ints(void) { return 32 * kCntStride; }

APR_API size_t apr_pairlist_bytes(int64_t n_out, int32_t K) {
  return 2 * align256((size_t)(n_out > 0 ? n_out : 1) * K * 4) + 256;
}

APR_API int apr_pairlist_build(const int32_t* nbr, int64_t n_out, int32_t K, int32_t* counters, void* plist,
                               size_t plist_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APR_CHECK_ARG(n_out > 0 && n_out < (1ll << 31) / 32 && K >= 1 && K <= 27,
                "apr_pairlist_build: needs 0 < n_out < 2^26 and 1 <= K <= 27");
  APR_CHECK_ARG(plist_bytes >= apr_pairlist_bytes(n_out, K), "apr_pairlist_build: blob too small");
  APR_CHECK_ARG(counters != nullptr && plist != nullptr, "apr_pairlist_build: null counters / plist");
  PairViews v = carve_pairs(counters, plist, n_out, K);
  hipLaunchKernelGGL(k_pairs_build, dim3((unsigned)cdiv64(n_out, kRows)), dim3(64 * kBuildWaves), 0, st, nbr,
                     (int)n_out, K, v);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

static int ws_fwd(const float* in, int64_t ldi, const int32_t* counters, const void* plist, int64_t n_out, int32_t K,
                  int32_t cin, int32_t cout, const float* w_packed, const void* w_bf3, const float* scale,
                  const float* shift, const float* residual, int64_t ldr, int32_t relu, float* out, int64_t ldo,
                  float* prod_scratch, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APR_CHECK_ARG(n_out > 0 && n_out < (1ll << 31) / 32 && K >= 1 && K <= 27, "apr_spconv_ws_fwd: bad n_out / K");
  APR_CHECK_ARG(cin % 64 == 0 && cin <= 512 && cout % 64 == 0,
                "apr_spconv_ws_fwd: needs cin %% 64 == 0, cin <= 512 and cout %% 64 == 0");
  APR_CHECK_ARG(ldi > 0 && ldi < (1ll << 31), "apr_spconv_ws_fwd: ldi out of range");
  APR_CHECK_ARG(ldi % 4 == 0 && ldo % 4 == 0 && ((((uintptr_t)in) | ((uintptr_t)out) | ((uintptr_t)prod_scratch)) & 15) == 0,
                "apr_spconv_ws_fwd: 16-byte aligned rows required");
  APR_CHECK_ARG(!residual || (ldr % 4 == 0 && (((uintptr_t)residual) & 15) == 0), "apr_spconv_ws_fwd: residual alignment");
  PairViews v = carve_pairs(const_cast<int32_t*>(counters), const_cast<void*>(plist), n_out, K);
  // The weight slice in LDS limits residency to floor(160 KB / slice) workgroups per CU; more units than resident
  // slots means a second round of workgroups behind the first (measured: 29 us vs 15 us on the 256-channel level).
  // target = the slots (<= 768); the kernel sizes its units from the real pair count to fit, and a grid of that
  // many workgroups strides over them.
  const size_t lds = (size_t)cin * 64 * 4;
  int64_t per_cu = (160 * 1024) / (int64_t)lds;
  if (per_cu > 8) per_cu = 8;
  int64_t target = 256 * per_cu;
  static const int s_target = env_int("APR_WS_TARGET", 768);   // A/B switch
  if (target > s_target) target = s_target;
  int64_t gx = cdiv64(target, cout / 64);
  const int64_t need = cdiv64(n_out * (int64_t)K, 64) + K;
  if (gx > need) gx = need;
  const unsigned units = (unsigned)gx;
  // this entry point is called from several host threads (one per stream): the 128 KB dynamic-LDS opt-in is set
  // once per device, under a lock
  {
    static std::mutex s_mu;
    static bool s_attr[64] = {};
    int dev = 0;
    APR_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(s_mu);
    if (dev >= 0 && dev < 64 && !s_attr[dev]) {
      APR_HIP(hipFuncSetAttribute((const void*)k_ws_gemm<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
      APR_HIP(hipFuncSetAttribute((const void*)k_ws_gemm<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
      APR_HIP(hipFuncSetAttribute((const void*)k_ws_gemm<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
      APR_HIP(hipFuncSetAttribute((const void*)k_ws_gemm<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
      APR_HIP(hipFuncSetAttribute((const void*)k_ws_gemm_bf3<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
      APR_HIP(hipFuncSetAttribute((const void*)k_ws_gemm_bf3<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
      APR_HIP(hipFuncSetAttribute((const void*)k_ws_gemm_bf3<4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
      APR_HIP(hipFuncSetAttribute((const void*)k_ws_gemm_bf3<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
      APR_HIP(hipFuncSetAttribute((const void*)k_ws_gemm_bf3<6, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
      s_attr[dev] = true;
    }
  }
  // 192 / 384 input channels: the transposed convolutions behind a skip concatenation in the wide variants (ResUNetFatBN:
  // conv2_tr 192 -> 128, conv3_tr 384 -> 128, FCGF_APR/model/resunet.py:224-227) -- same kernel, 3 / 6 chunks per group
  if (w_bf3 && (cin == 64 || cin == 128 || cin == 192 || cin == 256 || cin == 384)) {
    // bf16 3-way split path: slice = cin * 64 * 6 B
    const size_t lds3 = (size_t)cin * 64 * 6;
    int64_t per_cu3 = (160 * 1024) / (int64_t)lds3;
    if (per_cu3 > 8) per_cu3 = 8;
    int64_t target3 = 256 * per_cu3;
    if (target3 > s_target) target3 = s_target;
    int64_t gx3 = cdiv64(target3, cout / 64);
    if (gx3 > need) gx3 = need;
    static const int s_nw8 = env_int("APR_WS_NW8", 1);      // A/B switch: 0 = 4 waves for the 256-channel slice too
    if (cin == 256 && s_nw8)
      hipLaunchKernelGGL((k_ws_gemm_bf3<4, 8>), dim3((unsigned)(gx3 * (cout / 64))), dim3(512), lds3, st, in, ldi, v, K, cin, cout,
                         (const __bf16*)w_bf3, prod_scratch, (int)n_out, (int)target3);
    else if (cin == 384)      // 144 KB slice: one 8-wave workgroup per CU
      hipLaunchKernelGGL((k_ws_gemm_bf3<6, 8>), dim3((unsigned)(gx3 * (cout / 64))), dim3(512), lds3, st, in, ldi, v, K, cin, cout,
                         (const __bf16*)w_bf3, prod_scratch, (int)n_out, (int)target3);
    else {
      auto k3 = cin == 64 ? k_ws_gemm_bf3<1> : cin == 128 ? k_ws_gemm_bf3<2> : cin == 192 ? k_ws_gemm_bf3<3> : k_ws_gemm_bf3<4>;
      hipLaunchKernelGGL(k3, dim3((unsigned)(gx3 * (cout / 64))), dim3(256), lds3, st, in, ldi, v, K, cin, cout,
                         (const __bf16*)w_bf3, prod_scratch, (int)n_out, (int)target3);
    }
  } else {
  auto kern = cin == 64 ? k_ws_gemm<1> : cin == 128 ? k_ws_gemm<2> : cin == 256 ? k_ws_gemm<4> : k_ws_gemm<0>;
  hipLaunchKernelGGL(kern, dim3(units, cout / 64), dim3(256), lds, st, in, ldi, v, K, cin, cout, w_packed,
                     prod_scratch, (int)n_out, (int)target);
  }
  const dim3 rgrid((unsigned)cdiv64(n_out * (cout / 4), 256));
  if (K <= 8)
    hipLaunchKernelGGL(k_ws_reduce<8>, rgrid, dim3(256), 0, st, prod_scratch, v, n_out, K, cout, scale, shift, residual,
                       ldr, relu, out, ldo);
  else if (K <= 27)
    hipLaunchKernelGGL(k_ws_reduce<27>, rgrid, dim3(256), 0, st, prod_scratch, v, n_out, K, cout, scale, shift,
                       residual, ldr, relu, out, ldo);
  else
    hipLaunchKernelGGL(k_ws_reduce<32>, rgrid, dim3(256), 0, st, prod_scratch, v, n_out, K, cout, scale, shift,
                       residual, ldr, relu, out, ldo);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

// ---- triple pair lists (3^3 maps): half the product rows; see k_ws3_gemm_bf3 ----
APR_API size_t apr_pairlist3_bytes(int64_t n_out) {
  const size_t n = (size_t)(n_out > 0 ? n_out : 1);
  return align256(n * 9 * 12) + align256(n * 9 * 4) + 256;
}

APR_API int apr_pairlist3_build(const int32_t* nbr, int64_t n_out, int32_t K, int32_t* counters, void* plist3,
                                size_t plist3_bytes, void* stream) {
  APR_CHECK_ARG(n_out > 0 && n_out < (1ll << 31) / 32 && K == 27, "apr_pairlist3_build: needs 0 < n_out < 2^26 and K = 27");
  APR_CHECK_ARG(plist3_bytes >= apr_pairlist3_bytes(n_out), "apr_pairlist3_build: blob too small");
  APR_CHECK_ARG(nbr != nullptr && counters != nullptr && plist3 != nullptr, "apr_pairlist3_build: null argument");
  Pair3Views v = carve_pairs3(counters, plist3, n_out);
  hipLaunchKernelGGL(k_pairs3_build, dim3((unsigned)cdiv64(n_out, kRows3)), dim3(64 * kBuildWaves), 0, (hipStream_t)stream, nbr,
                     (int)n_out, v);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

// 1 if apr_spconv_ws3_fwd_bf3 takes the layer: a 27-offset map, 64 or 128 input channels, cout % 64 == 0
APR_API int apr_spconv_ws3_supported(int32_t K, int32_t cin, int32_t cout) {
  return K == 27 && (cin == 64 || cin == 128) && cout >= 64 && cout % 64 == 0;
}

// out = act((sum_k in[nbr[., k]] @ W[k]) * scale + shift + residual) over the TRIPLE pair lists of apr_pairlist3_build;
// w_bf3 from apr_spconv_pack_weights_bf3(w, 27, cin, cout); prod_scratch: f32 [9 * n_out, cout].
APR_API int apr_spconv_ws3_fwd_bf3(const float* in, int64_t ldi, const int32_t* counters, const void* plist3, int64_t n_out,
                                   int32_t cin, int32_t cout, const void* w_bf3, const float* scale, const float* shift,
                                   const float* residual, int64_t ldr, int32_t relu, float* out, int64_t ldo,
                                   float* prod_scratch, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APR_CHECK_ARG(in && counters && plist3 && w_bf3 && out && prod_scratch && n_out > 0 && n_out < (1ll << 31) / 32,
                "apr_spconv_ws3_fwd_bf3: bad n_out / null argument");
  APR_CHECK_ARG(apr_spconv_ws3_supported(27, cin, cout), "apr_spconv_ws3_fwd_bf3: needs cin 64 or 128 and cout %% 64 == 0");
  APR_CHECK_ARG(ldi > 0 && ldi < (1ll << 31) && ldi % 4 == 0 && ldo % 4 == 0 &&
                    ((((uintptr_t)in) | ((uintptr_t)out) | ((uintptr_t)prod_scratch)) & 15) == 0,
                "apr_spconv_ws3_fwd_bf3: 16-byte aligned rows required");
  APR_CHECK_ARG(!residual || (ldr % 4 == 0 && (((uintptr_t)residual) & 15) == 0), "apr_spconv_ws3_fwd_bf3: residual alignment");
  Pair3Views v = carve_pairs3(const_cast<int32_t*>(counters), const_cast<void*>(plist3), n_out);
  {
    static std::mutex s_mu;
    static bool s_attr[64] = {};
    int dev = 0;
    APR_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(s_mu);
    if (dev >= 0 && dev < 64 && !s_attr[dev]) {
      APR_HIP(hipFuncSetAttribute((const void*)k_ws3_gemm_bf3<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
      APR_HIP(hipFuncSetAttribute((const void*)k_ws3_gemm_bf3<2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
      s_attr[dev] = true;
    }
  }
  const size_t lds = (size_t)3 * cin * 64 * 6;
  const int64_t per_cu = cin == 64 ? 2 : 1;
  static const int s_target = env_int("APR_WS_TARGET", 768);
  int64_t target = 256 * per_cu;
  if (target > s_target) target = s_target;
  int64_t gx = cdiv64(target, cout / 64);
  const int64_t need = cdiv64(n_out * 9, 64) + 9;
  if (gx > need) gx = need;
  if (cin == 64)
    hipLaunchKernelGGL((k_ws3_gemm_bf3<1, 4>), dim3((unsigned)(gx * (cout / 64))), dim3(256), lds, st, in, ldi, v, cin, cout,
                       (const __bf16*)w_bf3, prod_scratch, (int)n_out, (int)target);
  else
    hipLaunchKernelGGL((k_ws3_gemm_bf3<2, 8>), dim3((unsigned)(gx * (cout / 64))), dim3(512), lds, st, in, ldi, v, cin, cout,
                       (const __bf16*)w_bf3, prod_scratch, (int)n_out, (int)target);
  PairViews rv;
  rv.hdr = v.hdr;
  rv.pair_in = nullptr;
  rv.pair_id = v.pair_id;
  hipLaunchKernelGGL(k_ws_reduce<9>, dim3((unsigned)cdiv64(n_out * (cout / 4), 256)), dim3(256), 0, st, prod_scratch, rv, n_out,
                     9, cout, scale, shift, residual, ldr, relu, out, ldo);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API int apr_spconv_ws_fwd(const float* in, int64_t ldi, const int32_t* counters, const void* plist, int64_t n_out,
                              int32_t K, int32_t cin, int32_t cout, const float* w_packed, const float* scale,
                              const float* shift, const float* residual, int64_t ldr, int32_t relu, float* out,
                              int64_t ldo, float* prod_scratch, void* stream) {
  return ws_fwd(in, ldi, counters, plist, n_out, K, cin, cout, w_packed, nullptr, scale, shift, residual, ldr, relu, out,
                ldo, prod_scratch, stream);
}

APR_API int apr_spconv_ws_fwd_bf3(const float* in, int64_t ldi, const int32_t* counters, const void* plist, int64_t n_out,
                                  int32_t K, int32_t cin, int32_t cout, const float* w_packed, const void* w_bf3,
                                  const float* scale, const float* shift, const float* residual, int64_t ldr,
                                  int32_t relu, float* out, int64_t ldo, float* prod_scratch, void* stream) {
  return ws_fwd(in, ldi, counters, plist, n_out, K, cin, cout, w_packed, w_bf3, scale, shift, residual, ldr, relu, out,
                ldo, prod_scratch, stream);
}
