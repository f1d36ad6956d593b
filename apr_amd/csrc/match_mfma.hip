// Feature-space nearest neighbour, filter + exact refine — same RESULT, bit for bit, as
// the brute-force kernel in match.hip (and oracle/nn_oracle.c), at ~1/8 of its cost.
//
// The exact answer is argmin_j d(i,j) with d the direct-form fp32 sum_c (a_c - b_c)^2 in the oracle's fixed order.
// Evaluating d for all n0*n1 pairs is ~6e9 packed-VALU lane-ops per KITTI pair (350 us).  Instead:
//   prep    rows -> split bf16: hi = bf16(a), lo = bf16(a - hi) (RNE), + fp32 norms |a|^2, |a|;
//   bound   bf16 MFMA (v_mfma_f32_16x16x32_bf16, K = 32: one instruction per 16x16 block of dot products and
//           term) accumulates a.b ~ lo.hi + hi.lo + hi.hi in fp32, so approx(i,j) = |a|^2 + |b|^2 - 2 a.b obeys the
//           RIGOROUS bound
//               |approx - d| <= eps(i,j) = 1e-4 |a||b| + 5e-5 (|a|^2 + |b|^2)   (used with |b| <= max_j |b_j|)
//           Derivation (worst case, every rounding at its bound): bf16 keeps 8 significant bits, so RNE gives
//           |x - bf16(x)| <= 2^-8 |x|, hence |lo| <= 2^-8 |a| and |a - hi - lo| <= 2^-16 |a| (vector norms).  The three
//           products drop lo.lo (<= 2^-16 |a||b|) and carry the two residuals (<= 2 * 2^-16 |a||b|): by Cauchy-Schwarz
//           the dot product is off by <= 3 * 2^-16 |a||b|, doubled by the factor 2 = 9.2e-5, plus <= 1.2e-5 for three
//           K=32 fp32 accumulations: 1.04e-4 |a||b|.  fp32 norm sums and the direct form's own rounding take
//           <= 3e-5 (|a|^2 + |b|^2) at C = 128 (less for smaller C); what is left of the second term,
//           2e-5 (|a|^2 + |b|^2) >= 4e-5 |a||b|, tops the first up to 1.4e-4 |a||b| >= 1.04e-4 |a||b|: the bound
//           holds with every error at its worst case at once (typical errors are ~10x smaller).  A single-bf16 bound (eps 0.008) is not
//           enough: untrained / weakly discriminative features put hundreds of targets inside that window.
//           Per query U_i = min over a quarter of the targets (the head of every chunk) of (approx + eps) is an upper
//           bound of the true minimum;
//   refine  the same MFMA pass again; only pairs with approx - eps <= U_i can be the arg-min (or tie with it): for
//           those — a handful per query — d is evaluated EXACTLY, in the oracle's order, and meets the others in the
//           same 64-bit atomicMin on (bits(d) << 32 | j) as the brute-force kernel.
// Candidates go to a list (wave-private LDS buffers, the wave's OWN region of the global list; a 64 x 16 block with 16 or
// more of them is listed as ONE dense block instead) and k_nn_resolve evaluates both kinds; if a list overflows — near-
// identical features everywhere — the same launch redoes the search exactly: the result is exact for ANY input.
// Four launches per search: prep (both sides), bound, refine, resolve.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr float kEpsRel = 1e-4f;     // with kEpsAbs: >= the 1.04e-4 |a||b| worst case derived above
constexpr float kEpsAbs = 5e-5f;    // fp32 norm sums, fp32 accumulation and the direct form's own rounding, C <= 128

__device__ inline unsigned short to_bf16_rne(float x) {
  unsigned u = __float_as_uint(x);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

// One group of C/4 lanes per row: 16-B loads, 8-B bf16 stores, shuffle-reduced norm.
// meta[i] = (kEpsRel |a|  or  |a| , |a|^2 + s, |a|^2 - s, 0) with s = kEpsAbs |a|^2; queries carry the eps factor and
// are stored as -2a, so the MFMA chain started from C = |b|^2 (+-s) yields |b|^2 - 2 a.b with no VALU work.
template <int C>
__global__ void k_nn_prep(const float* __restrict__ f, int64_t n, float len_scale, float row_scale,
                          unsigned short* __restrict__ fb, unsigned short* __restrict__ fl, f32x4* __restrict__ meta,
                          unsigned* __restrict__ u_init, unsigned long long* __restrict__ best_init,
                          unsigned* __restrict__ zero_me) {
  constexpr int LPR = C / 4;   // lanes per row
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = t / LPR;
  const int g = (int)(t - row * LPR);
  if (t == 0 && zero_me) {   // overflow flag, shared-list counter, dense-list counter
    zero_me[0] = 0u;
    zero_me[1] = 0u;
    zero_me[2] = 0u;   // dense-block list length
  }
  float s = 0.f;
  if (row < n) {
    f32x4 v = *reinterpret_cast<const f32x4*>(f + row * C + g * 4);
    s = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    v *= row_scale;   // queries: -2 (a power of two: the split is that of a, scaled), targets: 1
    ushort4 o, ol;
    o.x = to_bf16_rne(v[0]); o.y = to_bf16_rne(v[1]); o.z = to_bf16_rne(v[2]); o.w = to_bf16_rne(v[3]);
    ol.x = to_bf16_rne(v[0] - __uint_as_float((unsigned)o.x << 16));   // a - hi is exact in fp32
    ol.y = to_bf16_rne(v[1] - __uint_as_float((unsigned)o.y << 16));
    ol.z = to_bf16_rne(v[2] - __uint_as_float((unsigned)o.z << 16));
    ol.w = to_bf16_rne(v[3] - __uint_as_float((unsigned)o.w << 16));
    *reinterpret_cast<ushort4*>(fb + row * C + g * 4) = o;
    *reinterpret_cast<ushort4*>(fl + row * C + g * 4) = ol;
  }
#pragma unroll
  for (int d = 1; d < LPR; d <<= 1) s += __shfl_xor(s, d);
  if (row < n && g == 0) {
    const float nn = s;
    const float sl = kEpsAbs * nn;
    const float len = sqrtf(nn);
    meta[row] = (f32x4){len_scale * len, nn + sl, nn - sl, 0.f};
    if (u_init) u_init[row] = 0x7F800000u;
    if (best_init) best_init[row] = ~0ull;
  }
}

// Both sides of a search in ONE launch (round 5: one launch less per pair): threads [0, n0 * LPR) prepare the queries
// (scaled by -2, eps factor, U / best initialised), the rest the targets -- the same arithmetic per row as k_nn_prep
template <int C>
__global__ void k_nn_prep2(const float* __restrict__ f0, int64_t n0, const float* __restrict__ f1, int64_t n1,
                           unsigned short* __restrict__ qb, unsigned short* __restrict__ ql, f32x4* __restrict__ qmeta,
                           unsigned short* __restrict__ tb, unsigned short* __restrict__ tl, f32x4* __restrict__ tmeta,
                           unsigned* __restrict__ u_init, unsigned long long* __restrict__ best_init,
                           unsigned* __restrict__ zero_me) {
  constexpr int LPR = C / 4;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row_all = t / LPR;
  const int g = (int)(t - row_all * LPR);
  if (t == 0) {
    zero_me[0] = 0u;
    zero_me[1] = 0u;
    zero_me[2] = 0u;
  }
  // n0 * LPR is a multiple of LPR: the shuffle groups of LPR lanes never mix the two sides
  const bool is_q = row_all < n0;
  const int64_t row = is_q ? row_all : row_all - n0;
  const int64_t n = is_q ? n0 : n1;
  const float* f = is_q ? f0 : f1;
  const float row_scale = is_q ? -2.0f : 1.0f, len_scale = is_q ? kEpsRel : 1.0f;
  unsigned short* fb = is_q ? qb : tb;
  unsigned short* fl = is_q ? ql : tl;
  f32x4* meta = is_q ? qmeta : tmeta;
  float s = 0.f;
  if (row < n) {
    f32x4 v = *reinterpret_cast<const f32x4*>(f + row * C + g * 4);
    s = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    v *= row_scale;
    ushort4 o, ol;
    o.x = to_bf16_rne(v[0]); o.y = to_bf16_rne(v[1]); o.z = to_bf16_rne(v[2]); o.w = to_bf16_rne(v[3]);
    ol.x = to_bf16_rne(v[0] - __uint_as_float((unsigned)o.x << 16));
    ol.y = to_bf16_rne(v[1] - __uint_as_float((unsigned)o.y << 16));
    ol.z = to_bf16_rne(v[2] - __uint_as_float((unsigned)o.z << 16));
    ol.w = to_bf16_rne(v[3] - __uint_as_float((unsigned)o.w << 16));
    *reinterpret_cast<ushort4*>(fb + row * C + g * 4) = o;
    *reinterpret_cast<ushort4*>(fl + row * C + g * 4) = ol;
  }
#pragma unroll
  for (int d = 1; d < LPR; d <<= 1) s += __shfl_xor(s, d);
  if (row < n && g == 0) {
    const float nn = s;
    const float sl = kEpsAbs * nn;
    const float len = sqrtf(nn);
    meta[row] = (f32x4){len_scale * len, nn + sl, nn - sl, 0.f};
    if (is_q) {
      u_init[row] = 0x7F800000u;
      best_init[row] = ~0ull;
    }
  }
}

// REFINE = false: U_i = min_j (approx + eps).  REFINE = true: every pair with approx - eps <= U_i is appended to the
// candidate list (exact evaluation happens in k_nn_resolve).
// Workgroup = 4 waves x 64 queries; a wave holds its 4 query tiles (16 rows each, hi + lo) as MFMA A operands in
// registers.  Targets stream through LDS in chunks of 64 rows (hi, lo, meta), double buffered: the next chunk's
// global loads are in flight while the current one feeds the MFMAs, one barrier per chunk; B fragments are
// conflict-free ds_read_b128: the stage is laid out k-slice major, [k-slice][row 64][quad 4][16 B], so the 16 rows of a
// fragment are 16 x 64 B contiguous for every C (row-major [row][C] puts them C * 2 bytes apart: at C = 128 all 16 rows
// of a quad fall on the same banks, a 16-way conflict on every read -- APR's 128-d features ran the refine pass at a
// fifth of the rate of the 32-d ones per FLOP).
// Candidates: per-lane 16-bit masks, ranked by a DPP wave prefix sum into a wave-private LDS buffer (4-byte
// entries), flushed with coalesced stores into the wave's OWN region of the global list (kCandWave slots) — no
// global atomics at all (4000 waves bumping one counter at the end of the kernel cost ~15 us); the wave's count is
// written once.  A wave that fills its region spills into one shared list (one atomic per flush, heavy waves only);
// if that overflows too the flag arms the brute-force fallback.
// (Measured slower on the bench's 58 candidates per query: per-element wave masks with v_mbcnt ranks, and
// direct-to-global stores per candidate.)
constexpr int kCandBuf = 1280;    // LDS entries per wave (flushed above 256); a 16-target tile adds <= 64 x 16 = 1024
constexpr int kCandWave = 2048;   // global slots per wave
constexpr int kDenseMin = 16;     // candidates in one 64 x 16 block from which the block is evaluated densely
constexpr int kDlBuf = 64;        // dense-block ids a wave keeps in LDS before it has to flush on its own
constexpr unsigned kDenseListCap = 1u << 20;

template <int C, bool REFINE>
__global__ __launch_bounds__(256) void k_nn_mfma(const unsigned short* __restrict__ qb,
                                                 const unsigned short* __restrict__ ql,
                                                 const f32x4* __restrict__ qmeta, int64_t n0,
                                                 const unsigned short* __restrict__ tb,
                                                 const unsigned short* __restrict__ tl,
                                                 const f32x4* __restrict__ tmeta, int64_t n1, int chunk,
                                                 unsigned* __restrict__ U, unsigned long long* __restrict__ cand,
                                                 unsigned* __restrict__ cand_count, unsigned* __restrict__ overflow,
                                                 unsigned long long* __restrict__ shared_list, unsigned shared_cap,
                                                 unsigned* __restrict__ dense_list, unsigned* __restrict__ dense_count,
                                                 int dense_min) {
  // eps(i,j) <= kEpsRel |a_i| max_j|b_j| + s_i + s_j (max over THIS workgroup's target chunk) keeps the bound rigorous
  // and makes its |a||b| part a per-query constant, so the per-element epilogue is one min / one compare
  constexpr int KS = C / 32, QT = 4, TR = C >= 128 ? 32 : 64;   // TR target rows per LDS stage (wide rows: half the stage, two workgroups per CU)
  constexpr int LPT = (TR * C * 2) / 16 / 256;         // 16-B loads per thread per array and stage (C=32: 1)
  __shared__ __attribute__((aligned(16))) unsigned short s_h[2][TR * C];
  __shared__ __attribute__((aligned(16))) unsigned short s_l[2][TR * C];
  __shared__ f32x4 s_m[2][TR];
  __shared__ unsigned s_cand[REFINE ? 4 * kCandBuf : 1];   // (query within the workgroup) << 24 | (target - t0)
  __shared__ unsigned s_dl[REFINE ? 4 * kDlBuf : 1];       // ids of this workgroup's dense blocks, per wave
  __shared__ unsigned s_dn[4], s_dbase;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, lq = lane >> 4;
  const int64_t q0 = (int64_t)blockIdx.x * 256 + wave * 64;
  const bool wave_live = q0 < n0;        // dead waves still stage and hit the barriers
  bf16x8 a[QT][KS], al[QT][KS];
  float thr[QT][4], mn[QT][4];
  const int64_t t0 = (int64_t)blockIdx.y * chunk;
  // The bound pass looks at the leading 1/div of every chunk only (the `dense_min` argument carries div there): the
  // minimum over a SUBSET of the targets is still an upper bound of the minimum over all of them, so the refine pass
  // stays exact; it only lists a few more candidates (bench features, div 4: bound 43 -> 21 us, refine + exact +
  // dense +6 us).  Every chunk contributes, so the subset is spread over the whole target cloud.
  const int64_t t1 = min((long long)(t0 + (REFINE ? chunk : (chunk / (dense_min > 0 ? dense_min : 1) + 63) / 64 * 64)), (long long)n1);
  __shared__ float s_lmax[4];
  float len_max = 0.f;
  for (int64_t j = t0 + tid; j < t1; j += 256) len_max = fmaxf(len_max, tmeta[j][0]);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) len_max = fmaxf(len_max, __shfl_xor(len_max, d));
  if (lane == 0) s_lmax[wave] = len_max;
  __syncthreads();
  len_max = fmaxf(fmaxf(s_lmax[0], s_lmax[1]), fmaxf(s_lmax[2], s_lmax[3]));
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    int64_t ra = q0 + qt * 16 + l16;
    if (ra >= n0) ra = n0 - 1;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      a[qt][ks] = *reinterpret_cast<const bf16x8*>(qb + ra * C + ks * 32 + lq * 8);
      al[qt][ks] = *reinterpret_cast<const bf16x8*>(ql + ra * C + ks * 32 + lq * 8);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = q0 + qt * 16 + lq * 4 + r;
      const bool live = row < n0;
      const f32x4 m = qmeta[live ? row : n0 - 1];
      mn[qt][r] = __builtin_inff();
      // candidate test of the refine pass:  (|b|^2 - s_j - 2 a.b) <= U - (|a|^2 - s_i) + kEpsRel |a| max|b|
      thr[qt][r] = (REFINE && live) ? __uint_as_float(U[row]) - m[2] + m[0] * len_max : -__builtin_inff();
    }
  }
  int ncand = 0;                         // entries in this wave's LDS buffer (wave-uniform)
  int ndl = 0;                           // dense-block ids in this wave's LDS buffer (wave-uniform)
  unsigned* my_dl = s_dl + (REFINE ? wave * kDlBuf : 0);
  unsigned* my_cand = s_cand + (REFINE ? wave * kCandBuf : 0);

  const unsigned wave_id = (blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave;
  unsigned flushed = 0;                  // entries already in this wave's global region (wave-uniform)

  auto flush = [&]() {
    if (ncand == 0) return;
    auto entry = [&](int e) {
      const unsigned c = my_cand[e];
      return ((unsigned long long)(blockIdx.x * 256u + (c >> 24)) << 32) | (unsigned long long)(t0 + (c & 0xffffffu));
    };
    if (flushed + ncand <= (unsigned)kCandWave) {
      unsigned long long* dst = cand + (size_t)wave_id * kCandWave + flushed;
      for (int e = lane; e < ncand; e += 64) dst[e] = entry(e);
      flushed += ncand;
    } else {
      // this wave's region is full (heavy tail: clusters of near-identical features): the shared list takes the
      // rest, one atomic per flush; only if THAT overflows too the brute-force fallback is armed
      unsigned base = 0;
      if (lane == 0) base = atomicAdd(&overflow[1], (unsigned)ncand);
      base = __shfl(base, 0);
      if (base + (unsigned)ncand <= shared_cap) {
        for (int e = lane; e < ncand; e += 64) shared_list[base + e] = entry(e);
      } else if (lane == 0) {
        overflow[0] = 1u;
      }
    }
    ncand = 0;
  };

  // ---- staging registers for the next chunk
  uint4 rh[LPT], rl[LPT];
  f32x4 rm;
  auto fetch = [&](int64_t jt) {          // global -> registers, rows clamped to the range
#pragma unroll
    for (int u = 0; u < LPT; ++u) {
      const int e = u * 256 + tid;        // 16-B element of the [TR, C] bf16 slab
      int64_t j = jt + (e * 8) / C;
      if (j >= t1) j = t1 - 1;
      const int64_t off = j * C + (e * 8) % C;
      rh[u] = *reinterpret_cast<const uint4*>(tb + off);
      rl[u] = *reinterpret_cast<const uint4*>(tl + off);
    }
    if (tid < TR) {
      const int64_t j = jt + tid;
      rm = tmeta[j < t1 ? j : t1 - 1];
      if (j >= t1) {                      // a padded row can neither lower U nor become a candidate
        rm[1] = __builtin_inff();
        rm[2] = __builtin_inff();
      }
    }
  };
  auto commit = [&](int buf) {            // registers -> LDS
#pragma unroll
    for (int u = 0; u < LPT; ++u) {
      const int e = u * 256 + tid;
      const int row = (e * 8) / C, kc = ((e * 8) % C) / 8;      // 16-B piece kc of the row: k-slice kc >> 2, quad kc & 3
      const int at = (((kc >> 2) * TR + row) * 4 + (kc & 3)) * 8;
      *reinterpret_cast<uint4*>(&s_h[buf][at]) = rh[u];
      *reinterpret_cast<uint4*>(&s_l[buf][at]) = rl[u];
    }
    if (tid < TR) s_m[buf][tid] = rm;
  };

  if (t0 < t1) {
    fetch(t0);
    commit(0);
  }
  // every load issued so far (query fragments, meta) lands HERE: otherwise the compiler's counter bookkeeping puts
  // an s_waitcnt vmcnt(0) at their first use INSIDE the loop, which also drains the next chunk's prefetch every
  // iteration (vmcnt counts in order)
  __builtin_amdgcn_s_waitcnt(0x0f70);
  __syncthreads();
  int buf = 0;
  for (int64_t jt = t0; jt < t1; jt += TR, buf ^= 1) {
    const bool more = jt + TR < t1;
    if (more) fetch(jt + TR);
    if (wave_live) {
#pragma unroll
      for (int tt = 0; tt < TR / 16; ++tt) {
        if (jt + tt * 16 >= t1) break;
        bf16x8 b[KS], bl[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          b[ks] = *reinterpret_cast<const bf16x8*>(&s_h[buf][((ks * TR + tt * 16 + l16) * 4 + lq) * 8]);
          bl[ks] = *reinterpret_cast<const bf16x8*>(&s_l[buf][((ks * TR + tt * 16 + l16) * 4 + lq) * 8]);
        }
        const f32x4 tm = s_m[buf][tt * 16 + l16];
        // C operand = |b_j|^2 (+ s_j for the bound, - s_j for the refine test), the same for the lane's 4 rows; the
        // three terms of all four query tiles are issued term-major: 4 independent accumulators between consecutive
        // MFMAs on the same one (back-to-back dependent MFMAs would stall on the 8-pass latency)
        const float c0 = REFINE ? tm[2] : tm[1];
        const f32x4 cinit = {c0, c0, c0, c0};
        f32x4 acc[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
          acc[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[qt][0], b[0], cinit, 0, 0, 0);   // small terms first
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
          acc[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[qt][0], bl[0], acc[qt], 0, 0, 0);
#pragma unroll
        for (int ks = 1; ks < KS; ++ks) {
#pragma unroll
          for (int qt = 0; qt < QT; ++qt)
            acc[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[qt][ks], b[ks], acc[qt], 0, 0, 0);
#pragma unroll
          for (int qt = 0; qt < QT; ++qt)
            acc[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[qt][ks], bl[ks], acc[qt], 0, 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int qt = 0; qt < QT; ++qt)
            acc[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[qt][ks], b[ks], acc[qt], 0, 0, 0);
        if (!REFINE) {
#pragma unroll
          for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              // plain v_min_f32 (fminf adds a canonicalising v_max per operand: 3 VALU instead of 1)
              asm("v_min_f32 %0, %1, %2" : "=v"(mn[qt][r]) : "v"(mn[qt][r]), "v"(acc[qt][r]));
        } else {
          unsigned cmask = 0;
#pragma unroll
          for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (acc[qt][r] <= thr[qt][r]) cmask |= 1u << (qt * 4 + r);
          if (__any(cmask != 0)) {
            if (ncand > kCandBuf - 1024) flush();
            // wave prefix sum of the per-lane candidate counts -> ranks in the wave-private buffer
            const int c = __popc(cmask);
            const int incl = apr_wave_incl_scan(c);
            const int total = __builtin_amdgcn_readlane(incl, 63);
            if (total >= dense_min) {
              // a cluster of near-identical features: listing (and atomically reducing) dozens of pairs of this
              // 64-query x 16-target block costs more than evaluating the whole block once -> k_nn_dense.
              // The block's id goes to the wave's LDS buffer; the WORKGROUP appends its ids to the global list with one
              // atomic at the end of the kernel (a counter bumped per dense tile stalled every one of them ~1 us; flag
              // bytes + a compaction kernel were a store per tile and one more launch).
              if (ndl == kDlBuf) {       // only with more than kDlBuf dense tiles in one wave's chunk: flush alone
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(dense_count, (unsigned)ndl);
                base = __builtin_amdgcn_readfirstlane(base);
                for (int e = lane; e < ndl; e += 64) {
                  if (base + e < kDenseListCap) dense_list[base + e] = my_dl[e]; else overflow[0] = 1u;
                }
                ndl = 0;
              }
              if (lane == 0) my_dl[ndl] = wave_id * (unsigned)(chunk >> 4) + (unsigned)((jt - t0) >> 4) + tt;
              ++ndl;
              continue;
            }
            int pos = ncand + incl - c;
            const unsigned j = (unsigned)(jt - t0) + tt * 16 + l16;       // chunk < 2^24 (checked on the host)
            while (cmask) {
              const int e = __ffs((int)cmask) - 1;
              cmask &= cmask - 1;
              const unsigned i = wave * 64 + (e >> 2) * 16 + lq * 4 + (e & 3);
              my_cand[pos++] = (i << 24) | j;
            }
            ncand += total;
          }
        }
      }
    }
    if (more) commit(buf ^ 1);
    __syncthreads();
  }
  if (REFINE) {
    flush();
    if (lane == 0) {
      cand_count[wave_id] = flushed;
      s_dn[wave] = (unsigned)ndl;
    }
    __syncthreads();
    if (tid == 0) {
      const unsigned total = s_dn[0] + s_dn[1] + s_dn[2] + s_dn[3];
      s_dbase = total ? atomicAdd(dense_count, total) : 0u;
    }
    __syncthreads();
    unsigned base = s_dbase;
    for (int w = 0; w < wave; ++w) base += s_dn[w];
    for (int e = lane; e < ndl; e += 64) {
      if (base + e < kDenseListCap) dense_list[base + e] = my_dl[e]; else overflow[0] = 1u;   // the fallback redoes the search
    }
  }
  if (!REFINE && wave_live) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = mn[qt][r];
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) v = fminf(v, __shfl_xor(v, d));
        const int64_t row = q0 + qt * 16 + lq * 4 + r;
        if (l16 == 0 && row < n0) {
          // + (|a|^2 + s_i) + kEpsRel |a| max|b|; the true minimum is >= 0, so clamping keeps U an upper bound
          // and its bits ordered
          const f32x4 m = qmeta[row];
          const float u = fmaxf(v + m[1] + m[0] * len_max, 0.f);
          atomicMin(&U[row], __float_as_uint(u));
        }
      }
  }
}

// k_nn_resolve: everything behind the refine pass in ONE launch (round 5: the candidate kernel, the flag compaction, the
// dense-block kernel and the predicated brute-force launch were four).  Roles by block index:
//   blocks [0, nwg)            the 4 wave regions of refine workgroup b: one thread per candidate pair, the exact direct
//                              form in the oracle's order, 64-bit atomicMin like the brute-force kernel;
//   blocks [nwg, nexact)       the shared overflow list, the same way;
//   blocks [nexact, gridDim)   the dense blocks (64 queries x 16 targets the refine pass listed instead of >= 16 single
//                              pairs): the waves take the listed blocks in turn (grid stride), so a cluster of blocks is
//                              spread over the whole chip.  Per block: lane = query with its row in registers; the 16 target
//                              rows go through LDS ONCE (one vector load per lane instead of 16 dependent scalar fetches)
//                              and are read back as broadcasts; the direct form for all 1024 pairs, a running strict-<
//                              minimum per lane (ascending j: ties keep the smaller index), ONE atomicMin per query.
// If a list overflowed (overflow[0], final when the refine pass has ended) every wave of the grid turns to the fallback
// instead: query tile x slice of the targets through the same tile evaluator -- exact for ANY input, about twice the time
// of the dedicated brute-force kernel, and it has never been taken on encoder features.
template <int C>
__device__ inline void nn_tile_eval(float* my_t, const f32x4* xv, const float* __restrict__ f1, int64_t j0, int64_t j1,
                                    int lane, float& bd, int& bj) {
  // targets -> LDS (rows past j1 repeat row j1 - 1: a re-read never wins the strict <)
  constexpr int VPR = C / 4;   // 16-B vectors per row
#pragma unroll
  for (int v = lane; v < 16 * VPR; v += 64) {
    const int r = v / VPR, g = v - r * VPR;
    const int64_t row = (j0 + r < j1) ? j0 + r : j1 - 1;
    *reinterpret_cast<f32x4*>(my_t + r * C + g * 4) = *reinterpret_cast<const f32x4*>(f1 + row * C + g * 4);
  }
  // the wave's own LDS writes are visible to its reads in program order (one wave, LDS ops complete in order)
#pragma unroll 2
  for (int jj = 0; jj < 16; ++jj) {
    // plain v_sub_f32 + (compiler-packed) fma: packed adds (y + (-x)) measured SLOWER here (23 -> 30 us), packed
    // fp32 issues at the same lane rate and the pairing costs registers
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int g = 0; g < C / 4; ++g) {
      const f32x4 yv = *reinterpret_cast<const f32x4*>(my_t + jj * C + g * 4);   // broadcast read
      const float d0 = xv[g][0] - yv[0], d1 = xv[g][1] - yv[1];
      const float d2 = xv[g][2] - yv[2], d3 = xv[g][3] - yv[3];
      s0 = fmaf(d0, d0, s0);
      s1 = fmaf(d1, d1, s1);
      s2 = fmaf(d2, d2, s2);
      s3 = fmaf(d3, d3, s3);
    }
    const float dd = (s0 + s1) + (s2 + s3);
    const int64_t j = (j0 + jj < j1) ? j0 + jj : j1 - 1;
    if (dd < bd) {
      bd = dd;
      bj = (int)j;
    }
  }
}

template <int C>
__global__ __launch_bounds__(256) void k_nn_resolve(const unsigned long long* __restrict__ cand,
                                                    const unsigned* __restrict__ cand_count, unsigned nwg, unsigned nexact,
                                                    const unsigned long long* __restrict__ shared_list,
                                                    const unsigned* __restrict__ overflow, unsigned shared_cap,
                                                    const unsigned* __restrict__ dense_list, int chunk, unsigned qwaves,
                                                    const float* __restrict__ f0, int64_t n0,
                                                    const float* __restrict__ f1, int64_t n1,
                                                    unsigned long long* __restrict__ best) {
  __shared__ __attribute__((aligned(16))) float s_t[4][16 * C];   // per wave: a block's 16 target rows
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (overflow[0] != 0u) {
    // fallback: wave = (query tile, slice of the 16-target tiles), running minimum in registers, one atomicMin per query
    const int64_t nqt = (n0 + 63) >> 6, ntt = (n1 + 15) >> 4;
    const int64_t nw = (int64_t)gridDim.x * 4;
    const int64_t slices = nw / nqt > 0 ? nw / nqt : 1;
    const int64_t tps = (ntt + slices - 1) / slices;
    for (int64_t w = (int64_t)blockIdx.x * 4 + wave; w < nqt * slices; w += nw) {
      const int64_t q = (w % nqt) * 64 + lane;
      const int64_t tt0 = (w / nqt) * tps, tt1 = min((long long)(tt0 + tps), (long long)ntt);
      const float* x = f0 + (q < n0 ? q : n0 - 1) * C;
      f32x4 xv[C / 4];
#pragma unroll
      for (int g = 0; g < C / 4; ++g) xv[g] = *reinterpret_cast<const f32x4*>(x + g * 4);
      float bd = __builtin_inff();
      int bj = 0x7fffffff;
      for (int64_t tt = tt0; tt < tt1; ++tt)
        nn_tile_eval<C>(s_t[wave], xv, f1, tt * 16, min((long long)(tt * 16 + 16), (long long)n1), lane, bd, bj);
      if (q < n0 && bj != 0x7fffffff) {
        const unsigned long long mine = ((unsigned long long)__float_as_uint(bd) << 32) | (unsigned)bj;
        if (mine < __builtin_nontemporal_load(&best[q])) atomicMin(&best[q], mine);
      }
    }
    return;
  }
  if (blockIdx.x < nexact) {
    const bool shared = blockIdx.x >= nwg;
    const unsigned wave_id = blockIdx.x * 4 + (threadIdx.x >> 6);
    const unsigned n = shared ? min(overflow[1], shared_cap) : min(cand_count[wave_id], (unsigned)kCandWave);
    const unsigned long long* src = shared ? shared_list : cand + (size_t)wave_id * kCandWave;
    const unsigned first = shared ? (blockIdx.x - nwg) * 256u + threadIdx.x : (threadIdx.x & 63u);
    const unsigned step = shared ? (nexact - nwg) * 256u : 64u;
    for (unsigned t = first; t < n; t += step) {
      const unsigned long long ij = src[t];
      const int64_t i = (int64_t)(ij >> 32), j = (int64_t)(ij & 0xffffffffull);
      const float* x = f0 + i * C;
      const float* y = f1 + j * C;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
      for (int g = 0; g < C / 4; ++g) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + g * 4);
        const f32x4 yv = *reinterpret_cast<const f32x4*>(y + g * 4);
        const float d0 = xv[0] - yv[0], d1 = xv[1] - yv[1], d2 = xv[2] - yv[2], d3 = xv[3] - yv[3];
        s0 = fmaf(d0, d0, s0);
        s1 = fmaf(d1, d1, s1);
        s2 = fmaf(d2, d2, s2);
        s3 = fmaf(d3, d3, s3);
      }
      const float d = (s0 + s1) + (s2 + s3);
      // best[i] only ever decreases: a (possibly stale) read that is already <= this pair proves it cannot win and
      // saves the atomic — clusters put thousands of candidates on the same few hundred queries
      const unsigned long long mine = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)j;
      if (mine < __builtin_nontemporal_load(&best[i])) atomicMin(&best[i], mine);
    }
    return;
  }
  const unsigned tpc = (unsigned)chunk >> 4;
  const unsigned n = min(overflow[2], kDenseListCap);
  const unsigned nwaves = (gridDim.x - nexact) * 4;
  for (unsigned e = (blockIdx.x - nexact) * 4 + wave; e < n; e += nwaves) {
    const unsigned fb = __builtin_amdgcn_readfirstlane(dense_list[e]);
    const unsigned wave_id = fb / tpc, tile = fb % tpc;
    const unsigned dq = (wave_id % qwaves) * 64u;
    const unsigned t0 = (wave_id / qwaves) * (unsigned)chunk;
    const int64_t q = (int64_t)dq + lane;
    const int64_t j0 = (int64_t)t0 + (int64_t)tile * 16;
    const int64_t j1 = min(min((long long)(j0 + 16), (long long)((int64_t)t0 + chunk)), (long long)n1);
    const float* x = f0 + (q < n0 ? q : n0 - 1) * C;
    f32x4 xv[C / 4];
#pragma unroll
    for (int g = 0; g < C / 4; ++g) xv[g] = *reinterpret_cast<const f32x4*>(x + g * 4);
    float bd = __builtin_inff();
    int bj = 0x7fffffff;
    nn_tile_eval<C>(s_t[wave], xv, f1, j0, j1, lane, bd, bj);
    if (q < n0 && bj != 0x7fffffff) {
      const unsigned long long mine = ((unsigned long long)__float_as_uint(bd) << 32) | (unsigned)bj;
      if (mine < __builtin_nontemporal_load(&best[q])) atomicMin(&best[q], mine);
    }
  }
}

size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

int64_t shared_capacity(int64_t n0) {   // shared overflow list: 64 candidates per query
  int64_t c = 64 * n0;
  if (c < 65536) c = 65536;
  return c < (1ll << 31) ? c : (1ll << 31) - 1;
}

// refine grid: query blocks of 256 x target chunks (multiples of 64 rows) so that ONE resident round of workgroups exists:
// at C = 32 three workgroups fit a CU (~768: the count of earlier rounds, kept), at C = 64 / 128 two (176 / 244 registers),
// and there the chunk count is the largest that still fits 512 -- 14 k x 14 k x 128: 9 chunks (495 workgroups) 310 us,
// 14 (770: a second, half-empty round) 356, 11-13 354-360, 7 348; at C = 32 9 / 13 / 14 chunks are within 1 us of each other.
void nn_grid(int64_t n0, int64_t n1, int feat_c, int64_t* qblocks, int64_t* chunk, int64_t* nchunk) {
  static const int s_wgs = env_int("APR_NN_GRID_WGS", 768);      // A/B switches
  static const int s_want = env_int("APR_NN_WANT", 0);
  *qblocks = cdiv64(n0, 256);
  int64_t want = cdiv64(s_wgs > 0 && s_wgs < 768 ? s_wgs : 768, *qblocks);
  if (feat_c >= 64) {
    const int64_t fit = 512 / *qblocks;
    if (fit >= 1 && fit < want) want = fit;
  }
  if (s_want > 0 && s_want <= want) want = s_want;
  int64_t c = cdiv64(cdiv64(n1, want), 64) * 64;
  if (c < 256) c = 256;
  if (c >= (1 << 24)) c = (1 << 24) - 64;
  *chunk = c;
  *nchunk = cdiv64(n1, c);
}

size_t dense_list_bytes(int64_t qblocks, int64_t nchunk, int64_t chunk) {
  uint64_t nflag = (uint64_t)(qblocks * 4) * (uint64_t)nchunk * (uint64_t)(chunk / 16);
  if (nflag > kDenseListCap) nflag = kDenseListCap;
  return al256((size_t)nflag * 4 + 64);
}

template <int C>
int run_fast(const float* f0, int64_t n0, const float* f1, int64_t n1, unsigned long long* best, char* p,
             hipStream_t st) {
  int64_t qblocks, chunk, nchunk;
  nn_grid(n0, n1, C, &qblocks, &chunk, &nchunk);
  const int64_t nwaves = qblocks * nchunk * 4;
  unsigned short* qb = (unsigned short*)p;  p += al256((size_t)n0 * C * 2);
  unsigned short* ql = (unsigned short*)p;  p += al256((size_t)n0 * C * 2);
  unsigned short* tb = (unsigned short*)p;  p += al256((size_t)n1 * C * 2);
  unsigned short* tl = (unsigned short*)p;  p += al256((size_t)n1 * C * 2);
  f32x4* qmeta = (f32x4*)p;                 p += al256((size_t)n0 * 16);
  f32x4* tmeta = (f32x4*)p;                 p += al256((size_t)n1 * 16);
  unsigned* U = (unsigned*)p;               p += al256((size_t)n0 * 4);
  unsigned* overflow = (unsigned*)p;        p += 256;
  unsigned* cand_count = (unsigned*)p;      p += al256((size_t)nwaves * 4);
  unsigned long long* cand = (unsigned long long*)p;   p += al256((size_t)nwaves * kCandWave * 8);
  unsigned long long* shared_list = (unsigned long long*)p;   p += al256((size_t)shared_capacity(n0) * 8);
  // (the flag bytes of earlier rounds sat here; the scratch bound still counts them)
  p += al256((size_t)(qblocks * 4) * (size_t)nchunk * (size_t)(chunk / 16) + 64);
  unsigned* dense_list = (unsigned*)p;                        // ids of the dense blocks, appended by the refine pass
  const unsigned shared_cap = (unsigned)shared_capacity(n0);
  constexpr int LPR = C / 4;
  if ((uint64_t)qblocks * 4 * (uint64_t)nchunk * (uint64_t)(chunk / 16) + 64 >= (1ull << 32)) {
    apr_set_error("apr_feature_nn_fast: problem too large for the dense-block ids");
    return APR_EINVAL;
  }
  hipLaunchKernelGGL((k_nn_prep2<C>), dim3((unsigned)cdiv64((n0 + n1) * LPR, 256)), dim3(256), 0, st, f0, n0, f1, n1, qb, ql,
                     qmeta, tb, tl, tmeta, U, best, overflow);
  static const int s_dense_min = env_int("APR_NN_DENSE_MIN", kDenseMin);
  static const int s_bound_div = env_int("APR_NN_BOUND_DIV", 4);   // bound pass over 1/div of every target chunk
  const dim3 grid((unsigned)qblocks, (unsigned)nchunk);
  hipLaunchKernelGGL((k_nn_mfma<C, false>), grid, dim3(256), 0, st, qb, ql, qmeta, n0, tb, tl, tmeta, n1, (int)chunk, U,
                     cand, cand_count, overflow, shared_list, shared_cap, dense_list, overflow + 2, s_bound_div);
  hipLaunchKernelGGL((k_nn_mfma<C, true>), grid, dim3(256), 0, st, qb, ql, qmeta, n0, tb, tl, tmeta, n1, (int)chunk, U,
                     cand, cand_count, overflow, shared_list, shared_cap, dense_list, overflow + 2, s_dense_min);
  // candidates, dense blocks and -- if a list overflowed -- the exact fallback, in one launch: exact for ANY input
  const unsigned nwg = (unsigned)(qblocks * nchunk);
  const unsigned nexact = nwg + 256;
  hipLaunchKernelGGL((k_nn_resolve<C>), dim3(nexact + 1024), dim3(256), 0, st, cand, cand_count, nwg, nexact, shared_list,
                     overflow, shared_cap, dense_list, (int)chunk, (unsigned)(qblocks * 4), f0, n0, f1, n1, best);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

}  // namespace

// An upper bound of what run_fast carves, MONOTONE in n0 and in n1: a caller that sizes one buffer for a batch of
// pairs from the largest n0 and the largest n1 (apr_match_pose_batch) must be covered for every pair of the batch,
// and the exact need is not monotone (the refine grid trades query blocks against target chunks: a smaller n0 can
// mean more chunks).  Bounds: qblocks * nchunk <= 768 + qblocks (nchunk <= want = ceil(768 / qblocks));
// nchunk * chunk <= 2 * n1 + 320 (chunk <= max(256, n1 / want + 64)).
APR_API size_t apr_feature_nn_fast_scratch_bytes(int64_t n0, int64_t n1, int32_t c) {
  if (n0 < 0 || n1 < 0 || c <= 0) return 0;
  if (n0 < 1) n0 = 1;
  if (n1 < 1) n1 = 1;
  const size_t qblocks = (size_t)cdiv64(n0, 256);
  const size_t nwaves = 4 * (768 + qblocks);
  const size_t nflag = qblocks * 4 * ((2 * (size_t)n1 + 320) / 16 + 1);
  const size_t nlist = nflag > kDenseListCap ? (size_t)kDenseListCap : nflag;
  return 2 * al256((size_t)n0 * c * 2) + 2 * al256((size_t)n1 * c * 2) + al256((size_t)n0 * 16) + al256((size_t)n1 * 16) +
         al256((size_t)n0 * 4) + 256 + al256(nwaves * 4) + al256(nwaves * kCandWave * 8) +
         al256((size_t)shared_capacity(n0) * 8) + al256(nflag + 64) + al256(nlist * 4 + 64) + 512;
}

APR_API int apr_feature_nn_fast(const float* f0, int64_t n0, const float* f1, int64_t n1, int32_t c, uint64_t* best,
                                void* scratch, size_t scratch_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APR_CHECK_ARG(n0 >= 0 && n1 > 0 && n1 < (1ll << 31) && n0 < (1ll << 31), "apr_feature_nn_fast: bad shape");
  APR_CHECK_ARG(c == 32 || c == 64 || c == 128, "apr_feature_nn_fast: c=%d, supported: 32, 64, 128", c);
  // the scratch bound (apr_feature_nn_fast_scratch_bytes) assumes at most ceil(768 / qblocks) target chunks, which holds
  // while one chunk (capped at 2^24 - 64 rows) can cover n1 / that count
  APR_CHECK_ARG(n1 < (1ll << 24) - 64, "apr_feature_nn_fast: n1 = %lld, supported below 2^24 - 64 targets", (long long)n1);
  APR_CHECK_ARG(((((uintptr_t)f0) | ((uintptr_t)f1)) & 15) == 0, "apr_feature_nn_fast: 16-byte aligned rows required");
  APR_CHECK_ARG(scratch_bytes >= apr_feature_nn_fast_scratch_bytes(n0, n1, c), "apr_feature_nn_fast: scratch too small");
  if (n0 == 0) return APR_OK;
  char* p = (char*)(((uintptr_t)scratch + 255) & ~(uintptr_t)255);
  unsigned long long* b = (unsigned long long*)best;
  if (c == 32) return run_fast<32>(f0, n0, f1, n1, b, p, st);
  if (c == 64) return run_fast<64>(f0, n0, f1, n1, b, p, st);
  return run_fast<128>(f0, n0, f1, n1, b, p, st);
}
