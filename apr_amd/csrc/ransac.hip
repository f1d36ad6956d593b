// RANSAC + Kabsch pose fit and IRLS pose refinement on the GPU
// (K13).
//
// The reference hands N<=~15 k correspondences to open3d's CPU RANSAC
// (FCGF_APR/scripts/test_apr.py:148-156: ransac_n = 4, edge-length checker 0.9,
// distance checker = voxel_size, TransformationEstimationPointToPoint(False),
// up to 4 000 000 iterations).  Here one thread owns one hypothesis: a
// counter-based RNG picks its 4 correspondences (so the CPU oracle can replay the
// exact stream) and the cheap edge-length test runs first (k_sample_check; the
// ~99.9 % rejected there never reach a square root: squared lengths are compared
// with a 1e-12 guard band and only the band replays the oracle's sqrt form).
// Survivors are COMPACTED (wave-aggregated append) so that the expensive part
// runs on dense waves (k_fit_check): closed-form Kabsch (Horn quaternion, Jacobi
// eigen-solve of the 4x4 in fp64 registers), the distance checker, and an append
// to the hypothesis list.  A persistent grid of waves then scores each hypothesis
// against all correspondences (32-B packed {source, target} records, fp64
// transform, wave shuffle reduction), and a single workgroup keeps the
// lexicographic best (inliers desc, rmse asc, iteration asc).
//
// sqrt(d2) < m is evaluated as d2 < T2 with T2 = the smallest double whose
// correctly-rounded sqrt is >= m (found on the host): sqrt is monotone, so the two
// predicates are identical for every d2, and no fp64 sqrt runs per point.
#include <atomic>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>

#include "common.h"

namespace {

struct Hyp {
  double T[12];  // row-major [R | t], 3x4
  long long it;
  int inliers;
  int pad;
  double err2;
};

__host__ __device__ inline uint64_t splitmix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__host__ __device__ inline uint32_t sample_index(uint64_t seed, uint64_t it, int slot, uint32_t n) {
  uint64_t z = splitmix64(seed + (it * 4ull + (uint64_t)slot + 1ull) * 0x9E3779B97F4A7C15ull);
  return (uint32_t)(((z >> 32) * (uint64_t)n) >> 32);
}

// Horn's closed-form absolute orientation: R,t minimising sum |R s_i + t - t_i|^2.
// Equals Kabsch / Eigen::umeyama(with_scaling=false) whenever the optimum is unique.
__device__ inline void kabsch4(const double s[4][3], const double t[4][3], double T[12]) {
  double ms[3] = {0, 0, 0}, mt[3] = {0, 0, 0};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      ms[d] += s[i][d];
      mt[d] += t[i][d];
    }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    ms[d] *= 0.25;
    mt[d] *= 0.25;
  }
  double S[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) S[a][b] += (s[i][a] - ms[a]) * (t[i][b] - mt[b]);
  double A[4][4];
  A[0][0] = S[0][0] + S[1][1] + S[2][2];
  A[0][1] = S[1][2] - S[2][1];
  A[0][2] = S[2][0] - S[0][2];
  A[0][3] = S[0][1] - S[1][0];
  A[1][1] = S[0][0] - S[1][1] - S[2][2];
  A[1][2] = S[0][1] + S[1][0];
  A[1][3] = S[2][0] + S[0][2];
  A[2][2] = -S[0][0] + S[1][1] - S[2][2];
  A[2][3] = S[1][2] + S[2][1];
  A[3][3] = -S[0][0] - S[1][1] + S[2][2];
  A[1][0] = A[0][1]; A[2][0] = A[0][2]; A[3][0] = A[0][3];
  A[2][1] = A[1][2]; A[3][1] = A[1][3]; A[3][2] = A[2][3];
  double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  // cyclic Jacobi, fully unrolled so A and V stay in registers
  for (int sweep = 0; sweep < 12; ++sweep) {
    double off = 0.0;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int q2 = p + 1; q2 < 4; ++q2) off += A[p][q2] * A[p][q2];
    double diag = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2] + A[3][3] * A[3][3];
    if (off <= 1e-32 * diag || off == 0.0) break;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int q2 = p + 1; q2 < 4; ++q2) {
        double apq = A[p][q2];
        if (apq != 0.0) {
          double theta = (A[q2][q2] - A[p][p]) / (2.0 * apq);
          double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
          double c = 1.0 / sqrt(tt * tt + 1.0), sn = tt * c;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            double akp = A[k][p], akq = A[k][q2];
            A[k][p] = c * akp - sn * akq;
            A[k][q2] = sn * akp + c * akq;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            double apk = A[p][k], aqk = A[q2][k];
            A[p][k] = c * apk - sn * aqk;
            A[q2][k] = sn * apk + c * aqk;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            double vkp = V[k][p], vkq = V[k][q2];
            V[k][p] = c * vkp - sn * vkq;
            V[k][q2] = sn * vkp + c * vkq;
          }
        }
      }
  }
  // eigenvector of the largest eigenvalue
  double best = A[0][0];
  double qw = V[0][0], qx = V[1][0], qy = V[2][0], qz = V[3][0];
#pragma unroll
  for (int j = 1; j < 4; ++j)
    if (A[j][j] > best) {
      best = A[j][j];
      qw = V[0][j]; qx = V[1][j]; qy = V[2][j]; qz = V[3][j];
    }
  double nq = 1.0 / sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
  qw *= nq; qx *= nq; qy *= nq; qz *= nq;
  double R[3][3];
  R[0][0] = 1 - 2 * (qy * qy + qz * qz); R[0][1] = 2 * (qx * qy - qw * qz); R[0][2] = 2 * (qx * qz + qw * qy);
  R[1][0] = 2 * (qx * qy + qw * qz); R[1][1] = 1 - 2 * (qx * qx + qz * qz); R[1][2] = 2 * (qy * qz - qw * qx);
  R[2][0] = 2 * (qx * qz - qw * qy); R[2][1] = 2 * (qy * qz + qw * qx); R[2][2] = 1 - 2 * (qx * qx + qy * qy);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    T[a * 4 + 0] = R[a][0]; T[a * 4 + 1] = R[a][1]; T[a * 4 + 2] = R[a][2];
    T[a * 4 + 3] = mt[a] - (R[a][0] * ms[0] + R[a][1] * ms[1] + R[a][2] * ms[2]);
  }
}

// rec[2i] = (source_i, 0), rec[2i+1] = (target_i = xyz1[corr[i]], 0): one 32-B record per correspondence, so a
// random sample costs two 16-B loads from one cache line instead of six scattered dwords.
// rec2: the same correspondences for the count kernel's SCALAR loads, two per 48-B row, structure of arrays:
//   [sx0 sx1 | sy0 sy1 | sz0 sz1 | tx0 tx1 | ty0 ty1 | tz0 tz1], padded to whole mini-chunks of kMini rows with a
// correspondence that no transform maps within reach (target at 3e18).  maxn2: bits of max(|s|^2, |t|^2) (fp32, >= 0).
constexpr int kMini = 32;                        // rows of rec2 (= 64 correspondences) per mini-chunk
__host__ __device__ inline int64_t rec2_rows(int64_t n0) { return ((n0 + 1) / 2 + kMini - 1) / kMini * kMini; }

// corr_out (nullable): `corr` holds the NN search's PACKED results (bits(d^2) << 32 | j) and the plain indices are written
// here on the way -- the batch path's former apr_nn_unpack launch (round 5)
__global__ void k_pack_pairs(const float* __restrict__ xyz0, const float* __restrict__ xyz1, int64_t n1,
                             const long long* __restrict__ corr, int64_t n0, float4* __restrict__ rec,
                             float* __restrict__ rec2, unsigned* __restrict__ maxn2, float* __restrict__ rec2_live,
                             long long* __restrict__ corr_out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float sx = 0.f, sy = 0.f, sz = 0.f, tx = 3e18f, ty = 3e18f, tz = 3e18f, m = 0.f;
  if (i < n0) {
    long long j = corr[i];
    if (corr_out) {
      j &= 0xFFFFFFFFll;
      corr_out[i] = j;
    }
    if (j < 0 || j >= n1) j = 0;
    sx = xyz0[3 * i]; sy = xyz0[3 * i + 1]; sz = xyz0[3 * i + 2];
    tx = xyz1[3 * j]; ty = xyz1[3 * j + 1]; tz = xyz1[3 * j + 2];
    rec[2 * i] = make_float4(sx, sy, sz, 0.f);
    rec[2 * i + 1] = make_float4(tx, ty, tz, 0.f);
    m = fmaxf(sx * sx + sy * sy + sz * sz, tx * tx + ty * ty + tz * tz);
  }
  if (i < 2 * rec2_rows(n0)) {
    float* row = rec2 + (i >> 1) * 12 + (i & 1);
    row[0] = sx; row[2] = sy; row[4] = sz; row[6] = tx; row[8] = ty; row[10] = tz;
    // the pruned list of the count path (k_prune) starts out as all padding: whatever it does not overwrite is a
    // correspondence nothing reaches, so its last mini-chunk needs no finishing pass
    float* lrow = rec2_live + (i >> 1) * 12 + (i & 1);
    lrow[0] = 0.f; lrow[2] = 0.f; lrow[4] = 0.f; lrow[6] = 3e18f; lrow[8] = 3e18f; lrow[10] = 3e18f;
  }
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(maxn2, __float_as_uint(m));
}

__device__ inline void load_samples(const float4* __restrict__ rec, uint64_t seed, long long it, uint32_t n0,
                                    double s[4][3], double t[4][3]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t i = sample_index(seed, (uint64_t)it, j, n0);
    const float4 a = rec[2 * (int64_t)i], b = rec[2 * (int64_t)i + 1];
    s[j][0] = (double)a.x; s[j][1] = (double)a.y; s[j][2] = (double)a.z;
    t[j][0] = (double)b.x; t[j][1] = (double)b.y; t[j][2] = (double)b.z;
  }
}

// CorrespondenceCheckerBasedOnEdgeLength: reject if |s_a s_b| < r |t_a t_b| or |t_a t_b| < r |s_a s_b| (lengths via
// fp64 sqrt in the oracle).  Squared form first; inside the 1e-12 guard band the oracle's exact expression decides.
__device__ inline bool edge_reject(double a2, double b2, double r, double r2) {
  const double rb = b2 * r2;
  if (a2 < rb * (1.0 - 1e-12)) return true;
  if (a2 > rb * (1.0 + 1e-12)) return false;
  return sqrt(a2) < sqrt(b2) * r;
}

constexpr int kCandLists = 64;     // candidate sub-lists (power of two)

// The exact staged checker of one iteration: the checker is a conjunction over the 6 edges, so its value does not depend on
// the order they are looked at.  Correspondences 0 and 1 first; only the lanes whose first edge passes (a few per cent on
// real correspondence sets) fetch the third, and only the survivors of its two edges the fourth: ~4.3 instead of 8 16-B
// gathers per iteration.
__device__ inline bool sample_passes(const float4* __restrict__ rec, uint32_t n0, double edge_ratio, long long it,
                                     uint64_t seed) {
  float4 sf[4], tf[4];
  const double r2 = edge_ratio * edge_ratio;
  const float r2f = (float)r2;
  auto fetch = [&](int j) {
    const uint32_t i = sample_index(seed, (uint64_t)it, j, n0);
    sf[j] = rec[2 * (int64_t)i];
    tf[j] = rec[2 * (int64_t)i + 1];
  };
  // The squared lengths first in fp32: the coordinates ARE fp32, a difference is off by <= 1 ulp of itself, a sum of three
  // squares by < 8 u relative (u = 2^-24; all terms positive) -- so A = fl(|s_a - s_b|^2) and B likewise decide
  // a^2 < r^2 b^2 whenever they differ from equality by more than 4e-6 relative (8 u on each side + the rounding of r^2,
  // with a factor 4 to spare).  Only an edge inside that band goes through the oracle's fp64 expression: the same
  // decisions, a tenth of the fp64 work of the queue's exact checks (36 of k_sample_screen's 78 us at 30 % true matches).
  auto edge_ok = [&](int a, int b) {
    const float dsx = sf[a].x - sf[b].x, dsy = sf[a].y - sf[b].y, dsz = sf[a].z - sf[b].z;
    const float dtx = tf[a].x - tf[b].x, dty = tf[a].y - tf[b].y, dtz = tf[a].z - tf[b].z;
    const float A = dsx * dsx + dsy * dsy + dsz * dsz, B = dtx * dtx + dty * dty + dtz * dtz;
    constexpr float lo = 1.f - 4e-6f, hi = 1.f + 4e-6f;
    // reject if a^2 < r^2 b^2 or b^2 < r^2 a^2.  Outside the normal range (sums that overflowed, or so small that their
    // relative error is no longer 8 u -- exact zeros are fine) nothing is decided here
    const bool sound = (A == 0.f || A >= 1e-30f) && (B == 0.f || B >= 1e-30f) && A < 3e37f && B < 3e37f;
    if (sound && (A < r2f * B * lo || B < r2f * A * lo)) return false;      // certainly rejected
    if (sound && A > r2f * B * hi && B > r2f * A * hi) return true;         // certainly kept
    const double ds2 = ((double)sf[a].x - (double)sf[b].x) * ((double)sf[a].x - (double)sf[b].x) +
                       ((double)sf[a].y - (double)sf[b].y) * ((double)sf[a].y - (double)sf[b].y) +
                       ((double)sf[a].z - (double)sf[b].z) * ((double)sf[a].z - (double)sf[b].z);
    const double dt2 = ((double)tf[a].x - (double)tf[b].x) * ((double)tf[a].x - (double)tf[b].x) +
                       ((double)tf[a].y - (double)tf[b].y) * ((double)tf[a].y - (double)tf[b].y) +
                       ((double)tf[a].z - (double)tf[b].z) * ((double)tf[a].z - (double)tf[b].z);
    return !(edge_reject(ds2, dt2, edge_ratio, r2) || edge_reject(dt2, ds2, edge_ratio, r2));
  };
  fetch(0);
  fetch(1);
  if (!edge_ok(0, 1)) return false;
  fetch(2);
  if (!(edge_ok(0, 2) && edge_ok(1, 2))) return false;
  fetch(3);
  return edge_ok(0, 3) && edge_ok(1, 3) && edge_ok(2, 3);
}

// wave-aggregated append of the survivors' iteration numbers.  kCandLists sub-lists, each with its own counter (one
// per 4 B of a 256-B block: different L2 atomic slots are not needed, different ADDRESSES are): with trained
// descriptors nearly every wave has a survivor, and 62 k reservations on ONE counter serialise at ~10 ns each
// (0.63 ms of a 0.70 ms kernel at 50 % true matches).  Workgroup b appends to sub-list b % kCandLists, which can
// receive at most sub_cap entries.
__device__ inline void append_candidate(bool ok, long long it, long long* __restrict__ cand, int* __restrict__ n_cand,
                                        int sub_cap) {
  const unsigned long long m = __ballot(ok);
  if (m) {
    const int lane = threadIdx.x & 63;
    const int list = blockIdx.x & (kCandLists - 1);
    int base = 0;
    if (lane == __ffsll((long long)m) - 1) base = atomicAdd(n_cand + list, __popcll(m));
    base = __shfl(base, __ffsll((long long)m) - 1);
    if (ok) cand[(int64_t)list * sub_cap + base + __popcll(m & ((1ull << lane) - 1ull))] = it;
  }
}

// sub_cap = ceil(#workgroups / kCandLists) * 256 entries per sub-list
__global__ __launch_bounds__(256) void k_sample_check(const float4* __restrict__ rec, uint32_t n0, double edge_ratio,
                                                      long long it0, long long it1, uint64_t seed,
                                                      long long* __restrict__ cand, int* __restrict__ n_cand,
                                                      int sub_cap) {
  const long long it = it0 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool ok = it < it1 && sample_passes(rec, n0, edge_ratio, it, seed);
  append_candidate(ok, it, cand, n_cand, sub_cap);
}

// ---- the same selection with the FIRST edge screened out of LDS (4 M iterations: the FCGF call) ----------------------
// k_sample_check is bound by the L2's request rate: every iteration gathers at least correspondences 0 and 1 (4 random 16-B
// loads from a 450 KB table that no L1 holds), 16 M line requests per pair = 44.6 us whatever the features are, and with
// random-init features 95 % of the iterations die on that first edge.  Here every correspondence also exists as ONE 8-byte
// word -- source and target point quantised to 10 bits per axis over [-R, R], R^2 = maxn2 -- and the whole table (14 k
// correspondences: 112 KB) sits in the LDS of a 1024-thread workgroup (one per CU).  A thread screens its iteration's first
// edge on the quantised points; the band is rigorous: a coordinate is off by at most 0.5001 cells (floor of an fp32 product),
// so a distance by at most sqrt(3) * 1.0002 < 1.75 cells =: E, and "Ds + E < r (Dt - E)" in cell units implies ds < r dt for
// the exact lengths with 0.03 cells to spare (fp32 / sqrt rounding here: 1e-4 cells).  Only an iteration that is CERTAINLY
// rejected is dropped; the others (the passes plus a band of ~2 %) are queued in LDS and run sample_passes -- the exact
// checker on the full-precision records -- densely.  Same candidate set as k_sample_check, bit for bit.
constexpr int kScreenThreads = 1024;
constexpr int kScreenPer = 2;                                   // iterations per thread and round
constexpr int kScreenRound = kScreenThreads * kScreenPer;
constexpr int kScreenQueue = 2 * kScreenRound;                  // survivors wait here for the exact checker; drained when a
                                                                // further round might not fit (random features: once, at the end)
constexpr int kScreenBits = 10;
constexpr int64_t kScreenMaxN0 = (160 * 1024 - kScreenQueue * 4 - 512) / 8;     // 18 368 correspondences

// rec8[i] = source (x | y << 10 | z << 20) | target (same) << 32, cells of size 2 R / 1023 (R = 0: everything in cell 0 and
// the screen passes every iteration)
__global__ void k_pack_small(const float4* __restrict__ rec, int64_t n0, const unsigned* __restrict__ maxn2,
                             unsigned long long* __restrict__ rec8) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ((n0 + 31) & ~(int64_t)31)) return;
  if (i >= n0) {            // padding of the staged table (never sampled)
    rec8[i] = 0ull;
    return;
  }
  const float R = sqrtf(__uint_as_float(*maxn2)) * 1.000001f;
  const float inv_h = R > 0.f ? (float)((1 << kScreenBits) - 1) / (2.f * R) : 0.f;
  auto q = [&](float c) {
    int u = (int)floorf((c + R) * inv_h);
    u = u < 0 ? 0 : (u > (1 << kScreenBits) - 1 ? (1 << kScreenBits) - 1 : u);
    return (unsigned)u;
  };
  const float4 a = rec[2 * i], b = rec[2 * i + 1];
  const unsigned lo = q(a.x) | (q(a.y) << kScreenBits) | (q(a.z) << (2 * kScreenBits));
  const unsigned hi = q(b.x) | (q(b.y) << kScreenBits) | (q(b.z) << (2 * kScreenBits));
  rec8[i] = (unsigned long long)lo | ((unsigned long long)hi << 32);
}

__device__ inline float cell_dist(unsigned a, unsigned b) {
  constexpr unsigned M = (1u << kScreenBits) - 1u;
  const int dx = (int)(a & M) - (int)(b & M), dy = (int)((a >> kScreenBits) & M) - (int)((b >> kScreenBits) & M),
            dz = (int)((a >> (2 * kScreenBits)) & M) - (int)((b >> (2 * kScreenBits)) & M);
  return __builtin_sqrtf((float)(dx * dx + dy * dy + dz * dz));      // exact integer < 2^24 under a 1-ulp sqrt
}

// the two 64-bit multiplications of splitmix64's finaliser, from a pre-mixed counter z0 = seed + (4 it + slot + 1) * golden
__device__ inline uint32_t sample_index_from(uint64_t z, uint32_t n) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint32_t)(((z >> 32) * (uint64_t)n) >> 32);
}

// workgroup b owns iterations [it0 + b * per_wg, .. + per_wg) and appends to sub-list b % kCandLists (capacity sub_cap)
__global__ __launch_bounds__(kScreenThreads) void k_sample_screen(const float4* __restrict__ rec,
                                                                  const unsigned long long* __restrict__ rec8, uint32_t n0,
                                                                  double edge_ratio, long long it0, long long it1,
                                                                  long long per_wg, uint64_t seed,
                                                                  long long* __restrict__ cand, int* __restrict__ n_cand,
                                                                  int sub_cap) {
  extern __shared__ unsigned long long s_rec[];          // [n0 padded] records | queue [kScreenQueue] u32 | queue length
  const uint32_t n0p = (n0 + 31u) & ~31u;
  unsigned* const s_q = reinterpret_cast<unsigned*>(s_rec + n0p);
  int& s_nq = *reinterpret_cast<int*>(s_q + kScreenQueue);      // no static LDS: the dynamic opt-in covers all of it
  const int tid = threadIdx.x, lane = tid & 63;
  {   // stage the table: 16-B loads, 8 in flight per thread before the first LDS store (rec8 is padded to n0p entries)
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const u64x2* src = reinterpret_cast<const u64x2*>(rec8);
    u64x2* dst = reinterpret_cast<u64x2*>(s_rec);
    const uint32_t nv = n0p >> 1;
    for (uint32_t i0 = tid; i0 < nv; i0 += 8 * kScreenThreads) {
      u64x2 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u * kScreenThreads < nv) t[u] = src[i0 + u * kScreenThreads];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u * kScreenThreads < nv) dst[i0 + u * kScreenThreads] = t[u];
    }
  }
  if (tid == 0) s_nq = 0;
  const long long base = it0 + (long long)blockIdx.x * per_wg;
  const long long end = (base + per_wg < it1) ? base + per_wg : it1;
  const float r = (float)edge_ratio, E = 1.75f;
  constexpr uint64_t G = 0x9E3779B97F4A7C15ull;
  auto drain = [&]() {      // the exact checker, densely over the queue; callers synchronise before and after
    const int nq = s_nq;
    for (int c0 = 0; c0 < nq; c0 += kScreenThreads) {
      const int c = c0 + tid;
      const long long it = c < nq ? base + (long long)s_q[c] : 0;
      const bool ok = c < nq && sample_passes(rec, n0, edge_ratio, it, seed);
      append_candidate(ok, it, cand, n_cand, sub_cap);
    }
  };
  __syncthreads();                                        // table staged, counter cleared
  // counter of (iteration, slot 0): seed + (4 it + 1) G; slot 1 is one G further, the next iteration of this thread
  // 4 * kScreenThreads * G further -- additions only
  uint64_t z0 = seed + (uint64_t)(4 * (base + tid) + 1) * G;
  for (long long rb = base; rb < end; rb += kScreenRound) {
#pragma unroll
    for (int u = 0; u < kScreenPer; ++u) {
      const long long it = rb + u * kScreenThreads + tid;
      bool pass = false;
      if (it < end) {
        const unsigned long long a = s_rec[sample_index_from(z0, n0)];
        const unsigned long long b = s_rec[sample_index_from(z0 + G, n0)];
        const float Ds = cell_dist((unsigned)a, (unsigned)b), Dt = cell_dist((unsigned)(a >> 32), (unsigned)(b >> 32));
        pass = !(Ds + E < r * (Dt - E)) && !(Dt + E < r * (Ds - E));
      }
      z0 += (uint64_t)(4 * kScreenThreads) * G;
      const unsigned long long m = __ballot(pass);
      if (m) {
        int qb = 0;
        if (lane == __ffsll((long long)m) - 1) qb = atomicAdd(&s_nq, __popcll(m));
        qb = __shfl(qb, __ffsll((long long)m) - 1);
        if (pass) s_q[qb + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned)(it - base);
      }
    }
    __syncthreads();
    const int nq_now = s_nq;                              // read by everybody BEFORE anybody appends again
    __syncthreads();
    if (nq_now > kScreenQueue - kScreenRound) {           // the next round might not fit: run the exact checker now
      drain();
      __syncthreads();
      if (tid == 0) s_nq = 0;
      __syncthreads();
    }
  }
  drain();
}

// dense over the compacted candidates (grid-stride: the count only exists on the device)
__global__ __launch_bounds__(256) void k_fit_check(const float4* __restrict__ rec, uint32_t n0, double thr_gt,
                                                   uint64_t seed, const long long* __restrict__ cand,
                                                   const int* __restrict__ n_cand, int sub_cap,
                                                   Hyp* __restrict__ hyps, int* __restrict__ n_valid, int cap) {
  __shared__ int s_off[kCandLists + 1];      // exclusive offsets of the sub-lists in the concatenated candidate order
  if (threadIdx.x < 64) {
    const int c = n_cand[threadIdx.x];
    const int incl = apr_wave_incl_scan(c);
    s_off[threadIdx.x + 1] = incl;
    if (threadIdx.x == 0) s_off[0] = 0;
  }
  __syncthreads();
  const int nc = s_off[kCandLists];
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += gridDim.x * blockDim.x) {
    int list = 0;
#pragma unroll
    for (int step = kCandLists / 2; step >= 1; step >>= 1)
      if (s_off[list + step] <= c) list += step;
    const long long it = cand[(int64_t)list * sub_cap + (c - s_off[list])];
    double s[4][3], t[4][3];
    load_samples(rec, seed, it, n0, s, t);
    double T[12];
    kabsch4(s, t, T);
    // CorrespondenceCheckerBasedOnDistance: sqrt(d2) > max_dist  <=>  d2 > thr_gt
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double dx = T[0] * s[j][0] + T[1] * s[j][1] + T[2] * s[j][2] + T[3] - t[j][0];
      double dy = T[4] * s[j][0] + T[5] * s[j][1] + T[6] * s[j][2] + T[7] - t[j][1];
      double dz = T[8] * s[j][0] + T[9] * s[j][1] + T[10] * s[j][2] + T[11] - t[j][2];
      if (dx * dx + dy * dy + dz * dz > thr_gt) ok = false;
    }
    if (!ok) continue;
    const int slot = atomicAdd(n_valid, 1);
    if (slot < cap) {
#pragma unroll
      for (int k = 0; k < 12; ++k) hyps[slot].T[k] = T[k];
      hyps[slot].it = it;
      hyps[slot].inliers = 0;
      hyps[slot].err2 = 0.0;
    }
  }
}

constexpr int kGeoGrid = 2048;
struct GeoPart {
  double e2;
  int cnt;
  int pad;
};

__device__ inline int geo_parts(int nv, int64_t n0) {
  const int nchunk = (int)((n0 + 255) >> 8);
  int S = nv > 0 ? kGeoGrid / nv : 1;
  if (S > nchunk) S = nchunk;
  return S < 1 ? 1 : S;
}

constexpr int kScoreThreads = 256;   // 1024 threads per hypothesis cut the kernel from 14 to 12 us in isolation (4 instead of
                                     // 14 dependent L2 round trips per thread) but cost 2.4 % pairs/s with three steps in
                                     // flight: a 16-wave workgroup waits for a whole CU's worth of free slots
// A workgroup per hypothesis (grid-stride; every workgroup reaches the exit): only a few hundred hypotheses survive
// the checkers, so one WAVE per hypothesis would leave the chip idle behind ~220 serial iterations per wave.
// The per-thread partial sums are combined in a fixed order (lane tree, then the waves in order): bitwise reproducible.
// With few survivors (the usual case: tens to hundreds) a hypothesis is dealt to S = kGeoGrid / nv workgroups, each
// scoring a contiguous range of 1024-record blocks into part[h * S + p]; k_score_finish adds the S slots in order
// (a fixed order for given inputs).  With more than kGeoGrid survivors S = 1 and the workgroup writes the hypothesis
// itself, as before.
__device__ inline int score_parts(int nv, int64_t n0) {
  const int nblk = (int)((n0 + 4 * kScoreThreads - 1) / (4 * kScoreThreads));
  int S = nv > 0 ? kGeoGrid / nv : 1;
  if (S > nblk) S = nblk;
  return S < 1 ? 1 : S;
}

// The hypotheses scored are the picked ones: sel[0 .. sel_hdr[0]) (k_pick: all of them, or those at the maximum count).
__global__ __launch_bounds__(kScoreThreads) void k_score(const float4* __restrict__ rec, int64_t n0, double thr_lt,
                                               Hyp* __restrict__ hyps, const int* __restrict__ sel_hdr,
                                               const int* __restrict__ sel, GeoPart* __restrict__ part) {
  constexpr int NW = kScoreThreads / 64;
  __shared__ int s_cnt[NW];
  __shared__ double s_e2[NW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = sel_hdr[0];
  const int S = score_parts(nv, n0);
  const int64_t nblk = (n0 + 4 * kScoreThreads - 1) / (4 * kScoreThreads);
  for (int64_t item = blockIdx.x; item < (int64_t)nv * S; item += gridDim.x) {
    const int hs = (int)(item / S), pp = (int)(item - (int64_t)hs * S);
    const int h = sel[hs];
    const int64_t i_begin = (nblk * pp / S) * (4 * kScoreThreads);
    const int64_t i_end = min((long long)((nblk * (pp + 1) / S) * (4 * kScoreThreads)), (long long)n0);
    double T[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = hyps[h].T[k];
    int cnt = 0;
    double e2 = 0.0;
    for (int64_t i0 = i_begin + threadIdx.x; i0 < i_end; i0 += 4 * kScoreThreads) {   // 4 records in flight, consumed in index order
      float4 a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t i = i0 + u * kScoreThreads;
        if (i < i_end) {
          a[u] = rec[2 * i];
          b[u] = rec[2 * i + 1];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (i0 + u * kScoreThreads >= i_end) break;
        const double sx = a[u].x, sy = a[u].y, sz = a[u].z;
        double dx = T[0] * sx + T[1] * sy + T[2] * sz + T[3] - (double)b[u].x;
        double dy = T[4] * sx + T[5] * sy + T[6] * sz + T[7] - (double)b[u].y;
        double dz = T[8] * sx + T[9] * sy + T[10] * sz + T[11] - (double)b[u].z;
        double d2 = dx * dx + dy * dy + dz * dz;
        if (d2 < thr_lt) {   // <=> sqrt(d2) < max_dist
          ++cnt;
          e2 += d2;
        }
      }
    }
    for (int d = 32; d >= 1; d >>= 1) {
      cnt += __shfl_xor(cnt, d);
      e2 += __shfl_xor(e2, d);
    }
    if (lane == 0) {
      s_cnt[wave] = cnt;
      s_e2[wave] = e2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int c = 0;
      double e = 0.0;
      for (int w = 0; w < NW; ++w) {   // fixed order
        c += s_cnt[w];
        e += s_e2[w];
      }
      if (S == 1) {
        hyps[h].inliers = c;
        hyps[h].err2 = e;
      } else {
        part[item].cnt = c;
        part[item].e2 = e;
      }
    }
    __syncthreads();
  }
}

__global__ void k_score_finish(Hyp* __restrict__ hyps, const int* __restrict__ sel_hdr, const int* __restrict__ sel,
                               int64_t n0, const GeoPart* __restrict__ part) {
  const int nv = sel_hdr[0];
  const int S = score_parts(nv, n0);
  const int hs = blockIdx.x * blockDim.x + threadIdx.x;
  if (S == 1 || hs >= nv) return;          // S > 1 only when nv <= kGeoGrid
  int c = 0;
  double e = 0.0;
  for (int p = 0; p < S; ++p) {
    c += part[(int64_t)hs * S + p].cnt;
    e += part[(int64_t)hs * S + p].e2;
  }
  const int h = sel[hs];
  hyps[h].inliers = c;
  hyps[h].err2 = e;
}

// ---------------------------------------------------------------------------------------------------------------
// Many survivors (trained descriptors: 10^4 ... 10^6 valid hypotheses of 4 M).  Scoring every hypothesis in fp64 over
// all correspondences is survivors x n0 x ~30 fp64 operations and survivors x n0 x 32 B of L2 reads (6.4 ms per pair
// at 243 k survivors).  But the winner is decided by the INLIER COUNT; the rmse only orders hypotheses that tie on it.  So:
//   k_count      one THREAD per hypothesis, its transform in fp32 registers; the correspondences arrive through SCALAR
//                loads (the index is wave-uniform), two per row, and feed packed fp32 FMAs as SGPR operands: no LDS, no
//                reduction, no re-read per hypothesis beyond the scalar cache.  d2 is compared with lo = thr - b and
//                hi = thr + b, where b bounds |d2_fp32 - d2_exact| rigorously (below); per mini-chunk of 64
//                correspondences, equal counts under lo and hi prove that no correspondence sits in the band, and the
//                count is exact.  A mini-chunk with a band hit is flagged in a bit mask instead of being counted.
//   k_count_fix  the flagged (hypothesis, mini-chunk) cells again, in fp64 exactly as k_score does, a wave per cell
//                (lane = correspondence), added to the hypothesis' count with an integer atomic.
//   k_count_max / k_pick   the hypotheses that reach the maximum count: only they get their squared error summed
//                (k_score over the picked list, fp64, fixed order) -- usually one or two of 10^5.
// Integer counts and integer atomics only: the result (best transform, its inliers, its rmse) is bit-identical to
// scoring everything in fp64.  Up to kGeoGrid survivors the kernels return at once and everything is scored in fp64.
//
// The band.  x_f = fl(fl(T0f sx + fl(T1f sy + fl(T2f sz + T3f))) - tx) against x = T0 sx + T1 sy + T2 sz + T3 - tx in
// fp64: rounding T costs u (|s| + |T3|) (a rotation row has unit length), each of the four operations u times the size
// of its result <= |s| + |T3| + |tx|: |x_f - x| <= 5 u (2 Mn + tn) with Mn = max point norm, tn = max |translation|,
// u = 2^-24; taken as e1 = 8 u (2 Mn + tn).  |d2_f - d2| <= 2 sqrt(3 d2) e1 + 3 e1^2 + 4 u d2, increasing in d2 slower
// than d2 itself: d2_f < thr - b(thr) implies d2 < thr, d2_f >= thr + b(thr) implies d2 >= thr.
// ---------------------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------------------------
// Pruning the correspondences of the count (round 4).  With trained descriptors the survivors are 4-inlier samples: tens
// of thousands of transforms within a few metres of each other, each counted over ALL correspondences -- of which the
// outliers (70 % at a 30 % match share) are tens of metres away under every one of them.  Take a reference T_ref (the
// first survivor) and the residual r_c = |T_ref s_c - t_c| of every correspondence.  For a hypothesis h,
//     | |T_h s - t| - |T_ref s - t| | <= |T_h s - T_ref s| <= ||R_h - R_ref||_F |s| + |t_h - t_ref| <= delta_h
// with delta_h := ||R_h - R_ref||_F * max|s| + |t_h - t_ref|.  So if delta_h <= D, a correspondence with r_c >= m + D is
// at distance >= m from its target under T_h: not an inlier (inliers need < m).  k_prune keeps the correspondences with
// r_c < m + D as a second record set and sorts the survivors into NEAR (delta_h <= D: counted over the kept
// records only) and FAR (counted over all, as before -- also everything when T_ref is a poor draw: only speed is lost).
// fp64 throughout, D compared with 1e-6 m to spare; counts, picks and errors are those of the full evaluation, bit for bit
// (tests/test_match_pose_gpu.py).  APR_RANSAC_PRUNE=0: every survivor is FAR.
// ---------------------------------------------------------------------------------------------------------------
enum { kLiveCount = 0, kLiveMini = 1, kLiveWords = 2, kLiveNear = 3, kLiveFar = 4, kLiveInts = 8 };
constexpr double kPruneReach = 10.0;      // D, metres

// (the order of the kept records is free: the count is an integer sum, and the squared error of the picked hypotheses is
// summed over the FULL records by k_score.)  live[kLiveCount] is zero on entry (cleared with the round's counters) and
// rec2_live is all padding (k_pack_pairs): the list needs no finishing pass, its mini-chunk count follows from its length.
// (A first version let the last block to finish pad the list behind a __threadfence() + ticket; device-scope fences write
// back and invalidate the L2 under the other streams' kernels -- see k_select_part.)
__device__ inline void live_part(int blk, int nblk, const float4* __restrict__ rec, int64_t n0, double m_up,
                                 const Hyp* __restrict__ hyps, float4* __restrict__ rec_live, float* __restrict__ rec2_live,
                                 int* __restrict__ live) {
  const int lane = threadIdx.x & 63;
  const Hyp* hp = hyps;      // T_ref: the first survivor
  const double T0 = hp->T[0], T1 = hp->T[1], T2 = hp->T[2], T3 = hp->T[3], T4 = hp->T[4], T5 = hp->T[5];
  const double T6 = hp->T[6], T7 = hp->T[7], T8 = hp->T[8], T9 = hp->T[9], T10 = hp->T[10], T11 = hp->T[11];
  const double reach = m_up + kPruneReach, reach2 = reach * reach;
  const int64_t nround = (n0 + 255) & ~(int64_t)255;
  for (int64_t i = (int64_t)blk * blockDim.x + threadIdx.x; i < nround; i += (int64_t)nblk * blockDim.x) {
    bool keep = false;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (i < n0) {
      a = rec[2 * i];
      b = rec[2 * i + 1];
      const double sx = a.x, sy = a.y, sz = a.z;
      const double dx = T0 * sx + T1 * sy + T2 * sz + T3 - (double)b.x;
      const double dy = T4 * sx + T5 * sy + T6 * sz + T7 - (double)b.y;
      const double dz = T8 * sx + T9 * sy + T10 * sz + T11 - (double)b.z;
      keep = !(dx * dx + dy * dy + dz * dz >= reach2);      // NaN stays: never dropped on a maybe
    }
    const unsigned long long mk = __ballot(keep);
    int base = 0;
    if (lane == 0 && mk) base = atomicAdd(live + kLiveCount, __popcll(mk));
    base = __shfl(base, 0);
    if (keep) {
      const int pos = base + __popcll(mk & ((1ull << lane) - 1ull));
      rec_live[2 * (int64_t)pos] = a;
      rec_live[2 * (int64_t)pos + 1] = b;
      float* row = rec2_live + (int64_t)(pos >> 1) * 12 + (pos & 1);
      row[0] = a.x; row[2] = a.y; row[4] = a.z; row[6] = b.x; row[8] = b.y; row[10] = b.z;
    }
  }
}

__device__ inline void near_part(int blk, int nblk, const Hyp* __restrict__ hyps, int nv, const unsigned* __restrict__ maxn2,
                                 int prune, int* __restrict__ order_near, int* __restrict__ order_far,
                                 unsigned char* __restrict__ near_flag, int* __restrict__ live, unsigned* __restrict__ band,
                                 int nwords) {
  const int lane = threadIdx.x & 63;
  const double Mn = sqrt((double)__uint_as_float(*maxn2)) * (1.0 + 1e-6);
  const Hyp* r = hyps;
  const int nround = (nv + 255) & ~255;
  for (int h = blk * blockDim.x + threadIdx.x; h < nround; h += nblk * blockDim.x) {
    bool nr = false;
    if (h < nv && prune) {
      const Hyp* p = hyps + h;
      double f2 = 0.0, t2 = 0.0;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          const double d = p->T[4 * a + b] - r->T[4 * a + b];
          f2 += d * d;
        }
        const double dt = p->T[4 * a + 3] - r->T[4 * a + 3];
        t2 += dt * dt;
      }
      const double delta = (sqrt(f2) * Mn + sqrt(t2)) * (1.0 + 1e-9) + 1e-6;
      nr = delta <= kPruneReach;      // NaN: far
    }
    const bool fr = h < nv && !nr;
    if (h < nv) {
      near_flag[h] = nr ? 1 : 0;
      for (int w = 0; w < nwords; ++w) band[(int64_t)h * nwords + w] = 0u;
    }
    const unsigned long long mn = __ballot(nr), mf = __ballot(fr);
    int bn = 0, bf = 0;
    if (lane == 0) {
      if (mn) bn = atomicAdd(live + kLiveNear, __popcll(mn));
      if (mf) bf = atomicAdd(live + kLiveFar, __popcll(mf));
    }
    bn = __shfl(bn, 0);
    bf = __shfl(bf, 0);
    if (nr) order_near[bn + __popcll(mn & ((1ull << lane) - 1ull))] = h;
    if (fr) order_far[bf + __popcll(mf & ((1ull << lane) - 1ull))] = h;
  }
}

// ONE launch for both parts (with few survivors -- the random-feature case -- it is one empty launch more per pair, not
// two): the first n_live_blocks workgroups build the pruned correspondence list, the others sort the survivors.
__global__ __launch_bounds__(256) void k_prune(const float4* __restrict__ rec, int64_t n0, double m_up, Hyp* __restrict__ hyps,
                                               const int* __restrict__ n_valid, int cap, int few,
                                               const unsigned* __restrict__ maxn2, int prune, int n_live_blocks,
                                               float4* __restrict__ rec_live, float* __restrict__ rec2_live,
                                               int* __restrict__ live, int* __restrict__ order_near,
                                               int* __restrict__ order_far, unsigned char* __restrict__ near_flag,
                                               unsigned* __restrict__ band, int nwords) {
  const int nv = min(*n_valid, cap);
  if (nv <= few) return;
  if ((int)blockIdx.x < n_live_blocks)
    live_part((int)blockIdx.x, n_live_blocks, rec, n0, m_up, hyps, rec_live, rec2_live, live);
  else
    near_part((int)blockIdx.x - n_live_blocks, (int)gridDim.x - n_live_blocks, hyps, nv, maxn2, prune, order_near, order_far,
              near_flag, live, band, nwords);
}

__device__ inline int count_parts(int nblk, int nunits) {      // ranges per 256-hypothesis block: ~4096 items in all
  int S = (4096 + nblk - 1) / nblk;
  if (S > nunits) S = nunits;
  return S < 1 ? 1 : S;
}

// Both survivor lists in ONE launch: the NEAR list (order_near[0 .. live[kLiveNear])) over the pruned correspondences
// (rec2_live; its mini-chunk count follows from live[kLiveCount]), then the FAR list over all of them; band rows have the full stride `nwords` and
// were cleared by k_prune.
// Work items = (256 hypotheses, a contiguous range of MINI-CHUNKS).  A thread walks its range row by row behind scalar
// loads: ~115 us per 32 mini-chunks whatever else runs, so the ranges must be short enough that the grid is many waves
// per SIMD deep (ranges of whole 32-mini-chunk words left 248 items of 115 us each for the pruned list of a 30 % pair:
// no faster than the full list).  A range ORs its band bits in.
__global__ __launch_bounds__(256) void k_count(const float* __restrict__ rec2_full, const float* __restrict__ rec2_live,
                                               int nwords, int nmini_full, double thr_lt,
                                               const unsigned* __restrict__ maxn2, Hyp* __restrict__ hyps,
                                               const int* __restrict__ n_valid, int cap, unsigned* __restrict__ band,
                                               int few, const int* __restrict__ order_near,
                                               const int* __restrict__ order_far, const int* __restrict__ live) {
  const int nv_all = min(*n_valid, cap);
  if (nv_all <= few) return;
  const int nvN = live[kLiveNear], nvF = live[kLiveFar], nminiN = (int)(rec2_rows(live[kLiveCount]) / kMini);
  const int nblkN = (nvN + 255) >> 8, nblkF = (nvF + 255) >> 8;
  const int SN = count_parts(nblkN > 0 ? nblkN : 1, nminiN > 0 ? nminiN : 1);
  const int SF = count_parts(nblkF > 0 ? nblkF : 1, nmini_full > 0 ? nmini_full : 1);
  const int itemsN = nminiN > 0 ? nblkN * SN : 0, itemsF = nblkF * SF;
  const int nwords_full = nwords;
  const float Mn = sqrtf(__uint_as_float(*maxn2)) * 1.000001f;
  const float thr = (float)thr_lt;
  for (int item0 = blockIdx.x; item0 < itemsN + itemsF; item0 += gridDim.x) {
    const bool isN = item0 < itemsN;
    const int item = isN ? item0 : item0 - itemsN;
    const int S = isN ? SN : SF, nv = isN ? nvN : nvF, nmini = isN ? nminiN : nmini_full;
    const int* __restrict__ order = isN ? order_near : order_far;
    const float* __restrict__ rec2 = isN ? rec2_live : rec2_full;
    const int hb = item / S, part = item - hb * S;
    const int m_begin = (int)((int64_t)nmini * part / S), m_stop = (int)((int64_t)nmini * (part + 1) / S);
    const int w_begin = m_begin >> 5, w_end = (m_stop + 31) >> 5;
    const int slot = hb * 256 + threadIdx.x;
    const bool act = slot < nv;
    const int h = order[act ? slot : nv - 1];
    const Hyp* hp = hyps + h;
    float T[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = (float)hp->T[k];
    const float tn = fmaxf(fabsf(T[3]), fmaxf(fabsf(T[7]), fabsf(T[11]))) * 1.000001f;
    const float e1 = 8.f * 5.9604645e-8f * (2.f * Mn + tn);
    const float b = (2.f * sqrtf(3.f * thr) * e1 + 3.f * e1 * e1 + 4.f * 5.9604645e-8f * thr) * 1.01f + 1e-30f;
    const float lo = thr - b, hi = thr + b;
    const f32x2 t0 = {T[0], T[0]}, t1 = {T[1], T[1]}, t2 = {T[2], T[2]}, t3 = {T[3], T[3]};
    const f32x2 t4 = {T[4], T[4]}, t5 = {T[5], T[5]}, t6 = {T[6], T[6]}, t7 = {T[7], T[7]};
    const f32x2 t8 = {T[8], T[8]}, t9 = {T[9], T[9]}, t10 = {T[10], T[10]}, t11 = {T[11], T[11]};
    int total = 0;
    for (int w = w_begin; w < w_end; ++w) {
      unsigned mask = 0;
      const int m_first = max(0, m_begin - w * 32), m_end = min(32, m_stop - w * 32);
      for (int mc = m_first; mc < m_end; ++mc) {
        const float* __restrict__ row = rec2 + (int64_t)(w * 32 + mc) * (kMini * 12);   // wave-uniform: scalar loads
        int c_lo = 0, c_hi = 0;
#pragma unroll 4
        for (int pp = 0; pp < kMini; ++pp) {
          const float* __restrict__ q = row + pp * 12;
          const f32x2 sx = {q[0], q[1]}, sy = {q[2], q[3]}, sz = {q[4], q[5]};
          const f32x2 tx = {q[6], q[7]}, ty = {q[8], q[9]}, tz = {q[10], q[11]};
          f32x2 dx = __builtin_elementwise_fma(t2, sz, t3);
          f32x2 dy = __builtin_elementwise_fma(t6, sz, t7);
          f32x2 dz = __builtin_elementwise_fma(t10, sz, t11);
          dx = __builtin_elementwise_fma(t1, sy, dx);
          dy = __builtin_elementwise_fma(t5, sy, dy);
          dz = __builtin_elementwise_fma(t9, sy, dz);
          dx = __builtin_elementwise_fma(t0, sx, dx) - tx;
          dy = __builtin_elementwise_fma(t4, sx, dy) - ty;
          dz = __builtin_elementwise_fma(t8, sx, dz) - tz;
          const f32x2 d2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
          c_lo += (d2[0] < lo) + (d2[1] < lo);
          c_hi += (d2[0] < hi) + (d2[1] < hi);
        }
        if (c_lo == c_hi) total += c_lo;
        else mask |= 1u << mc;
      }
      if (act && mask) atomicOr(&band[(int64_t)h * nwords_full + w], mask);
    }
    if (act && total) atomicAdd(&hyps[h].inliers, total);
  }
}

// The flagged cells, exactly: a wave per (hypothesis, mini-chunk), lane = correspondence, the fp64 expression of k_score.
__global__ __launch_bounds__(256) void k_count_fix(const float4* __restrict__ rec, int64_t n0, double thr_lt,
                                                   Hyp* __restrict__ hyps, const int* __restrict__ n_valid, int cap,
                                                   const unsigned* __restrict__ band, int nwords, int few,
                                                   const float4* __restrict__ rec_live, const int* __restrict__ live,
                                                   const unsigned char* __restrict__ near_flag) {
  const int nv = min(*n_valid, cap);
  if (nv <= few) return;
  const int64_t n_live = live ? live[kLiveCount] : 0;
  const int lane = threadIdx.x & 63;
  const int64_t total = (int64_t)nv * nwords;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  // A wave takes kFixWords band words at a time and walks their flagged cells one after the other (~1.7 us each: the
  // hypothesis comes from memory, then 64 correspondences in fp64).  With 64 words per wave the 220 k words of a 30 %
  // pair were 3.4 k wave-loads of ~18 flagged cells each on an 8 k-wave grid: 31 us of serial cells.  16 words per wave
  // spread the same cells over four times as many waves.
  constexpr int kFixWords = 16;
  for (int64_t base = wave0 * kFixWords; base < total; base += nwave * kFixWords) {
    const int64_t idx = base + lane;
    unsigned mask = (lane < kFixWords && idx < total) ? band[idx] : 0u;
    unsigned long long pending = __ballot(mask != 0u);
    while (pending) {
      const int L = __builtin_ctzll(pending);
      pending &= pending - 1ull;
      const int64_t cell = base + L;
      const int h = (int)(cell / nwords), w = (int)(cell - (int64_t)h * nwords);
      unsigned bits = __shfl(mask, L);
      const Hyp* hp = hyps + h;
      const bool nr = near_flag && near_flag[h];      // a pruned-list hypothesis: its cells index the pruned records
      const float4* __restrict__ R = nr ? rec_live : rec;
      const int64_t nR = nr ? n_live : n0;
      const double T0 = hp->T[0], T1 = hp->T[1], T2 = hp->T[2], T3 = hp->T[3], T4 = hp->T[4], T5 = hp->T[5];
      const double T6 = hp->T[6], T7 = hp->T[7], T8 = hp->T[8], T9 = hp->T[9], T10 = hp->T[10], T11 = hp->T[11];
      int cnt = 0;
      while (bits) {
        const int mc = __builtin_ctz(bits);
        bits &= bits - 1u;
        const int64_t i = ((int64_t)w * 32 + mc) * 64 + lane;
        bool in = false;
        if (i < nR) {
          const float4 a = R[2 * i], bq = R[2 * i + 1];
          const double sx = a.x, sy = a.y, sz = a.z;
          const double dx = T0 * sx + T1 * sy + T2 * sz + T3 - (double)bq.x;
          const double dy = T4 * sx + T5 * sy + T6 * sz + T7 - (double)bq.y;
          const double dz = T8 * sx + T9 * sy + T10 * sz + T11 - (double)bq.z;
          in = dx * dx + dy * dy + dz * dz < thr_lt;
        }
        cnt += __popcll(__ballot(in));
      }
      if (lane == 0 && cnt) atomicAdd(&hyps[h].inliers, cnt);
    }
  }
}

// sel_hdr[0] = number of picked hypotheses, sel_hdr[1] = the maximum inlier count (both zero before k_count_max)
__global__ __launch_bounds__(256) void k_count_max(const Hyp* __restrict__ hyps, const int* __restrict__ n_valid, int cap,
                                                   int* __restrict__ sel_hdr, int few) {
  const int nv = min(*n_valid, cap);
  if (nv <= few) return;
  int m = 0;
  for (int h = blockIdx.x * blockDim.x + threadIdx.x; h < nv; h += gridDim.x * blockDim.x) m = max(m, hyps[h].inliers);
  for (int d = 32; d >= 1; d >>= 1) m = max(m, __shfl_xor(m, d));
  if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(&sel_hdr[1], m);
}

// few survivors: every hypothesis (sel = identity); many: the ones at the maximum count
__global__ __launch_bounds__(256) void k_pick(const Hyp* __restrict__ hyps, const int* __restrict__ n_valid, int cap,
                                              int* __restrict__ sel_hdr, int* __restrict__ sel, int few) {
  const int nv = min(*n_valid, cap);
  const bool all = nv <= few;
  const int cmax = sel_hdr[1];
  for (int h = blockIdx.x * blockDim.x + threadIdx.x; h < nv; h += gridDim.x * blockDim.x) {
    if (all) {
      sel[h] = h;
    } else if (hyps[h].inliers == cmax) {
      sel[atomicAdd(&sel_hdr[0], 1)] = h;      // the SET is what matters: k_select orders by (count, rmse, iteration)
    }
  }
  if (all && blockIdx.x == 0 && threadIdx.x == 0) sel_hdr[0] = nv;
}

__device__ inline bool better(int c1, double r1, long long i1, int c2, double r2, long long i2) {
  if (c1 != c2) return c1 > c2;
  if (r1 != r2) return r1 < r2;
  return i1 < i2;
}

// best[0] carries the running best over chunks (inliers < 0 == none yet)
__global__ void k_select(const Hyp* __restrict__ hyps, const int* __restrict__ n_valid, int cap,
                         Hyp* __restrict__ best, long long* __restrict__ total_valid) {
  __shared__ int s_c[1024];
  __shared__ double s_r[1024];
  __shared__ long long s_i[1024];
  __shared__ int s_h[1024];
  const int nv = min(*n_valid, cap);
  int bc = -1, bh = -1;
  double br = 0.0;
  long long bi = 0;
  for (int h = threadIdx.x; h < nv; h += blockDim.x) {
    int c = hyps[h].inliers;
    double r = c > 0 ? sqrt(hyps[h].err2 / (double)c) : 0.0;
    long long it = hyps[h].it;
    if (bh < 0 || better(c, r, it, bc, br, bi)) {
      bc = c; br = r; bi = it; bh = h;
    }
  }
  s_c[threadIdx.x] = bc; s_r[threadIdx.x] = br; s_i[threadIdx.x] = bi; s_h[threadIdx.x] = bh;
  __syncthreads();
  for (int stride = blockDim.x / 2; stride >= 1; stride >>= 1) {
    if (threadIdx.x < stride) {
      int o = threadIdx.x + stride;
      if (s_h[o] >= 0 && (s_h[threadIdx.x] < 0 ||
                          better(s_c[o], s_r[o], s_i[o], s_c[threadIdx.x], s_r[threadIdx.x], s_i[threadIdx.x]))) {
        s_c[threadIdx.x] = s_c[o]; s_r[threadIdx.x] = s_r[o]; s_i[threadIdx.x] = s_i[o]; s_h[threadIdx.x] = s_h[o];
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    *total_valid += *n_valid;
    if (s_h[0] >= 0) {
      double prev_r = best->inliers > 0 ? sqrt(best->err2 / (double)best->inliers) : 0.0;
      if (best->inliers < 0 || better(s_c[0], s_r[0], s_i[0], best->inliers, prev_r, best->it)) *best = hyps[s_h[0]];
    }
  }
}

// Two-stage form of k_select for long hypothesis lists (one workgroup reading 2^18 records of 128 B took 0.5 ms): kSelParts
// workgroups each keep the best of a slice, k_select_final the best of those and of the running best.  (One kernel whose
// last workgroup reduces the parts -- __threadfence() + ticket in all 256 workgroups -- cost the three-steps-in-flight
// headline 2.5 %: a device-scope fence writes back and invalidates the L2 under the other streams' kernels.)
constexpr int kSelParts = 256;
struct SelPart {
  double r;
  long long it;
  int c, h;
};

__global__ __launch_bounds__(256) void k_select_part(const Hyp* __restrict__ hyps, const int* __restrict__ n_valid, int cap,
                                                     SelPart* __restrict__ parts) {
  __shared__ SelPart s_p[256];
  const int nv = min(*n_valid, cap);
  SelPart b;
  b.c = -1; b.h = -1; b.r = 0.0; b.it = 0;
  for (int h = blockIdx.x * blockDim.x + threadIdx.x; h < nv; h += gridDim.x * blockDim.x) {
    const int c = hyps[h].inliers;
    const double r = c > 0 ? sqrt(hyps[h].err2 / (double)c) : 0.0;
    const long long it = hyps[h].it;
    if (b.h < 0 || better(c, r, it, b.c, b.r, b.it)) {
      b.c = c; b.r = r; b.it = it; b.h = h;
    }
  }
  s_p[threadIdx.x] = b;
  __syncthreads();
  for (int stride = blockDim.x / 2; stride >= 1; stride >>= 1) {
    if (threadIdx.x < stride) {
      const SelPart o = s_p[threadIdx.x + stride], m = s_p[threadIdx.x];
      if (o.h >= 0 && (m.h < 0 || better(o.c, o.r, o.it, m.c, m.r, m.it))) s_p[threadIdx.x] = o;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) parts[blockIdx.x] = s_p[0];
}

__global__ __launch_bounds__(kSelParts) void k_select_final(const Hyp* __restrict__ hyps, const int* __restrict__ n_valid,
                                                            const SelPart* __restrict__ parts, Hyp* __restrict__ best,
                                                            long long* __restrict__ total_valid, int* __restrict__ sel_hdr) {
  __shared__ SelPart s_p[kSelParts];
  s_p[threadIdx.x] = parts[threadIdx.x];
  __syncthreads();
  for (int stride = kSelParts / 2; stride >= 1; stride >>= 1) {
    if ((int)threadIdx.x < stride) {
      const SelPart o = s_p[threadIdx.x + stride], m = s_p[threadIdx.x];
      if (o.h >= 0 && (m.h < 0 || better(o.c, o.r, o.it, m.c, m.r, m.it))) s_p[threadIdx.x] = o;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    sel_hdr[0] = sel_hdr[1] = 0;      // ready for the next round's k_count_max / k_pick
    *total_valid += *n_valid;
    const SelPart w = s_p[0];
    if (w.h >= 0) {
      const double prev_r = best->inliers > 0 ? sqrt(best->err2 / (double)best->inliers) : 0.0;
      if (best->inliers < 0 || better(w.c, w.r, w.it, best->inliers, prev_r, best->it)) *best = hyps[w.h];
    }
  }
}

// also clears the norm bound of k_pack_pairs and the pick header of the scoring kernels (zero3[0..2]; NULL: nothing)
// `counters` (nullable): the n_valid .. n_cand words that launch_hypotheses would otherwise clear with a memset node of
// its own -- the single-round callers hand them over here (one launch less per pair)
__global__ void k_init_best(Hyp* best, long long* total_valid, unsigned* zero3, int* counters, int n_counters) {
  if (counters)
    for (int i = threadIdx.x; i < n_counters; i += blockDim.x) counters[i] = 0;
  if (threadIdx.x != 0) return;
#pragma unroll
  for (int k = 0; k < 12; ++k) best->T[k] = (k % 5 == 0) ? 1.0 : 0.0;  // identity (open3d's default result)
  best->it = -1;
  best->inliers = -1;
  best->err2 = 0.0;
  *total_valid = 0;
  if (zero3) zero3[0] = zero3[1] = zero3[2] = 0u;
}

// ---------------------------------------------------------------------------------
// IRLS: est_quad_linear_robust, one 1024-thread workgroup, 20 iterations in-kernel.
// ---------------------------------------------------------------------------------
__device__ inline void solve6(double M[6][7]) {
  // Gauss-Jordan with partial pivoting on the augmented 6x7 system (thread 0 only)
  for (int c = 0; c < 6; ++c) {
    int piv = c;
    double mx = fabs(M[c][c]);
    for (int r = c + 1; r < 6; ++r)
      if (fabs(M[r][c]) > mx) {
        mx = fabs(M[r][c]);
        piv = r;
      }
    if (piv != c)
      for (int k = 0; k < 7; ++k) {
        double tmp = M[c][k];
        M[c][k] = M[piv][k];
        M[piv][k] = tmp;
      }
    double inv = 1.0 / M[c][c];
    for (int k = 0; k < 7; ++k) M[c][k] *= inv;
    for (int r = 0; r < 6; ++r)
      if (r != c) {
        double f = M[r][c];
        for (int k = 0; k < 7; ++k) M[r][k] -= f * M[c][k];
      }
  }
}

__global__ __launch_bounds__(1024) void k_irls(const float* __restrict__ pts0, const float* __restrict__ pts1,
                                               const float* __restrict__ weight0, int64_t n,
                                               float* __restrict__ cur, float* __restrict__ w,
                                               float* __restrict__ T_out) {
  __shared__ double s_red[16][27];
  __shared__ double s_M[6][7];
  __shared__ float s_T[16];   // accumulated transform
  __shared__ float s_Tc[12];  // current step [R|t]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int64_t i = tid; i < n; i += 1024) {
    cur[3 * i] = pts0[3 * i]; cur[3 * i + 1] = pts0[3 * i + 1]; cur[3 * i + 2] = pts0[3 * i + 2];
    w[i] = weight0 ? weight0[i] : 1.f;
  }
  if (tid < 16) s_T[tid] = (tid % 5 == 0) ? 1.f : 0.f;
  __syncthreads();
  float par = 1.0f;
  for (int iter = 0; iter < 20; ++iter) {
    if (iter > 0 && iter % 5 == 0) par *= 0.5f;
    // normal equations: rows a0=[0,z,-y,1,0,0], a1=[-z,0,x,0,1,0], a2=[y,-x,0,0,0,1], all scaled by w
    double acc[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[k] = 0.0;
    for (int64_t i = tid; i < n; i += 1024) {
      double x = cur[3 * i], y = cur[3 * i + 1], z = cur[3 * i + 2];
      double ww = (double)w[i] * (double)w[i];
      double bx = (double)pts1[3 * i] - x, by = (double)pts1[3 * i + 1] - y, bz = (double)pts1[3 * i + 2] - z;
      // AtA upper triangle (21) in row-major order, then Atb (6)
      acc[0] += ww * (z * z + y * y);  // 00
      acc[1] += ww * (-x * y);         // 01
      acc[2] += ww * (-x * z);         // 02
      acc[3] += 0.0;                   // 03
      acc[4] += ww * (-z);             // 04
      acc[5] += ww * (y);              // 05
      acc[6] += ww * (z * z + x * x);  // 11
      acc[7] += ww * (-y * z);         // 12
      acc[8] += ww * (z);              // 13
      acc[9] += 0.0;                   // 14
      acc[10] += ww * (-x);            // 15
      acc[11] += ww * (y * y + x * x); // 22
      acc[12] += ww * (-y);            // 23
      acc[13] += ww * (x);             // 24
      acc[14] += 0.0;                  // 25
      acc[15] += ww;                   // 33
      acc[16] += 0.0;                  // 34
      acc[17] += 0.0;                  // 35
      acc[18] += ww;                   // 44
      acc[19] += 0.0;                  // 45
      acc[20] += ww;                   // 55
      acc[21] += ww * (-z * by + y * bz);
      acc[22] += ww * (z * bx - x * bz);
      acc[23] += ww * (-y * bx + x * by);
      acc[24] += ww * bx;
      acc[25] += ww * by;
      acc[26] += ww * bz;
    }
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      double v = acc[k];
      for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
      if (lane == 0) s_red[wave][k] = v;
    }
    __syncthreads();
    if (tid == 0) {
      double tot[27];
      for (int k = 0; k < 27; ++k) {
        double v = 0.0;
        for (int wv = 0; wv < 16; ++wv) v += s_red[wv][k];
        tot[k] = v;
      }
      int p = 0;
      for (int r = 0; r < 6; ++r)
        for (int c = r; c < 6; ++c) {
          s_M[r][c] = tot[p];
          s_M[c][r] = tot[p];
          ++p;
        }
      for (int r = 0; r < 6; ++r) s_M[r][6] = tot[21 + r];
      double M[6][7];
      for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 7; ++c) M[r][c] = s_M[r][c];
      solve6(M);
      float x0 = (float)M[0][6], x1 = (float)M[1][6], x2 = (float)M[2][6];
      float cx = cosf(x0), sx = sinf(x0), cy = cosf(x1), sy = sinf(x1), cz = cosf(x2), sz = sinf(x2);
      // R = rot_z(x2) rot_y(x1) rot_x(x0)
      float R[3][3];
      R[0][0] = cz * cy; R[0][1] = cz * sy * sx - sz * cx; R[0][2] = cz * sy * cx + sz * sx;
      R[1][0] = sz * cy; R[1][1] = sz * sy * sx + cz * cx; R[1][2] = sz * sy * cx - cz * sx;
      R[2][0] = -sy;     R[2][1] = cy * sx;                R[2][2] = cy * cx;
      for (int a = 0; a < 3; ++a) {
        s_Tc[a * 4 + 0] = R[a][0]; s_Tc[a * 4 + 1] = R[a][1]; s_Tc[a * 4 + 2] = R[a][2];
        s_Tc[a * 4 + 3] = (float)M[3 + a][6];
      }
      // trans = trans_curr @ trans
      float Tn[12];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 4; ++b) {
          float v = s_Tc[a * 4 + 0] * s_T[0 * 4 + b] + s_Tc[a * 4 + 1] * s_T[1 * 4 + b] + s_Tc[a * 4 + 2] * s_T[2 * 4 + b];
          if (b == 3) v += s_Tc[a * 4 + 3];
          Tn[a * 4 + b] = v;
        }
      for (int k = 0; k < 12; ++k) s_T[k] = Tn[k];
    }
    __syncthreads();
    for (int64_t i = tid; i < n; i += 1024) {
      float x = cur[3 * i], y = cur[3 * i + 1], z = cur[3 * i + 2];
      float nx = s_Tc[0] * x + s_Tc[1] * y + s_Tc[2] * z + s_Tc[3];
      float ny = s_Tc[4] * x + s_Tc[5] * y + s_Tc[6] * z + s_Tc[7];
      float nz = s_Tc[8] * x + s_Tc[9] * y + s_Tc[10] * z + s_Tc[11];
      cur[3 * i] = nx; cur[3 * i + 1] = ny; cur[3 * i + 2] = nz;
      float dx = nx - pts1[3 * i], dy = ny - pts1[3 * i + 1], dz = nz - pts1[3 * i + 2];
      w[i] = par / (sqrtf(dx * dx + dy * dy + dz * dz) + par);
    }
    __syncthreads();
  }
  if (tid < 16) T_out[tid] = s_T[tid];
}

// ---- open3d 0.10/0.11 semantics: geometric validation of the first `max_validation` survivors ----
// rank[h] = number of survivors with a smaller iteration index (survivors are appended unordered)
__global__ void k_rank_by_iteration(const Hyp* __restrict__ hyps, const int* __restrict__ n_valid, int cap,
                                    int max_validation, int* __restrict__ selected) {
  const int nv = min(*n_valid, cap);
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= nv) return;
  const long long it = hyps[h].it;
  int rank = 0;
  for (int o = 0; o < nv; ++o) rank += hyps[o].it < it ? 1 : 0;
  selected[h] = rank < max_validation ? 1 : 0;
}

// Every transformed source point looks for its nearest target point within max_dist in a uniform grid whose cell edge
// is 2*max_dist, so the 2x2x2 block of cells around (p - max_dist) covers the search ball with 8 hash probes: inlier
// count + sum of squared NN distances (GetRegistrationResultAndCorrespondences of open3d <= 0.11).
// Work split: a hypothesis is dealt to S = min(kGeoGrid / nv, #256-point chunks) workgroups (nv = survivors on the
// device; with the usual handful of survivors ONE workgroup per hypothesis left 5 workgroups walking 5000 points each
// through chains of dependent loads: 210 us).  A workgroup adds the (count, sum) of its chunks in chunk order into its
// slot part[h * S + p]; k_score_geometric_finish adds the S slots in order: a fixed summation order for given inputs.
__global__ __launch_bounds__(256) void k_score_geometric(const float* __restrict__ xyz0, int64_t n0,
                                                         const float* __restrict__ xyz1, AprSearchGrid g,
                                                         double max_dist, const Hyp* __restrict__ hyps,
                                                         const int* __restrict__ n_valid, int cap,
                                                         const int* __restrict__ selected, GeoPart* __restrict__ part) {
  __shared__ int s_cnt[4];
  __shared__ double s_e2[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = min(*n_valid, cap);
  const int S = geo_parts(nv, n0);
  const int nchunk = (int)((n0 + 255) >> 8);
  const float md2 = (float)(max_dist * max_dist);
  const float inv_cell = 1.0f / g.cell;
  for (int64_t item = blockIdx.x; item < (int64_t)nv * S; item += gridDim.x) {
    const int h = (int)(item / S), p = (int)(item - (int64_t)h * S);
    if (!selected[h]) continue;                      // workgroup-uniform
    double T[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = hyps[h].T[k];
    const int c0 = (int)((int64_t)nchunk * p / S), c1 = (int)((int64_t)nchunk * (p + 1) / S);
    int cnt_total = 0;
    double e2_total = 0.0;
    for (int c = c0; c < c1; ++c) {
      const int64_t i = (int64_t)c * 256 + threadIdx.x;
      int cnt = 0;
      double e2 = 0.0;
      if (i < n0) {
        const double sx = xyz0[3 * i], sy = xyz0[3 * i + 1], sz = xyz0[3 * i + 2];
        const float px = (float)(T[0] * sx + T[1] * sy + T[2] * sz + T[3]);
        const float py = (float)(T[4] * sx + T[5] * sy + T[6] * sz + T[7]);
        const float pz = (float)(T[8] * sx + T[9] * sy + T[10] * sz + T[11]);
        const int bx = (int)floorf((px - g.mins[0]) * inv_cell - 0.5f), by = (int)floorf((py - g.mins[1]) * inv_cell - 0.5f),
                  bz = (int)floorf((pz - g.mins[2]) * inv_cell - 0.5f);
        float best = md2;
        bool found = false;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
          const int x = bx + (o & 1), y = by + ((o >> 1) & 1), z = bz + (o >> 2);
          if (!apr_key_in_range(0, x, y, z)) continue;
          const int id = apr_table_lookup(g.keys, g.vals, g.mask, apr_pack_key(0, x, y, z));
          if (id < 0) continue;
          for (int a = g.start[id]; a < g.start[id + 1]; ++a) {
            const int j = g.sorted[a];
            const float ex = px - xyz1[3 * (int64_t)j], ey = py - xyz1[3 * (int64_t)j + 1],
                        ez = pz - xyz1[3 * (int64_t)j + 2];
            const float d2 = ex * ex + ey * ey + ez * ez;
            if (d2 < best) {
              best = d2;
              found = true;
            }
          }
        }
        if (found) {
          cnt = 1;
          e2 = (double)best;
        }
      }
      for (int d = 32; d >= 1; d >>= 1) {
        cnt += __shfl_xor(cnt, d);
        e2 += __shfl_xor(e2, d);
      }
      __syncthreads();
      if (lane == 0) {
        s_cnt[wave] = cnt;
        s_e2[wave] = e2;
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        cnt_total += s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        e2_total += (s_e2[0] + s_e2[1]) + (s_e2[2] + s_e2[3]);
      }
    }
    if (threadIdx.x == 0) {
      part[item].cnt = cnt_total;
      part[item].e2 = e2_total;
    }
  }
}

__global__ void k_score_geometric_finish(Hyp* __restrict__ hyps, const int* __restrict__ n_valid, int cap, int64_t n0,
                                         const int* __restrict__ selected, const GeoPart* __restrict__ part) {
  const int nv = min(*n_valid, cap);
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= nv) return;
  if (!selected[h]) {
    hyps[h].inliers = -1;
    return;
  }
  const int S = geo_parts(nv, n0);
  int cnt = 0;
  double e2 = 0.0;
  for (int p = 0; p < S; ++p) {
    cnt += part[(int64_t)h * S + p].cnt;
    e2 += part[(int64_t)h * S + p].e2;
  }
  hyps[h].inliers = cnt;
  hyps[h].err2 = e2;
}

constexpr int64_t kChunk = 1 << 20;

// smallest double t with sqrt(t) >= m: sqrt(x) < m <=> x < t for every x >= 0 (sqrt is monotone, correctly rounded)
static double sqrt_lt_threshold(double m) {
  double t = m * m;
  while (t > 0.0 && sqrt(t) >= m) t = nextafter(t, 0.0);
  while (sqrt(t) < m) t = nextafter(t, INFINITY);
  return t;
}
// largest double u with sqrt(u) <= m: sqrt(x) > m <=> x > u
static double sqrt_gt_threshold(double m) {
  double u = m * m;
  while (sqrt(u) <= m) u = nextafter(u, INFINITY);
  while (u > 0.0 && sqrt(u) > m) u = nextafter(u, 0.0);
  return u;
}

struct RansacScratch {
  Hyp* best;
  long long* total_valid;
  int* n_valid;
  int* n_cand;
  Hyp* hyps;
  float4* rec;
  long long* cand;
  GeoPart* part;   // [kGeoGrid] partial scores of k_score / k_score_geometric's split form
  float* rec2;     // [rec2_rows(n0)][12] the correspondences for k_count's scalar loads
  unsigned* maxn2; // bits of the largest squared point norm; sel_hdr = maxn2 + 1 (count of picked hypotheses, max inliers)
  int* sel_hdr;
  int* sel;        // [cap] picked hypotheses
  unsigned* band;  // [cap][band_words(n0)] flagged mini-chunks
  SelPart* selp;   // [kSelParts]
  unsigned long long* rec8;   // [n0] quantised correspondences for k_sample_screen
  float4* rec_live;           // [2 n0] the correspondences k_prune kept, both layouts, ...
  float* rec2_live;
  int* live;                  // ... their counts and the NEAR / FAR list lengths (kLiveInts)
  int *order_near, *order_far;      // [cap] each
  unsigned char* near_flag;   // [cap]
  char* end;
};

static int band_words(int64_t n0) { return (int)((rec2_rows(n0) / kMini + 31) / 32); }

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t ransac_core_bytes(int64_t n0, int64_t max_iter) {
  const int64_t cap = max_iter < kChunk ? max_iter : kChunk;
  const size_t c1 = (size_t)(cap < 1 ? 1 : cap);
  return 768 + align256(c1 * sizeof(Hyp)) + align256((size_t)n0 * 32) +
         align256(((size_t)(max_iter < 1 ? 1 : max_iter) + kCandLists * 256) * 8) + align256(kGeoGrid * sizeof(GeoPart)) +
         align256((size_t)rec2_rows(n0) * 48) + 256 + align256(c1 * 4) + align256(c1 * band_words(n0) * 4) +
         align256(kSelParts * sizeof(SelPart)) + align256((size_t)(n0 + 32) * 8) + 256 + align256((size_t)n0 * 32) +
         align256((size_t)rec2_rows(n0) * 48) + 256 + 2 * align256(c1 * 4) + align256(c1);
}

static RansacScratch carve_ransac(void* scratch, int64_t n0, int64_t max_iter) {
  const int64_t cap = max_iter < kChunk ? max_iter : kChunk;
  RansacScratch r;
  char* p = (char*)(((uintptr_t)scratch + 255) & ~(uintptr_t)255);
  r.best = (Hyp*)p;
  r.total_valid = (long long*)(p + sizeof(Hyp));
  r.n_valid = (int*)(p + sizeof(Hyp) + 8);
  r.n_cand = (int*)(p + 256);   // kCandLists counters; n_valid .. the last counter: ONE memset (kCountersBytes) clears them
  r.live = (int*)(p + 512);     // ... and the pruned list's counters behind them (counter_words covers both)
  p += 768;
  r.hyps = (Hyp*)p;
  p += align256((size_t)cap * sizeof(Hyp));
  r.rec = (float4*)p;
  p += align256((size_t)n0 * 32);
  r.cand = (long long*)p;
  p += align256(((size_t)max_iter + kCandLists * 256) * 8);
  r.part = (GeoPart*)p;
  p += align256(kGeoGrid * sizeof(GeoPart));
  r.rec2 = (float*)p;
  p += align256((size_t)rec2_rows(n0) * 48);
  r.maxn2 = (unsigned*)p;
  r.sel_hdr = (int*)p + 1;    // adjacent: one 12-byte memset clears the norm bound and the pick header
  p += 256;
  r.sel = (int*)p;
  p += align256((size_t)cap * 4);
  r.band = (unsigned*)p;
  p += align256((size_t)cap * band_words(n0) * 4);
  r.selp = (SelPart*)p;
  p += align256(kSelParts * sizeof(SelPart));
  r.rec8 = (unsigned long long*)p;
  p += align256((size_t)(n0 + 32) * 8);
  r.rec_live = (float4*)p;
  p += align256((size_t)n0 * 32);
  r.rec2_live = (float*)p;
  p += align256((size_t)rec2_rows(n0) * 48);
  r.order_near = (int*)p;
  p += align256((size_t)cap * 4);
  r.order_far = (int*)p;
  p += align256((size_t)cap * 4);
  r.near_flag = (unsigned char*)p;
  p += align256((size_t)cap);
  r.end = p;
  return r;
}

// correspondences -> records (both layouts) + the norm bound of k_count (cleared by k_init_best, which runs first)
static void launch_pack(const RansacScratch& r, const float* xyz0, const float* xyz1, int64_t n1, const int64_t* corr,
                        int64_t n0, hipStream_t st, int64_t* unpack_to = nullptr) {
  hipLaunchKernelGGL(k_pack_pairs, dim3((unsigned)cdiv64(2 * rec2_rows(n0), 256)), dim3(256), 0, st, xyz0, xyz1, n1,
                     (const long long*)corr, n0, r.rec, r.rec2, r.maxn2, r.rec2_live, (long long*)unpack_to);
  if (n0 <= kScreenMaxN0)      // the 8-byte records of k_sample_screen (needs the finished norm bound: its own launch)
    hipLaunchKernelGGL(k_pack_small, dim3((unsigned)cdiv64(n0 + 32, 256)), dim3(256), 0, st, r.rec, n0, r.maxn2, r.rec8);
}

// Which sampling kernel: k_sample_screen (default) takes the first edge out of a 112 KB table in LDS; it owns a CU's whole
// LDS while it runs.  Whether that costs a pipelined caller more than it saves could not be settled on this repo's
// benchmark (three same-box A/Bs: +0.8 %, -5 %, 0 %; with true matches in the set the screen is 6 % ahead), so the choice
// is the caller's: apr_ransac_set_screen; -1 = APR_RANSAC_SCREEN from the environment (default 1, read once: the A/B
// and test hook).
static std::atomic<int> g_ransac_screen{-1};

// A/B and test switches: set through apr_ransac_set_option (an atomic each), defaults from the environment read ONCE per
// process (round-4 advice: a getenv on every call from threads that run without the interpreter lock races with any
// setenv / putenv of the host application, and Python's os.environ assignments are putenv calls)
enum { kOptScreen = 0, kOptCount = 1, kOptPrune = 2, kOptForceRounds = 3, kOptN = 4 };
static std::atomic<int> g_ransac_opt[kOptN] = {{-1}, {-1}, {-1}, {-1}};
static int ransac_opt(int which) {
  const int v = g_ransac_opt[which].load(std::memory_order_relaxed);
  if (v >= 0) return v;
  static const int s_env[kOptN] = {env_int("APR_RANSAC_SCREEN", 1), env_int("APR_RANSAC_COUNT", 1),
                                   env_int("APR_RANSAC_PRUNE", 1), env_int("APR_RANSAC_FORCE_ROUNDS", 0)};
  return s_env[which];
}

// k_sample_screen's dynamic LDS (up to 160 KB) needs the per-device opt-in, once, under a lock (several host threads call in)
static bool screen_ready() {
  static std::mutex s_mu;
  static int s_state[64] = {};      // 0 = not tried, 1 = ok, 2 = failed
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  std::lock_guard<std::mutex> lk(s_mu);
  if (s_state[dev] == 0) {
    s_state[dev] = hipFuncSetAttribute((const void*)k_sample_screen, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024) == hipSuccess ? 1 : 2;
    if (s_state[dev] == 2) (void)hipGetLastError();      // the plain kernel takes over: do not leave the error for its check
  }
  return s_state[dev] == 1;
}

// inlier counts + squared errors of the hypothesis list, then the running best (see the comment above k_count)
static void launch_scoring(const RansacScratch& r, int64_t n0, double thr_lt, int cap, hipStream_t st) {
  const int nwords = band_words(n0), nmini = (int)(rec2_rows(n0) / kMini);
  // up to `few` survivors everything is scored in fp64 (option "count" = 0: always -- the A/B and test hook)
  const int few = ransac_opt(kOptCount) ? kGeoGrid : 0x7fffffff;
  // the survivors sorted into NEAR the first one (counted over the correspondences within reach of it) and FAR (over all)
  const int prune = ransac_opt(kOptPrune);               // A/B and test hook (apr_ransac_set_option)
  const double m_up = sqrt(thr_lt) * (1.0 + 1e-12);
  const int n_live_blocks = (int)cdiv64(n0, 256);
  hipLaunchKernelGGL(k_prune, dim3((unsigned)(n_live_blocks + 512)), dim3(256), 0, st, r.rec, n0, m_up, r.hyps, r.n_valid, cap, few,
                     r.maxn2, prune, n_live_blocks, r.rec_live, r.rec2_live, r.live, r.order_near, r.order_far, r.near_flag,
                     r.band, nwords);
  hipLaunchKernelGGL(k_count, dim3(4096), dim3(256), 0, st, r.rec2, r.rec2_live, nwords, nmini, thr_lt, r.maxn2, r.hyps,
                     r.n_valid, cap, r.band, few, r.order_near, r.order_far, r.live);
  hipLaunchKernelGGL(k_count_fix, dim3(4096), dim3(256), 0, st, r.rec, n0, thr_lt, r.hyps, r.n_valid, cap, r.band, nwords,
                     few, r.rec_live, r.live, r.near_flag);
  hipLaunchKernelGGL(k_count_max, dim3(256), dim3(256), 0, st, r.hyps, r.n_valid, cap, r.sel_hdr, few);
  hipLaunchKernelGGL(k_pick, dim3(256), dim3(256), 0, st, r.hyps, r.n_valid, cap, r.sel_hdr, r.sel, few);
  hipLaunchKernelGGL(k_score, dim3(kGeoGrid), dim3(kScoreThreads), 0, st, r.rec, n0, thr_lt, r.hyps, r.sel_hdr, r.sel, r.part);
  hipLaunchKernelGGL(k_score_finish, dim3(kGeoGrid / 256), dim3(256), 0, st, r.hyps, r.sel_hdr, r.sel, n0, r.part);
  hipLaunchKernelGGL(k_select_part, dim3(kSelParts), dim3(256), 0, st, r.hyps, r.n_valid, cap, r.selp);
  hipLaunchKernelGGL(k_select_final, dim3(1), dim3(kSelParts), 0, st, r.hyps, r.n_valid, r.selp, r.best, r.total_valid,
                     r.sel_hdr);
}

// sample + edge check over [it0, it1) -> compacted candidates -> Kabsch + distance check -> hypothesis list
// launches of the two sampling kernels since the library was loaded ([0] k_sample_check, [1] k_sample_screen): what a
// measurement reports instead of the switch it asked for (apr_ransac_sampling_launches)
static std::atomic<long long> g_sampling_launches[2];

static int counter_words(const RansacScratch& r) { return (int)((r.live + kLiveInts) - r.n_valid); }

static void launch_hypotheses(const RansacScratch& r, int64_t n0, double max_dist, double edge_ratio, int64_t it0,
                              int64_t it1, uint64_t seed, int cap, hipStream_t st, bool counters_cleared = false) {
  const int64_t nwg = cdiv64(it1 - it0, 256);
  int sub_cap = (int)(cdiv64(nwg, kCandLists) * 256);
  if (!counters_cleared) (void)hipMemsetAsync(r.n_valid, 0, (size_t)counter_words(r) * 4, st);
  // many iterations over a table that fits the LDS: the first edge is screened there (k_sample_screen; same candidates).
  // APR_RANSAC_SCREEN=0 / apr_ransac_set_screen(0) keeps the plain kernel.
  const int64_t niter = it1 - it0;
  const int screen_set = g_ransac_screen.load(std::memory_order_relaxed);
  const int want_screen = screen_set >= 0 ? screen_set : ransac_opt(kOptScreen);
  const bool screened = niter >= 64 * kScreenRound && n0 <= kScreenMaxN0 && want_screen && screen_ready();
  g_sampling_launches[screened ? 1 : 0].fetch_add(1, std::memory_order_relaxed);
  if (screened) {
    const int64_t swg = 256;                                     // one 1024-thread workgroup per CU; a multiple of kCandLists
    const int64_t per_wg = cdiv64(niter, swg);
    sub_cap = (int)((swg / kCandLists) * per_wg);                // <= niter / 64 + 4: inside the candidate region
    const size_t lds = (size_t)((n0 + 31) & ~(int64_t)31) * 8 + (size_t)kScreenQueue * 4 + 16;
    hipLaunchKernelGGL(k_sample_screen, dim3((unsigned)swg), dim3(kScreenThreads), lds, st, r.rec, r.rec8, (uint32_t)n0,
                       edge_ratio, (long long)it0, (long long)it1, (long long)per_wg, seed, r.cand, r.n_cand, sub_cap);
  } else
  hipLaunchKernelGGL(k_sample_check, dim3((unsigned)nwg), dim3(256), 0, st, r.rec, (uint32_t)n0, edge_ratio,
                     (long long)it0, (long long)it1, seed, r.cand, r.n_cand, sub_cap);
  hipLaunchKernelGGL(k_fit_check, dim3(512), dim3(256), 0, st, r.rec, (uint32_t)n0, sqrt_gt_threshold(max_dist), seed,
                     r.cand, r.n_cand, sub_cap, r.hyps, r.n_valid, cap);
}

static int fetch_result(const RansacScratch& r, double* result_host, long long* tv_out, hipStream_t st) {
  Hyp hb;
  long long tv = 0;
  APR_HIP(hipMemcpyAsync(&hb, r.best, sizeof(Hyp), hipMemcpyDeviceToHost, st));
  APR_HIP(hipMemcpyAsync(&tv, r.total_valid, 8, hipMemcpyDeviceToHost, st));
  APR_HIP(hipStreamSynchronize(st));
  for (int k = 0; k < 12; ++k) result_host[k] = hb.T[k];
  result_host[12] = 0.0; result_host[13] = 0.0; result_host[14] = 0.0; result_host[15] = 1.0;
  result_host[16] = (double)(hb.inliers < 0 ? 0 : hb.inliers);
  result_host[17] = hb.inliers > 0 ? sqrt(hb.err2 / (double)hb.inliers) : 0.0;
  result_host[18] = (double)hb.it;
  result_host[19] = (double)tv;
  *tv_out = tv;
  return APR_OK;
}

}  // namespace

APR_API size_t apr_ransac_geometric_scratch_bytes(int64_t n0, int64_t n1, int64_t max_iter) {
  const size_t cap = (size_t)(max_iter < kChunk ? max_iter : kChunk);
  return ransac_core_bytes(n0, max_iter) + align256(cap * 4) + 256 + apr_internal_grid_bytes(n1) + 256 +
         align256((cap + kGeoGrid) * sizeof(GeoPart));
}

namespace {
int geometric_enqueue(const float* xyz0, int64_t n0, const float* xyz1, int64_t n1, const int64_t* corr,
                      double max_dist, double edge_ratio, int64_t max_iter, int64_t max_validation, uint64_t seed,
                      void* scratch, size_t scratch_bytes, RansacScratch* r_out, hipStream_t st) {
  APR_CHECK_ARG(n0 > 0 && n0 < (1ll << 31) && n1 > 0 && n1 < (1ll << 31), "apr_ransac_pose_geometric: empty point set");
  APR_CHECK_ARG(max_iter > 0 && max_iter <= kChunk && max_validation > 0 && max_dist > 0,
                "apr_ransac_pose_geometric: need 0 < max_iter <= %lld and max_validation > 0", (long long)kChunk);
  APR_CHECK_ARG(scratch_bytes >= apr_ransac_geometric_scratch_bytes(n0, n1, max_iter),
                "apr_ransac_pose_geometric: scratch too small");
  const int64_t cap = max_iter;
  const RansacScratch r = carve_ransac(scratch, n0, max_iter);
  int* selected = (int*)r.end;
  void* grid_scratch = (void*)(r.end + align256((size_t)cap * 4));
  GeoPart* part = (GeoPart*)((char*)grid_scratch + align256(apr_internal_grid_bytes(n1)) + 256);
  AprSearchGrid g;
  int rc = apr_internal_search_grid(xyz1, n1, (float)(2.0 * max_dist), grid_scratch, &g, st);
  if (rc != APR_OK) return rc;
  hipLaunchKernelGGL(k_init_best, dim3(1), dim3(1), 0, st, r.best, r.total_valid, r.maxn2, (int*)nullptr, 0);
  launch_pack(r, xyz0, xyz1, n1, corr, n0, st);
  launch_hypotheses(r, n0, max_dist, edge_ratio, 0, max_iter, seed, (int)cap, st);
  hipLaunchKernelGGL(k_rank_by_iteration, dim3((unsigned)cdiv64(cap, 256)), dim3(256), 0, st, r.hyps, r.n_valid,
                     (int)cap, (int)(max_validation < (1ll << 30) ? max_validation : (1ll << 30)), selected);
  hipLaunchKernelGGL(k_score_geometric, dim3(kGeoGrid), dim3(256), 0, st, xyz0, n0, xyz1, g, max_dist, r.hyps, r.n_valid,
                     (int)cap, selected, part);
  hipLaunchKernelGGL(k_score_geometric_finish, dim3((unsigned)cdiv64(cap, 256)), dim3(256), 0, st, r.hyps, r.n_valid,
                     (int)cap, n0, selected, part);
  hipLaunchKernelGGL(k_select, dim3(1), dim3(1024), 0, st, r.hyps, r.n_valid, (int)cap, r.best, r.total_valid);
  APR_LAUNCH_CHECK();
  *r_out = r;
  return APR_OK;
}

void decode_result(const Hyp& hb, long long tv, double* result_host) {
  for (int k = 0; k < 12; ++k) result_host[k] = hb.T[k];
  result_host[12] = 0.0; result_host[13] = 0.0; result_host[14] = 0.0; result_host[15] = 1.0;
  result_host[16] = (double)(hb.inliers < 0 ? 0 : hb.inliers);
  result_host[17] = hb.inliers > 0 ? sqrt(hb.err2 / (double)hb.inliers) : 0.0;
  result_host[18] = (double)hb.it;
  result_host[19] = (double)tv;
}
}  // namespace

APR_API int apr_ransac_pose_geometric(const float* xyz0, int64_t n0, const float* xyz1, int64_t n1,
                                      const int64_t* corr, double max_dist, double edge_ratio, int64_t max_iter,
                                      int64_t max_validation, uint64_t seed, void* scratch, size_t scratch_bytes,
                                      double* result_host, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  RansacScratch r;
  int rc = geometric_enqueue(xyz0, n0, xyz1, n1, corr, max_dist, edge_ratio, max_iter, max_validation, seed, scratch,
                             scratch_bytes, &r, st);
  if (rc != APR_OK) return rc;
  long long tv;
  return fetch_result(r, result_host, &tv, st);
}

// The same without the host synchronisation: the raw result (apr_ransac_raw_bytes() bytes) is copied to `raw_dev`, a
// device buffer of the caller's, so that the scratch can serve the next pair at once; the caller fetches the raw
// results of a whole batch with one copy and decodes them on the host with apr_ransac_decode.
APR_API size_t apr_ransac_raw_bytes(void) { return sizeof(Hyp) + 8; }

APR_API int apr_ransac_pose_geometric_async(const float* xyz0, int64_t n0, const float* xyz1, int64_t n1,
                                            const int64_t* corr, double max_dist, double edge_ratio, int64_t max_iter,
                                            int64_t max_validation, uint64_t seed, void* scratch, size_t scratch_bytes,
                                            void* raw_dev, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APR_CHECK_ARG(raw_dev, "apr_ransac_pose_geometric_async: raw_dev is NULL");
  RansacScratch r;
  int rc = geometric_enqueue(xyz0, n0, xyz1, n1, corr, max_dist, edge_ratio, max_iter, max_validation, seed, scratch,
                             scratch_bytes, &r, st);
  if (rc != APR_OK) return rc;
  APR_HIP(hipMemcpyAsync(raw_dev, r.best, sizeof(Hyp), hipMemcpyDeviceToDevice, st));
  APR_HIP(hipMemcpyAsync((char*)raw_dev + sizeof(Hyp), r.total_valid, 8, hipMemcpyDeviceToDevice, st));
  return APR_OK;
}

APR_API int apr_ransac_decode(const void* raw_host, double* result_host) {
  APR_CHECK_ARG(raw_host && result_host, "apr_ransac_decode: NULL argument");
  Hyp hb;
  long long tv;
  memcpy(&hb, raw_host, sizeof(Hyp));
  memcpy(&tv, (const char*)raw_host + sizeof(Hyp), 8);
  decode_result(hb, tv, result_host);
  return APR_OK;
}

APR_API size_t apr_ransac_scratch_bytes(int64_t n0, int64_t max_iter) { return ransac_core_bytes(n0, max_iter); }

APR_API int apr_ransac_pose(const float* xyz0, int64_t n0, const float* xyz1, int64_t n1, const int64_t* corr,
                            double max_dist, double edge_ratio, int64_t max_iter, uint64_t seed, void* scratch,
                            size_t scratch_bytes, double* result_host, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APR_CHECK_ARG(n0 > 0 && n0 < (1ll << 31) && n1 > 0, "apr_ransac_pose: empty point set");
  APR_CHECK_ARG(max_iter > 0 && max_iter < (1ll << 31) && max_dist > 0, "apr_ransac_pose: bad max_iter / max_dist");
  APR_CHECK_ARG(scratch_bytes >= apr_ransac_scratch_bytes(n0, max_iter), "apr_ransac_pose: scratch too small");
  const int64_t cap = max_iter < kChunk ? max_iter : kChunk;
  const RansacScratch r = carve_ransac(scratch, n0, max_iter);
  const double thr_lt = sqrt_lt_threshold(max_dist);
  hipLaunchKernelGGL(k_init_best, dim3(1), dim3(1), 0, st, r.best, r.total_valid, r.maxn2, (int*)nullptr, 0);   // clears the norm bound + pick header
  launch_pack(r, xyz0, xyz1, n1, corr, n0, st);
  // Fast path: ALL iterations in one round.  The hypothesis list holds kChunk entries; only if more than that
  // survive both checkers (near-perfect correspondences) the rounds are replayed kChunk iterations at a time,
  // where the list cannot overflow.  total_valid tells which case it was.
  const int s_force_rounds = ransac_opt(kOptForceRounds);   // test hook (apr_ransac_set_option): skip the fast path
  bool packed_fresh = true;      // the pruned list of the count path must start out as all padding for EVERY scoring round
  for (int pass = s_force_rounds ? 1 : 0; pass < 2; ++pass) {
    const int64_t step = pass == 0 ? max_iter : cap;
    hipLaunchKernelGGL(k_init_best, dim3(1), dim3(1), 0, st, r.best, r.total_valid, (unsigned*)nullptr, (int*)nullptr, 0);
    for (int64_t it0 = 0; it0 < max_iter; it0 += step) {
      const int64_t it1 = (it0 + step < max_iter) ? it0 + step : max_iter;
      if (!packed_fresh) launch_pack(r, xyz0, xyz1, n1, corr, n0, st);
      packed_fresh = false;
      launch_hypotheses(r, n0, max_dist, edge_ratio, it0, it1, seed, (int)cap, st);
      launch_scoring(r, n0, thr_lt, (int)cap, st);
    }
    APR_LAUNCH_CHECK();
    long long tv;
    int rc = fetch_result(r, result_host, &tv, st);
    if (rc != APR_OK) return rc;
    if (tv <= cap || step == cap) break;   // nothing was dropped
  }
  return APR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// B pairs in ONE call: per pair the feature NN (filter + refine, or brute force for other channel counts) and the
// single-round RANSAC are enqueued back to back on `stream`; the big work buffers are shared (stream order
// serialises the pairs), every pair keeps its own correspondence array and a small result slot; ONE synchronise
// fetches all results.  A pair whose hypothesis list overflowed (more than 2^20 valid hypotheses) is redone through
// apr_ransac_pose, whose chunked rounds cannot overflow.  Same results as B x (apr_feature_nn_fast, apr_nn_unpack,
// apr_ransac_pose); removes 2 host round trips and ~15 library calls per pair from the caller.
// ---------------------------------------------------------------------------------------------------------------
namespace {

// The pairs of a batch are independent and most of their ~25 kernels are small (a 28 k-point pair: grids of a few
// hundred workgroups, 5-60 us each), so run back to back on one stream they leave the card mostly idle.  The batch is
// therefore dealt over `lanes` streams: lane 0 is the caller's stream, the others are library-owned streams forked from
// it by an event and joined back before the result copy -- to the caller it is still ONE stream's worth of ordering.
// Each lane has its own matching / RANSAC scratch (pairs of a lane reuse it in stream order).
// OFF by default (APR_MATCH_LANES=1): it is a remedy for a caller with ONE stream in flight (6-pair FCGF step on one
// stream: 1558 -> 1640 pairs/s with 3 lanes, 1718 with the host three steps ahead as well); a caller that already keeps
// several steps in flight on its own streams has the card full, and more concurrent small kernels cost it throughput
// (three streams: 2380 pairs/s with 1 lane, 2260 with 3; 2040 with 3 lanes and 8 hardware queues).
constexpr int kMaxLanes = 4;

std::atomic<int> g_match_lanes{0};      // 0 = not set: APR_MATCH_LANES, else 1

int match_lanes(int32_t B) {
  static const int s_env = env_int("APR_MATCH_LANES", 1);
  const int set = g_match_lanes.load(std::memory_order_relaxed);
  const int want = set > 0 ? set : s_env;
  const int l = want < 1 ? 1 : (want > kMaxLanes ? kMaxLanes : want);
  return B < l ? B : l;
}

struct BatchLayout {
  size_t nn_scratch, best, ransac, lane, slots, corr_each, total;   // lane = nn_scratch + best + ransac, once per lane
  int lanes;
};

BatchLayout batch_layout(int32_t B, int64_t n0_max, int64_t n1_max, int32_t c, int64_t max_iter) {
  BatchLayout L;
  L.lanes = match_lanes(B);
  L.nn_scratch = align256(apr_feature_nn_fast_scratch_bytes(n0_max, n1_max, c) + 256);
  L.best = align256((size_t)n0_max * 8);
  L.ransac = align256(ransac_core_bytes(n0_max, max_iter) + 256);
  L.lane = L.nn_scratch + L.best + L.ransac;
  L.slots = align256((size_t)B * (sizeof(Hyp) + 64));
  L.corr_each = align256((size_t)n0_max * 8);
  L.total = (size_t)L.lanes * L.lane + L.slots + (size_t)B * L.corr_each + 512;
  return L;
}

// Scratch layout: [result slots][correspondences of the B pairs][lane 0][lane 1] ...  Everything apr_match_pose_batch_finish
// reads (slots, correspondences, lane 0's RANSAC scratch for the overflow replay) sits at positions that do NOT depend on
// the lane count, and the enqueue call uses as many lanes as the scratch it was handed holds -- so a change of
// apr_match_pose_set_lanes between scratch_bytes / enqueue / finish of a batch in flight can cost lanes, never correctness.
struct BatchViews {
  char *slots, *corr_base, *lane_base;
  int lanes;
};

BatchViews batch_views(const BatchLayout& L, int32_t B, void* scratch, size_t scratch_bytes) {
  BatchViews v;
  char* p = (char*)(((uintptr_t)scratch + 255) & ~(uintptr_t)255);
  v.slots = p;                          p += L.slots;
  v.corr_base = p;                      p += (size_t)B * L.corr_each;
  v.lane_base = p;
  const size_t used = (size_t)(p - (char*)scratch);
  const size_t fit = scratch_bytes > used ? (scratch_bytes - used) / L.lane : 0;
  v.lanes = (int)(fit < (size_t)L.lanes ? fit : (size_t)L.lanes);
  return v;
}

// Library-owned side streams, kMaxLanes - 1 per (device, caller stream), created on first use and kept for the life of
// the process (a caller stream's side streams carry only work forked from it, so two caller streams never share one).
struct SideStreams { hipStream_t s[kMaxLanes - 1]; };

int side_streams(hipStream_t caller, SideStreams* out) {
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, SideStreams> pools;
  int dev = 0;
  APR_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  auto it = pools.find({dev, caller});
  if (it == pools.end()) {
    SideStreams ss;
    for (int i = 0; i < kMaxLanes - 1; ++i) APR_HIP(hipStreamCreateWithFlags(&ss.s[i], hipStreamNonBlocking));
    it = pools.emplace(std::make_pair(dev, caller), ss).first;
  }
  *out = it->second;
  return APR_OK;
}

}  // namespace

// Streams the pairs of a batch are dealt over (1 .. 4; see match_lanes above).  Takes effect for the calls that follow; a
// batch enqueued with a scratch sized under a smaller setting simply runs on the lanes that scratch holds, and the finish
// call of a batch in flight does not depend on the setting (layout above).
// 1: k_sample_screen (lowest latency alone on the card), 0: k_sample_check (friendlier to other streams' kernels), -1: the
// environment's APR_RANSAC_SCREEN (default 1).  Same candidates either way.  Takes effect for the calls that follow.
APR_API int apr_ransac_set_screen(int32_t mode) {
  APR_CHECK_ARG(mode >= -1 && mode <= 1, "apr_ransac_set_screen: mode -1, 0 or 1");
  g_ransac_screen.store(mode, std::memory_order_relaxed);
  return APR_OK;
}

// The sampling kernel that actually RAN: launch counts since load, out[0] = k_sample_check, out[1] = k_sample_screen (the
// library falls back to the former when the table does not fit the LDS, the call has few iterations, or the 160 KB opt-in
// failed -- whatever apr_ransac_set_screen / APR_RANSAC_SCREEN asked for).
// option: 0 = screen (as apr_ransac_set_screen), 1 = count (0: score every survivor in fp64), 2 = prune (0: count over all
// correspondences), 3 = force_rounds (1: skip the single-round fast path).  value -1: back to the environment's default
// (APR_RANSAC_SCREEN / _COUNT / _PRUNE / _FORCE_ROUNDS, read once per process).  Same results whatever the settings.
APR_API int apr_ransac_set_option(int32_t option, int32_t value) {
  APR_CHECK_ARG(option >= 0 && option < kOptN && value >= -1 && value <= 1, "apr_ransac_set_option: option 0 .. 3, value -1, 0 or 1");
  if (option == kOptScreen) g_ransac_screen.store(value, std::memory_order_relaxed);
  g_ransac_opt[option].store(value, std::memory_order_relaxed);
  return APR_OK;
}

APR_API int apr_ransac_sampling_launches(int64_t* out2) {
  APR_CHECK_ARG(out2 != nullptr, "apr_ransac_sampling_launches: NULL output");
  out2[0] = g_sampling_launches[0].load(std::memory_order_relaxed);
  out2[1] = g_sampling_launches[1].load(std::memory_order_relaxed);
  return APR_OK;
}

APR_API int apr_match_pose_set_lanes(int32_t lanes) {
  APR_CHECK_ARG(lanes >= 1 && lanes <= kMaxLanes, "apr_match_pose_set_lanes: 1 .. %d lanes", kMaxLanes);
  g_match_lanes.store(lanes, std::memory_order_relaxed);
  return APR_OK;
}

APR_API size_t apr_match_pose_batch_scratch_bytes(int32_t B, int64_t n0_max, int64_t n1_max, int32_t c,
                                                  int64_t max_iter) {
  if (B <= 0 || n0_max <= 0 || n1_max <= 0 || c <= 0 || max_iter <= 0) return 0;
  return batch_layout(B, n0_max, n1_max, c, max_iter).total;
}

APR_API size_t apr_match_pose_batch_slot_bytes(int32_t B) { return B > 0 ? (size_t)B * (sizeof(Hyp) + 64) : 0; }

// Everything of apr_match_pose_batch up to the copy of the B result slots into slots_host (pinned memory of
// apr_match_pose_batch_slot_bytes(B) bytes for the copy to be asynchronous): NO host synchronisation.
APR_API int apr_match_pose_batch_enqueue(const apr_pair_desc* pairs, int32_t B, int32_t c, double max_dist,
                                         double edge_ratio, int64_t max_iter, void* scratch, size_t scratch_bytes,
                                         void* slots_host, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APR_CHECK_ARG(B > 0 && B <= 4096 && pairs && slots_host, "apr_match_pose_batch: bad arguments");
  APR_CHECK_ARG(max_iter > 0 && max_iter < (1ll << 31) && max_dist > 0 && c > 0, "apr_match_pose_batch: bad parameters");
  int64_t n0_max = 0, n1_max = 0;
  for (int i = 0; i < B; ++i) {
    APR_CHECK_ARG(pairs[i].n0 > 0 && pairs[i].n1 > 0 && pairs[i].n0 < (1ll << 31) && pairs[i].n1 < (1ll << 31),
                  "apr_match_pose_batch: empty point set in pair %d", i);
    n0_max = pairs[i].n0 > n0_max ? pairs[i].n0 : n0_max;
    n1_max = pairs[i].n1 > n1_max ? pairs[i].n1 : n1_max;
  }
  const BatchLayout L = batch_layout(B, n0_max, n1_max, c, max_iter);
  const BatchViews V = batch_views(L, B, scratch, scratch_bytes);
  APR_CHECK_ARG(V.lanes >= 1, "apr_match_pose_batch: scratch too small");
  const int nlanes = V.lanes;
  char* const lane_base = V.lane_base;
  char* const slots = V.slots;
  char* const corr_base = V.corr_base;
  const int64_t cap = max_iter < kChunk ? max_iter : kChunk;
  const double thr_lt = sqrt_lt_threshold(max_dist);
  const bool fast_nn = (c == 32 || c == 64 || c == 128);
  // fork: the side lanes start behind everything the caller has queued so far (the descriptors, the points)
  hipStream_t lane_st[kMaxLanes] = {st};
  struct EventGuard {      // destroyed on every exit path (the runtime releases an event once it has completed)
    hipEvent_t e = nullptr;
    ~EventGuard() { if (e) (void)hipEventDestroy(e); }
  } fork;
  if (nlanes > 1) {
    SideStreams ss;
    const int rc = side_streams(st, &ss);
    if (rc != APR_OK) return rc;
    APR_HIP(hipEventCreateWithFlags(&fork.e, hipEventDisableTiming));
    APR_HIP(hipEventRecord(fork.e, st));
    for (int l = 1; l < nlanes; ++l) {
      lane_st[l] = ss.s[l - 1];
      APR_HIP(hipStreamWaitEvent(lane_st[l], fork.e, 0));
    }
  }
  int rc = APR_OK;
  for (int i = 0; i < B && rc == APR_OK; ++i) {
    const apr_pair_desc& d = pairs[i];
    const int l = i % nlanes;
    hipStream_t ls = lane_st[l];
    char* lp = lane_base + (size_t)l * L.lane;
    void* nn_scratch = lp;
    uint64_t* best = (uint64_t*)(lp + L.nn_scratch);
    void* ransac_scratch = lp + L.nn_scratch + L.best;
    rc = fast_nn ? apr_feature_nn_fast(d.f0, d.n0, d.f1, d.n1, c, best, nn_scratch, L.nn_scratch, ls)
                 : apr_feature_nn(d.f0, d.n0, d.f1, d.n1, c, best, ls);
    if (rc != APR_OK) break;
    int64_t* corr = (int64_t*)(corr_base + (size_t)i * L.corr_each);     // plain indices: written by k_pack_pairs below
    // single-round RANSAC into this pair's result slot
    RansacScratch r = carve_ransac(ransac_scratch, d.n0, max_iter);
    r.best = (Hyp*)(slots + (size_t)i * (sizeof(Hyp) + 64));
    r.total_valid = (long long*)((char*)r.best + sizeof(Hyp));
    hipLaunchKernelGGL(k_init_best, dim3(1), dim3(128), 0, ls, r.best, r.total_valid, r.maxn2, r.n_valid, counter_words(r));
    launch_pack(r, d.xyz0, d.xyz1, d.n1, (const int64_t*)best, d.n0, ls, corr);
    launch_hypotheses(r, d.n0, max_dist, edge_ratio, 0, max_iter, d.seed, (int)cap, ls, true);
    launch_scoring(r, d.n0, thr_lt, (int)cap, ls);
  }
  // join (also on an error above: whatever reached a side lane is ordered before the caller's next work)
  if (nlanes > 1) {
    for (int l = 1; l < nlanes; ++l) {
      EventGuard join;
      APR_HIP(hipEventCreateWithFlags(&join.e, hipEventDisableTiming));
      APR_HIP(hipEventRecord(join.e, lane_st[l]));
      APR_HIP(hipStreamWaitEvent(st, join.e, 0));
    }
  }
  if (rc != APR_OK) return rc;
  APR_LAUNCH_CHECK();
  APR_HIP(hipMemcpyAsync(slots_host, slots, (size_t)B * (sizeof(Hyp) + 64), hipMemcpyDeviceToHost, st));
  return APR_OK;
}

// Decode the slots once the stream work of the matching enqueue call has completed (the caller waited on the stream or
// on an event recorded behind it); `pairs`, `scratch` and the parameters must be the ones given to the enqueue call.
// A pair whose hypothesis list overflowed is redone through apr_ransac_pose (chunked rounds; synchronises).
APR_API int apr_match_pose_batch_finish(const apr_pair_desc* pairs, int32_t B, int32_t c, double max_dist,
                                        double edge_ratio, int64_t max_iter, void* scratch, size_t scratch_bytes,
                                        const void* slots_host, double* results_host, void* stream) {
  APR_CHECK_ARG(B > 0 && B <= 4096 && pairs && slots_host && results_host, "apr_match_pose_batch: bad arguments");
  int64_t n0_max = 0, n1_max = 0;
  for (int i = 0; i < B; ++i) {
    n0_max = pairs[i].n0 > n0_max ? pairs[i].n0 : n0_max;
    n1_max = pairs[i].n1 > n1_max ? pairs[i].n1 : n1_max;
  }
  const BatchLayout L = batch_layout(B, n0_max, n1_max, c, max_iter);
  const BatchViews V = batch_views(L, B, scratch, scratch_bytes);
  APR_CHECK_ARG(V.lanes >= 1, "apr_match_pose_batch: scratch too small");
  void* ransac_scratch = V.lane_base + L.nn_scratch + L.best;      // lane 0's
  char* const corr_base = V.corr_base;
  const int64_t cap = max_iter < kChunk ? max_iter : kChunk;
  const size_t slot_bytes = sizeof(Hyp) + 64;
  const char* host = (const char*)slots_host;
  int rc = APR_OK;
  for (int i = 0; i < B && rc == APR_OK; ++i) {
    Hyp hb;
    long long tv;
    memcpy(&hb, host + (size_t)i * slot_bytes, sizeof(Hyp));
    memcpy(&tv, host + (size_t)i * slot_bytes + sizeof(Hyp), 8);
    double* out = results_host + (size_t)i * 20;
    if (tv > cap && max_iter > cap) {   // hypothesis list overflowed: chunked rounds through the regular entry point
      const apr_pair_desc& d = pairs[i];
      rc = apr_ransac_pose(d.xyz0, d.n0, d.xyz1, d.n1, (const int64_t*)(corr_base + (size_t)i * L.corr_each), max_dist,
                           edge_ratio, max_iter, d.seed, ransac_scratch, L.ransac, out, stream);
      continue;
    }
    for (int k = 0; k < 12; ++k) out[k] = hb.T[k];
    out[12] = 0.0; out[13] = 0.0; out[14] = 0.0; out[15] = 1.0;
    out[16] = (double)(hb.inliers < 0 ? 0 : hb.inliers);
    out[17] = hb.inliers > 0 ? sqrt(hb.err2 / (double)hb.inliers) : 0.0;
    out[18] = (double)hb.it;
    out[19] = (double)tv;
  }
  return rc;
}

APR_API int apr_match_pose_batch(const apr_pair_desc* pairs, int32_t B, int32_t c, double max_dist, double edge_ratio,
                                 int64_t max_iter, void* scratch, size_t scratch_bytes, double* results_host,
                                 void* stream) {
  APR_CHECK_ARG(B > 0 && B <= 4096 && pairs && results_host, "apr_match_pose_batch: bad arguments");
  const size_t nb = apr_match_pose_batch_slot_bytes(B);
  char* host = (char*)malloc(nb);
  if (!host) {
    apr_set_error("apr_match_pose_batch: out of host memory");
    return APR_EINVAL;
  }
  int rc = apr_match_pose_batch_enqueue(pairs, B, c, max_dist, edge_ratio, max_iter, scratch, scratch_bytes, host, stream);
  if (rc == APR_OK && hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
    apr_set_error("apr_match_pose_batch: result fetch failed");
    rc = APR_EHIP;
  }
  if (rc == APR_OK)
    rc = apr_match_pose_batch_finish(pairs, B, c, max_dist, edge_ratio, max_iter, scratch, scratch_bytes, host,
                                     results_host, stream);
  free(host);
  return rc;
}

APR_API size_t apr_irls_scratch_bytes(int64_t n) { return (size_t)n * 16 + 256 + 64; }

APR_API int apr_irls_pose(const float* pts0, const float* pts1, const float* weight, int64_t n, float* T_host,
                          void* scratch, size_t scratch_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APR_CHECK_ARG(n > 0, "apr_irls_pose: n <= 0");
  APR_CHECK_ARG(scratch_bytes >= apr_irls_scratch_bytes(n), "apr_irls_pose: scratch too small");
  float* T_dev = (float*)scratch;
  float* cur = (float*)((char*)scratch + 64);
  float* w = cur + 3 * n;
  hipLaunchKernelGGL(k_irls, dim3(1), dim3(1024), 0, st, pts0, pts1, weight, n, cur, w, T_dev);
  APR_LAUNCH_CHECK();
  APR_HIP(hipMemcpyAsync(T_host, T_dev, 64, hipMemcpyDeviceToHost, st));
  APR_HIP(hipStreamSynchronize(st));
  return APR_OK;
}
