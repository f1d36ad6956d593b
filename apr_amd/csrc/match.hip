// Feature-space nearest neighbour (squared L2), SURVEY 8(a) row F9 / K9.
//
// The reference materialises a [500, N, C] broadcast difference per chunk
// (FCGF_APR/lib/metrics.py:22-29, lib/eval.py:18-48).  Here nothing is
// materialised: each thread keeps QPT query rows in registers, the target rows
// stream through the SCALAR path (wave-uniform s_load into SGPRs) and
// a running arg-min lives in registers; the target range is split over
// blockIdx.y so ~15 k queries still fill 256 CUs, partial results meet in one
// 64-bit atomicMin on (bits(d2) << 32 | j).
//
// The inner loop is packed fp32 (v_pk_add_f32 + v_pk_fma_f32, two channels per instruction): the query rows are
// held NEGATED so that the difference is the packed add b + (-a) = -(a - b), whose square is bit-identical.
//
// d2 is the direct form sum_c (a_c - b_c)^2 accumulated in a FIXED order:
// four fma chains (c mod 4, ascending c) combined as (s0+s1)+(s2+s3).  fp32
// MFMA runs at the fp32 VALU rate on gfx950, so the |a|^2 - 2ab + |b|^2 GEMM form
// would be no faster, and the fixed order makes the arg-min bit-reproducible by
// the C oracle (oracle/nn_oracle.c).
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kTR = 64;  // granularity of the target split

template <int C, int QPT>
__global__ __launch_bounds__(256) void k_feature_nn(const float* __restrict__ f0, int64_t n0,
                                                    const float* __restrict__ f1, int64_t n1,
                                                    int chunk, unsigned long long* __restrict__ best,
                                                    const unsigned* __restrict__ run_if) {
  if (run_if && *run_if == 0u) return;   // predicated launch (fallback of the filter + refine path)
  const int tid = threadIdx.x;
  const int64_t qbase = (int64_t)blockIdx.x * 256 * QPT;
  f32x2 nq[QPT][C / 2];   // NEGATED query rows: d = b + (-a) = -(a - b), d*d is bit-identical
  float bd[QPT];
  int bj[QPT];
#pragma unroll
  for (int u = 0; u < QPT; ++u) {
    int64_t qi = qbase + u * 256 + tid;
    bd[u] = __builtin_inff();
    bj[u] = 0x7fffffff;
    if (qi >= n0) qi = n0 - 1;
#pragma unroll
    for (int g = 0; g < C / 4; ++g) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(f0 + qi * C + g * 4);
      nq[u][2 * g] = (f32x2){-v[0], -v[1]};
      nq[u][2 * g + 1] = (f32x2){-v[2], -v[3]};
    }
  }
  const int64_t t0 = (int64_t)blockIdx.y * chunk;
  const int64_t t1 = min((long long)(t0 + chunk), (long long)n1);
  // the target row index is wave-uniform: the row arrives through scalar loads (s_load_dwordx16) into SGPRs
  // and feeds the packed VALU ops directly — no LDS staging, no barrier
  // (two SGPR row sets ping-pong: row j+1 is in flight while row j is consumed, no copies)
  float ra[C], rb[C];
  auto fetch = [&](float (&dst)[C], int64_t j) {
    // scalar loads return out of order, the only wait is lgkmcnt(0): drain the PREVIOUS fetch (issued one row
    // of VALU work ago) before issuing this one, so the wait in front of the next consume never stalls on it
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_sched_barrier(0);
    const float* __restrict__ row = f1 + (j < t1 ? j : t1 - 1) * C;
#pragma unroll
    for (int c = 0; c < C; ++c) dst[c] = row[c];
    __builtin_amdgcn_sched_barrier(0);   // keep the s_loads ahead of the VALU block they overlap with
  };
  auto consume = [&](const float (&cur)[C], int64_t j) {
    f32x2 sa[QPT], sb[QPT];   // chains (c mod 4) = {0,1} and {2,3}
#pragma unroll
    for (int u = 0; u < QPT; ++u) sa[u] = sb[u] = (f32x2){0.f, 0.f};
#pragma unroll
    for (int g = 0; g < C / 4; ++g) {
      const f32x2 b01 = {cur[4 * g], cur[4 * g + 1]}, b23 = {cur[4 * g + 2], cur[4 * g + 3]};
#pragma unroll
      for (int u = 0; u < QPT; ++u) {
        const f32x2 d01 = b01 + nq[u][2 * g], d23 = b23 + nq[u][2 * g + 1];
        sa[u] = __builtin_elementwise_fma(d01, d01, sa[u]);
        sb[u] = __builtin_elementwise_fma(d23, d23, sb[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < QPT; ++u) {
      const float d = (sa[u][0] + sa[u][1]) + (sb[u][0] + sb[u][1]);
      if (d < bd[u]) {
        bd[u] = d;
        bj[u] = (int)j;
      }
    }
  };
  if (t0 < t1) fetch(ra, t0);
  for (int64_t j = t0; j < t1; j += 2) {
    fetch(rb, j + 1);
    consume(ra, j);
    fetch(ra, j + 2);
    consume(rb, j + 1);   // odd tail: fetch() clamped to row t1-1 = row j again, equal d2 never replaces (strict <)
  }
#pragma unroll
  for (int u = 0; u < QPT; ++u) {
    int64_t qi = qbase + u * 256 + tid;
    if (qi < n0 && bj[u] != 0x7fffffff) {
      unsigned long long p = ((unsigned long long)__float_as_uint(bd[u]) << 32) | (unsigned)bj[u];
      atomicMin(&best[qi], p);
    }
  }
}

// wide features (C >= 64): a row no longer fits two SGPR sets, the (negated) target tile streams through LDS
// (broadcast ds_read_b128) instead
template <int C, int QPT>
__global__ __launch_bounds__(256) void k_feature_nn_lds(const float* __restrict__ f0, int64_t n0,
                                                        const float* __restrict__ f1, int64_t n1,
                                                        int chunk, unsigned long long* __restrict__ best,
                                                        const unsigned* __restrict__ run_if) {
  if (run_if && *run_if == 0u) return;
  __shared__ __attribute__((aligned(16))) float s_t[kTR * C];
  const int tid = threadIdx.x;
  const int64_t qbase = (int64_t)blockIdx.x * 256 * QPT;
  f32x4 qv[QPT][C / 4];
  float bd[QPT];
  int bj[QPT];
#pragma unroll
  for (int u = 0; u < QPT; ++u) {
    int64_t qi = qbase + u * 256 + tid;
    bd[u] = __builtin_inff();
    bj[u] = 0x7fffffff;
    if (qi >= n0) qi = n0 - 1;
#pragma unroll
    for (int g = 0; g < C / 4; ++g) qv[u][g] = *reinterpret_cast<const f32x4*>(f0 + qi * C + g * 4);
  }
  const int64_t t0 = (int64_t)blockIdx.y * chunk;
  const int64_t t1 = min((long long)(t0 + chunk), (long long)n1);
  for (int64_t tb = t0; tb < t1; tb += kTR) {
    const int rows = (int)min((long long)kTR, (long long)(t1 - tb));
    __syncthreads();
    for (int e = tid; e < rows * (C / 4); e += 256)
      *reinterpret_cast<f32x4*>(&s_t[e * 4]) = -*reinterpret_cast<const f32x4*>(f1 + tb * C + (int64_t)e * 4);
    __syncthreads();
    for (int r = 0; r < rows; ++r) {
      f32x2 sa[QPT], sb[QPT];
#pragma unroll
      for (int u = 0; u < QPT; ++u) sa[u] = sb[u] = (f32x2){0.f, 0.f};
#pragma unroll
      for (int g = 0; g < C / 4; ++g) {
        const f32x4 nb = *reinterpret_cast<const f32x4*>(&s_t[r * C + g * 4]);   // = -b
        const f32x2 nb01 = {nb[0], nb[1]}, nb23 = {nb[2], nb[3]};
#pragma unroll
        for (int u = 0; u < QPT; ++u) {
          const f32x2 q01 = {qv[u][g][0], qv[u][g][1]}, q23 = {qv[u][g][2], qv[u][g][3]};
          const f32x2 d01 = q01 + nb01, d23 = q23 + nb23;
          sa[u] = __builtin_elementwise_fma(d01, d01, sa[u]);
          sb[u] = __builtin_elementwise_fma(d23, d23, sb[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < QPT; ++u) {
        const float d = (sa[u][0] + sa[u][1]) + (sb[u][0] + sb[u][1]);
        if (d < bd[u]) {
          bd[u] = d;
          bj[u] = (int)(tb + r);
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < QPT; ++u) {
    int64_t qi = qbase + u * 256 + tid;
    if (qi < n0 && bj[u] != 0x7fffffff) {
      unsigned long long p = ((unsigned long long)__float_as_uint(bd[u]) << 32) | (unsigned)bj[u];
      atomicMin(&best[qi], p);
    }
  }
}

// any C (multiple of 4 not required): queries re-read from global, targets from LDS
__global__ __launch_bounds__(256) void k_feature_nn_generic(const float* __restrict__ f0, int64_t n0,
                                                            const float* __restrict__ f1, int64_t n1,
                                                            int c, int chunk,
                                                            unsigned long long* __restrict__ best) {
  const int64_t qi = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (qi >= n0) return;
  const int64_t t0 = (int64_t)blockIdx.y * chunk;
  const int64_t t1 = min((long long)(t0 + chunk), (long long)n1);
  float bd = __builtin_inff();
  int bj = 0x7fffffff;
  for (int64_t j = t0; j < t1; ++j) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < c; ++k) {
      float d = f0[qi * c + k] - f1[j * c + k];
      s[k & 3] = fmaf(d, d, s[k & 3]);
    }
    float d = (s[0] + s[1]) + (s[2] + s[3]);
    if (d < bd) {
      bd = d;
      bj = (int)j;
    }
  }
  if (bj != 0x7fffffff)
    atomicMin(&best[qi], ((unsigned long long)__float_as_uint(bd) << 32) | (unsigned)bj);
}

__global__ void k_nn_unpack(const unsigned long long* __restrict__ best, int64_t n,
                            long long* __restrict__ idx, float* __restrict__ d2) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long p = best[i];
  if (idx) idx[i] = (long long)(p & 0xffffffffull);
  if (d2) d2[i] = __uint_as_float((unsigned)(p >> 32));
}

template <int C, int QPT>
int launch_nn_q(const float* f0, int64_t n0, const float* f1, int64_t n1, unsigned long long* best,
                const unsigned* run_if, hipStream_t st) {
  const int64_t qblocks = cdiv64(n0, 256 * QPT);
  // enough target chunks to give >= ~4 workgroups per CU, at least one 64-row granule each
  int64_t want = cdiv64(1024, qblocks);
  int64_t chunk = cdiv64(cdiv64(n1, want), kTR) * kTR;
  if (chunk < 4 * kTR) chunk = 4 * kTR;
  const int64_t msplit = cdiv64(n1, chunk);
  if (C <= 32)
    hipLaunchKernelGGL((k_feature_nn<C, QPT>), dim3((unsigned)qblocks, (unsigned)msplit), dim3(256), 0, st, f0,
                       n0, f1, n1, (int)chunk, best, run_if);
  else
    hipLaunchKernelGGL((k_feature_nn_lds<C, QPT>), dim3((unsigned)qblocks, (unsigned)msplit), dim3(256), 0, st,
                       f0, n0, f1, n1, (int)chunk, best, run_if);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

template <int C>
int launch_nn(const float* f0, int64_t n0, const float* f1, int64_t n1, unsigned long long* best,
              const unsigned* run_if, hipStream_t st) {
  static const int s_qpt = env_int("APR_NN_QPT", 0);
  if (C <= 32 && s_qpt == 4) return launch_nn_q<C, 4>(f0, n0, f1, n1, best, run_if, st);
  return launch_nn_q<C, (C <= 64) ? 2 : 1>(f0, n0, f1, n1, best, run_if, st);
}

}  // namespace

// brute force into an ALREADY initialised `best`, executed only if *run_if != 0 (device side); c in 16/32/64/128
int apr_internal_nn_brute(const float* f0, int64_t n0, const float* f1, int64_t n1, int32_t c, uint64_t* best,
                          const unsigned* run_if, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* b = (unsigned long long*)best;
  if (c == 16) return launch_nn<16>(f0, n0, f1, n1, b, run_if, st);
  if (c == 32) return launch_nn<32>(f0, n0, f1, n1, b, run_if, st);
  if (c == 64) return launch_nn<64>(f0, n0, f1, n1, b, run_if, st);
  if (c == 128) return launch_nn<128>(f0, n0, f1, n1, b, run_if, st);
  apr_set_error("apr_internal_nn_brute: unsupported channel count %d", c);
  return APR_EINVAL;
}

APR_API int apr_feature_nn(const float* f0, int64_t n0, const float* f1, int64_t n1, int32_t c,
                           uint64_t* best, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APR_CHECK_ARG(n0 >= 0 && n1 > 0 && n1 < (1ll << 31) && c > 0, "apr_feature_nn: bad shape n0=%lld n1=%lld c=%d",
                (long long)n0, (long long)n1, c);
  if (n0 == 0) return APR_OK;
  APR_HIP(hipMemsetAsync(best, 0xFF, (size_t)n0 * 8, st));
  unsigned long long* b = (unsigned long long*)best;
  const bool aligned = ((((uintptr_t)f0) | ((uintptr_t)f1)) & 15) == 0;
  if (aligned && c == 16) return launch_nn<16>(f0, n0, f1, n1, b, nullptr, st);
  if (aligned && c == 32) return launch_nn<32>(f0, n0, f1, n1, b, nullptr, st);
  if (aligned && c == 64) return launch_nn<64>(f0, n0, f1, n1, b, nullptr, st);
  if (aligned && c == 128) return launch_nn<128>(f0, n0, f1, n1, b, nullptr, st);
  const int64_t qblocks = cdiv64(n0, 256);
  int64_t chunk = cdiv64(n1, cdiv64(2048, qblocks));
  if (chunk < 64) chunk = 64;
  hipLaunchKernelGGL(k_feature_nn_generic, dim3((unsigned)qblocks, (unsigned)cdiv64(n1, chunk)), dim3(256), 0,
                     st, f0, n0, f1, n1, c, (int)chunk, b);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API int apr_nn_unpack(const uint64_t* best, int64_t n, int64_t* idx, float* d2, void* stream) {
  APR_CHECK_ARG(n >= 0, "apr_nn_unpack: n < 0");
  if (n == 0) return APR_OK;
  hipLaunchKernelGGL(k_nn_unpack, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned long long*)best, n, (long long*)idx, d2);
  APR_LAUNCH_CHECK();
  return APR_OK;
}
