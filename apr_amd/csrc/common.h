// Shared host/device helpers for libapr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/apr_hip.h"

// tuning / A-B switches read once from the environment
static inline int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

#define APR_API extern "C" __attribute__((visibility("default")))

void apr_set_error(const char* fmt, ...);

#define APR_CHECK_ARG(cond, ...)      \
  do {                                \
    if (!(cond)) {                    \
      apr_set_error(__VA_ARGS__);     \
      return APR_EINVAL;              \
    }                                 \
  } while (0)

#define APR_HIP(call)                                                            \
  do {                                                                           \
    hipError_t e_ = (call);                                                      \
    if (e_ != hipSuccess) {                                                      \
      apr_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      return APR_EHIP;                                                           \
    }                                                                            \
  } while (0)

#define APR_LAUNCH_CHECK() APR_HIP(hipGetLastError())

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// match.hip: brute-force NN into an initialised `best`, run only if *run_if != 0 (NULL: always)
int apr_internal_nn_brute(const float* f0, int64_t n0, const float* f1, int64_t n1, int32_t c, uint64_t* best,
                          const unsigned* run_if, void* stream);

// hash.hip: memset as a kernel of the library's own (APR_FILL_KERNEL=0: hipMemsetAsync)
int apr_internal_fill(void* ptr, int32_t byte_value, size_t bytes, hipStream_t st);

// dense.hip: [M, cin] x [cin, cout] with the sparse conv's epilogue (identity kernel map); _ok = shape is supported
bool apr_internal_dense_ok(int64_t M, int32_t cin, int32_t cout);
bool apr_internal_dense_rows_route(int64_t M, int32_t cin, int32_t cout);      // dense_rows.hip
int apr_internal_dense_gemm(const float* in, int64_t ldi, int64_t M, int32_t cin, int32_t cout, const float* wp,
                            const float* scale, const float* shift, const float* residual, int64_t ldr, int32_t relu,
                            float* out, int64_t ldo, hipStream_t st);

// norm.hip: y = act((x - mean) * rstd (+ residual)) per row segment, mean / rstd rebuilt from per-block column sums
// partial[blk][2][c] (fp64; segment s owns blocks blk0[s] .. blk0[s + 1]) that another kernel left behind
int apr_internal_norm_apply_partials(const float* x, int64_t ldx, int32_t c, const int64_t* seg_row0, const int* seg_blk0,
                                     int32_t nseg, const double* partial, float eps, const float* residual, int64_t ldr,
                                     int32_t act_mode, float negative_slope, float* y, int64_t ldy, hipStream_t st);

#ifdef __HIPCC__
// Inclusive prefix sum over the 64 lanes with DPP adds only (no ds_bpermute round trips, ~6 VALU instead of 6
// LDS-crossbar shuffles): 4 shifts inside each row of 16 lanes, then the row totals are broadcast into the
// following rows (gfx9 row_bcast:15 / row_bcast:31).
__device__ inline int apr_wave_incl_scan(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);   // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);   // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);   // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);   // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, true);   // row_bcast:15 -> rows 1, 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, true);   // row_bcast:31 -> rows 2, 3
  return x;
}
#endif

// ---- packed voxel key: (b:10 | x:18 | y:18 | z:18), x/y/z biased by 2^17 ----
#define APR_KEY_EMPTY 0xFFFFFFFFFFFFFFFFull
#define APR_AXIS_BIAS (1 << 17)
#define APR_AXIS_RANGE (1 << 18)

// batch index 1023 is reserved: (1023, max, max, max) would pack to APR_KEY_EMPTY
__host__ __device__ static inline bool apr_key_in_range(int b, int x, int y, int z) {
  return (unsigned)b < 1023u && (unsigned)(x + APR_AXIS_BIAS) < (unsigned)APR_AXIS_RANGE &&
         (unsigned)(y + APR_AXIS_BIAS) < (unsigned)APR_AXIS_RANGE &&
         (unsigned)(z + APR_AXIS_BIAS) < (unsigned)APR_AXIS_RANGE;
}

__host__ __device__ static inline uint64_t apr_pack_key(int b, int x, int y, int z) {
  return ((((uint64_t)(unsigned)b << 18 | (uint64_t)(unsigned)(x + APR_AXIS_BIAS)) << 18 |
           (uint64_t)(unsigned)(y + APR_AXIS_BIAS))
          << 18) |
         (uint64_t)(unsigned)(z + APR_AXIS_BIAS);
}

__host__ __device__ static inline uint32_t apr_hash_u64(uint64_t k) {
  // murmur3 finaliser
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return (uint32_t)k;
}

// floor division / floor-to-multiple for possibly negative ints, m > 0
__host__ __device__ static inline int apr_floor_to(int v, int m) {
  int q = v / m;
  if ((v % m) != 0 && v < 0) --q;
  return q * m;
}

// ---- internal helpers shared between translation units (not exported) ----
struct AprSearchGrid {
  const unsigned long long* keys;
  const int* vals;
  uint32_t mask;
  const int* start;   // [ncell + 1]
  const int* sorted;  // point indices bucketed by cell
  const float* mins;  // [3] cloud minimum = grid origin
  float cell;
  const int4* cell_coords;   // [n_cells] (cloud, cx, cy, cz) of every occupied cell, by cell id
  const int* n_cells;        // device-side count of occupied cells
};
size_t apr_internal_grid_bytes(int64_t n);
int apr_internal_search_grid(const float* pts, int64_t n, float cell, void* scratch, AprSearchGrid* out,
                             hipStream_t st);
int apr_internal_search_grid_batch(const float* pts, int64_t n, const int32_t* lengths_host, int32_t nb, float cell, void* scratch,
                                   AprSearchGrid* out, hipStream_t st);

__device__ static inline int apr_table_lookup(const unsigned long long* __restrict__ keys,
                                              const int* __restrict__ vals, uint32_t mask,
                                              unsigned long long key) {
  uint32_t slot = apr_hash_u64(key) & mask;
  for (uint32_t probe = 0; probe <= mask; ++probe) {
    unsigned long long k = keys[slot];
    if (k == key) return vals[slot];
    if (k == APR_KEY_EMPTY) return -1;
    slot = (slot + 1) & mask;
  }
  return -1;
}
