// bf16 3-way split of fp32 operands for v_mfma_f32_16x16x32_bf16 with fp32-equivalent accuracy (shared by the
// weight-stationary sparse conv, spconv_ws.hip, and the dense GEMM, dense.hip).
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x4_b3 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

// x = h + m + l EXACTLY, each piece the top 8 significant bits of what is left (truncation: an fp32 has 24 significant
// bits, three 8-bit pieces hold them all; the subtractions are exact).  Pairs of pieces are packed with one byte
// permute: 2 x (and + sub) per element + 3 permutes per pair = 5.5 VALU per element (the compiler's own
// float -> bf16 -> float round trips cost ~8.5: one v_cvt_pk per ELEMENT plus unpack and repack).
__device__ inline unsigned apr_pack_hi16(float x1, float x0) {      // [bf16(x0) | bf16(x1) << 16], truncating
  return __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
}

__device__ inline void apr_split3(const f32x4_b3& a, const f32x4_b3& b, bf16x8& h, bf16x8& m, bf16x8& l) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  u32x4 hp, mp, lp;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x0 = x[2 * i], x1 = x[2 * i + 1];
    hp[i] = apr_pack_hi16(x1, x0);
    const float r0 = x0 - __uint_as_float(__float_as_uint(x0) & 0xFFFF0000u);
    const float r1 = x1 - __uint_as_float(__float_as_uint(x1) & 0xFFFF0000u);
    mp[i] = apr_pack_hi16(r1, r0);
    const float t0 = r0 - __uint_as_float(__float_as_uint(r0) & 0xFFFF0000u);
    const float t1 = r1 - __uint_as_float(__float_as_uint(r1) & 0xFFFF0000u);
    lp[i] = apr_pack_hi16(t1, t0);
  }
  h = __builtin_bit_cast(bf16x8, hp);
  m = __builtin_bit_cast(bf16x8, mp);
  l = __builtin_bit_cast(bf16x8, lp);
}

