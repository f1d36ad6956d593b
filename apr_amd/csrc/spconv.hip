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
#include <stdlib.h>

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


// ---------------------------------------------------------------------------------------
// v2: pair-compacted, wave-autonomous kernel (the default).
//
// A workgroup still owns TM output rows x CN output channels, but instead of multiplying
// zero-padded [TM x Cin] blocks for every offset it first compacts, per offset k, the
// (input row, output row) pairs that exist (wave64 ballot + popcount into LDS lists), then
//   * work item = (offset k, Cin chunk); items are dealt statically to the 4 waves;
//   * a wave loads the item's weight piece W[k][chunk][CN] ONCE into registers (CK*CN*4 B,
//     coalesced 16-B loads of the pre-packed layout) and walks that offset's pairs in groups
//     of 16: the A fragments are gathered straight from global memory into the MFMA register
//     layout (lane (r,q) loads in[idx_r][16j+4q..+3]; no LDS staging, no barrier),
//     CK/4 x CN/16 v_mfma_f32_16x16x4_f32, and the 16 x CN result is added (ds_add_f32) into
//     the wave's PRIVATE accumulator tile in LDS at the pairs' output rows;
//   * one barrier, then the epilogue sums the 4 private tiles in fixed order, applies
//     scale/shift/residual/ReLU and stores whole rows with 16-B stores.
// MFMA work is padded only to 16 pairs per (tile, offset) (~1.3x) instead of ~2.2-2.7x for
// the dense-tile form, there is no barrier in the main loop, and the summation order is
// fixed (static item deal + in-order LDS adds per wave) => bitwise reproducible.
// ---------------------------------------------------------------------------------------
constexpr int kKMax = 27;


template <int TM, int CN, int CK, int NW>
__global__ __launch_bounds__(64 * NW, (NW == 8 ? 4 : 2)) void k_spconv_pairs(
    const float* __restrict__ in, int64_t ldi, const int* __restrict__ nbr, int n_out, int K,
    int cin, int cout, const float* __restrict__ wp, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ residual, int64_t ldr, int relu,
    float* __restrict__ out, int64_t ldo, int l2norm) {
  constexpr int CB = CN / 16;   // 16-col blocks per wave item
  constexpr int NJ = CK / 16;   // 16-wide k groups per chunk
  constexpr int LDO = CN + 4;   // accumulator row stride (floats), 16-B aligned rows
  constexpr int NT = 64 * NW;   // threads per workgroup (NW waves, each with a private accumulator tile)
  static_assert(TM * kKMax <= NW * TM * LDO, "nbr staging must fit in the accumulator region");

  __shared__ __attribute__((aligned(16))) float s_acc[NW * TM * LDO];
  __shared__ int s_in[kKMax * TM];
  __shared__ __attribute__((aligned(4))) unsigned char s_row[kKMax * TM];
  __shared__ int s_cnt[32];
  __shared__ int s_act[32];
  __shared__ int s_nact;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, q = lane >> 4;

  const int nb = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, loc = bid >> 3;
  const int lin = xcd * (nb >> 3) + min(xcd, nb & 7) + loc;
  const int ncol = cout / CN;
  const int tile_m = lin / ncol, tile_n = lin - tile_m * ncol;
  const int row0 = tile_m * TM;
  const int col0 = tile_n * CN;

  // 1. stage this tile's slice of the neighbour table (coalesced) in the accumulator region
  int* s_stage = reinterpret_cast<int*>(s_acc);
  for (int t = tid; t < TM * K; t += NT) {
    int r = t / K;
    int row = row0 + r;
    int v = -1;
    if (row < n_out) v = nbr ? nbr[(int64_t)row0 * K + t] : row;
    s_stage[t] = v;
  }
  __syncthreads();
  // 2. per-offset compaction of (input row, local output row) pairs: ballot + popcount
  for (int k = wave; k < K; k += NW) {
    int cnt = 0;
#pragma unroll
    for (int base = 0; base < TM; base += 64) {
      int r = base + lane;
      int v = (r < TM) ? s_stage[r * K + k] : -1;
      unsigned long long m = __ballot(v >= 0);
      if (v >= 0) {
        int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
        s_in[k * TM + pos] = v;
        s_row[k * TM + pos] = (unsigned char)r;
      }
      cnt += __popcll(m);
    }
    if (lane == 0) s_cnt[k] = cnt;
  }
  __syncthreads();
  // 3. active offset list (wave 0) + zero the accumulators (everyone)
  if (wave == 0) {
    int c = (lane < K) ? s_cnt[lane] : 0;
    unsigned long long m = __ballot(c > 0);
    if (c > 0) s_act[__popcll(m & ((1ull << lane) - 1ull))] = lane;
    if (lane == 0) s_nact = __popcll(m);
  }
  for (int t = tid; t < NW * TM * LDO / 4; t += NT)
    reinterpret_cast<f32x4*>(s_acc)[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  const int nact = s_nact;
  const int nchunk = cin / CK;
  const int cinG = cin >> 2;
  float* my_acc = s_acc + wave * TM * LDO;

  // 4. wave-autonomous main loop: no barriers.  A step = (item, 16-pair group).  Every step issues
  //    exactly CB*NJ weight-fragment loads + NJ gather loads for the NEXT step into the other
  //    register buffer before running its own MFMAs (the loop is unrolled by two so the buffers
  //    swap by name, and the fixed load count lets the compiler wait with a counted vmcnt instead
  //    of vmcnt(0)); a multi-group item re-reads its weight piece (L1 hit) to keep the count fixed.
  const int nitems = nact * nchunk;
  struct Desc { int item, g0, k, chunk, cnt; };
  auto decode = [&](Desc& d) {
    const int ai = d.item / nchunk;
    d.chunk = d.item - ai * nchunk;
    d.k = __builtin_amdgcn_readfirstlane(s_act[ai]);
    d.cnt = __builtin_amdgcn_readfirstlane(s_cnt[d.k]);
  };
  // returns false when there is no further step (d is left unchanged so dummy loads stay valid)
  auto advance = [&](Desc& d) -> bool {
    if (d.g0 + 16 < d.cnt) {
      d.g0 += 16;
      return true;
    }
    if (d.item + NW >= nitems) return false;
    d.item += NW;
    d.g0 = 0;
    decode(d);
    return true;
  };
  // hand strength-reduced addressing: one 64-bit lane base per step, scalar strides, immediate offsets
  // (letting the compiler expand the full index expression cost ~150 VALU incl. quarter-rate v_mul_lo
  // per step and made the loop issue-bound)
  const int lane_w_off = (q * cout + r16) * 4;           // floats, lane-constant
  const int64_t wstride = (int64_t)16 * cout;            // floats between consecutive j (4 cin groups)
  auto load = [&](const Desc& d, f32x4 (&bw)[CB][NJ], f32x4 (&aw)[NJ]) {
    const int64_t wsc = ((int64_t)(d.k * cinG + d.chunk * (CK / 4)) * cout + col0) * 4;   // scalar part
    const float* wb = wp + wsc + lane_w_off;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const float* wj = wb + j * wstride;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        bw[cb][j] = *reinterpret_cast<const f32x4*>(wj + cb * 64);
      }
    }
    // padded pairs (p >= cnt) gather pair 0's row: their columns of D^T are never written back
    const int p = d.g0 + r16;
    const int idx = s_in[d.k * TM + (p < d.cnt ? p : 0)];
    const float* ab = in + (int64_t)idx * ldi + (d.chunk * CK + q * 4);
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      aw[j] = *reinterpret_cast<const f32x4*>(ab + j * 16);
  };
  auto compute = [&](const Desc& d, const f32x4 (&bw)[CB][NJ], const f32x4 (&aw)[NJ]) {
    // D^T = W^T . A^T : lane (r16 = pair, q) ends up with 4 consecutive output channels of one pair
    f32x4 acc[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
          acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[cb][j][t], aw[j][t], acc[cb], 0, 0, 0);
    // read-modify-write into the wave-private tile (rows of one group are distinct: no atomics)
    if (d.g0 + r16 < d.cnt) {
      float* dst = my_acc + (int)s_row[d.k * TM + d.g0 + r16] * LDO + q * 4;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        f32x4* d4 = reinterpret_cast<f32x4*>(dst + cb * 16);
        *d4 = *d4 + acc[cb];
      }
    }
  };

  if (wave < nitems) {
    f32x4 b0[CB][NJ], a0[NJ], b1[CB][NJ], a1[NJ];
    Desc d0;
    d0.item = wave;
    d0.g0 = 0;
    decode(d0);
    load(d0, b0, a0);
    while (true) {
      Desc d1 = d0;
      const bool more1 = advance(d1);
      load(d1, b1, a1);          // prefetch (a harmless re-load of d0 when there is no next step)
      compute(d0, b0, a0);
      if (!more1) break;
      d0 = d1;
      const bool more0 = advance(d0);
      load(d0, b0, a0);
      compute(d1, b1, a1);
      if (!more0) break;
    }
  }
  __syncthreads();

  // 5. epilogue: fixed-order sum of the 4 private tiles, fused affine/residual/ReLU, 16-B row stores
  for (int e = tid; e < TM * (CN / 4); e += NT) {
    const int r = e / (CN / 4), c4 = e - r * (CN / 4);
    const int row = row0 + r;
    if (row >= n_out) continue;
    const int col = col0 + c4 * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(&s_acc[(0 * TM + r) * LDO + c4 * 4]);
#pragma unroll
    for (int w = 1; w < NW; ++w) v += *reinterpret_cast<const f32x4*>(&s_acc[(w * TM + r) * LDO + c4 * 4]);
    if (scale) v *= *reinterpret_cast<const f32x4*>(scale + col);
    if (shift) v += *reinterpret_cast<const f32x4*>(shift + col);
    if (residual) v += *reinterpret_cast<const f32x4*>(residual + (int64_t)row * ldr + col);
    if (relu) {
      v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
    }
    if (l2norm) {
      // row / |row|_2 (the encoder's normalize_feature, FCGF_APR/model/resunet.py:139-142) fused behind the last layer:
      // the host launches this only with cout == CN, so the row's CN channels sit in the CN / 4 consecutive lanes of
      // this pass.  The sum of squares follows k_l2_normalize's butterfly (channel distance CN/2 ... 4 across lanes,
      // then 2 and 1 inside the lane) and the division is the same: the same bits as the separate kernel.
      f32x4 sq;      // rounded squares (k_l2_normalize's fmaf(v, v, 0)): must not contract into the adds below
      sq[0] = fmaf(v[0], v[0], 0.f); sq[1] = fmaf(v[1], v[1], 0.f);
      sq[2] = fmaf(v[2], v[2], 0.f); sq[3] = fmaf(v[3], v[3], 0.f);
#pragma unroll
      for (int dl = CN / 8; dl >= 1; dl >>= 1) {
        sq[0] += __shfl_xor(sq[0], dl); sq[1] += __shfl_xor(sq[1], dl);
        sq[2] += __shfl_xor(sq[2], dl); sq[3] += __shfl_xor(sq[3], dl);
      }
      const float nrm = sqrtf((sq[0] + sq[2]) + (sq[1] + sq[3]));
      v[0] = v[0] / nrm; v[1] = v[1] / nrm; v[2] = v[2] / nrm; v[3] = v[3] / nrm;
    }
    *reinterpret_cast<f32x4*>(out + (int64_t)row * ldo + col) = v;
  }
}

// Small-Cin path (conv1: Cin = 1 or 3, K = 125 / 343): 0.5 FLOP/B, pure gather/stream work.
// A workgroup stages its 64 rows' slice of the neighbour table in LDS with coalesced loads,
// then 4 threads per row walk the offsets (one broadcast LDS read each) and keep 8 output
// channels in registers; weights ([K,cin,cout], <= 32 KB) are read through L1.
template <int CIN>
__global__ __launch_bounds__(256) void k_spconv_smallcin(
    const float* __restrict__ in, int64_t ldi, const int* __restrict__ nbr, int n_out, int K, int cout,
    const float* __restrict__ w, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ residual, int64_t ldr, int relu, float* __restrict__ out, int64_t ldo) {
  extern __shared__ int s_nb[];   // [64][K]
  const int tid = threadIdx.x;
  const int row0 = blockIdx.x * 64;
  const int rows = min(64, n_out - row0);
  for (int t = tid; t < rows * K; t += 256) s_nb[t] = nbr[(int64_t)row0 * K + t];
  __syncthreads();
  const int r = tid >> 2, cg = tid & 3;
  if (r >= rows) return;
  for (int c0 = cg * 8; c0 < cout; c0 += 32) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; ++k) {
      const int idx = s_nb[r * K + k];
      if (idx < 0) continue;
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) {
        const float x = in[(int64_t)idx * ldi + ci];
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + ((int64_t)k * CIN + ci) * cout + c0);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(w + ((int64_t)k * CIN + ci) * cout + c0 + 4);
        acc[0] = fmaf(x, w0[0], acc[0]); acc[1] = fmaf(x, w0[1], acc[1]);
        acc[2] = fmaf(x, w0[2], acc[2]); acc[3] = fmaf(x, w0[3], acc[3]);
        acc[4] = fmaf(x, w1[0], acc[4]); acc[5] = fmaf(x, w1[1], acc[5]);
        acc[6] = fmaf(x, w1[2], acc[6]); acc[7] = fmaf(x, w1[3], acc[7]);
      }
    }
    const int64_t row = row0 + r;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = c0 + i;
      float v = acc[i] * (scale ? scale[c] : 1.f) + (shift ? shift[c] : 0.f);
      if (residual) v += residual[row * ldr + c];
      if (relu) v = fmaxf(v, 0.f);
      out[row * ldo + c] = v;
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

// wt[k'][co][ci] = w[k][ci][co], k' = K-1-k when flip: the kernel of a convolution's input gradient (the same operator
// over the reverse map: mirrored offsets on a same-level map, transposed channel matrix)
__global__ void k_flip_transpose(const float* __restrict__ w, int K, int cin, int cout, int flip, float* __restrict__ wt) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = (int64_t)K * cin * cout;
  if (t >= total) return;
  int ci = (int)(t % cin);
  int co = (int)((t / cin) % cout);
  int k = (int)(t / ((int64_t)cout * cin));
  wt[t] = w[((int64_t)(flip ? K - 1 - k : k) * cin + ci) * cout + co];
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

template <int TM, int CN, int CK, int NW>
int launch_pairs(const float* in, int64_t ldi, const int* nbr, int64_t n_out, int K, int cin,
                 int cout, const float* wp, const float* scale, const float* shift,
                 const float* residual, int64_t ldr, int relu, float* out, int64_t ldo,
                 hipStream_t st, int l2norm = 0) {
  int64_t tiles = cdiv64(n_out, TM) * (cout / CN);
  if (l2norm && cout != CN) {
    apr_set_error("spconv: fused row normalisation needs the whole row in one tile (cout %d, tile %d)", cout, CN);
    return APR_EINVAL;
  }
  hipLaunchKernelGGL((k_spconv_pairs<TM, CN, CK, NW>), dim3((unsigned)tiles), dim3(64 * NW), 0, st, in, ldi,
                     nbr, (int)n_out, K, cin, cout, wp, scale, shift, residual, ldr, relu, out, ldo, l2norm);
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

APR_API int apr_weights_flip_transpose(const float* w, int32_t K, int32_t cin, int32_t cout, int32_t flip, float* wt,
                                       void* stream) {
  APR_CHECK_ARG(w && wt && K > 0 && cin > 0 && cout > 0, "apr_weights_flip_transpose: bad arguments");
  const int64_t total = (int64_t)K * cin * cout;
  hipLaunchKernelGGL(k_flip_transpose, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, w, K, cin, cout,
                     flip, wt);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

// apr_spconv_fwd + optional row normalisation of the result (out[j] /= |out[j]|_2): fused into the tile kernel's
// epilogue when that kernel runs the layer with the whole row in one tile (cout 32 or 64), else a second launch.
static int spconv_fwd_impl(const float* in, int64_t ldi, const int32_t* nbr, int64_t n_out, int32_t K, int32_t cin,
                           int32_t cout, const float* w_packed, const float* scale, const float* shift,
                           const float* residual, int64_t ldr, int32_t relu, float* out, int64_t ldo, void* stream,
                           int l2norm);

APR_API int apr_spconv_fwd(const float* in, int64_t ldi, const int32_t* nbr, int64_t n_out, int32_t K,
                           int32_t cin, int32_t cout, const float* w_packed, const float* scale,
                           const float* shift, const float* residual, int64_t ldr, int32_t relu,
                           float* out, int64_t ldo, void* stream) {
  return spconv_fwd_impl(in, ldi, nbr, n_out, K, cin, cout, w_packed, scale, shift, residual, ldr, relu, out, ldo, stream, 0);
}

static int spconv_fwd_plain(const float* in, int64_t ldi, const int32_t* nbr, int64_t n_out, int32_t K, int32_t cin,
                            int32_t cout, const float* w_packed, const float* scale, const float* shift,
                            const float* residual, int64_t ldr, int32_t relu, float* out, int64_t ldo, void* stream,
                            int l2f, bool* fused);

static int spconv_fwd_impl(const float* in, int64_t ldi, const int32_t* nbr, int64_t n_out, int32_t K, int32_t cin,
                           int32_t cout, const float* w_packed, const float* scale, const float* shift,
                           const float* residual, int64_t ldr, int32_t relu, float* out, int64_t ldo, void* stream,
                           int l2norm) {
  bool fused = false;
  int rc = spconv_fwd_plain(in, ldi, nbr, n_out, K, cin, cout, w_packed, scale, shift, residual, ldr, relu, out, ldo,
                            stream, l2norm, &fused);
  if (rc != APR_OK || !l2norm || fused || n_out == 0) return rc;
  return apr_l2_normalize(out, ldo, n_out, cout, out, ldo, stream);
}

static int spconv_fwd_plain(const float* in, int64_t ldi, const int32_t* nbr, int64_t n_out, int32_t K, int32_t cin,
                            int32_t cout, const float* w_packed, const float* scale, const float* shift,
                            const float* residual, int64_t ldr, int32_t relu, float* out, int64_t ldo, void* stream,
                            int l2f, bool* fused) {
  hipStream_t st = (hipStream_t)stream;
  APR_CHECK_ARG(n_out >= 0 && n_out < (1ll << 31), "apr_spconv_fwd: n_out=%lld", (long long)n_out);
  APR_CHECK_ARG(K >= 1 && cin >= 1 && cout >= 1, "apr_spconv_fwd: bad K/cin/cout");
  APR_CHECK_ARG(nbr != nullptr || K == 1, "apr_spconv_fwd: identity map needs K == 1");
  APR_CHECK_ARG(ldi >= cin && ldo >= cout, "apr_spconv_fwd: leading dimension smaller than channels");
  APR_CHECK_ARG(!residual || ldr >= cout, "apr_spconv_fwd: ldr < cout");
  if (n_out == 0) return APR_OK;
  static const int s_impl = env_int("APR_SPCONV_IMPL", 2);      // 1 = dense-tile kernel, 2 = pair-compacted
  static const int s_tm = env_int("APR_SPCONV_TM", 0);           // 0 = heuristic
  const bool vec_ok = (ldi % 4) == 0 && (((uintptr_t)in) & 15) == 0 && (ldo % 4) == 0 &&
                      (((uintptr_t)out) & 15) == 0 &&
                      (!residual || ((ldr % 4) == 0 && (((uintptr_t)residual) & 15) == 0));
  // identity map (dense layer: K = 1 convolutions, Predator's Linear layers and KPConv's second step): dense.hip
  if (nbr == nullptr && K == 1 && vec_ok && apr_internal_dense_ok(n_out, cin, cout) &&
      ((((uintptr_t)scale) | ((uintptr_t)shift) | ((uintptr_t)w_packed)) & 15) == 0)
    return apr_internal_dense_gemm(in, ldi, n_out, cin, cout, w_packed, scale, shift, residual, ldr, relu, out, ldo, st);
  if (s_impl == 2 && use_mfma(K, cin, cout) && K <= kKMax && vec_ok) {
    // the row normalisation rides in this kernel's epilogue when a tile holds the whole row (CN == cout)
    const int l2k = (l2f && (cout == 32 || cout == 64)) ? 1 : 0;
    static const int s_nw = env_int("APR_SPCONV_NW", 4);
    if (l2k && !(s_nw == 8 && cout % 64 == 0) && !(cout == 64 && env_int("APR_SPCONV_CN", 0) == 32)) *fused = true;
#define APR_PAIRS(TM_, CN_, CK_)                                                                      \
  return launch_pairs<TM_, CN_, CK_, 4>(in, ldi, nbr, n_out, K, cin, cout, w_packed, scale, shift,    \
                                        residual, ldr, relu, out, ldo, st, *fused ? 1 : 0)
    if (s_nw == 8 && cout % 64 == 0)   // 8-wave tiles: 32-row tile, 32-channel pieces (<= 128 VGPRs, 4 waves/SIMD)
      return launch_pairs<32, 64, 32, 8>(in, ldi, nbr, n_out, K, cin, cout, w_packed, scale, shift, residual, ldr,
                                         relu, out, ldo, st);
    static const int s_ck = env_int("APR_SPCONV_CK", 0);
    static const int s_cn = env_int("APR_SPCONV_CN", 0);
    const bool cn64 = (cout % 64 == 0) && s_cn != 32;
    const bool ck64 = (cin % 64 == 0) && s_ck != 32;
    // 64-row tiles halve the weight re-reads but need >= ~400 workgroups to fill 256 CUs; below that 32-row
    // tiles win.  16-row tiles / 32-channel slices (APR_SPCONV_TM=16, APR_SPCONV_CN=32) measured slower on
    // every layer of the encoder: the deep layers are bound by the weight stream, not by parallelism — they
    // go through the weight-stationary path (spconv_ws.hip) instead.
    int tm = s_tm;
    if (!tm) tm = cdiv64(n_out, 64) * (cout / (cn64 ? 64 : 32)) >= 400 ? 64 : 32;
    if (tm == 64) {
      if (cn64 && ck64) APR_PAIRS(64, 64, 64);
      if (cn64) APR_PAIRS(64, 64, 32);
      if (ck64) APR_PAIRS(64, 32, 64);
      APR_PAIRS(64, 32, 32);
    }
    if (tm == 16) {
      if (cn64 && ck64) APR_PAIRS(16, 64, 64);
      if (cn64) APR_PAIRS(16, 64, 32);
      if (ck64) APR_PAIRS(16, 32, 64);
      APR_PAIRS(16, 32, 32);
    }
    if (cn64 && ck64) APR_PAIRS(32, 64, 64);
    if (cn64) APR_PAIRS(32, 64, 32);
    if (ck64) APR_PAIRS(32, 32, 64);
    APR_PAIRS(32, 32, 32);
#undef APR_PAIRS
  }
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
  if (nbr && (cin == 1 || cin == 3) && cout % 32 == 0 && (size_t)64 * K * 4 <= 64 * 1024 &&
      (((uintptr_t)w_packed) & 15) == 0) {
    const unsigned grid = (unsigned)cdiv64(n_out, 64);
    const size_t lds = (size_t)64 * K * 4;
    if (cin == 1)
      hipLaunchKernelGGL(k_spconv_smallcin<1>, dim3(grid), dim3(256), lds, st, in, ldi, nbr, (int)n_out, K, cout,
                         w_packed, scale, shift, residual, ldr, relu, out, ldo);
    else
      hipLaunchKernelGGL(k_spconv_smallcin<3>, dim3(grid), dim3(256), lds, st, in, ldi, nbr, (int)n_out, K, cout,
                         w_packed, scale, shift, residual, ldr, relu, out, ldo);
    APR_LAUNCH_CHECK();
    return APR_OK;
  }
  int64_t total = n_out * cout;
  hipLaunchKernelGGL(k_spconv_generic, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, in, ldi, nbr,
                     n_out, K, cin, cout, w_packed, scale, shift, residual, ldr, relu, out, ldo);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

static int spconv_batch_one(const apr_spconv_desc& d, void* stream, hipEvent_t e0, hipEvent_t e1) {
  hipStream_t st = (hipStream_t)stream;
  bool dense_bf3 = false, rows_bf3 = false;
  if (d.os_pairs && d.w_bf3) {   // output-stationary path (spconv_os.hip): tile lists built on first use of the map
    if (d.os_build_bytes > 0) {
      int rcb = apr_spconv_os_pairs_build(d.nbr, d.n_out, d.os_n_in, d.K, (int32_t)d.os_rows, d.os_pairs,
                                          (size_t)d.os_build_bytes, stream);
      if (rcb != APR_OK) return rcb;
    }
    if (e0) APR_HIP(hipEventRecord(e0, st));
    int rco = apr_spconv_os_fwd(d.in, d.ldi, d.os_pairs, d.n_out, d.K, (int32_t)d.os_rows, d.cin, d.cout, d.w_bf3, d.scale,
                                d.shift, d.residual, d.ldr, d.relu, d.out, d.ldo, stream);
    if (rco != APR_OK) return rco;
  } else if (d.plist && d.ws3 && d.w_bf3) {   // triple pair lists: half the product rows (spconv_ws.hip)
    if (d.plist_bytes > 0) {
      int rcb = apr_pairlist3_build(d.nbr, d.n_out, d.K, d.counters, d.plist, (size_t)d.plist_bytes, stream);
      if (rcb != APR_OK) return rcb;
    }
    if (e0) APR_HIP(hipEventRecord(e0, st));
    int rcw = apr_spconv_ws3_fwd_bf3(d.in, d.ldi, d.counters, d.plist, d.n_out, d.cin, d.cout, d.w_bf3, d.scale, d.shift,
                                     d.residual, d.ldr, d.relu, d.out, d.ldo, d.prod_scratch, stream);
    if (rcw != APR_OK) return rcw;
  } else if (d.plist) {
    if (d.plist_bytes > 0) {   // the pair-list build is per map, not part of the timed conv layer
      int rcb = apr_pairlist_build(d.nbr, d.n_out, d.K, d.counters, d.plist, (size_t)d.plist_bytes, stream);
      if (rcb != APR_OK) return rcb;
    }
    if (e0) APR_HIP(hipEventRecord(e0, st));
    int rcw = apr_spconv_ws_fwd_bf3(d.in, d.ldi, d.counters, d.plist, d.n_out, d.K, d.cin, d.cout, d.w_packed, d.w_bf3,
                                    d.scale, d.shift, d.residual, d.ldr, d.relu, d.out, d.ldo, d.prod_scratch, stream);
    if (rcw != APR_OK) return rcw;
  } else if (!d.nbr && d.K == 1 && d.w_bf3 && apr_internal_dense_rows_route(d.n_out, d.cin, d.cout)) {
    // K = 1 layer with few input channels over many rows (conv1_tr / final on the finest level, resunet.py:126-140): the
    // weight-resident row stream (dense_rows.hip), the row normalisation in its epilogue
    rows_bf3 = true;
    if (e0) APR_HIP(hipEventRecord(e0, st));
    int rcd = apr_dense_rows_bf3(d.in, d.ldi, d.n_out, d.cin, d.cout, d.w_bf3, d.scale, d.shift, d.residual, d.ldr, d.relu,
                                 d.l2norm, d.out, d.ldo, stream);
    if (rcd != APR_OK) return rcd;
  } else if (!d.nbr && d.K == 1 && d.w_bf3 && d.cin % 64 == 0 && d.cout % 64 == 0 && d.n_out > 0) {
    // K = 1 layer with 64-multiple widths and split weights (the wide variants' conv1_tr / final, resunet.py:224-251): the
    // dense GEMM on the bf16 split (dense.hip), 1.5-1.7x the exact-fp32 MFMA kernel
    dense_bf3 = true;
    if (e0) APR_HIP(hipEventRecord(e0, st));
    int rcd = apr_dense_gemm_bf3(d.in, d.ldi, d.n_out, d.cin, d.cout, d.w_bf3, d.scale, d.shift, d.residual, d.ldr, d.relu,
                                 d.out, d.ldo, stream);
    if (rcd != APR_OK) return rcd;
  } else {
    if (e0) APR_HIP(hipEventRecord(e0, st));
    int rc = spconv_fwd_impl(d.in, d.ldi, d.nbr, d.n_out, d.K, d.cin, d.cout, d.w_packed, d.scale, d.shift, d.residual,
                             d.ldr, d.relu, d.out, d.ldo, stream, d.l2norm);
    if (rc != APR_OK) return rc;
  }
  if (d.l2norm && !rows_bf3 && (d.plist || (d.os_pairs && d.w_bf3) || dense_bf3) && d.n_out > 0) {   // other conv families: a second launch
    int rcn = apr_l2_normalize(d.out, d.ldo, d.n_out, d.cout, d.out, d.ldo, stream);
    if (rcn != APR_OK) return rcn;
  }
  if (e1) APR_HIP(hipEventRecord(e1, st));
  return APR_OK;
}

APR_API int apr_spconv_fwd_batch(const apr_spconv_desc* d, int32_t n, void* stream) {
  APR_CHECK_ARG(n >= 0 && (d != nullptr || n == 0), "apr_spconv_fwd_batch: bad arguments");
  for (int i = 0; i < n; ++i) {
    int rc = spconv_batch_one(d[i], stream, nullptr, nullptr);
    if (rc != APR_OK) return rc;
  }
  return APR_OK;
}

APR_API int apr_spconv_fwd_batch_timed(const apr_spconv_desc* d, int32_t n, float* layer_ms, void* stream) {
  APR_CHECK_ARG(n >= 0 && n <= 4096 && (d != nullptr || n == 0) && (layer_ms != nullptr || n == 0),
                "apr_spconv_fwd_batch_timed: bad arguments");
  hipEvent_t* ev = (hipEvent_t*)malloc(sizeof(hipEvent_t) * 2 * (size_t)(n > 0 ? n : 1));
  int made = 0, rc = APR_OK;
  for (; made < 2 * n; ++made)
    if (hipEventCreate(&ev[made]) != hipSuccess) {
      apr_set_error("apr_spconv_fwd_batch_timed: hipEventCreate failed");
      rc = APR_EHIP;
      break;
    }
  for (int i = 0; rc == APR_OK && i < n; ++i) rc = spconv_batch_one(d[i], stream, ev[2 * i], ev[2 * i + 1]);
  if (rc == APR_OK && hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
    apr_set_error("apr_spconv_fwd_batch_timed: stream synchronize failed");
    rc = APR_EHIP;
  }
  for (int i = 0; rc == APR_OK && i < n; ++i)
    if (hipEventElapsedTime(&layer_ms[i], ev[2 * i], ev[2 * i + 1]) != hipSuccess) layer_ms[i] = -1.f;
  for (int i = 0; i < made; ++i) (void)hipEventDestroy(ev[i]);
  free(ev);
  return rc;
}
