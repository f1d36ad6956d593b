// Dense [M, cin] x [cin, cout] contraction with the sparse conv's fused epilogue (SURVEY 8(a) rows P4 step 2, P5 and
// the K = 1 layers of F5): the unary / bottleneck Linear layers of KPFCNN (`Predator_APR/models/blocks.py:499-504,
// 653-681`), KPConv's second step `[N, 15*Cin] @ [15*Cin, Cout]` (`blocks.py:347-374`) and the K = 1
// MinkowskiConvolutions (`FCGF_APR/model/resunet.py:126-140`).  Reached through apr_spconv_fwd with an identity map
// (nbr == NULL, K == 1): same packed weights Wp[cin/4][cout][4], same out = act(acc * scale + shift + residual).
//
// The pair-compacted tile kernel treats these as a sparse conv with one offset: every 16-row group re-reads its
// 16 KB weight piece through the L1 (20 KB of vector-memory traffic per 64 MFMAs, ~60 % of a CU's L1 rate) and
// the partial sums take a trip through LDS.  Here a workgroup owns 64*G rows x 64 columns and walks cin in
// 64-channel chunks: the chunk's weight slice (16 KB) is staged ONCE per workgroup in LDS (double-buffered, the
// next chunk's global loads are issued before the current chunk's MFMAs), every wave keeps the accumulators of its
// G 16-row groups in registers for the whole K loop (B fragments from LDS are shared by the G groups), A rows
// stream straight from global memory into MFMA operand registers one chunk ahead, and the epilogue stores 16-B
// row pieces.  Exact fp32 (v_mfma_f32_16x16x4_f32, D^T form: a lane ends with 4 consecutive channels of one row).
#include "common.h"
#include "bf3.h"
#include <mutex>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int G>
__global__ __launch_bounds__(256) void k_dense_gemm(const float* __restrict__ in, int64_t ldi, int M, int cin,
                                                    int cout, const float* __restrict__ wp,
                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                    const float* __restrict__ residual, int64_t ldr, int relu,
                                                    float* __restrict__ out, int64_t ldo) {
  __shared__ __attribute__((aligned(16))) float s_w[2][16 * 64 * 4];   // [buf][g = k/4][col 64][4]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, q = lane >> 4;
  const int ncol = cout >> 6;
  const int tile_m = blockIdx.x / ncol, tile_n = blockIdx.x - tile_m * ncol;   // neighbours share the A rows in L2
  const int row0 = tile_m * (64 * G) + wave * (16 * G);
  const int col0 = tile_n * 64;
  const int nchunk = cin >> 6;

  // this lane's A row of each group (clamped: rows past M are computed and dropped)
  const float* arow[G];
#pragma unroll
  for (int gi = 0; gi < G; ++gi) {
    const int r = row0 + gi * 16 + r16;
    arow[gi] = in + (int64_t)(r < M ? r : M - 1) * ldi + q * 4;
  }
  // weight staging: thread t copies 4 x 16 B of the chunk: g = (t >> 6) + 4u, lane-contiguous 1 KB rows
  const float* wsrc = wp + ((int64_t)wave * cout + col0) * 4 + lane * 4;
  const int64_t wstep_g = (int64_t)4 * cout * 4;      // 4 g-rows further
  const int64_t wstep_chunk = (int64_t)16 * cout * 4;   // next 64-channel chunk
  f32x4 wreg[4];
  f32x4 abuf[2][G][4];
  f32x4 acc[G][4];
#pragma unroll
  for (int gi = 0; gi < G; ++gi)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[gi][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int u = 0; u < 4; ++u) wreg[u] = *reinterpret_cast<const f32x4*>(wsrc + u * wstep_g);
#pragma unroll
  for (int gi = 0; gi < G; ++gi)
#pragma unroll
    for (int j = 0; j < 4; ++j) abuf[0][gi][j] = *reinterpret_cast<const f32x4*>(arow[gi] + j * 16);
#pragma unroll
  for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4*>(&s_w[0][((wave + 4 * u) * 64 + lane) * 4]) = wreg[u];
  __syncthreads();

#define APR_DENSE_CHUNK(cur, nxt, c)                                                                              \
  {                                                                                                               \
    const bool more = (c) + 1 < nchunk;                                                                           \
    if (more) {                                                                                                   \
      const float* ws = wsrc + ((c) + 1) * wstep_chunk;                                                           \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) wreg[u] = *reinterpret_cast<const f32x4*>(ws + u * wstep_g);  \
      _Pragma("unroll") for (int gi = 0; gi < G; ++gi)                                                            \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                             \
          abuf[nxt][gi][j] = *reinterpret_cast<const f32x4*>(arow[gi] + ((c) + 1) * 64 + j * 16);                \
    }                                                                                                             \
    const float* wb = &s_w[(c) & 1][0];                                                                           \
    f32x4 bq[2][4];                                                                                               \
    _Pragma("unroll") for (int cb = 0; cb < 4; ++cb)                                                              \
      bq[0][cb] = *reinterpret_cast<const f32x4*>(wb + (q * 64 + cb * 16 + r16) * 4);                             \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                               \
      if (j < 3) {                                                                                                \
        _Pragma("unroll") for (int cb = 0; cb < 4; ++cb)                                                          \
          bq[(j + 1) & 1][cb] = *reinterpret_cast<const f32x4*>(wb + (((j + 1) * 4 + q) * 64 + cb * 16 + r16) * 4); \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
      }                                                                                                           \
      _Pragma("unroll") for (int gi = 0; gi < G; ++gi)                                                            \
        _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                             \
          _Pragma("unroll") for (int cb = 0; cb < 4; ++cb)                                                        \
            acc[gi][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[j & 1][cb][t], abuf[cur][gi][j][t],             \
                                                               acc[gi][cb], 0, 0, 0);                             \
    }                                                                                                             \
    if (more) {                                                                                                   \
      _Pragma("unroll") for (int u = 0; u < 4; ++u)                                                               \
        *reinterpret_cast<f32x4*>(&s_w[((c) + 1) & 1][((wave + 4 * u) * 64 + lane) * 4]) = wreg[u];               \
    }                                                                                                             \
    __syncthreads();                                                                                              \
  }

  int c = 0;
  for (; c + 2 <= nchunk; c += 2) {
    APR_DENSE_CHUNK(0, 1, c)
    APR_DENSE_CHUNK(1, 0, c + 1)
  }
  if (c < nchunk) APR_DENSE_CHUNK(0, 1, c)
#undef APR_DENSE_CHUNK

  // epilogue: lane (r16 = row, q) holds channels col0 + cb*16 + 4q .. +3 of its row
#pragma unroll
  for (int gi = 0; gi < G; ++gi) {
    const int r = row0 + gi * 16 + r16;
    if (r >= M) continue;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const int col = col0 + cb * 16 + q * 4;
      f32x4 v = acc[gi][cb];
      if (scale) v *= *reinterpret_cast<const f32x4*>(scale + col);
      if (shift) v += *reinterpret_cast<const f32x4*>(shift + col);
      if (residual) v += *reinterpret_cast<const f32x4*>(residual + (int64_t)r * ldr + col);
      if (relu) {
        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
      }
      *reinterpret_cast<f32x4*>(out + (int64_t)r * ldo + col) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// k_dense_gemm_bf3: the same tile (64 G rows x 64 columns per workgroup, cin walked in 64-channel chunks), with the
// contraction on the bf16 MFMA in the 3-way split of bf3.h (fp32-equivalent: 6 of the 9 cross terms, small terms
// first, fp32 accumulate -- see spconv_ws.hip).  The exact-fp32 kernel above runs the deep KPConv contractions
// ([1382, 7680] x [7680, 512], [3770, 3840] x [3840, 256], [9918, 1920] x [1920, 128]: 46 of the ~90 GFLOP of one
// KPFCNN forward) at 36-44 % of the 157 TFLOP/s fp32-MFMA roof, and the roof itself is the limit: 48 bf16 MFMAs x 16
// cycles per 16 rows x 64 channels x 64 columns instead of 64 x 32.
// Weights: apr_spconv_pack_weights_bf3 with K = 1 ([column block][plane h/m/l][32-channel step][col 64][quad'][8]):
// a chunk is 3 x 8 KB contiguous pieces, staged by straight copies into a double-buffered 2 x 24 KB LDS image; the
// fragment reads are the conflict-free ds_read_b128 of the sparse kernel.  A rows stream from global memory one chunk
// ahead (lane (row r16, quad q): 8 consecutive channels per 32-channel step) and are split in registers per step.
// ---------------------------------------------------------------------------------------------------------------
// STATS (round 5): the rows come in SEGMENTS (the scan pairs stacked into one KPFCNN forward; one segment otherwise) and row
// tiles never straddle a segment: tile_m -> (segment, tile within it) by a scan of <= 32 entries; the epilogue also leaves the
// tile's per-column sum and sum of squares (fp64, rows in order) in partial[tile][2][cout] -- exactly what k_bn_partial would
// compute in a launch of its own for the InstanceNorm that follows every Linear of KPFCNN (blocks.py:459-468).
struct DenseSegs {
  int nseg;
  int row0[33];
  int tile0[33];
};

// ASTG (round 5): the rows reach the MFMA operands THROUGH LDS.  By ablation (16 KPFCNN shapes, 1427 us in all) the
// fragment-shaped row loads are the largest single cost of this kernel -- without them 972 us, without the split 1341, without
// the weight staging 1259, without the barrier 1289: a lane's two 16-B pieces per 32-channel step make every instruction touch
// 16 cache lines for 1 KB, every line twice.  Here a wave copies ITS 16 G rows of the chunk (256 B each) into its own LDS region
// by LDS-DMA in full lines (4 rows x 256 B per instruction, the 16-B pieces XOR-swizzled by row through the per-lane SOURCE
// address so that the fragment reads are conflict-free), reads its fragments back at the top of the chunk and then sends for
// the next chunk's rows into the same region: wave-private, no extra barrier, one register set.
template <int G, bool STATS, bool ASTG = false>
__global__ __launch_bounds__(256, 2) void k_dense_gemm_bf3(const float* __restrict__ in, int64_t ldi, int M, int cin,
                                                           int cout, const unsigned char* __restrict__ wp3,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ residual, int64_t ldr, int relu,
                                                           float* __restrict__ out, int64_t ldo, DenseSegs sg,
                                                           double* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) unsigned char s_w[2][3 * 8192];   // [buf][plane][step 2][col 64][quad 4][16 B]
  extern __shared__ __attribute__((aligned(16))) unsigned char s_a[];       // ASTG: [wave 4][row 16 G][piece 16 (swizzled)][16 B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, q = lane >> 4;
  const int ncol = cout >> 6;
  // Workgroups are dealt round-robin over the 8 XCDs, each with its own L2.  Consecutive LINEAR tile ids (the column
  // blocks of one row tile, then the next row tile) are therefore given to ONE XCD: the ncol workgroups that read the
  // same A rows share them through that L2 instead of fetching them ncol times from the Infinity Cache / HBM
  // (16 KPFCNN shapes: 1319 -> 1250 us in all; [14692, 3840] x [3840, 256] 198 -> 181 us).
  const int nb = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, loc = bid >> 3;
  const int lin = xcd * (nb >> 3) + min(xcd, nb & 7) + loc;
  const int tile_m = lin / ncol, tile_n = lin - tile_m * ncol;
  int tile_base = tile_m * (64 * G);      // first row of the workgroup's tile; rows >= row_end are not there
  int row_end = M;
  if (STATS) {
    int sgi = 0;
    while (sgi + 1 < sg.nseg && tile_m >= sg.tile0[sgi + 1]) ++sgi;
    tile_base = sg.row0[sgi] + (tile_m - sg.tile0[sgi]) * (64 * G);
    row_end = sg.row0[sgi + 1];
  }
  const int row0 = tile_base + wave * (16 * G);
  const int col0 = tile_n * 64;
  const int nchunk = cin >> 6;
  const int64_t plane_bytes = (int64_t)(cin >> 5) * 4096;
  const unsigned char* wsrc = wp3 + (int64_t)tile_n * 3 * plane_bytes + tid * 16;
  const int frag_off = (r16 * 4 + ((r16 & 8) ? (q ^ 3) : q)) * 16;

  const float* arow[G];
#pragma unroll
  for (int gi = 0; gi < G; ++gi) {
    const int r = row0 + gi * 16 + r16;
    arow[gi] = in + (int64_t)(r < row_end ? r : row_end - 1) * ldi + q * 8;
  }
  f32x4 abuf[2][G][4];
  f32x4 acc[G][4];
#pragma unroll
  for (int gi = 0; gi < G; ++gi)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[gi][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // chunk c of plane pl: 8 KB at pl * plane_bytes + c * 8192; thread t copies 16 B at t * 16 and 4096 + t * 16, by
  // LDS-DMA (no staging registers: wave-uniform LDS base + lane * 16, which is exactly the straight copy)
#define APR_DENSE3_STAGE(buf, c)                                                                                   \
  _Pragma("unroll") for (int u = 0; u < 6; ++u)                                                                    \
    __builtin_amdgcn_global_load_lds(                                                                              \
        (const __attribute__((address_space(1))) void*)(wsrc + (int64_t)(c) * 8192 + (u >> 1) * plane_bytes +      \
                                                         (u & 1) * 4096),                                          \
        (__attribute__((address_space(3))) void*)&s_w[buf][(u >> 1) * 8192 + (u & 1) * 4096 + wave * 1024], 16, 0, 0);
  // ASTG: instruction u of a wave brings rows 4 u .. 4 u + 3 of its 16 G rows; lane l -> row 4 u + l / 16, LDS slot l % 16,
  // which holds the row's 16-B piece (l % 16) ^ (row % 16)
  const float* asrc[ASTG ? 4 * G : 1];
  unsigned char* my_a = s_a + (ASTG ? wave * (G * 4096) : 0);
  if (ASTG) {
#pragma unroll
    for (int u = 0; u < 4 * G; ++u) {
      const int rloc = 4 * u + (lane >> 4);
      const int r = row0 + rloc;
      asrc[u] = in + (int64_t)(r < row_end ? r : row_end - 1) * ldi + (((lane & 15) ^ (rloc & 15)) << 2);
    }
  }
#define APR_DENSE3_ASTAGE(c)                                                                                       \
  _Pragma("unroll") for (int u = 0; u < 4 * G; ++u)                                                                \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[u] + (int64_t)(c) * 64),  \
                                     (__attribute__((address_space(3))) void*)(my_a + u * 1024), 16, 0, 0);
  // this lane's fragments of the staged chunk: piece (j >> 1) * 8 + 2 q + (j & 1) of row gi * 16 + r16, in slot piece ^ r16
#define APR_DENSE3_AREAD(buf)                                                                                      \
  _Pragma("unroll") for (int gi = 0; gi < G; ++gi)                                                                 \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                  \
      abuf[buf][gi][j] = *reinterpret_cast<const f32x4*>(my_a + (gi * 16 + r16) * 256 +                            \
                                                         ((((j >> 1) * 8 + 2 * q + (j & 1)) ^ r16) << 4));
  APR_DENSE3_STAGE(0, 0)
  if (ASTG) {
    APR_DENSE3_ASTAGE(0)
  } else {
#pragma unroll
    for (int gi = 0; gi < G; ++gi)
#pragma unroll
      for (int j = 0; j < 4; ++j) abuf[0][gi][j] = *reinterpret_cast<const f32x4*>(arow[gi] + (j >> 1) * 32 + (j & 1) * 4);
  }
  __syncthreads();

#define APR_DENSE3_CHUNK(cur_, nxt, c)                                                                             \
  {                                                                                                                \
    const bool more = (c) + 1 < nchunk;                                                                            \
    constexpr int cur = ASTG ? 0 : cur_;                                                                           \
    if (ASTG) {                                                                                                    \
      APR_DENSE3_AREAD(0)                                                                                          \
      if (more) { APR_DENSE3_STAGE(((c) + 1) & 1, (c) + 1) }                                                       \
      __builtin_amdgcn_s_waitcnt(0xc07f);   /* lgkmcnt(0): the fragments are in registers, the region is free */   \
      if (more) { APR_DENSE3_ASTAGE((c) + 1) }                                                                     \
    } else if (more) {                                                                                             \
      APR_DENSE3_STAGE(((c) + 1) & 1, (c) + 1)                                                                     \
      _Pragma("unroll") for (int gi = 0; gi < G; ++gi)                                                             \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                              \
          abuf[nxt][gi][j] =                                                                                       \
              *reinterpret_cast<const f32x4*>(arow[gi] + ((c) + 1) * 64 + (j >> 1) * 32 + (j & 1) * 4);            \
    }                                                                                                              \
    const unsigned char* wb = &s_w[(c) & 1][frag_off];                                                             \
    bf16x8 wf[3][3];                                                                                               \
    _Pragma("unroll") for (int i0 = 0; i0 < 2; ++i0)                                                               \
      _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                             \
        wf[i0][pl] = *reinterpret_cast<const bf16x8*>(wb + (i0 * 16) * 64 + pl * 8192);                            \
    bf16x8 ah[G], am[G], al[G];                                                                                    \
    _Pragma("unroll") for (int gi = 0; gi < G; ++gi)                                                               \
      apr_split3(abuf[cur][gi][0], abuf[cur][gi][1], ah[gi], am[gi], al[gi]);                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                \
      const int cb = i & 3;                                                                                        \
      if (i < 6) {                                                                                                 \
        const int s2 = (i + 2) >> 2, cb2 = (i + 2) & 3;                                                            \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                           \
          wf[(i + 2) % 3][pl] = *reinterpret_cast<const bf16x8*>(wb + (s2 * 64 + cb2 * 16) * 64 + pl * 8192);      \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
      }                                                                                                            \
      if (i == 4) {                                                                                                \
        _Pragma("unroll") for (int gi = 0; gi < G; ++gi)                                                           \
          apr_split3(abuf[cur][gi][2], abuf[cur][gi][3], ah[gi], am[gi], al[gi]);                                  \
      }                                                                                                            \
      const bf16x8 wh = wf[i % 3][0], wm = wf[i % 3][1], wl = wf[i % 3][2];                                        \
      _Pragma("unroll") for (int gi = 0; gi < G; ++gi) {                                                           \
        f32x4 t = acc[gi][cb];                                                                                     \
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, ah[gi], t, 0, 0, 0);                                       \
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, al[gi], t, 0, 0, 0);                                       \
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, am[gi], t, 0, 0, 0);                                       \
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, ah[gi], t, 0, 0, 0);                                       \
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, am[gi], t, 0, 0, 0);                                       \
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ah[gi], t, 0, 0, 0);                                       \
        acc[gi][cb] = t;                                                                                           \
      }                                                                                                            \
      __builtin_amdgcn_sched_barrier(0);                                                                           \
    }                                                                                                              \
    __syncthreads();   /* drains the LDS-DMA of the next chunk (vmcnt(0)) and frees this chunk's buffer */        \
  }

  int c = 0;
  for (; c + 2 <= nchunk; c += 2) {
    APR_DENSE3_CHUNK(0, 1, c)
    APR_DENSE3_CHUNK(1, 0, c + 1)
  }
  if (c < nchunk) APR_DENSE3_CHUNK(0, 1, c)
#undef APR_DENSE3_CHUNK
#undef APR_DENSE3_STAGE
#undef APR_DENSE3_ASTAGE
#undef APR_DENSE3_AREAD

  if (STATS) {
    // the tile's image [64 G rows][64 columns] fp32 through the weight buffers (free: the last chunk ended in a barrier),
    // absent rows as zeros; then thread = (column, quarter of the rows): fp64 sums in row order, quarters combined in order
    float* s_t = reinterpret_cast<float*>(&s_w[0][0]);
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
      const int rl = wave * (16 * G) + gi * 16 + r16;
      const bool live = row0 + gi * 16 + r16 < row_end;
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
        *reinterpret_cast<f32x4*>(s_t + rl * 64 + cb * 16 + q * 4) = live ? acc[gi][cb] : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    const int cc = tid & 63, part = tid >> 6;
    double a = 0.0, b = 0.0;
#pragma unroll 4
    for (int rr = 0; rr < 16 * G; ++rr) {
      const double d = (double)s_t[(part * (16 * G) + rr) * 64 + cc];
      a += d;
      b += d * d;
    }
    __syncthreads();
    double* s_d = reinterpret_cast<double*>(&s_w[1][0]);
    s_d[(part * 64 + cc) * 2] = a;
    s_d[(part * 64 + cc) * 2 + 1] = b;
    __syncthreads();
    if (tid < 128) {
      const int c2 = tid & 63, which = tid >> 6;
      const double t = ((s_d[(0 * 64 + c2) * 2 + which] + s_d[(1 * 64 + c2) * 2 + which]) + s_d[(2 * 64 + c2) * 2 + which]) +
                       s_d[(3 * 64 + c2) * 2 + which];
      partial[((int64_t)tile_m * 2 + which) * cout + col0 + c2] = t;
    }
  }
#pragma unroll
  for (int gi = 0; gi < G; ++gi) {
    const int r = row0 + gi * 16 + r16;
    if (r >= row_end) continue;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const int col = col0 + cb * 16 + q * 4;
      f32x4 v = acc[gi][cb];
      if (scale) v *= *reinterpret_cast<const f32x4*>(scale + col);
      if (shift) v += *reinterpret_cast<const f32x4*>(shift + col);
      if (residual) v += *reinterpret_cast<const f32x4*>(residual + (int64_t)r * ldr + col);
      if (relu) {
        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
      }
      *reinterpret_cast<f32x4*>(out + (int64_t)r * ldo + col) = v;
    }
  }
}

}  // namespace

// true if the dense kernel takes this shape (the caller has checked the 16-B alignment of in / out / residual rows)
bool apr_internal_dense_ok(int64_t M, int32_t cin, int32_t cout) {
  static const int s_on = env_int("APR_DENSE_GEMM", 1);   // A/B switch: 0 = identity maps stay on the tile kernel
  // One workgroup walks the whole of cin for its 64 rows x 64 columns: below ~144 workgroups (deep levels: 1-4 k rows
  // against cin = 1920 ... 3840) the chip is underfilled and the tile kernel, which deals the cin chunks of a tile to
  // its 4 waves, is faster (measured: 116 workgroups 47 vs 27 us, 160 workgroups 26 vs 33 us).
  return s_on && M > 0 && M < (1ll << 31) && cin % 64 == 0 && cout % 64 == 0 && cdiv64(M, 64) * (cout / 64) >= 144;
}

int apr_internal_dense_gemm(const float* in, int64_t ldi, int64_t M, int32_t cin, int32_t cout, const float* wp,
                            const float* scale, const float* shift, const float* residual, int64_t ldr, int32_t relu,
                            float* out, int64_t ldo, hipStream_t st) {
  // 128-row tiles halve the weight staging per row but need enough workgroups to fill 256 CUs (>= 2 per CU)
  const int64_t ncol = cout / 64;
  const bool big = cdiv64(M, 128) * ncol >= 512;
  if (big)
    hipLaunchKernelGGL(k_dense_gemm<2>, dim3((unsigned)(cdiv64(M, 128) * ncol)), dim3(256), 0, st, in, ldi, (int)M, cin,
                       cout, wp, scale, shift, residual, ldr, relu, out, ldo);
  else
    hipLaunchKernelGGL(k_dense_gemm<1>, dim3((unsigned)(cdiv64(M, 64) * ncol)), dim3(256), 0, st, in, ldi, (int)M, cin,
                       cout, wp, scale, shift, residual, ldr, relu, out, ldo);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

// rows through LDS (k_dense_gemm_bf3<.., ASTG>): APR_DENSE_ASTG=0 is the A/B switch back to fragment-shaped loads
static bool dense_astg(int32_t cin) {
  static const int s_on = env_int("APR_DENSE_ASTG", 1);
  return s_on != 0 && cin >= 128;
}
static int dense_astg_attr() {      // 48 KB static + 16 G KB dynamic LDS: above 64 KB needs the opt-in, once per device
  static std::mutex s_mu;
  static bool s_attr[64] = {};
  int dev = 0;
  APR_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(s_mu);
  if (dev >= 0 && dev < 64 && !s_attr[dev]) {
    APR_HIP(hipFuncSetAttribute((const void*)k_dense_gemm_bf3<2, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768));
    APR_HIP(hipFuncSetAttribute((const void*)k_dense_gemm_bf3<2, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768));
    APR_HIP(hipFuncSetAttribute((const void*)k_dense_gemm_bf3<1, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 16384));
    APR_HIP(hipFuncSetAttribute((const void*)k_dense_gemm_bf3<1, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 16384));
    s_attr[dev] = true;
  }
  return APR_OK;
}

// The same contraction on the bf16 3-way split (k_dense_gemm_bf3): out = act((in @ W) * scale + shift + residual) for
// an identity map.  w_bf3: apr_spconv_pack_weights_bf3(w, K = 1, cin, cout).  cin % 64 == 0, cout % 64 == 0, rows of
// in / out / residual 16-B aligned (ld % 4 == 0).
APR_API int apr_dense_gemm_bf3(const float* in, int64_t ldi, int64_t M, int32_t cin, int32_t cout, const void* w_bf3,
                               const float* scale, const float* shift, const float* residual, int64_t ldr, int32_t relu,
                               float* out, int64_t ldo, void* stream) {
  APR_CHECK_ARG(in && out && w_bf3 && M > 0 && M < (1ll << 31) && cin >= 64 && cin % 64 == 0 && cout >= 64 &&
                    cout % 64 == 0,
                "apr_dense_gemm_bf3: needs M > 0, cin %% 64 == 0, cout %% 64 == 0");
  APR_CHECK_ARG(ldi >= cin && ldo >= cout && ldi % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)in & 15) == 0 &&
                    ((uintptr_t)out & 15) == 0,
                "apr_dense_gemm_bf3: rows of in / out must be 16-byte aligned");
  APR_CHECK_ARG(!residual || (ldr >= cout && ldr % 4 == 0 && ((uintptr_t)residual & 15) == 0),
                "apr_dense_gemm_bf3: residual rows must be 16-byte aligned");
  APR_CHECK_ARG((!scale || ((uintptr_t)scale & 15) == 0) && (!shift || ((uintptr_t)shift & 15) == 0),
                "apr_dense_gemm_bf3: scale / shift must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int64_t ncol = cout / 64;
  static const int s_g = env_int("APR_DENSE_G", 0);      // A/B switch: force 64-row (1) or 128-row (2) tiles
  const bool g2 = s_g == 2 || (s_g == 0 && cdiv64(M, 128) * ncol >= 512);
  const bool astg = dense_astg(cin);
  if (astg) {
    int rca = dense_astg_attr();
    if (rca != APR_OK) return rca;
  }
  auto kern = g2 ? (astg ? k_dense_gemm_bf3<2, false, true> : k_dense_gemm_bf3<2, false, false>)
                 : (astg ? k_dense_gemm_bf3<1, false, true> : k_dense_gemm_bf3<1, false, false>);
  hipLaunchKernelGGL(kern, dim3((unsigned)(cdiv64(M, g2 ? 128 : 64) * ncol)), dim3(256), astg ? (g2 ? 32768 : 16384) : 0, st, in,
                     ldi, (int)M, cin, cout, (const unsigned char*)w_bf3, scale, shift, residual, ldr, relu, out, ldo, DenseSegs{},
                     (double*)nullptr);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

// out = act(instance_norm(in @ W) (+ residual)) per row segment in TWO launches: the GEMM leaves the raw product in `out`
// and the per-tile column sums in scratch (k_dense_gemm_bf3<G, true>), norm.hip's apply kernel rebuilds mean / rstd from them
// and normalises in place.  Replaces apr_dense_gemm_bf3 + apr_instance_norm_act[_seg] (three launches).
APR_API size_t apr_dense_gemm_bf3_norm_scratch_bytes(int64_t M, int32_t cout, int32_t nseg) {
  return (size_t)(cdiv64(M > 0 ? M : 1, 64) + (nseg > 0 ? nseg : 1)) * 2 * (size_t)cout * sizeof(double) + 256;
}

APR_API int apr_dense_gemm_bf3_norm_act(const float* in, int64_t ldi, int64_t M, int32_t cin, int32_t cout, const void* w_bf3,
                                        float eps, const float* residual, int64_t ldr, int32_t act_mode, float negative_slope,
                                        float* out, int64_t ldo, const int64_t* seg_offsets_host, int32_t nseg, void* scratch,
                                        size_t scratch_bytes, void* stream) {
  APR_CHECK_ARG(in && out && w_bf3 && M > 0 && M < (1ll << 31) && cin >= 64 && cin % 64 == 0 && cout >= 64 && cout % 64 == 0,
                "apr_dense_gemm_bf3_norm_act: needs M > 0, cin %% 64 == 0, cout %% 64 == 0");
  APR_CHECK_ARG(ldi >= cin && ldo >= cout && ldi % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0,
                "apr_dense_gemm_bf3_norm_act: rows of in / out must be 16-byte aligned");
  APR_CHECK_ARG(!residual || ldr >= cout, "apr_dense_gemm_bf3_norm_act: ldr < cout");
  APR_CHECK_ARG(eps >= 0.f && act_mode >= 0 && act_mode <= 2, "apr_dense_gemm_bf3_norm_act: bad eps / act_mode");
  const int ns = (seg_offsets_host && nseg > 1) ? nseg : 1;
  APR_CHECK_ARG(ns <= 32, "apr_dense_gemm_bf3_norm_act: at most 32 segments");
  APR_CHECK_ARG(scratch && scratch_bytes >= apr_dense_gemm_bf3_norm_scratch_bytes(M, cout, ns), "apr_dense_gemm_bf3_norm_act: scratch too small");
  hipStream_t st = (hipStream_t)stream;
  const int64_t ncol = cout / 64;
  static const int s_g = env_int("APR_DENSE_G", 0);
  const int G = (s_g == 2 || (s_g == 0 && cdiv64(M, 128) * ncol >= 512)) ? 2 : 1;
  DenseSegs sg;
  sg.nseg = ns;
  sg.row0[0] = 0;
  sg.tile0[0] = 0;
  int64_t offs[33];
  for (int i = 0; i <= ns; ++i) offs[i] = ns == 1 ? (i == 0 ? 0 : M) : seg_offsets_host[i];
  APR_CHECK_ARG(offs[0] == 0 && offs[ns] == M, "apr_dense_gemm_bf3_norm_act: segment offsets must run 0 .. M");
  for (int i = 0; i < ns; ++i) {
    APR_CHECK_ARG(offs[i + 1] > offs[i], "apr_dense_gemm_bf3_norm_act: empty segment %d", i);
    sg.row0[i + 1] = (int)offs[i + 1];
    sg.tile0[i + 1] = sg.tile0[i] + (int)cdiv64(offs[i + 1] - offs[i], 64 * G);
  }
  const int ntile = sg.tile0[ns];
  double* partial = (double*)(((uintptr_t)scratch + 255) & ~(uintptr_t)255);
  const bool astg = dense_astg(cin);
  if (astg) {
    int rca = dense_astg_attr();
    if (rca != APR_OK) return rca;
  }
  auto kern = G == 2 ? (astg ? k_dense_gemm_bf3<2, true, true> : k_dense_gemm_bf3<2, true, false>)
                     : (astg ? k_dense_gemm_bf3<1, true, true> : k_dense_gemm_bf3<1, true, false>);
  hipLaunchKernelGGL(kern, dim3((unsigned)(ntile * ncol)), dim3(256), astg ? (G == 2 ? 32768 : 16384) : 0, st, in, ldi, (int)M, cin,
                     cout, (const unsigned char*)w_bf3, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                     (int64_t)0, 0, out, ldo, sg, partial);
  APR_LAUNCH_CHECK();
  return apr_internal_norm_apply_partials(out, ldo, cout, offs, sg.tile0, ns, partial, eps, residual, ldr, act_mode, negative_slope,
                                          out, ldo, st);
}
