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
#include <atomic>
#include <mutex>

#include "common.h"
#include "bf3.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWaves = 8;
constexpr int kThreads = 64 * kWaves;
constexpr int kInBits = 23;                      // input row in the low 23 bits of a pair word, local output row above
constexpr unsigned kInMask = (1u << kInBits) - 1u;
constexpr int kSched = 128;                      // schedule words per tile (27 offsets x <= 4 slots)
constexpr int kSlice64 = 24576;                  // bf16-split slice W[k][:, 64 columns] per 64 input channels: 3 x 8 KB
constexpr int kRing = 8;                         // pair-word ring: slots of 64 B per wave

// schedule word of one ITEM = (offset k, slot s): in slot s wave w owns the 16-pair group w + 8 s of the offset's list
//   bits 0-4 k | 5-7 s | 8 last slot of the offset | 9-13 next live offset (31: none) | 16-31 pairs of the offset
struct OsViews {
  int* cnt;          // [ntiles][32]  pairs of (tile, offset); entry 31 = items of the tile
  unsigned* sched;   // [ntiles][kSched]
  unsigned* pair;    // [ntiles][K][R]
};

__host__ __device__ inline size_t os_align256(size_t x) { return (x + 255) & ~(size_t)255; }

__host__ __device__ inline OsViews os_carve(void* blob, int64_t ntiles) {
  OsViews v;
  v.cnt = (int*)blob;
  v.sched = (unsigned*)((char*)blob + os_align256((size_t)ntiles * 32 * 4));
  v.pair = (unsigned*)((char*)v.sched + os_align256((size_t)ntiles * kSched * 4));
  return v;
}

// One workgroup per tile: the [rows, K] slab through LDS (coalesced), wave w ranks offsets w, w + 8, ... by ballot +
// popcount over 64-row chunks and writes the compact list of (tile, offset) in row order; wave 0 then lays out the
// tile's item schedule.
__global__ __launch_bounds__(kThreads) void k_os_build(const int* __restrict__ nbr, int n_out, int K, int R, OsViews v) {
  extern __shared__ int s_nbr[];   // [R][K] + 32 counts
  int* const s_cnt = s_nbr + R * K;
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
    if (lane == 0) s_cnt[k] = run;
  }
  __syncthreads();
  if (wave == 0) {
    const int c = (lane < 32) ? s_cnt[lane] : 0;
    const unsigned long long live = __ballot(c > 0);
    const int ns = (((c + 15) >> 4) + kWaves - 1) / kWaves;      // slots of this offset (0: no pairs)
    int first = apr_wave_incl_scan(ns) - ns;
    const int items = __shfl(first + ns, 63);
    const unsigned long long above = (lane >= 63) ? 0ull : (live & ~((2ull << lane) - 1ull));
    const unsigned knext = above ? (unsigned)__builtin_ctzll(above) : 31u;
    for (int s2 = 0; s2 < ns; ++s2)
      v.sched[(int64_t)tile * kSched + first + s2] =
          (unsigned)lane | ((unsigned)s2 << 5) | ((s2 == ns - 1) ? 256u : 0u) | (knext << 9) | ((unsigned)c << 16);
    if (lane < 31) v.cnt[tile * 32 + lane] = c;
    if (lane == 31) v.cnt[tile * 32 + 31] = items;
  }
}

// ---- the row pipeline: loads issued and waited for by hand (inline asm, counted s_waitcnt).  Left to the compiler, the
// merge of its load scoreboard over the paths of the item loop ends in vmcnt(0) at the top of two items out of three.  The
// destination registers are ordinary values (three buffers used in turn, the loop body written three times): the wait
// statement takes them as in/out operands, so every use of a row is ordered behind its wait.  apr_amd/build.py checks the
// disassembly for register copies out of these buffers (a copy made while the load is in flight would read stale data).
#define APR_OS_LOAD16(DST, PTR, OFF) asm volatile("global_load_dwordx4 %0, %1, off offset:" #OFF : "=v"(DST) : "v"(PTR))
#define APR_OS_ROWS(BUF, PTR)                                                                                       \
  {                                                                                                                \
    APR_OS_LOAD16(BUF[0], PTR, 0);                                                                                 \
    APR_OS_LOAD16(BUF[1], PTR, 16);                                                                                \
    APR_OS_LOAD16(BUF[2], PTR, 128);                                                                               \
    APR_OS_LOAD16(BUF[3], PTR, 144);                                                                               \
    if (NCH == 2) {                                                                                                \
      APR_OS_LOAD16(BUF[4 * (NCH - 1) + 0], PTR, 256);                                                             \
      APR_OS_LOAD16(BUF[4 * (NCH - 1) + 1], PTR, 272);                                                             \
      APR_OS_LOAD16(BUF[4 * (NCH - 1) + 2], PTR, 384);                                                             \
      APR_OS_LOAD16(BUF[4 * (NCH - 1) + 3], PTR, 400);                                                             \
    }                                                                                                              \
  }
// everything but the youngest 4 NCH + 1 vector-memory operations has landed; BUF is then safe to read
#define APR_OS_WAIT_ITEM(BUF)                                                                                       \
  if (NCH == 1)                                                                                                    \
    asm volatile("s_waitcnt vmcnt(5)" : "+v"(BUF[0]), "+v"(BUF[1]), "+v"(BUF[2]), "+v"(BUF[3])::"memory");         \
  else                                                                                                             \
    asm volatile("s_waitcnt vmcnt(9)"                                                                              \
                 : "+v"(BUF[0]), "+v"(BUF[1]), "+v"(BUF[2]), "+v"(BUF[3]), "+v"(BUF[4 * (NCH - 1) + 0]),             \
                   "+v"(BUF[4 * (NCH - 1) + 1]), "+v"(BUF[4 * (NCH - 1) + 2]), "+v"(BUF[4 * (NCH - 1) + 3])::"memory")
#define APR_OS_WAIT_ALL(BUF)                                                                                        \
  if (NCH == 1)                                                                                                    \
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(BUF[0]), "+v"(BUF[1]), "+v"(BUF[2]), "+v"(BUF[3])::"memory");         \
  else                                                                                                             \
    asm volatile("s_waitcnt vmcnt(0)"                                                                              \
                 : "+v"(BUF[0]), "+v"(BUF[1]), "+v"(BUF[2]), "+v"(BUF[3]), "+v"(BUF[4 * (NCH - 1) + 0]),             \
                   "+v"(BUF[4 * (NCH - 1) + 1]), "+v"(BUF[4 * (NCH - 1) + 2]), "+v"(BUF[4 * (NCH - 1) + 3])::"memory")

// cin = 64 NCH.  LDS: two weight slices (48 KB x NCH) | pair-word ring (4 KB) | schedule | accumulators [R][64] f32.
// DBG: the diagnostics build (timeline stamps, ablation switches); the production instantiation carries neither.
template <int NCH, bool DBG>
__global__ __launch_bounds__(kThreads, 2) void k_os_conv(const float* __restrict__ in, int64_t ldi, OsViews v, int n_out,
                                                         int R, int K, int cout, const unsigned char* __restrict__ wp3,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ residual, int64_t ldr, int relu,
                                                         float* __restrict__ out, int64_t ldo, int ablate_arg,
                                                         unsigned long long* __restrict__ trace) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  const int ablate = DBG ? ablate_arg : 0;
  constexpr int nstep = 2 * NCH;                      // 32-channel steps
  constexpr int kSlice = NCH * kSlice64;
  constexpr int plane_bytes = nstep * 4096;           // one split plane of a slice: [step][col 64][quad 4][8 bf16]
  unsigned char* const s_w = s_raw;                                        // two slices
  unsigned char* const s_ring = s_raw + 2 * kSlice;                        // [wave][kRing][16 words]
  unsigned* const s_sched = reinterpret_cast<unsigned*>(s_ring + kWaves * kRing * 64);   // the tile's schedule words
  unsigned char* const s_acc = s_ring + kWaves * kRing * 64 + kSched * 4;  // [R][16 chunks of 16 B], chunk c at c ^ (row & 15)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, q = lane >> 4;
  const int tile = blockIdx.x, cblk = blockIdx.y;
  const int row0 = tile * R;
  const int rows = min(R, n_out - row0);
  const int nitems = __builtin_amdgcn_readfirstlane(v.cnt[tile * 32 + 31]);
  if (tid < kSched) s_sched[tid] = (tid < nitems) ? v.sched[(int64_t)tile * kSched + tid] : 0u;
  const unsigned* const lists = v.pair + (int64_t)tile * K * R;
  const unsigned ldi32 = (unsigned)ldi;
  const int frag_off = (r16 * 4 + ((r16 & 8) ? (q ^ 3) : q)) * 16;
  unsigned char* const my_ring = s_ring + wave * (kRing * 64);

  // diagnostics (APR_OS_TRACE=1): one workgroup stamps s_memtime at fixed points of every wave's timeline
  const bool tracing = DBG && trace != nullptr && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0;
  int tr_n = 0;
  auto stamp = [&](int tag) {
    if (DBG && tracing && lane == 0 && tr_n < 510) {
      trace[wave * 512 + tr_n] = ((unsigned long long)tag << 56) | (__builtin_amdgcn_s_memtime() & 0x00FFFFFFFFFFFFFFull);
      ++tr_n;
      trace[wave * 512 + 511] = tr_n;
    }
  };
  stamp(0);
  for (int o = tid * 16; o < R * 256; o += kThreads * 16) *reinterpret_cast<f32x4*>(s_acc + o) = (f32x4){0.f, 0.f, 0.f, 0.f};

  // schedule word of item i; past the end: an empty slot of the last item's offset (no group, no slice, no barrier)
  // (read from LDS five items ahead of its use; the value is the same in every lane)
  auto item = [&](int i) {
    const unsigned w = __builtin_amdgcn_readfirstlane(s_sched[i < nitems ? i : nitems - 1]);
    return i < nitems ? w : ((w & 31u) | (31u << 9));      // no group, no next offset (31): nothing is staged for it
  };
  // this wave's 16 pair words of an item -> ring slot (LDS-DMA, lanes 0-15); lanes past the list's end (and waves without a
  // group in the slot) fetch the offset's first word: a valid pair, masked at the accumulator write-back
  auto fetch_words = [&](unsigned w, int ring_slot) {
    const int k = (int)(w & 31u), cnt = (int)(w >> 16);
    const int p = (wave + kWaves * (int)((w >> 5) & 7u)) * 16 + r16;
    const unsigned* src = lists + (int64_t)k * R + (p < cnt ? p : 0);
    if (lane < 16)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(my_ring + ring_slot * 64), 4, 0, 0);
  };
  auto stage_piece = [&](int k, int buf, int piece) {   // 1 KB of the slice of offset k -> LDS buffer `buf`
    const unsigned char* src = wp3 + ((int64_t)k * (cout >> 6) + cblk) * kSlice + piece * 1024 + lane * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(s_w + buf * kSlice + piece * 1024), 16, 0, 0);
  };
  auto ring_word = [&](int ring_slot) { return *reinterpret_cast<const unsigned*>(my_ring + ring_slot * 64 + r16 * 4); };
  auto row_ptr = [&](unsigned word) { return in + (uint64_t)((ablate & 1) ? 0u : (word & kInMask)) * ldi32 + q * 8; };

  __syncthreads();   // schedule in LDS
  if (nitems > 0) {
    unsigned w0 = item(0), w1 = item(1), w2 = item(2), w3 = item(3), w4 = item(4), w5 = item(5);
#pragma unroll
    for (int u = 0; u < kSlice / 1024 / kWaves; ++u) stage_piece((int)(w0 & 31u), 0, wave + kWaves * u);
    fetch_words(w0, 0);
    fetch_words(w1, 1);
    fetch_words(w2, 2);
    fetch_words(w3, 3);
    fetch_words(w4, 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f32x4 rowA[4 * NCH], rowB[4 * NCH];
    {
      const float* p0 = row_ptr(ring_word(0));
      const float* p1 = row_ptr(ring_word(1));
      APR_OS_ROWS(rowA, p0)
      APR_OS_ROWS(rowB, p1)
    }
    APR_OS_WAIT_ALL(rowA);
    APR_OS_WAIT_ALL(rowB);
    __syncthreads();   // accumulators zeroed, first slice landed
    stamp(1);

    int buf = 0, i = 0;
    // Item i: its rows sit in buffer CUR (requested two items ago), are split into bf16 pieces at the top of the item, and
    // the rows of item i + 2 are then requested into CUR itself (two buffers used in turn: the loop body is written
    // twice); the pair words of item i + 5 go to ring slot (i + 5) & 7.  Issue order inside an item: slice pieces, pair
    // words, 4 NCH row loads -- so "all but the 4 NCH + 1 youngest" = everything up to and including the previous item's
    // slice pieces.
#define APR_OS_ITEM(CUR)                                                                                            \
    {                                                                                                              \
      const int slot = (int)((w0 >> 5) & 7u), cnt_k = (int)(w0 >> 16);                     \
      const bool last = (w0 & 256u) != 0;                                                                          \
      const int g16 = (wave + kWaves * slot) * 16;                                                                 \
      const bool real = g16 < cnt_k;            /* this wave has a group in this slot */                           \
      const bool valid = g16 + r16 < cnt_k;                                                                        \
      const unsigned char* const wbuf = s_w + buf * kSlice + frag_off;                                             \
      stamp(4);                                                                                                    \
      if (DBG && (ablate & 16)) { APR_OS_WAIT_ALL(CUR); } /* checking mode: full drain instead of the counted wait */ \
      APR_OS_WAIT_ITEM(CUR);                                                                                       \
      const unsigned wd_i = ring_word(i & 7), wd_i2 = ring_word((i + 2) & 7);                                      \
      const int orow = (int)(wd_i >> kInBits);                                                                     \
      unsigned char* const arow = s_acc + orow * 256;                                                              \
      const int sw = orow & 15;                                                                                    \
      bf16x8 ah[nstep], am[nstep], al[nstep];                                                                      \
      f32x4 acc[4];                                                                                                \
      if (real) {                                                                                                  \
        _Pragma("unroll") for (int s2 = 0; s2 < nstep; ++s2)                                                       \
          apr_split3(CUR[2 * s2], CUR[2 * s2 + 1], ah[s2], am[s2], al[s2]);                                        \
        _Pragma("unroll") for (int cb = 0; cb < 4; ++cb)                                                           \
          acc[cb] = *reinterpret_cast<const f32x4*>(arow + (((cb * 4 + q) ^ sw) << 4));                            \
      }                                                                                                            \
      __builtin_amdgcn_sched_barrier(0);                                                                           \
      stamp(5);                                                                                                    \
      /* ---- issue: the next offset's slice (first slot of an offset: by the waves that have no group in it when   \
         there are at least two of them, else by everybody), the pair words of item + 5, the rows of item + 2 */    \
      if (slot == 0 && !(ablate & 2)) {                                                                            \
        const int knext = (int)((w0 >> 9) & 31u);                                                                  \
        if (knext < 31) {                                                                                          \
          const int ng = (cnt_k + 15) >> 4;                                                                        \
          const int nbusy = ng < kWaves ? ng : kWaves;                                                             \
          const int nidle = kWaves - nbusy;                                                                        \
          if (nidle >= 2) {                                                                                        \
            if (wave >= nbusy)                                                                                     \
              for (int pc = wave - nbusy; pc < kSlice / 1024; pc += nidle) stage_piece(knext, buf ^ 1, pc);        \
          } else {                                                                                                 \
            _Pragma("unroll") for (int u = 0; u < kSlice / 1024 / kWaves; ++u)                                     \
              stage_piece(knext, buf ^ 1, wave + kWaves * u);                                                      \
          }                                                                                                        \
        }                                                                                                          \
      }                                                                                                            \
      fetch_words(w5, (i + 5) & 7);                                                                                \
      {                                                                                                            \
        const float* pr = row_ptr(wd_i2);                                                                          \
        APR_OS_ROWS(CUR, pr)                                                                                       \
      }                                                                                                            \
      __builtin_amdgcn_sched_barrier(0);                                                                           \
      stamp(6);                                                                                                    \
      if (real && !(ablate & 4)) {                                                                                 \
        /* 48 MFMAs; W fragments of (step s, 16-column block cb) fetched two (s, cb) ahead */                      \
        bf16x8 wf[3][3];                                                                                           \
        _Pragma("unroll") for (int i0 = 0; i0 < 2; ++i0)                                                           \
          _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                         \
            wf[i0][pl] = *reinterpret_cast<const bf16x8*>(wbuf + (i0 * 16) * 64 + pl * plane_bytes);               \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        _Pragma("unroll") for (int ii = 0; ii < 4 * nstep; ++ii) {                                                 \
          const int s2 = ii >> 2, cb = ii & 3;                                                                     \
          if (ii + 2 < 4 * nstep) {                                                                                \
            const int s3 = (ii + 2) >> 2, cb3 = (ii + 2) & 3;                                                      \
            _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                       \
              wf[(ii + 2) % 3][pl] =                                                                               \
                  *reinterpret_cast<const bf16x8*>(wbuf + (s3 * 64 + cb3 * 16) * 64 + pl * plane_bytes);           \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
          }                                                                                                        \
          const bf16x8 wh = wf[ii % 3][0], wm = wf[ii % 3][1], wl = wf[ii % 3][2];                                 \
          f32x4 t = acc[cb];                                                                                       \
          t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, ah[s2], t, 0, 0, 0);                                     \
          t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, al[s2], t, 0, 0, 0);                                     \
          t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, am[s2], t, 0, 0, 0);                                     \
          t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, ah[s2], t, 0, 0, 0);                                     \
          t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, am[s2], t, 0, 0, 0);                                     \
          t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ah[s2], t, 0, 0, 0);                                     \
          acc[cb] = t;                                                                                             \
          __builtin_amdgcn_sched_barrier(0);                                                                       \
        }                                                                                                          \
        if (valid && !(ablate & 8)) {                                                                              \
          _Pragma("unroll") for (int cb = 0; cb < 4; ++cb)                                                         \
            *reinterpret_cast<f32x4*>(arow + (((cb * 4 + q) ^ sw) << 4)) = acc[cb];                                \
        }                                                                                                          \
      }                                                                                                            \
      stamp(7);                                                                                                    \
      if (last) {                                                                                                  \
        /* the pieces of the next slice this wave issued (in the offset's first slot, before that item's 5 younger  \
           loads) have landed; accumulator write-backs are done; then the barrier orders this offset's updates      \
           before the next one's and publishes the slice */                                                         \
        if (NCH == 1) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory");                     \
        else asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)\n\ts_barrier" ::: "memory");                              \
        buf ^= 1;                                                                                                  \
        stamp(8);                                                                                                  \
      }                                                                                                            \
      w0 = w1; w1 = w2; w2 = w3; w3 = w4; w4 = w5;                                                                 \
      ++i;                                                                                                         \
      w5 = item(i + 5);                                                                                            \
    }
    do {      // items past the end are empty slots: one exit, at the bottom
      APR_OS_ITEM(rowA)
      APR_OS_ITEM(rowB)
    } while (i < nitems);
#undef APR_OS_ITEM
    APR_OS_WAIT_ALL(rowA);      // nothing of the pipeline may outlive the loop
    APR_OS_WAIT_ALL(rowB);
  } else {
    __syncthreads();
  }

  stamp(9);
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
  stamp(10);
}

unsigned long long* g_trace = nullptr;   // device buffer of the diagnostics trace (8 waves x 512 stamps)
std::atomic<int> g_os_debug{0};          // apr_spconv_os_set_debug

// LDS: two slices (48 KB per 64 input channels) + the pair-word ring (4 KB) + the schedule (512 B) + 256 B per
// accumulator row, <= 156 KB
inline size_t os_fixed_lds(int cin) { return (size_t)2 * (cin / 64) * kSlice64 + kWaves * kRing * 64 + kSched * 4; }
inline int os_rmax(int cin) {
  return (cin == 64 || cin == 128) ? (int)((156 * 1024 - os_fixed_lds(cin)) / 256) / 16 * 16 : 0;
}

}  // namespace

// Rows per tile for an [n_out]-row map feeding a cin -> cout layer: the smallest whole number m of rounds of 256
// workgroups (one per CU: the tile's accumulators + two weight slices take most of the 160 KB of LDS) whose tiles fit.
APR_API int32_t apr_spconv_os_tile_rows(int64_t n_out, int32_t cin, int32_t cout) {
  if (n_out <= 0 || (cin != 64 && cin != 128) || cout < 64 || cout % 64 != 0) return 0;
  static const int s_cus = env_int("APR_OS_CUS", 256);
  const int64_t ncol = cout / 64;
  const int rmax = os_rmax(cin);
  if (rmax < 64) return 0;
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
  return os_align256((size_t)ntiles * 32 * 4) + os_align256((size_t)ntiles * kSched * 4) + (size_t)ntiles * K * R * 4 + 256;
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
  hipLaunchKernelGGL(k_os_build, dim3((unsigned)ntiles), dim3(kThreads), (size_t)(R * K + 32) * 4, (hipStream_t)stream, nbr,
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
  APR_CHECK_ARG((cin == 64 || cin == 128) && cout >= 64 && cout % 64 == 0,
                "apr_spconv_os_fwd: needs cin 64 or 128 and cout %% 64 == 0");
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
      APR_HIP(hipFuncSetAttribute((const void*)k_os_conv<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      APR_HIP(hipFuncSetAttribute((const void*)k_os_conv<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      APR_HIP(hipFuncSetAttribute((const void*)k_os_conv<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      APR_HIP(hipFuncSetAttribute((const void*)k_os_conv<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      s_attr[dev] = true;
    }
  }
  const int64_t ntiles = cdiv64(n_out, R);
  OsViews v = os_carve(const_cast<void*>(os_pairs), ntiles);
  static const int s_ablate = env_int("APR_OS_ABLATE", 0);   // timing experiments only (wrong results when non-zero)
  const size_t lds = os_fixed_lds(cin) + (size_t)R * 256;
  const dim3 grid((unsigned)ntiles, (unsigned)(cout / 64));
  static const int s_trace = env_int("APR_OS_TRACE", 0);
  if (s_trace && !g_trace) {
    APR_HIP(hipMalloc(&g_trace, 8 * 512 * sizeof(unsigned long long)));
    APR_HIP(hipMemset(g_trace, 0, 8 * 512 * sizeof(unsigned long long)));
  }
  const int dbg_mode = g_os_debug.load(std::memory_order_relaxed);
  const bool dbg = s_trace || s_ablate || dbg_mode;
  const int ablate = s_ablate | (dbg_mode == 2 ? 16 : 0);
  auto kern = cin == 64 ? (dbg ? k_os_conv<1, true> : k_os_conv<1, false>) : (dbg ? k_os_conv<2, true> : k_os_conv<2, false>);
  hipLaunchKernelGGL(kern, grid, dim3(kThreads), lds, st, in, ldi, v, (int)n_out, R, K, cout, (const unsigned char*)w_bf3,
                     scale, shift, residual, ldr, relu, out, ldo, ablate, s_trace ? g_trace : nullptr);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

// Run-time guard of the hand-counted row pipeline (tests/test_spconv_gpu.py): mode 1 launches the diagnostics
// instantiation k_os_conv<*, true> with no ablation (other registers, other schedule, the same counted waits), mode 2
// the same with every counted wait preceded by a full drain (s_waitcnt vmcnt(0): no load can be in flight when its
// registers are read, whatever the compiler did around the asm statements); mode 0 = production.  All three must give
// the same bits.  Process-wide; not for concurrent use with other callers of apr_spconv_os_fwd.
APR_API int apr_spconv_os_set_debug(int32_t mode) {
  APR_CHECK_ARG(mode >= 0 && mode <= 2, "apr_spconv_os_set_debug: mode 0, 1 or 2");
  g_os_debug.store(mode, std::memory_order_relaxed);
  return APR_OK;
}

// Diagnostics: with APR_OS_TRACE=1 in the environment, the middle workgroup of every apr_spconv_os_fwd launch stamps
// s_memtime (tag << 56 | cycles) at fixed points of each wave's timeline; this copies the last launch's stamps
// (8 waves x 512 entries, entry 511 = count) to the host.  Returns APR_EINVAL when tracing is off.
APR_API int apr_spconv_os_trace(uint64_t* host_out, int32_t n) {
  APR_CHECK_ARG(g_trace != nullptr && host_out != nullptr && n == 8 * 512, "apr_spconv_os_trace: tracing is off (APR_OS_TRACE=1) or bad buffer");
  APR_HIP(hipDeviceSynchronize());
  APR_HIP(hipMemcpy(host_out, g_trace, 8 * 512 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return APR_OK;
}
