// Point-set index builds for the KPConv encoder (N1-N3):
// barycentre grid subsampling, batched radius neighbours, brute-force kNN.
//
// The reference does these on the CPU, single-threaded, inside DataLoader workers
// (unordered_map voxel grid: cpp_subsampling/grid_subsampling/grid_subsampling.cpp:5-107;
// nanoflann KD-tree radius search: cpp_neighbors/neighbors/neighbors.cpp:211-333).  On the GPU
// both become uniform-grid problems on top of the voxel hash of hash.hip:
//   * cells = hash-unique (batch, ix, iy, iz) keys in first-occurrence order;
//   * points are bucketed per cell (count -> exclusive scan -> fill);
//   * grid subsample: one thread per cell re-orders its few points by input index and sums them
//     in fp32 in that order, so the barycentres are BIT-IDENTICAL to the reference's sequential
//     `point += p` accumulation (row order differs: libstdc++ unordered_map iteration order is
//     not reproduced, parity is per cell);
//   * radius search: one wave per query probes the 27 surrounding cells (cell edge = radius),
//     lanes stride over the candidates, hits are ballot-compacted into LDS and rank-sorted by
//     (d2, index); d2 = ((dx*dx) + dy*dy) + dz*dz in fp32 without contraction, `d2 < r*r`
//     strictly -- nanoflann's L2_Simple_Adaptor / RadiusResultSet arithmetic.
// All of it is integer / latency-bound work on a few 10^4 points: HBM traffic is negligible, the
// point is to keep the index build on the device, stream-ordered in front of the encoder.
#include "common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxBatch = 64;
constexpr int kHitCap = 1024;  // max neighbours per query the radius kernels can rank

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// ---- per-cloud bounding-box minimum -------------------------------------------------------
constexpr int kMinThreads = 1024;   // one workgroup per cloud: 1024 threads x 4 independent points in flight
__global__ __launch_bounds__(kMinThreads) void k_cloud_min(const float* __restrict__ pts, const int* __restrict__ starts,
                                                           int nb, float* __restrict__ mins /*[nb,3]*/) {
  __shared__ float s[3][kMinThreads / 64];
  const int b = blockIdx.x;
  const int lo = starts[b], hi = starts[b + 1];
  float m[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
  for (int i0 = lo + threadIdx.x; i0 < hi; i0 += 4 * kMinThreads) {
    float v[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * kMinThreads;
#pragma unroll
      for (int d = 0; d < 3; ++d) v[u][d] = (i < hi) ? pts[3 * (int64_t)i + d] : __builtin_inff();
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int d = 0; d < 3; ++d) m[d] = fminf(m[d], v[u][d]);
  }
#pragma unroll
  for (int d = 0; d < 3; ++d)
    for (int o = 32; o >= 1; o >>= 1) m[d] = fminf(m[d], __shfl_xor(m[d], o));
  if ((threadIdx.x & 63) == 0)
    for (int d = 0; d < 3; ++d) s[d][threadIdx.x >> 6] = m[d];
  __syncthreads();
  if (threadIdx.x < 3) {
    float r = s[threadIdx.x][0];
    for (int w = 1; w < kMinThreads / 64; ++w) r = fminf(r, s[threadIdx.x][w]);
    mins[3 * b + threadIdx.x] = r;
  }
}

__device__ inline int batch_of(const int* __restrict__ starts, int nb, int i) {
  int b = 0;
  while (b + 1 < nb && i >= starts[b + 1]) ++b;
  return b;
}

// cell coordinates.  mode 0 (grid subsample): origin = floor(min * (1/dl)) * dl per cloud and
// i = floor((p - origin) / dl), every step a separately rounded fp32 op as in the reference.
// mode 1 (search grid): i = floor((p - min_b) / cell) -- any consistent grid works there.
__global__ void k_cell_coords(const float* __restrict__ pts, int64_t n, const int* __restrict__ starts, int nb,
                              const float* __restrict__ mins, float dl, int mode, int4* __restrict__ coords) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int b = batch_of(starts, nb, (int)i);
  int c[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    float o = mins[3 * b + d];
    if (mode == 0) o = __fmul_rn(floorf(__fmul_rn(o, __fdiv_rn(1.0f, dl))), dl);
    c[d] = (int)floorf(__fdiv_rn(__fsub_rn(pts[3 * i + d], o), dl));
  }
  coords[i] = make_int4(b, c[0], c[1], c[2]);
}

__device__ inline int table_lookup(const unsigned long long* __restrict__ keys, const int* __restrict__ vals,
                                   uint32_t mask, unsigned long long key) {
  uint32_t slot = apr_hash_u64(key) & mask;
  for (uint32_t probe = 0; probe <= mask; ++probe) {
    unsigned long long k = keys[slot];
    if (k == key) return vals[slot];
    if (k == APR_KEY_EMPTY) return -1;
    slot = (slot + 1) & mask;
  }
  return -1;
}

// Wave-aggregated counting: the lanes of a wave that fall into the same cell issue ONE atomic between them (a raw scan
// walks its cells in order, and an object at the sensor puts 20 k points into one cell: per-point atomics on one
// counter serialised k_cell_of and k_fill to 30 us each).  Returns the lane's rank inside its group, the group size and
// the group's leader (its lowest lane; an inactive lane is its own leader with size 0).
__device__ inline int wave_group_rank(int id, bool active, int* group_size, int* leader_lane) {
  const int lane = threadIdx.x & 63;
  int rank = 0;
  *group_size = 0;
  *leader_lane = lane;
  unsigned long long todo = __ballot(active);
  while (todo) {
    const int first = __ffsll((long long)todo) - 1;
    const int fid = __shfl(id, first);
    const unsigned long long grp = __ballot(active && id == fid) & todo;
    if (active && id == fid && ((todo >> lane) & 1ull)) {
      rank = __popcll(grp & ((1ull << lane) - 1ull));
      *group_size = __popcll(grp);
      *leader_lane = first;
    }
    todo &= ~grp;
  }
  return rank;
}

__global__ void k_cell_of(const int4* __restrict__ coords, int64_t n, const unsigned long long* __restrict__ keys,
                          const int* __restrict__ vals, uint32_t mask, int* __restrict__ cell,
                          int* __restrict__ cnt) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int id = -1;
  if (i < n) {
    int4 c = coords[i];
    id = table_lookup(keys, vals, mask, apr_pack_key(c.x, c.y, c.z, c.w));
    cell[i] = id;
  }
  int gsz, lead;
  wave_group_rank(id, id >= 0, &gsz, &lead);
  if (id >= 0 && lead == (int)(threadIdx.x & 63)) atomicAdd(&cnt[id], gsz);
}

// exclusive scan of cnt[0..n) -> start[0..n] in two passes of many workgroups (n lives on the device: the grids are
// sized by the host's upper bound, workgroups past n leave at once).  A single-workgroup scan pays one dependent
// load -> barrier -> store round trip of ~7 us per 8192 cells: 101 us for the 112 k cells of a stacked raw-scan batch.
//   k_scan_sums : block b adds its kScanBlock counts -> sums[b]
//   k_scan_apply: block b adds sums[0..b) (fixed order), scans its counts on top, writes start; block 0 also clears
//                 the big-cell counter, the last block writes start[n]
constexpr int kScanBlock = 4096;   // counts per workgroup: 256 threads x 16

__device__ inline int block_sum_256(int v, int* s_w) {      // sum over the 256 threads of a workgroup, to every thread
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  if (lane == 0) s_w[wave] = v;
  __syncthreads();
  const int t = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
  __syncthreads();
  return t;
}

__global__ __launch_bounds__(256) void k_scan_sums(const int* __restrict__ cnt, const int* __restrict__ n_dev,
                                                   int* __restrict__ sums) {
  __shared__ int s_w[4];
  const int n = *n_dev;
  const int base = blockIdx.x * kScanBlock;
  if (base >= n) return;
  int t = 0;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int i = base + u * 256 + threadIdx.x;      // coalesced
    t += i < n ? cnt[i] : 0;
  }
  t = block_sum_256(t, s_w);
  if (threadIdx.x == 0) sums[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void k_scan_apply(const int* __restrict__ cnt, const int* __restrict__ n_dev,
                                                    const int* __restrict__ sums, int* __restrict__ start,
                                                    int* __restrict__ big_count) {
  __shared__ int s_w[4];
  __shared__ int s_wave[4];
  const int n = *n_dev;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *big_count = 0;   // cells too large for k_barycentre's wave-local sort (listed there)
    if (n == 0) start[0] = 0;
  }
  const int base = blockIdx.x * kScanBlock;
  if (base >= n) return;
  int off = 0;
  for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) off += sums[b];
  off = block_sum_256(off, s_w);
  // thread t owns 16 CONSECUTIVE counts: serial prefix in registers, wave scan + wave offsets on top
  const int i0 = base + threadIdx.x * 16;
  int v[16], tsum = 0;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    v[u] = (i0 + u < n) ? cnt[i0 + u] : 0;
    tsum += v[u];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = tsum;
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d);
    if (lane >= d) incl += t;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int woff = 0;
  for (int w = 0; w < wave; ++w) woff += s_wave[w];
  int run = off + woff + incl - tsum;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    if (i0 + u < n) start[i0 + u] = run;
    run += v[u];
  }
  if (i0 <= n - 1 && n - 1 < i0 + 16) start[n] = run;   // the owner of the last count: counts past n are 0, run = total
}

__global__ void k_fill(const int* __restrict__ cell, int64_t n, const int* __restrict__ start,
                       int* __restrict__ cursor, int* __restrict__ sorted) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int id = i < n ? cell[i] : -1;
  int gsz, lead;
  const int rank = wave_group_rank(id, id >= 0, &gsz, &lead);
  int base = 0;
  if (id >= 0 && lead == (int)(threadIdx.x & 63)) base = atomicAdd(&cursor[id], gsz);
  base = __shfl(base, lead);            // the leader's range start reaches the members of its group
  if (id >= 0) sorted[start[id] + base + rank] = (int)i;
}

// one WAVE per cell: rank-sort the cell's point indices in LDS (cells of a raw scan near the sensor hold
// hundreds of points, a per-thread insertion sort made this kernel 8 ms), then lane 0 sums them
// sequentially in fp32 in index order -- the reference's `point += p` order -- and scales by
// (float)(1.0 / count)  (grid_subsampling.cpp:60-89).  Also counts cells per cloud.
constexpr int kCellCap = 1024;
__global__ __launch_bounds__(256) void k_barycentre(const float* __restrict__ pts, const int4* __restrict__ cell_coords,
                                                    const int* __restrict__ n_cells_dev, const int* __restrict__ start,
                                                    int* __restrict__ sorted, const float* __restrict__ feats, int fdim,
                                                    float* __restrict__ out_pts, float* __restrict__ out_feats,
                                                    int* __restrict__ out_len, int* __restrict__ big_count,
                                                    int* __restrict__ big_list) {
  __shared__ int s_raw[4][kCellCap];
  __shared__ int s_ord[4][kCellCap];
  __shared__ int s_hist[kMaxBatch];      // cells per cloud seen by this workgroup (one global atomic per bin at the end;
                                         // an atomic per CELL on 2 addresses serialised the whole kernel: 134 -> 40 us)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < kMaxBatch) s_hist[threadIdx.x] = 0;
  __syncthreads();
  const int ncell = *n_cells_dev;
  // grid-stride over the cells (the host only knows an upper bound): wave-uniform loop, no barrier inside
  for (int c = blockIdx.x * 4 + wave; c < ncell; c += gridDim.x * 4) {
  const int lo = start[c], hi = start[c + 1];
  const int m = hi - lo;
  const int* ord;
  if (m <= kCellCap) {
    // rank sort: a lane keeps its (up to 16) values in registers and every LDS read of another value is compared
    // against all of them (one LDS read per 16 comparisons; the dense cells next to the sensor hold ~1000 points
    // and set the kernel's duration)
    int v[kCellCap / 64], rank[kCellCap / 64];
#pragma unroll
    for (int u = 0; u < kCellCap / 64; ++u) {
      const int e = lane + 64 * u;
      v[u] = e < m ? sorted[lo + e] : 0x7fffffff;
      rank[u] = 0;
      if (e < m) s_raw[wave][e] = v[u];
    }
    __builtin_amdgcn_wave_barrier();
    const int nu = (m + 63) >> 6;                    // wave-uniform
    for (int o = 0; o < m; ++o) {
      const int x = s_raw[wave][o];
#pragma unroll
      for (int u = 0; u < kCellCap / 64; ++u)
        if (u < nu) rank[u] += x < v[u] ? 1 : 0;
    }
#pragma unroll
    for (int u = 0; u < kCellCap / 64; ++u)
      if (lane + 64 * u < m) s_ord[wave][rank[u]] = v[u];
    __builtin_amdgcn_wave_barrier();
    ord = s_ord[wave];
  } else {   // very dense cell (an object right at the sensor): left to k_barycentre_big, a whole workgroup per cell
    if (lane == 0) {
      big_list[atomicAdd(big_count, 1)] = c;
      atomicAdd(&s_hist[cell_coords[c].x], 1);
    }
    continue;
  }
  // the points of the cell are fetched by all lanes in parallel into LDS (the rank buffer is free again), lane 0 then
  // adds them in index order from LDS: same fp32 sum order as the reference, without one dependent global round
  // trip per point
  const bool staged = 3 * m <= kCellCap;
  float* s_val = reinterpret_cast<float*>(s_raw[wave]);
  if (staged) {
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < m; e += 64) {
      const int64_t i = ord[e];
      s_val[3 * e] = pts[3 * i];
      s_val[3 * e + 1] = pts[3 * i + 1];
      s_val[3 * e + 2] = pts[3 * i + 2];
    }
    __builtin_amdgcn_wave_barrier();
  }
  // lanes 0, 1, 2 add x, y, z: three independent chains in ONE instruction stream (a lone lane running the three
  // chains one after the other issues 3x the instructions; each wave64 instruction costs 4 cycles whatever the mask)
  if (lane < 3) {
    float acc1 = 0.f;
    if (staged) {
      for (int a = 0; a < m; ++a) acc1 = __fadd_rn(acc1, s_val[3 * a + lane]);
    } else {
      for (int a = 0; a < m; ++a) acc1 = __fadd_rn(acc1, pts[3 * (int64_t)ord[a] + lane]);
    }
    const float inv = (float)(1.0 / (double)m);
    out_pts[3 * (int64_t)c + lane] = __fmul_rn(acc1, inv);
  }
  if (lane != 0) continue;
  if (feats) {
    const float fc = (float)m;
    for (int f = 0; f < fdim; ++f) {
      float sacc = 0.f;
      for (int a = 0; a < m; ++a) sacc = __fadd_rn(sacc, feats[(int64_t)ord[a] * fdim + f]);
      out_feats[(int64_t)c * fdim + f] = __fdiv_rn(sacc, fc);
    }
  }
  atomicAdd(&s_hist[cell_coords[c].x], 1);
  }
  __syncthreads();
  if (threadIdx.x < kMaxBatch && s_hist[threadIdx.x]) atomicAdd(&out_len[threadIdx.x], s_hist[threadIdx.x]);
}

// Cells with more than kCellCap points (listed by k_barycentre): one 1024-thread workgroup per cell sorts the cell's
// point indices with a bitonic network whose comparators all point the same way (so the padding up to a power of two
// is virtual: +inf never moves down), in LDS up to kBigLds indices and in place in global memory beyond, then stages
// the points 1024 at a time in LDS for thread 0 to add in index order -- the same fp32 sum as the reference's
// sequential loop (grid_subsampling.cpp:60-89), in bounded time: a single lane's insertion sort over 20 k points
// took a second.
constexpr int kBigLds = 8192;
__global__ __launch_bounds__(1024) void k_barycentre_big(const float* __restrict__ pts, const int* __restrict__ big_count,
                                                         const int* __restrict__ big_list, const int* __restrict__ start,
                                                         int* __restrict__ sorted, const float* __restrict__ feats,
                                                         int fdim, float* __restrict__ out_pts,
                                                         float* __restrict__ out_feats) {
  __shared__ int s_keys[kBigLds];
  __shared__ float s_val[3 * 1024];
  const int nbig = *big_count;
  const int t = threadIdx.x;
  for (int b = blockIdx.x; b < nbig; b += gridDim.x) {   // workgroup-uniform
    const int c = big_list[b];
    const int lo = start[c], m = start[c + 1] - lo;
    int* glob = sorted + lo;
    const bool in_lds = m <= kBigLds;
    if (in_lds)
      for (int e = t; e < m; e += 1024) s_keys[e] = glob[e];
    __syncthreads();
    int* arr = in_lds ? s_keys : glob;
    int np2 = 1;
    while (np2 < m) np2 <<= 1;
    for (int k = 2; k <= np2; k <<= 1) {
      // merge of two sorted runs of k/2: first the mirrored comparators, then the half-cleaners
      for (int i = t; i < (np2 >> 1); i += 1024) {
        const int blk = i / (k >> 1), off = i - blk * (k >> 1);
        const int a = blk * k + off, bb = blk * k + k - 1 - off;
        if (bb < m) {
          const int va = arr[a], vb = arr[bb];
          if (va > vb) {
            arr[a] = vb;
            arr[bb] = va;
          }
        }
      }
      __syncthreads();
      for (int j = k >> 2; j >= 1; j >>= 1) {
        for (int i = t; i < (np2 >> 1); i += 1024) {
          const int a = (i / j) * 2 * j + (i % j), bb = a + j;
          if (bb < m) {
            const int va = arr[a], vb = arr[bb];
            if (va > vb) {
              arr[a] = vb;
              arr[bb] = va;
            }
          }
        }
        __syncthreads();
      }
    }
    float acc1 = 0.f;     // threads 0, 1, 2: the x, y, z chains
    for (int e0 = 0; e0 < m; e0 += 1024) {
      const int cnt = m - e0 < 1024 ? m - e0 : 1024;
      if (t < cnt) {
        const int64_t i = arr[e0 + t];
        s_val[3 * t] = pts[3 * i];
        s_val[3 * t + 1] = pts[3 * i + 1];
        s_val[3 * t + 2] = pts[3 * i + 2];
      }
      __syncthreads();
      if (t < 3)
        for (int a = 0; a < cnt; ++a) acc1 = __fadd_rn(acc1, s_val[3 * a + t]);
      __syncthreads();
    }
    if (t < 3) out_pts[3 * (int64_t)c + t] = __fmul_rn(acc1, (float)(1.0 / (double)m));
    for (int f = 0; feats && f < fdim; ++f) {
      float sacc = 0.f;
      for (int e0 = 0; e0 < m; e0 += 1024) {
        const int cnt = m - e0 < 1024 ? m - e0 : 1024;
        if (t < cnt) s_val[t] = feats[(int64_t)arr[e0 + t] * fdim + f];
        __syncthreads();
        if (t == 0)
          for (int a = 0; a < cnt; ++a) sacc = __fadd_rn(sacc, s_val[a]);
        __syncthreads();
      }
      if (t == 0) out_feats[(int64_t)c * fdim + f] = __fdiv_rn(sacc, (float)m);
    }
    __syncthreads();
  }
}

// ---- radius search ---------------------------------------------------------------------------
struct Grid {
  const unsigned long long* keys;
  const int* vals;
  uint32_t mask;
  const int* start;   // [ncell + 1]
  const int* sorted;  // support indices bucketed by cell
  const float* mins;  // [nb,3] per-cloud support minimum
  float cell;
};

// MODE 0: count only.  MODE 1: fill + sort.  MODE 2: fill + sort, and the query's neighbour count to counts[].
template <int MODE>
__global__ __launch_bounds__(256) void k_radius(const float* __restrict__ q, int64_t nq, const int* __restrict__ qstarts,
                                                const float* __restrict__ s, int nb, Grid g, float r2,
                                                int* __restrict__ counts, int* __restrict__ out, int width,
                                                int64_t ld, int pad_value, int* __restrict__ status) {
  __shared__ float s_d[4][kHitCap];
  __shared__ int s_i[4][kHitCap];
  __shared__ int s_lo[4][32], s_incl[4][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t qi = (int64_t)blockIdx.x * 4 + wave;
  if (qi >= nq) return;
  const int b = batch_of(qstarts, nb, (int)qi);
  const float qx = q[3 * qi], qy = q[3 * qi + 1], qz = q[3 * qi + 2];
  const int cx = (int)floorf(__fdiv_rn(__fsub_rn(qx, g.mins[3 * b]), g.cell));
  const int cy = (int)floorf(__fdiv_rn(__fsub_rn(qy, g.mins[3 * b + 1]), g.cell));
  const int cz = (int)floorf(__fdiv_rn(__fsub_rn(qz, g.mins[3 * b + 2]), g.cell));
  // lanes 0..26 each probe one neighbouring cell; the 27 (start, inclusive-count) pairs go to LDS
  int c_lo = 0, c_n = 0;
  if (lane < 27) {
    const int x = cx + lane % 3 - 1, y = cy + (lane / 3) % 3 - 1, z = cz + lane / 9 - 1;
    if (apr_key_in_range(b, x, y, z)) {
      int id = table_lookup(g.keys, g.vals, g.mask, apr_pack_key(b, x, y, z));
      if (id >= 0) {
        c_lo = g.start[id];
        c_n = g.start[id + 1] - c_lo;
      }
    }
  }
  int incl = c_n;
  for (int d = 1; d < 32; d <<= 1) {   // all 64 lanes take part: no shuffle from an inactive lane
    int t = __shfl_up(incl, d);
    if (lane >= d) incl += t;
  }
  if (lane < 27) {
    s_lo[wave][lane] = c_lo;
    s_incl[wave][lane] = incl;
  }
  const int total = __shfl(incl, 26);
  int nhit = 0;
  // Four 64-candidate batches per turn: a query has ~260 candidates in its 27 cells (6.4x the in-radius ones), and a batch
  // is a chain of dependent fetches (cell range from LDS -> sorted[] -> the support point's coordinates).  Taken one batch
  // at a time the wave sat through ~5 such chains back to back; here the index loads of four batches are issued together,
  // then their coordinate loads, and the hits are appended in the same candidate order as before (same tables out).
  for (int base = 0; base < total; base += 256) {
    int sidx[4];
    float px[4], py[4], pz[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = base + u * 64 + lane;
      sidx[u] = -1;
      if (t < total) {
        // cell range holding candidate t: first L with incl[L] > t (27 sorted entries: 5 LDS probes instead of a scan
        // that averages 13)
        int L = 0;
#pragma unroll
        for (int step = 16; step >= 1; step >>= 1)
          if (L + step <= 26 && s_incl[wave][L + step - 1] <= t) L += step;
        const int before = L ? s_incl[wave][L - 1] : 0;
        sidx[u] = g.sorted[s_lo[wave][L] + (t - before)];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      px[u] = py[u] = pz[u] = 0.f;
      if (sidx[u] >= 0) {
        px[u] = s[3 * (int64_t)sidx[u]];
        py[u] = s[3 * (int64_t)sidx[u] + 1];
        pz[u] = s[3 * (int64_t)sidx[u] + 2];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (base + u * 64 >= total) break;               // wave-uniform
      const float dx = __fsub_rn(qx, px[u]), dy = __fsub_rn(qy, py[u]), dz = __fsub_rn(qz, pz[u]);
      const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      const bool hit = sidx[u] >= 0 && d2 < r2;
      const unsigned long long m = __ballot(hit);
      if (MODE >= 1 && hit) {
        const int pos = nhit + __popcll(m & ((1ull << lane) - 1ull));
        if (pos < kHitCap) {
          s_d[wave][pos] = d2;
          s_i[wave][pos] = sidx[u];
        }
      }
      nhit += __popcll(m);
    }
  }
  if (MODE == 0) {
    if (lane == 0) counts[qi] = nhit;
    return;
  }
  if (MODE == 2 && lane == 0) counts[qi] = nhit;
  if (nhit > kHitCap) {
    // More neighbours than the rank buffer holds (a scan taken a few centimetres from a wall).  Only the `width`
    // nearest are wanted: when they fit, select them exactly — bisection on the bit pattern of d2 (non-negative
    // floats order like their bits) for the smallest T with at least `width` candidates at d2 <= T, all candidates
    // below T plus the smallest-index ones AT T (a second bisection) make up exactly `width` entries — and rank
    // those.  A few dozen extra sweeps over the candidates, for the rare query that needs them.
    if (width > kHitCap) {
      if (lane == 0) *status = 1;       // cannot rank more columns than the buffer holds
      nhit = kHitCap;
    } else {
      auto sweep = [&](auto&& f) {       // f(d2, index) for every in-radius candidate of this query
        for (int base = 0; base < total; base += 64) {
          const int t = base + lane;
          bool hit = false;
          float d2 = 0.f;
          int sidx = -1;
          if (t < total) {
            int L = 0;
#pragma unroll
            for (int step = 16; step >= 1; step >>= 1)
              if (L + step <= 26 && s_incl[wave][L + step - 1] <= t) L += step;
            const int before = L ? s_incl[wave][L - 1] : 0;
            sidx = g.sorted[s_lo[wave][L] + (t - before)];
            const float dx = __fsub_rn(qx, s[3 * (int64_t)sidx]), dy = __fsub_rn(qy, s[3 * (int64_t)sidx + 1]),
                        dz = __fsub_rn(qz, s[3 * (int64_t)sidx + 2]);
            d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            hit = d2 < r2;
          }
          f(hit, d2, sidx);
        }
      };
      unsigned lo = 0u, hi = __float_as_uint(r2);
      while (lo < hi) {
        const unsigned mid = lo + ((hi - lo) >> 1);
        int c = 0;
        sweep([&](bool hit, float d2, int) { c += __popcll(__ballot(hit && __float_as_uint(d2) <= mid)); });
        if (c >= width) hi = mid; else lo = mid + 1;
      }
      int below = 0;
      sweep([&](bool hit, float d2, int) { below += __popcll(__ballot(hit && __float_as_uint(d2) < lo)); });
      const int need = width - below;    // >= 1 entries to take at d2 == T, smallest indices first
      int ilo = 0, ihi = 0x7fffffff;
      while (ilo < ihi) {
        const int mid = ilo + ((ihi - ilo) >> 1);
        int c = 0;
        sweep([&](bool hit, float d2, int id) { c += __popcll(__ballot(hit && __float_as_uint(d2) == lo && id <= mid)); });
        if (c >= need) ihi = mid; else ilo = mid + 1;
      }
      int got = 0;
      sweep([&](bool hit, float d2, int id) {
        const unsigned bits = __float_as_uint(d2);
        const bool take = hit && (bits < lo || (bits == lo && id <= ilo));
        const unsigned long long m = __ballot(take);
        if (take) {
          const int pos = got + __popcll(m & ((1ull << lane) - 1ull));
          s_d[wave][pos] = d2;
          s_i[wave][pos] = id;
        }
        got += __popcll(m);
      });
      nhit = got;                         // == width
    }
  }
  // rank sort by (d2, index); wave-private LDS rows, so no barrier is needed
  for (int e = lane; e < nhit; e += 64) {
    const float d = s_d[wave][e];
    const int id = s_i[wave][e];
    int rank = 0;
    for (int o = 0; o < nhit; ++o) {
      const float od = s_d[wave][o];
      rank += (od < d || (od == d && s_i[wave][o] < id)) ? 1 : 0;
    }
    if (rank < width) out[qi * ld + rank] = id;
  }
  for (int e = nhit + lane; e < width; e += 64) out[qi * ld + e] = pad_value;
}

__global__ void k_max_int(const int* __restrict__ v, int64_t n, int* __restrict__ out) {
  int m = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    m = max(m, v[i]);
  for (int d = 32; d >= 1; d >>= 1) m = max(m, __shfl_xor(m, d));
  if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// ---- brute-force kNN (k <= 16) for the overlap-attention graph ---------------------------------
template <int KMAX>
__global__ __launch_bounds__(256) void k_knn(const float* __restrict__ pts, int n, int k, int skip_first,
                                             int* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qi = blockIdx.x * 4 + wave;
  if (qi >= n) return;
  const float qx = pts[3 * qi], qy = pts[3 * qi + 1], qz = pts[3 * qi + 2];
  float bd[KMAX];
  int bi[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    bd[j] = __builtin_inff();
    bi[j] = 0x7fffffff;
  }
  const int want = k + skip_first;
  for (int i = lane; i < n; i += 64) {
    const float dx = qx - pts[3 * i], dy = qy - pts[3 * i + 1], dz = qz - pts[3 * i + 2];
    float d = fmaxf(dx * dx + dy * dy + dz * dz, 1e-12f);
    int id = i;
    // insert into the lane-local sorted list (ascending by (d, index))
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      if (j < want && (d < bd[j] || (d == bd[j] && id < bi[j]))) {
        float td = bd[j]; int ti = bi[j];
        bd[j] = d; bi[j] = id;
        d = td; id = ti;
      }
    }
  }
  // merge: `want` rounds of wave arg-min over the lanes' list heads
  for (int r = 0; r < want; ++r) {
    float hd = bd[0];
    int hi = bi[0];
    float md = hd;
    int mi = hi;
    for (int dd = 32; dd >= 1; dd >>= 1) {
      float od = __shfl_xor(md, dd);
      int oi = __shfl_xor(mi, dd);
      if (od < md || (od == md && oi < mi)) { md = od; mi = oi; }
    }
    if (hd == md && hi == mi) {  // this lane owns the winner: pop
#pragma unroll
      for (int j = 0; j + 1 < KMAX; ++j) { bd[j] = bd[j + 1]; bi[j] = bi[j + 1]; }
      bd[KMAX - 1] = __builtin_inff();
      bi[KMAX - 1] = 0x7fffffff;
    }
    if (lane == 0 && r >= skip_first) out[qi * k + (r - skip_first)] = (mi == 0x7fffffff) ? qi : mi;
  }
}

struct GridWork {
  int4* coords;
  unsigned long long* keys;
  int* vals;
  int64_t cap;
  int4* cell_coords;
  int* n_cells;
  int* status;
  int* cell;
  int* cnt;
  int* start;
  int* cursor;
  int* sorted;
  void* map_scratch;
  size_t map_scratch_bytes;
  int* starts_dev;
  float* mins;
  int* big;   // [0]: number of cells with more than kCellCap points; their ids are listed in `cnt` (dead after the scan)
};

size_t grid_work_bytes(int64_t n) {
  const int64_t cap = apr_hash_capacity(n);
  return align256(n * 16) + align256(cap * 8) + align256(cap * 4) + align256(n * 16) + 256 + 256 + align256(n * 4) +
         align256(n * 4) + align256((n + 1) * 4) + align256(n * 4) + align256(n * 4) +
         align256(apr_map_scratch_bytes(n)) + align256((kMaxBatch + 1) * 4) + align256(kMaxBatch * 12) + 1024;
}

GridWork carve(void* scratch, int64_t n) {
  GridWork w;
  char* p = (char*)scratch;
  w.cap = apr_hash_capacity(n);
  auto take = [&](size_t bytes) {
    void* r = p;
    p += align256(bytes);
    return r;
  };
  w.coords = (int4*)take(n * 16);
  w.keys = (unsigned long long*)take(w.cap * 8);
  w.vals = (int*)take(w.cap * 4);
  w.cell_coords = (int4*)take(n * 16);
  w.n_cells = (int*)take(256);
  w.status = (int*)take(256);
  w.cell = (int*)take(n * 4);
  w.cnt = (int*)take(n * 4);
  w.start = (int*)take((n + 1) * 4);
  w.cursor = (int*)take(n * 4);
  w.sorted = (int*)take(n * 4);
  w.map_scratch_bytes = apr_map_scratch_bytes(n);
  w.map_scratch = take(w.map_scratch_bytes);
  w.starts_dev = (int*)take((kMaxBatch + 1) * 4);
  w.mins = (float*)take(kMaxBatch * 12);
  w.big = (int*)take(256);
  return w;
}

// Batch start offsets reach the device as a kernel argument, not by hipMemcpyAsync from a stack array: the
// synchronisation-free entry point returns before its stream work runs.
struct BatchStarts {
  int v[kMaxBatch + 1];
};
__global__ void k_set_starts(int* __restrict__ dst, BatchStarts s, int n) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = s.v[threadIdx.x];
}

// points -> cells -> buckets.  mode as in k_cell_coords.
int build_grid(const float* pts, int64_t n, const int32_t* lengths_host, int nb, float cell, int mode, GridWork& w,
               hipStream_t st) {
  BatchStarts bs;
  int* starts = bs.v;
  starts[0] = 0;
  for (int b = 0; b < nb; ++b) starts[b + 1] = starts[b] + lengths_host[b];
  APR_CHECK_ARG(starts[nb] == n, "batch lengths sum to %d, expected %lld points", starts[nb], (long long)n);
  hipLaunchKernelGGL(k_set_starts, dim3(1), dim3(128), 0, st, w.starts_dev, bs, nb + 1);
  hipLaunchKernelGGL(k_cloud_min, dim3(nb), dim3(kMinThreads), 0, st, pts, w.starts_dev, nb, w.mins);
  const unsigned nblk = (unsigned)cdiv64(n, kBlock);
  hipLaunchKernelGGL(k_cell_coords, dim3(nblk), dim3(kBlock), 0, st, pts, n, w.starts_dev, nb, w.mins, cell, mode,
                     w.coords);
  int rc = apr_map_build((const int32_t*)w.coords, n, nullptr, 0, (uint64_t*)w.keys, w.vals, w.cap,
                         (int32_t*)w.cell_coords, nullptr, w.n_cells, w.status, w.map_scratch, w.map_scratch_bytes,
                         st);
  if (rc != APR_OK) return rc;
  APR_HIP(hipMemsetAsync(w.cnt, 0, n * 4, st));
  APR_HIP(hipMemsetAsync(w.cursor, 0, n * 4, st));
  hipLaunchKernelGGL(k_cell_of, dim3(nblk), dim3(kBlock), 0, st, w.coords, n, w.keys, w.vals, (uint32_t)(w.cap - 1),
                     w.cell, w.cnt);
  // block sums in w.sorted (free until k_fill); n cells at most
  const unsigned nscan = (unsigned)cdiv64(n, kScanBlock);
  hipLaunchKernelGGL(k_scan_sums, dim3(nscan), dim3(256), 0, st, w.cnt, w.n_cells, w.sorted);
  hipLaunchKernelGGL(k_scan_apply, dim3(nscan), dim3(256), 0, st, w.cnt, w.n_cells, w.sorted, w.start, w.big);
  hipLaunchKernelGGL(k_fill, dim3(nblk), dim3(kBlock), 0, st, w.cell, n, w.start, w.cursor, w.sorted);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

}  // namespace

APR_API size_t apr_grid_subsample_scratch_bytes(int64_t n) { return grid_work_bytes(n > 0 ? n : 1); }

namespace {
int grid_subsample_enqueue(const float* pts, int64_t n, const int32_t* lengths_host, int32_t nb, float dl,
                           const float* feats, int32_t fdim, float* out_pts, float* out_feats, void* scratch,
                           size_t scratch_bytes, GridWork* w_out, hipStream_t st) {
  APR_CHECK_ARG(n > 0 && n < (1ll << 31) && nb > 0 && nb <= kMaxBatch && dl > 0.f, "apr_grid_subsample: bad arguments");
  APR_CHECK_ARG(scratch_bytes >= grid_work_bytes(n), "apr_grid_subsample: scratch too small");
  for (int b = 0; b < nb; ++b) APR_CHECK_ARG(lengths_host[b] > 0, "apr_grid_subsample: empty cloud in batch");
  GridWork w = carve(scratch, n);
  int rc = build_grid(pts, n, lengths_host, nb, dl, 0, w, st);
  if (rc != APR_OK) return rc;
  int* out_len_dev = w.cursor;  // cursor is dead after k_fill; reuse its first nb ints
  APR_HIP(hipMemsetAsync(out_len_dev, 0, kMaxBatch * 4, st));
  hipLaunchKernelGGL(k_barycentre, dim3((unsigned)(cdiv64(n, 4) < 4096 ? cdiv64(n, 4) : 4096)), dim3(256), 0, st, pts, w.cell_coords,
                     w.n_cells, w.start, w.sorted, feats, fdim, out_pts, out_feats, out_len_dev, w.big, w.cnt);
  hipLaunchKernelGGL(k_barycentre_big, dim3(64), dim3(1024), 0, st, pts, w.big, w.cnt, w.start, w.sorted, feats, fdim,
                     out_pts, out_feats);
  APR_LAUNCH_CHECK();
  *w_out = w;
  return APR_OK;
}
}  // namespace

APR_API int apr_grid_subsample(const float* pts, int64_t n, const int32_t* lengths_host, int32_t nb, float dl,
                               const float* feats, int32_t fdim, float* out_pts, float* out_feats,
                               int32_t* out_lengths_host, void* scratch, size_t scratch_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  GridWork w;
  int rc = grid_subsample_enqueue(pts, n, lengths_host, nb, dl, feats, fdim, out_pts, out_feats, scratch, scratch_bytes,
                                  &w, st);
  if (rc != APR_OK) return rc;
  int status = 0;
  APR_HIP(hipMemcpyAsync(out_lengths_host, w.cursor, nb * 4, hipMemcpyDeviceToHost, st));
  APR_HIP(hipMemcpyAsync(&status, w.status, 4, hipMemcpyDeviceToHost, st));
  APR_HIP(hipStreamSynchronize(st));
  if (status != 0) {
    apr_set_error("apr_grid_subsample: cell index outside the packed-key range");
    return APR_ERANGE;
  }
  return APR_OK;
}

// The same without the host synchronisation: the cloud lengths and the status word go to `lengths_status_dev`
// (int32[nb + 1] on the device: nb subsampled lengths, then 0 or the out-of-range flag), which the caller fetches
// when convenient; out_pts / out_feats must hold n rows (the subsampled rows are the first sum(lengths) of them).
APR_API int apr_grid_subsample_async(const float* pts, int64_t n, const int32_t* lengths_host, int32_t nb, float dl,
                                     const float* feats, int32_t fdim, float* out_pts, float* out_feats,
                                     int32_t* lengths_status_dev, void* scratch, size_t scratch_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APR_CHECK_ARG(lengths_status_dev, "apr_grid_subsample_async: lengths_status_dev is NULL");
  GridWork w;
  int rc = grid_subsample_enqueue(pts, n, lengths_host, nb, dl, feats, fdim, out_pts, out_feats, scratch, scratch_bytes,
                                  &w, st);
  if (rc != APR_OK) return rc;
  APR_HIP(hipMemcpyAsync(lengths_status_dev, w.cursor, nb * 4, hipMemcpyDeviceToDevice, st));
  APR_HIP(hipMemcpyAsync(lengths_status_dev + nb, w.status, 4, hipMemcpyDeviceToDevice, st));
  return APR_OK;
}

// internal (not part of the C ABI): uniform search grid over one cloud, used by the geometric
// RANSAC validation in ransac.hip
size_t apr_internal_grid_bytes(int64_t n) { return grid_work_bytes(n > 0 ? n : 1); }
int apr_internal_search_grid(const float* pts, int64_t n, float cell, void* scratch, AprSearchGrid* out,
                             hipStream_t st) {
  GridWork w = carve(scratch, n);
  int32_t len = (int32_t)n;
  int rc = build_grid(pts, n, &len, 1, cell, 1, w, st);
  if (rc != APR_OK) return rc;
  out->keys = w.keys;
  out->vals = w.vals;
  out->mask = (uint32_t)(w.cap - 1);
  out->start = w.start;
  out->sorted = w.sorted;
  out->mins = w.mins;
  out->cell = cell;
  out->cell_coords = w.cell_coords;
  out->n_cells = w.n_cells;
  return APR_OK;
}

// the same over a BATCH of clouds stacked in pts (lengths_host[nb]): cells are keyed (cloud, x, y, z), origins per cloud
int apr_internal_search_grid_batch(const float* pts, int64_t n, const int32_t* lengths_host, int32_t nb, float cell, void* scratch,
                                   AprSearchGrid* out, hipStream_t st) {
  if (nb < 1 || nb > kMaxBatch) {
    apr_set_error("search_grid_batch: 1 .. %d clouds", kMaxBatch);
    return APR_EINVAL;
  }
  GridWork w = carve(scratch, n);
  int rc = build_grid(pts, n, lengths_host, nb, cell, 1, w, st);
  if (rc != APR_OK) return rc;
  out->keys = w.keys;
  out->vals = w.vals;
  out->mask = (uint32_t)(w.cap - 1);
  out->start = w.start;
  out->sorted = w.sorted;
  out->mins = w.mins;
  out->cell = cell;
  out->cell_coords = w.cell_coords;
  out->n_cells = w.n_cells;
  return APR_OK;
}

APR_API size_t apr_radius_scratch_bytes(int64_t nq, int64_t ns) {
  return grid_work_bytes(ns > 0 ? ns : 1) + align256((nq > 0 ? nq : 1) * 4) + align256((kMaxBatch + 1) * 4) + 512;
}

APR_API int apr_radius_neighbors(const float* queries, int64_t nq, const float* supports, int64_t ns,
                                 const int32_t* q_lengths_host, const int32_t* s_lengths_host, int32_t nb, float radius,
                                 int32_t limit, int32_t* out, int64_t out_ld, int32_t* width_host,
                                 void* scratch, size_t scratch_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APR_CHECK_ARG(nq > 0 && ns > 0 && nq < (1ll << 31) && ns < (1ll << 31) && nb > 0 && nb <= kMaxBatch && radius > 0.f,
                "apr_radius_neighbors: bad arguments");
  APR_CHECK_ARG(scratch_bytes >= apr_radius_scratch_bytes(nq, ns), "apr_radius_neighbors: scratch too small");
  GridWork w = carve(scratch, ns);
  char* p = (char*)scratch + grid_work_bytes(ns);
  int* counts = (int*)p;
  p += align256(nq * 4);
  int* qstarts = (int*)p;
  p += align256((kMaxBatch + 1) * 4);
  int* maxc = (int*)p;
  int rc = build_grid(supports, ns, s_lengths_host, nb, radius, 1, w, st);
  if (rc != APR_OK) return rc;
  int qs[kMaxBatch + 1];
  qs[0] = 0;
  for (int b = 0; b < nb; ++b) qs[b + 1] = qs[b] + q_lengths_host[b];
  APR_CHECK_ARG(qs[nb] == nq, "apr_radius_neighbors: query batch lengths sum to %d, expected %lld", qs[nb], (long long)nq);
  APR_HIP(hipMemcpyAsync(qstarts, qs, (nb + 1) * 4, hipMemcpyHostToDevice, st));
  Grid g{w.keys, w.vals, (uint32_t)(w.cap - 1), w.start, w.sorted, w.mins, radius};
  const float r2 = radius * radius;
  const unsigned grid = (unsigned)cdiv64(nq, 4);
  // counting pass: the reference pads every row to the global maximum neighbour count and the
  // caller then keeps the first `limit` columns, so width = min(max_count, limit)
  int width = 0;
  APR_HIP(hipMemsetAsync(maxc, 0, 4, st));
  hipLaunchKernelGGL(k_radius<0>, dim3(grid), dim3(256), 0, st, queries, nq, qstarts, supports, nb, g, r2, counts,
                     (int*)nullptr, 0, (int64_t)0, 0, w.status);
  hipLaunchKernelGGL(k_max_int, dim3(64), dim3(256), 0, st, counts, nq, maxc);
  APR_HIP(hipMemcpyAsync(&width, maxc, 4, hipMemcpyDeviceToHost, st));
  APR_HIP(hipStreamSynchronize(st));
  if (limit > 0 && width > limit) width = limit;
  *width_host = width;
  if (out == nullptr) return APR_OK;  // size query only
  APR_CHECK_ARG(width <= out_ld, "apr_radius_neighbors: need %d columns, buffer rows hold %lld", width,
                (long long)out_ld);
  if (width == 0) return APR_OK;
  APR_HIP(hipMemsetAsync(w.status, 0, 4, st));
  hipLaunchKernelGGL(k_radius<1>, dim3(grid), dim3(256), 0, st, queries, nq, qstarts, supports, nb, g, r2, counts, out,
                     width, out_ld, (int)ns, w.status);
  APR_LAUNCH_CHECK();
  int status = 0;
  APR_HIP(hipMemcpyAsync(&status, w.status, 4, hipMemcpyDeviceToHost, st));
  APR_HIP(hipStreamSynchronize(st));
  if (status != 0) {
    apr_set_error("apr_radius_neighbors: a query has more than %d neighbours within the radius", kHitCap);
    return APR_ERANGE;
  }
  return APR_OK;
}

// Host-synchronisation-free variant for callers that know the column limit (KPConv's calibrated neighbourhood limits):
// one fill pass writes limit columns per query (sorted by distance, padded with ns) AND the query's full neighbour
// count; flags_dev[0] = max count over all queries, flags_dev[1] != 0 if a query overflowed the candidate buffer.
// The reference's width is min(max count, limit): a caller that needs it reads flags_dev when convenient (one
// synchronisation for a whole pyramid of tables) and drops the all-padding columns [max count, limit) if any.
static int radius_async(const float* queries, int64_t nq, const float* supports, int64_t ns,
                        const int32_t* q_lengths_host, const int32_t* s_lengths_host, int32_t nb, float radius,
                        int32_t limit, int32_t* out, int64_t out_ld, int32_t* flags_dev, void* scratch,
                        size_t scratch_bytes, void* stream, bool reuse_grid) {
  hipStream_t st = (hipStream_t)stream;
  APR_CHECK_ARG(nq > 0 && ns > 0 && nq < (1ll << 31) && ns < (1ll << 31) && nb > 0 && nb <= kMaxBatch && radius > 0.f,
                "apr_radius_neighbors_async: bad arguments");
  APR_CHECK_ARG(limit > 0 && out != nullptr && out_ld >= limit && flags_dev != nullptr,
                "apr_radius_neighbors_async: needs limit > 0, an output of >= limit columns and the flag words");
  APR_CHECK_ARG(scratch_bytes >= apr_radius_scratch_bytes(nq, ns), "apr_radius_neighbors_async: scratch too small");
  GridWork w = carve(scratch, ns);
  char* p = (char*)scratch + grid_work_bytes(ns);
  int* counts = (int*)p;
  p += align256(nq * 4);
  int* qstarts = (int*)p;
  if (!reuse_grid) {
    int rc = build_grid(supports, ns, s_lengths_host, nb, radius, 1, w, st);
    if (rc != APR_OK) return rc;
  }
  BatchStarts qb;
  int* qs = qb.v;
  qs[0] = 0;
  for (int b = 0; b < nb; ++b) qs[b + 1] = qs[b] + q_lengths_host[b];
  APR_CHECK_ARG(qs[nb] == nq, "apr_radius_neighbors_async: query batch lengths sum to %d, expected %lld", qs[nb],
                (long long)nq);
  hipLaunchKernelGGL(k_set_starts, dim3(1), dim3(128), 0, st, qstarts, qb, nb + 1);
  Grid g{w.keys, w.vals, (uint32_t)(w.cap - 1), w.start, w.sorted, w.mins, radius};
  APR_HIP(hipMemsetAsync(flags_dev, 0, 8, st));
  hipLaunchKernelGGL(k_radius<2>, dim3((unsigned)cdiv64(nq, 4)), dim3(256), 0, st, queries, nq, qstarts, supports, nb, g,
                     radius * radius, counts, out, (int)limit, out_ld, (int)ns, flags_dev + 1);
  hipLaunchKernelGGL(k_max_int, dim3(64), dim3(256), 0, st, counts, nq, flags_dev);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API int apr_radius_neighbors_async(const float* queries, int64_t nq, const float* supports, int64_t ns,
                                       const int32_t* q_lengths_host, const int32_t* s_lengths_host, int32_t nb,
                                       float radius, int32_t limit, int32_t* out, int64_t out_ld, int32_t* flags_dev,
                                       void* scratch, size_t scratch_bytes, void* stream) {
  return radius_async(queries, nq, supports, ns, q_lengths_host, s_lengths_host, nb, radius, limit, out, out_ld,
                      flags_dev, scratch, scratch_bytes, stream, false);
}

// Same, searching the grid the PREVIOUS apr_radius_neighbors_async call left in `scratch`: same supports, same
// s_lengths, same radius, same stream (KPConv's collate queries every level's points twice with one radius: once from
// the level itself, once from the pooled level: the ~15 launches of the second grid build are saved).
APR_API int apr_radius_neighbors_regrid_async(const float* queries, int64_t nq, const float* supports, int64_t ns,
                                              const int32_t* q_lengths_host, const int32_t* s_lengths_host, int32_t nb,
                                              float radius, int32_t limit, int32_t* out, int64_t out_ld,
                                              int32_t* flags_dev, void* scratch, size_t scratch_bytes, void* stream) {
  return radius_async(queries, nq, supports, ns, q_lengths_host, s_lengths_host, nb, radius, limit, out, out_ld,
                      flags_dev, scratch, scratch_bytes, stream, true);
}

APR_API int apr_knn(const float* pts, int32_t n, int32_t k, int32_t skip_first, int32_t* out, void* stream) {
  APR_CHECK_ARG(n > 0 && k > 0 && k + skip_first <= 16, "apr_knn: need n > 0 and k + skip_first <= 16");
  hipLaunchKernelGGL(k_knn<16>, dim3((unsigned)cdiv64(n, 4)), dim3(256), 0, (hipStream_t)stream, pts, n, k,
                     skip_first, out);
  APR_LAUNCH_CHECK();
  return APR_OK;
}
