// Adjacent components of the hot path: APG aggregation
// (rigid transform of the complement frames, crop to the key frame's radius, compaction) and the
// 1-NN squared-distance sums of the Chamfer loss.  Pure HBM-bound streaming + a tiled brute-force NN.
#include "common.h"

namespace {

constexpr int kBlock = 256;

// out = pts @ R^T + t  (FCGF_APR/lib/complement_data_loader.py:65-70, float32 like the reference)
__global__ void k_transform(const float* __restrict__ pts, int64_t n, const float* __restrict__ T /*[16] row-major*/,
                            float* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
  out[3 * i] = x * T[0] + y * T[1] + z * T[2] + T[3];
  out[3 * i + 1] = x * T[4] + y * T[5] + z * T[6] + T[7];
  out[3 * i + 2] = x * T[8] + y * T[9] + z * T[10] + T[11];
}

// max over points of |p|^2 (bit pattern of a non-negative float orders like an unsigned int)
__global__ void k_max_sqnorm(const float* __restrict__ pts, int64_t n, unsigned* __restrict__ out_bits) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    m = fmaxf(m, x * x + y * y + z * z);
  }
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
  if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __float_as_uint(m));
}

// keep[i] = |p_i|^2 < limit ; per-block counts (compaction pass 1)
__global__ void k_crop_flags(const float* __restrict__ pts, int64_t n, const unsigned* __restrict__ limit_bits,
                             uint8_t* __restrict__ flags, int* __restrict__ block_counts) {
  __shared__ int wave_cnt[kBlock / 64];
  const float limit = __uint_as_float(*limit_bits);
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool f = false;
  if (i < n) {
    const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    f = (x * x + y * y + z * z) < limit;
    flags[i] = f;
  }
  unsigned long long b = __ballot(f);
  if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = __popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

__global__ void k_scan_counts(const int* __restrict__ counts, int nblk, int* __restrict__ offsets, int* __restrict__ total) {
  __shared__ int wave_sum[16];
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int base = 0; base < nblk; base += blockDim.x) {
    int idx = base + threadIdx.x;
    int v = idx < nblk ? counts[idx] : 0;
    int incl = v;
    for (int d = 1; d < 64; d <<= 1) {
      int t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wave_sum[w];
    int carry = carry_s;
    if (idx < nblk) offsets[idx] = carry + woff + incl - v;
    __syncthreads();
    if (threadIdx.x == blockDim.x - 1) carry_s = carry + woff + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry_s;
}

__global__ void k_compact_points(const float* __restrict__ pts, int64_t n, const uint8_t* __restrict__ flags,
                                 const int* __restrict__ block_offsets, float* __restrict__ out) {
  __shared__ int wave_cnt[kBlock / 64];
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool f = (i < n) && flags[i];
  unsigned long long b = __ballot(f);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_cnt[wave] = __popcll(b);
  __syncthreads();
  if (!f) return;
  int pos = block_offsets[blockIdx.x] + __popcll(b & ((1ull << lane) - 1ull));
  for (int w = 0; w < wave; ++w) pos += wave_cnt[w];
  out[3 * (int64_t)pos] = pts[3 * i];
  out[3 * (int64_t)pos + 1] = pts[3 * i + 1];
  out[3 * (int64_t)pos + 2] = pts[3 * i + 2];
}

// sum_i min_j |a_i - b_j|^2 : one query per thread, targets tiled through LDS, fp64 sum of the minima
constexpr int kTile = 512;
__global__ __launch_bounds__(256) void k_nn3_min(const float* __restrict__ a, int64_t n, const float* __restrict__ b,
                                                 int64_t m, int chunk, unsigned* __restrict__ best_bits) {
  __shared__ float s_b[kTile * 3];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < n;
  const float x = live ? a[3 * i] : 0.f, y = live ? a[3 * i + 1] : 0.f, z = live ? a[3 * i + 2] : 0.f;
  float best = __builtin_inff();
  const int64_t t0 = (int64_t)blockIdx.y * chunk, t1 = min((long long)(t0 + chunk), (long long)m);
  for (int64_t tb = t0; tb < t1; tb += kTile) {
    const int rows = (int)min((long long)kTile, (long long)(t1 - tb));
    __syncthreads();
    for (int e = threadIdx.x; e < rows * 3; e += 256) s_b[e] = b[tb * 3 + e];
    __syncthreads();
    for (int r = 0; r < rows; ++r) {
      const float dx = x - s_b[3 * r], dy = y - s_b[3 * r + 1], dz = z - s_b[3 * r + 2];
      best = fminf(best, dx * dx + dy * dy + dz * dz);
    }
  }
  if (live) atomicMin(&best_bits[i], __float_as_uint(best));
}

__device__ inline float d2_rn(float ax, float ay, float az, float bx, float by, float bz);

// the same search keeping the arg-min: packed (bits(d^2) << 32 | j), 64-bit atomicMin (ties: the smallest j)
__global__ __launch_bounds__(256) void k_nn3_arg(const float* __restrict__ a, int64_t n, const float* __restrict__ b,
                                                 int64_t m, int chunk, unsigned long long* __restrict__ best) {
  __shared__ float s_b[kTile * 3];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < n;
  const float x = live ? a[3 * i] : 0.f, y = live ? a[3 * i + 1] : 0.f, z = live ? a[3 * i + 2] : 0.f;
  float bd = __builtin_inff();
  unsigned bj = 0xFFFFFFFFu;
  const int64_t t0 = (int64_t)blockIdx.y * chunk, t1 = min((long long)(t0 + chunk), (long long)m);
  for (int64_t tb = t0; tb < t1; tb += kTile) {
    const int rows = (int)min((long long)kTile, (long long)(t1 - tb));
    __syncthreads();
    for (int e = threadIdx.x; e < rows * 3; e += 256) s_b[e] = b[tb * 3 + e];
    __syncthreads();
    for (int r = 0; r < rows; ++r) {
      const float d = d2_rn(x, y, z, s_b[3 * r], s_b[3 * r + 1], s_b[3 * r + 2]);
      if (d < bd) {            // ascending j: the first of equal distances stays
        bd = d;
        bj = (unsigned)(tb + r);
      }
    }
  }
  if (live && bj != 0xFFFFFFFFu) atomicMin(&best[i], ((unsigned long long)__float_as_uint(bd) << 32) | bj);
}

// k_nn3_arg over a LIST of queries whose length lives on the device (the queries the grid passes left open): workgroup =
// 256 listed queries x one chunk of targets, targets tiled through LDS; workgroups past the end of the list leave at once
__global__ __launch_bounds__(256) void k_nn3_arg_list(const float* __restrict__ a, const int* __restrict__ list,
                                                      const int* __restrict__ list_n, const float* __restrict__ b, int64_t m,
                                                      int chunk, unsigned long long* __restrict__ best) {
  __shared__ float s_b[kTile * 3];
  const int cnt = *list_n;
  if ((int64_t)blockIdx.x * 256 >= cnt) return;              // workgroup-uniform
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = q < cnt;
  const int64_t i = live ? list[q] : 0;
  const float x = live ? a[3 * i] : 0.f, y = live ? a[3 * i + 1] : 0.f, z = live ? a[3 * i + 2] : 0.f;
  float bd = __builtin_inff();
  unsigned bj = 0xFFFFFFFFu;
  const int64_t t0 = (int64_t)blockIdx.y * chunk, t1 = min((long long)(t0 + chunk), (long long)m);
  for (int64_t tb = t0; tb < t1; tb += kTile) {
    const int rows = (int)min((long long)kTile, (long long)(t1 - tb));
    __syncthreads();
    for (int e = threadIdx.x; e < rows * 3; e += 256) s_b[e] = b[tb * 3 + e];
    __syncthreads();
    for (int r = 0; r < rows; ++r) {
      const float d = d2_rn(x, y, z, s_b[3 * r], s_b[3 * r + 1], s_b[3 * r + 2]);
      if (d < bd) {
        bd = d;
        bj = (unsigned)(tb + r);
      }
    }
  }
  if (live && bj != 0xFFFFFFFFu) atomicMin(&best[i], ((unsigned long long)__float_as_uint(bd) << 32) | bj);
}

// ---------------------------------------------------------------------------------------------------------------------
// Grid-accelerated exact 1-NN (round 5).  The brute-force search above is n x m distance evaluations (56 k generated points
// against a 50-56 k-point APG cloud, both directions, both frames: 3 ms of a training iteration); nearly every query has its
// neighbour within a voxel or two -- in ONE direction.  Generated points sit on the key frame's voxels, so 97 % of them find
// an APG point within 0.6 m; the APG cloud also covers what the key frame never saw, and 40 % of ITS points have their nearest
// generated point more than 1.8 m away, 15 % more than 9 m (scripts/dbg_nn_dist.py).  One uniform grid over the targets (cell
// c: points.hip's search grid), two passes, both exact:
//   A  thread per query: the 2^3 cells around the query -- everything within 0.49 c; a query whose best distance is inside
//      that radius is DONE (no closer point can exist outside the cells read), the others go to a list (wave-aggregated
//      append; the order of the list does not matter);
//   B  wave per listed query: ring after ring of the same grid (the cube of 2k cells per axis covers (k - 0.51) c), the lanes
//      sharing a shell's cells, until the wave-wide best distance is inside the covered radius; what is still open after
//      kMaxRing rings (6.6 m at c = 1.2 m) goes to a second list;
//   C  the tiled full search (k_nn3_arg's loop) over that list.
//   (Tried and dropped: the 4^3 shell inside the thread pass -- divergent, 380 us per call; a second, coarser grid -- its
//   build is ~20 launches per call; a wave-per-query full search for everything pass A leaves -- every wave re-reads all
//   targets from L2, 420 us per call.)
// Distances are (dx^2 + dy^2) + dz^2 with explicitly rounded operations in every pass, ties go to the smaller index: the
// packed result is bit-identical to k_nn3_arg's whatever path a query took (tested).
// ---------------------------------------------------------------------------------------------------------------------
__device__ inline float d2_rn(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

__device__ inline void better(float d, unsigned j, float& bd, unsigned& bj) {
  if (d < bd || (d == bd && j < bj)) {
    bd = d;
    bj = j;
  }
}

// several independent searches in one call: cloud s of the queries (rows a0[s] .. a0[s + 1]) is searched in cloud s of the
// targets (rows b0[s] .. b0[s + 1]) only; the grid's cells carry the cloud index in their key
constexpr int kNn3MaxSeg = 32;
struct Nn3Segs {
  int nb;
  int a0[kNn3MaxSeg + 1];
  int b0[kNn3MaxSeg + 1];
  double scale[kNn3MaxSeg];      // the per-cloud sum is multiplied by this on the way out (1, or 1 / rows for a mean)
};

__device__ inline int seg_of(const Nn3Segs& sg, int64_t i) {
  int s = 0;
  while (s + 1 < sg.nb && i >= sg.a0[s + 1]) ++s;
  return s;
}

// cells [base - (ring - 1), base + ring]^3 minus the cube of the previous ring; base = floor(u - 0.5), u = (p - min) / cell
__device__ inline void cell_base(const AprSearchGrid& g, int seg, float x, float y, float z, int* base) {
  base[0] = (int)floorf(__fsub_rn(__fdiv_rn(__fsub_rn(x, g.mins[3 * seg]), g.cell), 0.5f));
  base[1] = (int)floorf(__fsub_rn(__fdiv_rn(__fsub_rn(y, g.mins[3 * seg + 1]), g.cell), 0.5f));
  base[2] = (int)floorf(__fsub_rn(__fdiv_rn(__fsub_rn(z, g.mins[3 * seg + 2]), g.cell), 0.5f));
}

__global__ __launch_bounds__(256) void k_nn3_grid_thread(const float* __restrict__ a, int64_t n, const float* __restrict__ b,
                                                         AprSearchGrid g, Nn3Segs sg, unsigned long long* __restrict__ best,
                                                         int* __restrict__ list, int* __restrict__ list_n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < n;
  bool unresolved = false;
  if (live) {
    const float x = a[3 * i], y = a[3 * i + 1], z = a[3 * i + 2];
    const int seg = seg_of(sg, i);
    int base[3];
    cell_base(g, seg, x, y, z, base);
    float bd = __builtin_inff();
    unsigned bj = 0xFFFFFFFFu;
    bool done = false;
#pragma unroll
    for (int o = 0; o < 8; ++o) {                               // ring 1: the 2^3 cells around the query
      const int X = base[0] + (o & 1), Y = base[1] + ((o >> 1) & 1), Z = base[2] + (o >> 2);
      if (!apr_key_in_range(seg, X, Y, Z)) continue;
      const int id = apr_table_lookup(g.keys, g.vals, g.mask, apr_pack_key(seg, X, Y, Z));
      if (id < 0) continue;
      const int e1 = g.start[id + 1];
      for (int e = g.start[id]; e < e1; ++e) {
        const unsigned j = (unsigned)g.sorted[e];
        better(d2_rn(x, y, z, b[3 * (int64_t)j], b[3 * (int64_t)j + 1], b[3 * (int64_t)j + 2]), j, bd, bj);
      }
    }
    {
      const float r = 0.49f * g.cell;                           // every target within r lies in the cells read
      done = bd <= r * r;
    }
    if (done) best[i] = ((unsigned long long)__float_as_uint(bd) << 32) | bj;
    unresolved = !done;
  }
  const unsigned long long bal = __ballot(unresolved);
  if (bal) {
    const int lane = threadIdx.x & 63;
    int pos0 = 0;
    if (lane == 0) pos0 = atomicAdd(list_n, __popcll(bal));
    pos0 = __shfl(pos0, 0);
    if (unresolved) list[pos0 + __popcll(bal & ((1ull << lane) - 1ull))] = (int)i;
  }
}

// wave per listed query: the shells of rings 2 .. kMaxRing of the same grid, the LANES taking the shell's cells in turn (a far
// query's cost is hash probes into mostly empty cells: 64 of them in flight per wave), a wave-wide minimum after every
// ring; what is still open after the last ring -- or when the grid is absent (g.cell == 0) -- takes every target
constexpr int kMaxRing = 6;
__global__ __launch_bounds__(256) void k_nn3_grid_wave(const float* __restrict__ a, const float* __restrict__ b, int64_t m,
                                                       AprSearchGrid g, Nn3Segs sg, const int* __restrict__ list_in,
                                                       const int* __restrict__ list_in_n, unsigned long long* __restrict__ best,
                                                       int* __restrict__ list_out, int* __restrict__ list_out_n) {
  const int lane = threadIdx.x & 63;
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= *list_in_n) return;                               // wave-uniform
  const int64_t i = list_in[q];
  const float x = a[3 * i], y = a[3 * i + 1], z = a[3 * i + 2];
  const int seg = seg_of(sg, i);
  float bd = __builtin_inff();
  unsigned bj = 0xFFFFFFFFu;
  bool done = false;
  if (g.cell > 0.f) {
    int base[3];
    cell_base(g, seg, x, y, z, base);
    // ring 1 again (the thread pass does not hand its candidate over): 8 cells, lanes over the cells' points
    for (int o = 0; o < 8; ++o) {
      const int X = base[0] + (o & 1), Y = base[1] + ((o >> 1) & 1), Z = base[2] + (o >> 2);
      if (!apr_key_in_range(seg, X, Y, Z)) continue;
      const int id = apr_table_lookup(g.keys, g.vals, g.mask, apr_pack_key(seg, X, Y, Z));
      if (id < 0) continue;
      const int e1 = g.start[id + 1];
      for (int e = g.start[id] + lane; e < e1; e += 64) {
        const unsigned j = (unsigned)g.sorted[e];
        better(d2_rn(x, y, z, b[3 * (int64_t)j], b[3 * (int64_t)j + 1], b[3 * (int64_t)j + 2]), j, bd, bj);
      }
    }
    for (int ring = 2; ring <= kMaxRing && !done; ++ring) {
      const int L = 2 * ring, lo = -(ring - 1);
      for (int t = lane; t < L * L * L; t += 64) {
        const int cx = lo + t % L, cy = lo + (t / L) % L, cz = lo + t / (L * L);
        if (cx > lo && cx < ring && cy > lo && cy < ring && cz > lo && cz < ring) continue;      // the previous rings' cube
        const int X = base[0] + cx, Y = base[1] + cy, Z = base[2] + cz;
        if (!apr_key_in_range(seg, X, Y, Z)) continue;
        const int id = apr_table_lookup(g.keys, g.vals, g.mask, apr_pack_key(seg, X, Y, Z));
        if (id < 0) continue;
        const int e1 = g.start[id + 1];
        for (int e = g.start[id]; e < e1; ++e) {
          const unsigned j = (unsigned)g.sorted[e];
          better(d2_rn(x, y, z, b[3 * (int64_t)j], b[3 * (int64_t)j + 1], b[3 * (int64_t)j + 2]), j, bd, bj);
        }
      }
      float wd = bd;
      for (int d = 32; d >= 1; d >>= 1) wd = fminf(wd, __shfl_xor(wd, d));
      const float r = ((float)ring - 0.51f) * g.cell;          // every target within r lies in the cube read so far
      done = wd <= r * r;
    }
  }
  if (!done && list_out) {                                   // one cloud: the tiled full search takes it (k_nn3_arg_list)
    if (lane == 0) {
      best[i] = ~0ull;
      list_out[atomicAdd(list_out_n, 1)] = (int)i;
    }
    return;
  }
  if (!done && g.cell > 0.f) {
    // several clouds, nothing proven within kMaxRing rings (a complement cloud reaches far beyond the frame it is compared
    // with: 11-18 % of its points are > 9 m from every generated point): the OCCUPIED CELLS of the query's own cloud, lanes
    // over the cells -- a box's distance first, its points only if the box can hold a point at least as near as the best
    // so far.  ~10 k box tests instead of ~56 k point tests per query (the wave used to walk every target of the cloud).
    // The box is widened by 1 % of a cell on every side: a point sits in its cell up to the rounding of (p - min) / cell.
    const int nc = *g.n_cells;
    const float ox = g.mins[3 * seg], oy = g.mins[3 * seg + 1], oz = g.mins[3 * seg + 2];
    const float m = 0.01f * g.cell;
    auto box_d2 = [&](const int4 cc) {
      const float lx = ox + (float)cc.y * g.cell - m, ly = oy + (float)cc.z * g.cell - m, lz = oz + (float)cc.w * g.cell - m;
      const float hx = lx + g.cell + 2.f * m, hy = ly + g.cell + 2.f * m, hz = lz + g.cell + 2.f * m;
      const float dx = fmaxf(fmaxf(lx - x, x - hx), 0.f), dy = fmaxf(fmaxf(ly - y, y - hy), 0.f);
      const float dz = fmaxf(fmaxf(lz - z, z - hz), 0.f);
      return dx * dx + dy * dy + dz * dz;
    };
    float T = bd;                                            // what the rings found (not proven minimal), wave-wide
    for (int d = 32; d >= 1; d >>= 1) T = fminf(T, __shfl_xor(T, d));
    if (T == __builtin_inff()) {                             // nothing yet: the nearest box's points give the first bound
      float nd = __builtin_inff();
      int nid = -1;
      for (int id = lane; id < nc; id += 64) {
        const int4 cc = g.cell_coords[id];
        if (cc.x != seg) continue;
        const float d2 = box_d2(cc);
        if (d2 < nd) { nd = d2; nid = id; }
      }
      for (int d = 32; d >= 1; d >>= 1) {
        const float od = __shfl_xor(nd, d);
        const int oi = __shfl_xor(nid, d);
        if (od < nd || (od == nd && oi >= 0 && (nid < 0 || oi < nid))) { nd = od; nid = oi; }
      }
      if (nid >= 0) {
        const int e1 = g.start[nid + 1];
        for (int e = g.start[nid] + lane; e < e1; e += 64) {
          const unsigned j = (unsigned)g.sorted[e];
          better(d2_rn(x, y, z, b[3 * (int64_t)j], b[3 * (int64_t)j + 1], b[3 * (int64_t)j + 2]), j, bd, bj);
        }
        T = bd;
        for (int d = 32; d >= 1; d >>= 1) T = fminf(T, __shfl_xor(T, d));
      }
    }
    for (int id = lane; id < nc; id += 64) {
      const int4 cc = g.cell_coords[id];
      if (cc.x != seg) continue;
      if (box_d2(cc) > fminf(T, bd)) continue;               // <=: a tie with a smaller index may sit in this box
      const int e1 = g.start[id + 1];
      for (int e = g.start[id]; e < e1; ++e) {
        const unsigned j = (unsigned)g.sorted[e];
        better(d2_rn(x, y, z, b[3 * (int64_t)j], b[3 * (int64_t)j + 1], b[3 * (int64_t)j + 2]), j, bd, bj);
      }
    }
  } else if (!done) {                                        // no grid: every target of the query's own cloud
    bd = __builtin_inff();
    bj = 0xFFFFFFFFu;
    for (int64_t j = sg.b0[seg] + lane; j < sg.b0[seg + 1]; j += 64)
      better(d2_rn(x, y, z, b[3 * j], b[3 * j + 1], b[3 * j + 2]), (unsigned)j, bd, bj);
  }
  for (int d = 32; d >= 1; d >>= 1) {
    const float od = __shfl_xor(bd, d);
    const unsigned oj = (unsigned)__shfl_xor((int)bj, d);
    better(od, oj, bd, bj);
  }
  if (lane == 0) best[i] = ((unsigned long long)__float_as_uint(bd) << 32) | bj;
}

// sum of the distances of a packed arg-min array in a FIXED order (one workgroup: lane-strided partials, LDS tree)
__global__ __launch_bounds__(1024) void k_sum_packed(const unsigned long long* __restrict__ best, int64_t n,
                                                     double* __restrict__ out) {
  __shared__ double part[1024];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) s += (double)__uint_as_float((unsigned)(best[i] >> 32));
  part[threadIdx.x] = s;
  __syncthreads();
  for (int d = 512; d >= 1; d >>= 1) {
    if ((int)threadIdx.x < d) part[threadIdx.x] += part[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = part[0];
}

// the same per cloud: workgroup s sums the minima of rows a0[s] .. a0[s + 1]
__global__ __launch_bounds__(1024) void k_sum_packed_seg(const unsigned long long* __restrict__ best, Nn3Segs sg,
                                                         double* __restrict__ out) {
  __shared__ double part[1024];
  const int s0 = blockIdx.x;
  double s = 0.0;
  for (int64_t i = sg.a0[s0] + threadIdx.x; i < sg.a0[s0 + 1]; i += 1024) s += (double)__uint_as_float((unsigned)(best[i] >> 32));
  part[threadIdx.x] = s;
  __syncthreads();
  for (int d = 512; d >= 1; d >>= 1) {
    if ((int)threadIdx.x < d) part[threadIdx.x] += part[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[s0] = part[0] * sg.scale[s0];
}

__global__ void k_sum_bits(const unsigned* __restrict__ bits, int64_t n, double* __restrict__ out) {
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    s += (double)__uint_as_float(bits[i]);
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

APR_API int apr_transform_points(const float* pts, int64_t n, const float* T16_dev, float* out, void* stream) {
  APR_CHECK_ARG(n >= 0, "apr_transform_points: n < 0");
  if (n == 0) return APR_OK;
  hipLaunchKernelGGL(k_transform, dim3((unsigned)cdiv64(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, pts, n,
                     T16_dev, out);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API size_t apr_crop_scratch_bytes(int64_t n) {
  const int64_t nblk = cdiv64(n > 0 ? n : 1, kBlock);
  return align256(n) + 2 * align256(nblk * 4) + 512;
}

APR_API int apr_crop_to_radius(const float* key_pts, int64_t n_key, const float* pts, int64_t n, float* out,
                               int32_t* n_out_dev, void* scratch, size_t scratch_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APR_CHECK_ARG(n_key > 0 && n >= 0 && n < (1ll << 31), "apr_crop_to_radius: bad sizes");
  APR_CHECK_ARG(scratch_bytes >= apr_crop_scratch_bytes(n), "apr_crop_to_radius: scratch too small");
  char* p = (char*)scratch;
  unsigned* limit = (unsigned*)p;
  p += 256;
  uint8_t* flags = (uint8_t*)p;
  p += align256(n);
  const int nblk = (int)cdiv64(n > 0 ? n : 1, kBlock);
  int* cnt = (int*)p;
  p += align256((size_t)nblk * 4);
  int* off = (int*)p;
  APR_HIP(hipMemsetAsync(limit, 0, 4, st));
  hipLaunchKernelGGL(k_max_sqnorm, dim3(256), dim3(kBlock), 0, st, key_pts, n_key, limit);
  if (n == 0) {
    APR_HIP(hipMemsetAsync(n_out_dev, 0, 4, st));
    return APR_OK;
  }
  hipLaunchKernelGGL(k_crop_flags, dim3(nblk), dim3(kBlock), 0, st, pts, n, limit, flags, cnt);
  hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, st, cnt, nblk, off, n_out_dev);
  hipLaunchKernelGGL(k_compact_points, dim3(nblk), dim3(kBlock), 0, st, pts, n, flags, off, out);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API int apr_chamfer_sum(const float* a, int64_t n, const float* b, int64_t m, double* out_dev, void* scratch,
                            size_t scratch_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APR_CHECK_ARG(n > 0 && m > 0, "apr_chamfer_sum: empty cloud");
  APR_CHECK_ARG(scratch_bytes >= (size_t)n * 4, "apr_chamfer_sum: scratch too small (need 4*n bytes)");
  unsigned* best = (unsigned*)scratch;
  APR_HIP(hipMemsetAsync(best, 0x7F, (size_t)n * 4, st));   // 0x7F7F7F7F = 3.4e38 as float
  APR_HIP(hipMemsetAsync(out_dev, 0, 8, st));
  const int64_t qb = cdiv64(n, 256);
  int64_t want = cdiv64(2048, qb);
  int64_t chunk = cdiv64(cdiv64(m, want), kTile) * kTile;
  if (chunk < kTile) chunk = kTile;
  hipLaunchKernelGGL(k_nn3_min, dim3((unsigned)qb, (unsigned)cdiv64(m, chunk)), dim3(256), 0, st, a, n, b, m, (int)chunk,
                     best);
  hipLaunchKernelGGL(k_sum_bits, dim3(256), dim3(256), 0, st, best, n, out_dev);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API size_t apr_nn3_scratch_bytes(int64_t n, int64_t m) {
  return align256(apr_internal_grid_bytes(m)) + 2 * align256((size_t)(n > 0 ? n : 1) * 4) + 1024;
}

static int nn3_impl(const float* a, int64_t n, const int64_t* a_off, const float* b, int64_t m, const int64_t* b_off, int nb,
                    float cell, uint64_t* out_packed, double* sums_dev, const double* sum_scale_host, void* scratch,
                    size_t scratch_bytes, hipStream_t st) {
  unsigned long long* best = (unsigned long long*)out_packed;
  Nn3Segs sg;
  sg.nb = nb;
  for (int i = 0; i < nb; ++i) sg.scale[i] = sum_scale_host ? sum_scale_host[i] : 1.0;
  int32_t blen[kNn3MaxSeg];
  for (int i = 0; i <= nb; ++i) {
    sg.a0[i] = (int)a_off[i];
    sg.b0[i] = (int)b_off[i];
    if (i) blen[i - 1] = (int32_t)(b_off[i] - b_off[i - 1]);
  }
  if (cell > 0.f) {
    APR_CHECK_ARG(scratch && scratch_bytes >= apr_nn3_scratch_bytes(n, m), "apr_nn3: scratch too small");
    char* p = (char*)(((uintptr_t)scratch + 255) & ~(uintptr_t)255);
    void* g1s = p;
    p += align256(apr_internal_grid_bytes(m));
    int* list1 = (int*)p;
    p += align256((size_t)n * 4);
    int* list2 = (int*)p;
    p += align256((size_t)n * 4);
    int* counts = (int*)p;                                     // [0] list1, [1] list2
    AprSearchGrid g1;
    int rc = apr_internal_search_grid_batch(b, m, blen, nb, cell, g1s, &g1, st);
    if (rc != APR_OK) return rc;
    APR_HIP(hipMemsetAsync(counts, 0, 8, st));
    hipLaunchKernelGGL(k_nn3_grid_thread, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, a, n, b, g1, sg, best, list1, counts);
    if (nb == 1) {
      hipLaunchKernelGGL(k_nn3_grid_wave, dim3((unsigned)cdiv64(n, 4)), dim3(256), 0, st, a, b, m, g1, sg, list1, counts, best,
                         list2, counts + 1);
      const int64_t qb = cdiv64(n, 256);
      int64_t want = cdiv64(2048, qb);
      int64_t chunk = cdiv64(cdiv64(m, want), kTile) * kTile;
      if (chunk < kTile) chunk = kTile;
      hipLaunchKernelGGL(k_nn3_arg_list, dim3((unsigned)qb, (unsigned)cdiv64(m, chunk)), dim3(256), 0, st, a, list2, counts + 1, b,
                         m, (int)chunk, best);
    } else {
      hipLaunchKernelGGL(k_nn3_grid_wave, dim3((unsigned)cdiv64(n, 4)), dim3(256), 0, st, a, b, m, g1, sg, list1, counts, best,
                         (int*)nullptr, (int*)nullptr);
    }
  } else {
    APR_CHECK_ARG(nb == 1, "apr_nn3_batch: cell = 0 (brute force) searches one cloud at a time");
    APR_HIP(hipMemsetAsync(out_packed, 0xFF, (size_t)n * 8, st));
    const int64_t qb = cdiv64(n, 256);
    int64_t want = cdiv64(2048, qb);
    int64_t chunk = cdiv64(cdiv64(m, want), kTile) * kTile;
    if (chunk < kTile) chunk = kTile;
    hipLaunchKernelGGL(k_nn3_arg, dim3((unsigned)qb, (unsigned)cdiv64(m, chunk)), dim3(256), 0, st, a, n, b, m, (int)chunk, best);
  }
  if (sums_dev) hipLaunchKernelGGL(k_sum_packed_seg, dim3((unsigned)nb), dim3(1024), 0, st, (const unsigned long long*)best, sg, sums_dev);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API int apr_nn3(const float* a, int64_t n, const float* b, int64_t m, float cell, uint64_t* out_packed, double* sum_dev,
                    void* scratch, size_t scratch_bytes, void* stream) {
  APR_CHECK_ARG(n > 0 && m > 0 && m < (1ll << 31) - 1 && n < (1ll << 31), "apr_nn3: empty or oversized cloud");
  const int64_t ao[2] = {0, n}, bo[2] = {0, m};
  return nn3_impl(a, n, ao, b, m, bo, 1, cell, out_packed, sum_dev, nullptr, scratch, scratch_bytes, (hipStream_t)stream);
}

APR_API int apr_nn3_batch(const float* a, const int64_t* a_offsets_host, const float* b, const int64_t* b_offsets_host, int32_t nb,
                          float cell, uint64_t* out_packed, double* sums_dev, const double* sum_scale_host, void* scratch,
                          size_t scratch_bytes, void* stream) {
  APR_CHECK_ARG(a && b && a_offsets_host && b_offsets_host && nb >= 1 && nb <= kNn3MaxSeg, "apr_nn3_batch: 1 .. %d clouds",
                kNn3MaxSeg);
  APR_CHECK_ARG(a_offsets_host[0] == 0 && b_offsets_host[0] == 0, "apr_nn3_batch: offsets start at 0");
  for (int i = 0; i < nb; ++i)
    APR_CHECK_ARG(a_offsets_host[i + 1] > a_offsets_host[i] && b_offsets_host[i + 1] > b_offsets_host[i],
                  "apr_nn3_batch: empty cloud %d", i);
  const int64_t n = a_offsets_host[nb], m = b_offsets_host[nb];
  APR_CHECK_ARG(m < (1ll << 31) - 1 && n < (1ll << 31), "apr_nn3_batch: oversized clouds");
  APR_CHECK_ARG(cell > 0.f || nb == 1, "apr_nn3_batch: several clouds need the grid (cell > 0)");
  return nn3_impl(a, n, a_offsets_host, b, m, b_offsets_host, nb, cell, out_packed, sums_dev, sum_scale_host, scratch,
                  scratch_bytes, (hipStream_t)stream);
}
