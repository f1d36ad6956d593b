// Voxel hashing, coordinate maps and kernel maps.
//
// HBM layout: an open-addressing table of `cap` uint64 keys + `cap` int32 values
// (cap = power of two >= 2n, ~24 B per voxel), coordinates as int32 [n,4] rows
// (one 16-B load per voxel) and the dense neighbour table nbr int32 [n_out,K].
// All kernels are HBM/L2-latency bound integer work: one thread per row (or per
// (row,offset) probe), 16-B coalesced row loads, wave64 ballot + popcount for
// the compaction prefix sums.
#include "common.h"

namespace {

constexpr int kBlock = 256;

__global__ void k_voxelize(const float* __restrict__ xyz, int64_t n, float vs, int batch,
                           int4* __restrict__ coords) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // fp32 division, correctly rounded: identical to numpy/torch float32 `xyz / vs`
  float x = xyz[3 * i + 0] / vs, y = xyz[3 * i + 1] / vs, z = xyz[3 * i + 2] / vs;
  coords[i] = make_int4(batch, (int)floorf(x), (int)floorf(y), (int)floorf(z));
}

// the same for the concatenated frames of a batch: the batch index of point i is the segment [offsets[b], offsets[b+1])
// it falls into (<= a dozen segments: a short scan)
__global__ void k_voxelize_segments(const float* __restrict__ xyz, int64_t n, float vs,
                                    const long long* __restrict__ offsets, int nseg, int4* __restrict__ coords) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int b = 0;
  while (b + 1 < nseg && i >= offsets[b + 1]) ++b;
  float x = xyz[3 * i + 0] / vs, y = xyz[3 * i + 1] / vs, z = xyz[3 * i + 2] / vs;
  coords[i] = make_int4(b, (int)floorf(x), (int)floorf(y), (int)floorf(z));
}

__device__ inline int4 load_coord(const int4* coords, int64_t i, int floor_to) {
  int4 c = coords[i];
  if (floor_to > 0) {
    c.y = apr_floor_to(c.y, floor_to);
    c.z = apr_floor_to(c.z, floor_to);
    c.w = apr_floor_to(c.w, floor_to);
  }
  return c;
}

// Insert every row; the slot keeps the smallest row index of its key.
// Consecutive rows of a LiDAR scan mostly fall into the same voxel (a beam sweeps ~8 points through a 0.3 m voxel at
// 10 m range): inside a wave only the FIRST lane of every run of equal keys probes the table and issues the atomics
// (it also holds the smallest row index of the run); the other lanes of the run receive the slot by shuffle.
__global__ void k_insert(const int4* __restrict__ coords, int64_t n, const int* __restrict__ n_dev,
                         int floor_to, unsigned long long* __restrict__ keys, int* __restrict__ vals,
                         uint32_t mask, int* __restrict__ slot_of, int* __restrict__ status) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  if (n_dev) n = min((long long)n, (long long)*n_dev);
  const bool live = i < n;
  unsigned long long key = APR_KEY_EMPTY;     // never a real key: dead and out-of-range lanes start their own run
  bool ok = false;
  if (live) {
    const int4 c = load_coord(coords, i, floor_to);
    ok = apr_key_in_range(c.x, c.y, c.z, c.w);
    if (ok) key = apr_pack_key(c.x, c.y, c.z, c.w);
    else *status = 1;
  }
  const unsigned long long prev = __shfl_up(key, 1);
  const bool head = lane == 0 || key != prev || !ok;
  const unsigned long long heads = __ballot(head);
  const int leader = 63 - __clzll((long long)(heads & ((2ull << lane) - 1ull)));   // first lane of this lane's run
  int slot = -1;
  if (head && ok) {
    uint32_t sl = apr_hash_u64(key) & mask;
    for (uint32_t probe = 0; probe <= mask; ++probe) {
      unsigned long long p = keys[sl];
      if (p != key) {
        if (p != APR_KEY_EMPTY) {
          sl = (sl + 1) & mask;
          continue;
        }
        p = atomicCAS(&keys[sl], APR_KEY_EMPTY, key);
        if (p != APR_KEY_EMPTY && p != key) {
          sl = (sl + 1) & mask;
          continue;
        }
      }
      atomicMin(&vals[sl], (int)i);
      slot = (int)sl;
      break;
    }
    if (slot < 0) *status = 2;  // table full (cannot happen with cap >= 2n)
  }
  slot = __shfl(slot, leader);
  if (live) slot_of[i] = ok ? slot : -1;
}

// flag[i] = row i is the first occurrence of its key; per-block counts.
__global__ void k_flag_count(const int* __restrict__ vals, const int* __restrict__ slot_of,
                             int64_t n, const int* __restrict__ n_dev, uint8_t* __restrict__ flags,
                             int* __restrict__ block_counts) {
  __shared__ int wave_cnt[kBlock / 64];
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n_dev) n = min((long long)n, (long long)*n_dev);
  bool f = false;
  if (i < n) {
    int s = slot_of[i];
    f = (s >= 0) && (vals[s] == (int)i);
    flags[i] = f;
  }
  unsigned long long b = __ballot(f);
  if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = __popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
    for (int w = 0; w < kBlock / 64; ++w) s += wave_cnt[w];
    block_counts[blockIdx.x] = s;
  }
}

// Single-workgroup exclusive scan of the block counts; total -> *n_out.
__global__ void k_scan_blocks(const int* __restrict__ counts, int nblk, int* __restrict__ offsets,
                              int* __restrict__ n_out) {
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
  if (threadIdx.x == 0) *n_out = carry_s;
}

// Compact the first-occurrence rows in row order and re-point the table at output rows.
__global__ void k_compact(const int4* __restrict__ coords, int64_t n, const int* __restrict__ n_dev,
                          int floor_to, const uint8_t* __restrict__ flags,
                          const int* __restrict__ block_offsets, const int* __restrict__ slot_of,
                          int* __restrict__ vals, int4* __restrict__ out_coords,
                          long long* __restrict__ out_first) {
  __shared__ int wave_cnt[kBlock / 64];
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n_dev) n = min((long long)n, (long long)*n_dev);
  bool f = (i < n) && flags[i];
  unsigned long long b = __ballot(f);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_cnt[wave] = __popcll(b);
  __syncthreads();
  if (!f) return;
  int pos = block_offsets[blockIdx.x] + __popcll(b & ((1ull << lane) - 1ull));
  for (int w = 0; w < wave; ++w) pos += wave_cnt[w];
  out_coords[pos] = load_coord(coords, i, floor_to);
  if (out_first) out_first[pos] = (long long)i;
  vals[slot_of[i]] = pos;
}

__device__ inline int table_lookup(const unsigned long long* __restrict__ keys,
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

// keys <- ~0 (empty), vals <- INT-ish max (atomicMin target), status/n_out <- 0: one launch instead of four memsets
__global__ void k_table_init(unsigned long long* __restrict__ keys, int* __restrict__ vals, int64_t cap,
                             int* __restrict__ status, int* __restrict__ n_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cap) {
    keys[i] = ~0ull;
    vals[i] = 0x7F7F7F7F;
  }
  if (i == 0) {
    *status = 0;
    if (n_out) *n_out = 0;
  }
}

// counts[b] = number of map rows whose first input row lies in [offsets[b], offsets[b+1]): `first` is ascending
// (first-occurrence order), so two binary searches per segment over its valid prefix [0, *n_dev)
__global__ void k_segment_counts(const long long* __restrict__ first, const int* __restrict__ n_dev,
                                 const long long* __restrict__ offsets, int nseg, int* __restrict__ counts) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nseg) return;
  const int n = *n_dev;
  auto lower = [&](long long v) {
    int lo = 0, hi = n;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (first[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  counts[b] = lower(offsets[b + 1]) - lower(offsets[b]);
}

// One thread per (out row, offset) probe; consecutive lanes -> consecutive offsets of
// one row, so the nbr store is fully coalesced and the 16-B coordinate load is a
// broadcast within the wave.
__global__ void k_kernel_map(const int4* __restrict__ out_coords, int64_t n_out,
                             const int* __restrict__ n_out_dev,
                             const unsigned long long* __restrict__ keys,
                             const int* __restrict__ vals, uint32_t mask, int ks, int scale,
                             int* __restrict__ nbr) {
  const int K = ks * ks * ks, h = ks / 2;
  int64_t total = n_out * K;
  if (n_out_dev) total = (int64_t)min((long long)n_out, (long long)*n_out_dev) * K;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    int64_t j = t / K;
    int o = (int)(t - j * K);
    int ox = o % ks - h, oy = (o / ks) % ks - h, oz = o / (ks * ks) - h;
    int4 c = out_coords[j];
    int x = c.y + ox * scale, y = c.z + oy * scale, z = c.w + oz * scale;
    int r = -1;
    if (apr_key_in_range(c.x, x, y, z)) r = table_lookup(keys, vals, mask, apr_pack_key(c.x, x, y, z));
    nbr[t] = r;
  }
}

// Same-level map (the input map IS the output map, scale = its tensor stride): the relation is symmetric,
// nbr[j][o] = i  <=>  nbr[i][K-1-o] = j, and the centre offset is the voxel itself.  Only the offsets below the
// centre are probed; a hit also stores its mirror entry (unique writer: row i never probes offset K-1-o), the misses of
// the upper half were pre-filled with -1 (k_fill_upper).  Half the hash probes of k_kernel_map.
__global__ void k_kernel_map_sym(const int4* __restrict__ coords, int64_t n, const unsigned long long* __restrict__ keys,
                                 const int* __restrict__ vals, uint32_t mask, int ks, int scale,
                                 int* __restrict__ nbr) {
  const int K = ks * ks * ks, h = ks / 2, half = K / 2;   // offsets 0 .. half-1 probed, `half` = the centre
  const int64_t total = n * (half + 1);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = t / (half + 1);
    const int o = (int)(t - j * (half + 1));
    if (o == half) {
      nbr[j * K + o] = (int)j;
      continue;
    }
    const int ox = o % ks - h, oy = (o / ks) % ks - h, oz = o / (ks * ks) - h;
    const int4 c = coords[j];
    const int x = c.y + ox * scale, y = c.z + oy * scale, z = c.w + oz * scale;
    int r = -1;
    if (apr_key_in_range(c.x, x, y, z)) r = table_lookup(keys, vals, mask, apr_pack_key(c.x, x, y, z));
    nbr[j * K + o] = r;
    if (r >= 0) nbr[(int64_t)r * K + (K - 1 - o)] = (int)j;
  }
}

__global__ void k_fill_upper(int* __restrict__ nbr, int64_t n, int K, int half) {
  const int64_t total = n * half;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = t / half;
    nbr[j * K + half + 1 + (int)(t - j * half)] = -1;
  }
}

// nbrT[i][k] = j  for every entry nbr[j][k] = i >= 0 (each (i, k) has at most one source: plain stores)
__global__ void k_map_transpose(const int* __restrict__ nbr, int64_t total, int K, int* __restrict__ nbr_t) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int i = nbr[t];
  if (i >= 0) nbr_t[(int64_t)i * K + (int)(t % K)] = (int)(t / K);
}

// ---- a batch of frames without a concatenated copy: the frame pointers and start offsets travel in the kernel arguments ----
constexpr int kMaxFrames = APR_MAX_FRAMES;
struct FrameTable {
  const float* ptr[kMaxFrames];
  long long off[kMaxFrames + 1];      // off[b] = points before frame b; off[nseg] = total
  int nseg;
};

__device__ inline int frame_of(const FrameTable& ft, long long i) {
  int b = 0;
  while (b + 1 < ft.nseg && i >= ft.off[b + 1]) ++b;
  return b;
}

// coords[i] = (frame of point i, floor(xyz / vs)) over the virtual concatenation of the frames; thread 0 also leaves the
// offsets on the device for the kernels behind it (apr_segment_counts)
__global__ void k_voxelize_frames(FrameTable ft, int64_t n, float vs, int4* __restrict__ coords,
                                  long long* __restrict__ offsets_out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x <= ft.nseg && offsets_out) offsets_out[threadIdx.x] = ft.off[threadIdx.x];
  if (i >= n) return;
  const int b = frame_of(ft, i);
  const float* p = ft.ptr[b] + 3 * (i - ft.off[b]);
  float x = p[0] / vs, y = p[1] / vs, z = p[2] / vs;
  coords[i] = make_int4(b, (int)floorf(x), (int)floorf(y), (int)floorf(z));
}

// pts[r] = the point (of the virtual concatenation) that first[r] names, r < *n_dev
__global__ void k_gather_frame_points(FrameTable ft, const long long* __restrict__ first, const int* __restrict__ n_dev,
                                      float* __restrict__ pts) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= *n_dev) return;
  const long long i = first[r];
  const int b = frame_of(ft, i);
  const float* p = ft.ptr[b] + 3 * (i - ft.off[b]);
  pts[3 * r + 0] = p[0];
  pts[3 * r + 1] = p[1];
  pts[3 * r + 2] = p[2];
}

// Small int32 device arrays -> one contiguous block (the sizes / flags a step fetches in one copy), and a block of words
// cleared on the way (the pair-list counters of the encode that follows)
constexpr int kMaxPack = APR_MAX_PACK;
struct PackTable {
  const int* src[kMaxPack];
  int begin[kMaxPack + 1];
  int nsrc;
};
__global__ void k_pack_i32(PackTable pt, int* __restrict__ dst, int* __restrict__ zero_ptr, long long zero_words) {
  for (int s = 0; s < pt.nsrc; ++s) {
    const int n = pt.begin[s + 1] - pt.begin[s];
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) dst[pt.begin[s] + e] = pt.src[s][e];
  }
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < zero_words; e += (long long)gridDim.x * blockDim.x)
    zero_ptr[e] = 0;
}

}  // namespace

APR_API int64_t apr_hash_capacity(int64_t n) {
  int64_t cap = 1024;
  while (cap < 2 * n) cap <<= 1;
  return cap;
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

APR_API size_t apr_map_scratch_bytes(int64_t n) {
  int64_t nblk = cdiv64(n > 0 ? n : 1, kBlock);
  return align256(n * 4) + align256(n) + 2 * align256(nblk * 4) + 256;
}

APR_API int apr_voxelize(const float* xyz, int64_t n, float voxel_size, int32_t batch,
                         int32_t* coords, void* stream) {
  APR_CHECK_ARG(n >= 0 && voxel_size > 0.f, "apr_voxelize: bad n=%lld or voxel_size", (long long)n);
  APR_CHECK_ARG(batch >= 0 && batch < 1023, "apr_voxelize: batch index %d outside [0,1023)", batch);
  if (n == 0) return APR_OK;
  hipLaunchKernelGGL(k_voxelize, dim3((unsigned)cdiv64(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                     xyz, n, voxel_size, batch, (int4*)coords);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API int apr_map_build(const int32_t* coords_in, int64_t n, const int32_t* n_dev, int32_t floor_to,
                          uint64_t* keys,
                          int32_t* vals, int64_t cap, int32_t* out_coords, int64_t* out_first,
                          int32_t* n_out, int32_t* status, void* scratch, size_t scratch_bytes,
                          void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APR_CHECK_ARG(n >= 0 && n < (1ll << 31), "apr_map_build: n=%lld out of range", (long long)n);
  APR_CHECK_ARG(cap >= 2 * n && (cap & (cap - 1)) == 0, "apr_map_build: cap=%lld must be a power of two >= 2n",
                (long long)cap);
  APR_CHECK_ARG(scratch_bytes >= apr_map_scratch_bytes(n), "apr_map_build: scratch too small");
  APR_CHECK_ARG(floor_to >= 0, "apr_map_build: floor_to < 0");
  hipLaunchKernelGGL(k_table_init, dim3((unsigned)cdiv64(cap, 256)), dim3(256), 0, st, (unsigned long long*)keys, vals,
                     cap, status, n == 0 ? n_out : (int*)nullptr);
  if (n == 0) {
    APR_LAUNCH_CHECK();
    return APR_OK;
  }
  const int nblk = (int)cdiv64(n, kBlock);
  char* p = (char*)scratch;
  int* slot_of = (int*)p;
  p += align256(n * 4);
  uint8_t* flags = (uint8_t*)p;
  p += align256(n);
  int* blk_cnt = (int*)p;
  p += align256((size_t)nblk * 4);
  int* blk_off = (int*)p;
  const uint32_t mask = (uint32_t)(cap - 1);
  hipLaunchKernelGGL(k_insert, dim3(nblk), dim3(kBlock), 0, st, (const int4*)coords_in, n, n_dev,
                     floor_to, (unsigned long long*)keys, vals, mask, slot_of, status);
  hipLaunchKernelGGL(k_flag_count, dim3(nblk), dim3(kBlock), 0, st, vals, slot_of, n, n_dev, flags, blk_cnt);
  hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, st, blk_cnt, nblk, blk_off, n_out);
  hipLaunchKernelGGL(k_compact, dim3(nblk), dim3(kBlock), 0, st, (const int4*)coords_in, n, n_dev, floor_to,
                     flags, blk_off, slot_of, vals, (int4*)out_coords, (long long*)out_first);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

static int map_transpose(const int32_t* nbr, int64_t n_out, int32_t K, int64_t n_in, int32_t* nbr_t, bool prefilled,
                         void* stream) {
  APR_CHECK_ARG(n_out >= 0 && n_in >= 0 && K >= 1 && (nbr || n_out == 0) && (nbr_t || n_in == 0),
                "apr_kernel_map_transpose: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (n_in > 0 && !prefilled) {      // -1 everywhere
    if (int rcf = apr_internal_fill(nbr_t, 0xFF, (size_t)n_in * K * 4, st)) return rcf;
  }
  if (n_out > 0)
    hipLaunchKernelGGL(k_map_transpose, dim3((unsigned)cdiv64(n_out * K, kBlock)), dim3(kBlock), 0, st, nbr,
                       n_out * (int64_t)K, K, nbr_t);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API int apr_kernel_map_transpose(const int32_t* nbr, int64_t n_out, int32_t K, int64_t n_in, int32_t* nbr_t,
                                     void* stream) {
  return map_transpose(nbr, n_out, K, n_in, nbr_t, false, stream);
}

// the same into a table the caller has already filled with -1 (apr_fill_bytes 0xFF): the transposed tables of an encoder's
// three decoder levels then share ONE fill
APR_API int apr_kernel_map_transpose_prefilled(const int32_t* nbr, int64_t n_out, int32_t K, int64_t n_in, int32_t* nbr_t,
                                               void* stream) {
  return map_transpose(nbr, n_out, K, n_in, nbr_t, true, stream);
}

// A fill as a kernel of the library's own (16-byte stores, grid-stride; the unaligned head and tail bytes by the first
// workgroup): the two fills of a step (conv1's occupancy bitmap, the transposed tables' pool) show up as the library's
// kernels in a trace instead of runtime blits.  Throughput-neutral against hipMemsetAsync (2658 vs 2644 pairs/s over
// 4 x 200 steps each; APR_FILL_KERNEL=0 switches back).
static __global__ __launch_bounds__(256) void k_fill_bytes(unsigned char* __restrict__ p, unsigned v32, unsigned long long head,
                                              unsigned long long n16, unsigned long long tail) {
  uint4* const q = (uint4*)(p + head);
  const uint4 v = make_uint4(v32, v32, v32, v32);
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16;
       i += (unsigned long long)gridDim.x * blockDim.x)
    q[i] = v;
  if (blockIdx.x == 0) {
    if (threadIdx.x < head) p[threadIdx.x] = (unsigned char)v32;
    if (threadIdx.x < tail) p[head + n16 * 16 + threadIdx.x] = (unsigned char)v32;
  }
}

int apr_internal_fill(void* ptr, int32_t byte_value, size_t bytes, hipStream_t st) {
  static const int s_kernel = env_int("APR_FILL_KERNEL", 1);
  if (bytes == 0) return APR_OK;
  if (!s_kernel) {
    APR_HIP(hipMemsetAsync(ptr, byte_value, bytes, st));
    return APR_OK;
  }
  const unsigned b = (unsigned)byte_value & 0xFFu;
  const unsigned v32 = b * 0x01010101u;
  size_t head = (16 - ((uintptr_t)ptr & 15)) & 15;
  if (head > bytes) head = bytes;
  const size_t n16 = (bytes - head) / 16, tail = bytes - head - n16 * 16;
  int64_t nblk = cdiv64((int64_t)n16, 256 * 8);
  nblk = nblk < 1 ? 1 : (nblk > 2048 ? 2048 : nblk);
  hipLaunchKernelGGL(k_fill_bytes, dim3((unsigned)nblk), dim3(256), 0, st, (unsigned char*)ptr, v32, (unsigned long long)head,
                     (unsigned long long)n16, (unsigned long long)tail);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API int apr_fill_bytes(void* ptr, int32_t byte_value, size_t bytes, void* stream) {
  APR_CHECK_ARG(ptr || bytes == 0, "apr_fill_bytes: null pointer");
  return apr_internal_fill(ptr, byte_value, bytes, (hipStream_t)stream);
}

APR_API int apr_voxelize_segments(const float* xyz, int64_t n, float voxel_size, const int64_t* offsets, int32_t nseg,
                                  int32_t* coords, void* stream) {
  APR_CHECK_ARG(n >= 0 && voxel_size > 0.f && nseg >= 1 && nseg <= 1023 && offsets, "apr_voxelize_segments: bad arguments");
  if (n == 0) return APR_OK;
  hipLaunchKernelGGL(k_voxelize_segments, dim3((unsigned)cdiv64(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, xyz, n,
                     voxel_size, (const long long*)offsets, nseg, (int4*)coords);
  APR_LAUNCH_CHECK();
  return APR_OK;
}


static int fill_frame_table(FrameTable& ft, const float* const* frames, const int64_t* offsets, int32_t nseg) {
  APR_CHECK_ARG(frames && offsets && nseg >= 1 && nseg <= kMaxFrames, "frame table: 1 <= frames <= %d", kMaxFrames);
  APR_CHECK_ARG(offsets[0] == 0, "frame table: offsets[0] must be 0");
  for (int b = 0; b < nseg; ++b) {
    APR_CHECK_ARG(offsets[b + 1] >= offsets[b] && (frames[b] || offsets[b + 1] == offsets[b]), "frame table: bad frame %d", b);
    ft.ptr[b] = frames[b];
    ft.off[b] = offsets[b];
  }
  ft.off[nseg] = offsets[nseg];
  ft.nseg = nseg;
  return APR_OK;
}

// The voxelisation of a BATCH of frames (FCGF_APR/lib/complement_data_loader.py:788-812 per frame) without first copying
// them into one array: frames_host[b] = device pointer of frame b (f32 [n_b, 3], contiguous), offsets_host[b] = points
// before frame b (offsets_host[nseg] = total).  coords int32 [total, 4] <- (b, floor(xyz / voxel_size)) in frame order;
// offsets_dev (optional, int64 [nseg + 1]) receives the offsets for apr_segment_counts.
APR_API int apr_voxelize_frames(const float* const* frames_host, const int64_t* offsets_host, int32_t nseg, float voxel_size,
                                int32_t* coords, int64_t* offsets_dev, void* stream) {
  FrameTable ft;
  if (int rc = fill_frame_table(ft, frames_host, offsets_host, nseg)) return rc;
  APR_CHECK_ARG(voxel_size > 0.f && coords, "apr_voxelize_frames: bad arguments");
  const int64_t n = offsets_host[nseg];
  hipLaunchKernelGGL(k_voxelize_frames, dim3((unsigned)(n > 0 ? cdiv64(n, kBlock) : 1)), dim3(kBlock), 0, (hipStream_t)stream, ft,
                     n, voxel_size, (int4*)coords, (long long*)offsets_dev);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

// pts f32 [*n_dev, 3] <- the input point behind every row of a map built with want_first over the same frames (first:
// index into their virtual concatenation); n_max = rows allocated (upper bound of *n_dev, sizes the grid).
APR_API int apr_gather_frame_points(const float* const* frames_host, const int64_t* offsets_host, int32_t nseg,
                                    const int64_t* first, const int32_t* n_dev, int64_t n_max, float* pts, void* stream) {
  FrameTable ft;
  if (int rc = fill_frame_table(ft, frames_host, offsets_host, nseg)) return rc;
  APR_CHECK_ARG(first && n_dev && pts && n_max >= 0, "apr_gather_frame_points: bad arguments");
  if (n_max == 0) return APR_OK;
  hipLaunchKernelGGL(k_gather_frame_points, dim3((unsigned)cdiv64(n_max, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, ft,
                     (const long long*)first, n_dev, pts);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

// dst <- srcs_host[0][0 .. counts_host[0]) ++ srcs_host[1][...] ++ ... (device int32 arrays, nsrc <= APR_MAX_PACK), and
// zero_words int32 words at zero_ptr cleared (NULL / 0: nothing): the map sizes, status flags, frame row counts and bounding
// box a step brings to the host travel as ONE block, gathered by ONE launch.
APR_API int apr_pack_i32(const int32_t* const* srcs_host, const int32_t* counts_host, int32_t nsrc, int32_t* dst,
                         int32_t* zero_ptr, int64_t zero_words, void* stream) {
  APR_CHECK_ARG(nsrc >= 0 && nsrc <= kMaxPack && (nsrc == 0 || (srcs_host && counts_host && dst)) && zero_words >= 0 &&
                    (zero_words == 0 || zero_ptr),
                "apr_pack_i32: bad arguments (at most %d sources)", kMaxPack);
  PackTable pt;
  pt.nsrc = nsrc;
  pt.begin[0] = 0;
  int64_t words = 0;
  for (int s = 0; s < nsrc; ++s) {
    APR_CHECK_ARG(counts_host[s] >= 0 && (srcs_host[s] || counts_host[s] == 0), "apr_pack_i32: bad source %d", s);
    pt.src[s] = srcs_host[s];
    words += counts_host[s];
    APR_CHECK_ARG(words < (1 << 30), "apr_pack_i32: too many words");
    pt.begin[s + 1] = (int)words;
  }
  if (nsrc == 0 && zero_words == 0) return APR_OK;
  int64_t nblk = cdiv64(zero_words > words ? zero_words : words, 256 * 8);
  nblk = nblk < 1 ? 1 : (nblk > 64 ? 64 : nblk);
  hipLaunchKernelGGL(k_pack_i32, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, pt, dst, zero_ptr, (long long)zero_words);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API int apr_segment_counts(const int64_t* first, const int32_t* n_dev, const int64_t* offsets, int32_t nseg,
                               int32_t* counts, void* stream) {
  APR_CHECK_ARG(nseg >= 0 && first && n_dev && offsets && counts, "apr_segment_counts: bad arguments");
  if (nseg == 0) return APR_OK;
  hipLaunchKernelGGL(k_segment_counts, dim3((unsigned)cdiv64(nseg, 64)), dim3(64), 0, (hipStream_t)stream,
                     (const long long*)first, n_dev, (const long long*)offsets, nseg, counts);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API int apr_kernel_map_same(const int32_t* coords, int64_t n, const uint64_t* keys, const int32_t* vals, int64_t cap,
                                int32_t kernel_size, int32_t scale, int32_t* nbr, void* stream) {
  APR_CHECK_ARG(kernel_size >= 3 && (kernel_size & 1) && kernel_size <= 7,
                "apr_kernel_map_same: kernel_size=%d must be odd, 3 ... 7", kernel_size);
  APR_CHECK_ARG((cap & (cap - 1)) == 0 && cap > 0, "apr_kernel_map_same: cap must be a power of two");
  APR_CHECK_ARG(n >= 0 && n < (1ll << 31), "apr_kernel_map_same: bad n");
  if (n == 0) return APR_OK;
  hipStream_t st = (hipStream_t)stream;
  const int K = kernel_size * kernel_size * kernel_size, half = K / 2;
  // -1 in the upper half of every row (offsets above the centre): hits are scattered into it by the kernel
  int64_t fblk = cdiv64(n * half, kBlock);
  if (fblk > 16384) fblk = 16384;
  hipLaunchKernelGGL(k_fill_upper, dim3((unsigned)fblk), dim3(kBlock), 0, st, nbr, n, K, half);
  int64_t nblk = cdiv64(n * (half + 1), kBlock);
  if (nblk > 65536) nblk = 65536;
  hipLaunchKernelGGL(k_kernel_map_sym, dim3((unsigned)nblk), dim3(kBlock), 0, st, (const int4*)coords, n,
                     (const unsigned long long*)keys, vals, (uint32_t)(cap - 1), kernel_size, scale, nbr);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API int apr_kernel_map(const int32_t* out_coords, int64_t n_out, const int32_t* n_out_dev,
                           const uint64_t* in_keys, const int32_t* in_vals, int64_t cap,
                           int32_t kernel_size, int32_t scale, int32_t* nbr, void* stream) {
  APR_CHECK_ARG(kernel_size >= 1 && (kernel_size & 1) && kernel_size <= 7,
                "apr_kernel_map: kernel_size=%d must be odd and <= 7", kernel_size);
  APR_CHECK_ARG((cap & (cap - 1)) == 0 && cap > 0, "apr_kernel_map: cap must be a power of two");
  APR_CHECK_ARG(n_out >= 0, "apr_kernel_map: n_out < 0");
  if (n_out == 0) return APR_OK;
  const int K = kernel_size * kernel_size * kernel_size;
  int64_t total = n_out * K;
  int64_t nblk = cdiv64(total, kBlock);
  if (nblk > 65536) nblk = 65536;
  hipLaunchKernelGGL(k_kernel_map, dim3((unsigned)nblk), dim3(kBlock), 0, (hipStream_t)stream,
                     (const int4*)out_coords, n_out, n_out_dev, (const unsigned long long*)in_keys, in_vals,
                     (uint32_t)(cap - 1), kernel_size, scale, nbr);
  APR_LAUNCH_CHECK();
  return APR_OK;
}
