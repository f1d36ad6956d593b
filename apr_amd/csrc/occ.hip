// The encoder's first convolution when the input feature is the constant 1 (FCGF's voxel features ARE ones:
// FCGF_APR/lib/complement_data_loader.py:805-812 `feats = np.ones((len(coords), 1))`, model/resunet.py:57 conv1 with
// in_channels = 1, kernel 5 or 7): out[j] = sum over the offsets k whose neighbour EXISTS of W[k, 0, :].  The values
// gathered through a kernel map are all 1, so the map's row indices are never needed -- only the occupancy of the
// ks^3 cells around every voxel.  The generic path builds a [n, ks^3] int32 neighbour table for it (189 k voxels x 125
// offsets: 62 hash probes per voxel, 94 MB written, then read back by k_spconv_smallcin -- 190 us per 12-frame step,
// the most expensive index structure of the encoder, used by this one layer); here the voxels are scattered into a
// dense bitmap over the batch's bounding box (1 bit per cell, x along the bits of a word: 12 KITTI frames = 17 MB,
// cleared by one memset) and a voxel reads its 5 x-neighbours of a (dy, dz) row with ONE 8-byte load: 25 loads instead
// of 62 probes, no table.
// Accumulation runs over k ascending with acc += W[k] (== fmaf(1, W[k], acc) of k_spconv_smallcin) and the epilogue is
// the same expression: the same bits as the generic path on all-ones features.
#include <climits>

#include "common.h"

namespace {

struct OccGrid {
  int minx, miny, minz;      // cell (x, y, z) lives at (x - minx, y - miny, z - minz); the box is padded by ks / 2
  int dy, dz, wx;            // rows per slab, slabs per frame, 32-bit words per row (one spare word at the end)
  int nx, nb;                // cells along x (padding included), frames
};

// The box is host-supplied (CoordinateManager.set_bbox is public): a voxel outside it must neither write nor read past the
// bitmap.  True if the voxel itself (not its padded neighbourhood) lies inside the unpadded box of `h`-padded grid g.
__device__ inline bool occ_inside(const OccGrid& g, const int4& c, int h) {
  const int x = c.y - g.minx, y = c.z - g.miny, z = c.w - g.minz;
  return c.x >= 0 && c.x < g.nb && x >= h && x < g.nx - h && y >= h && y < g.dy - h && z >= h && z < g.dz - h;
}

__device__ inline int64_t occ_word(const OccGrid& g, int b, int x, int y, int z) {
  return (((int64_t)b * g.dz + (z - g.minz)) * g.dy + (y - g.miny)) * g.wx + ((x - g.minx) >> 5);
}

__global__ void k_bbox_init(int* __restrict__ bbox) {
  if (threadIdx.x < 3) bbox[threadIdx.x] = INT_MAX;
  else if (threadIdx.x < 7) bbox[threadIdx.x] = INT_MIN;
  else if (threadIdx.x == 7) bbox[7] = 0;
}

// bbox[0..2] = min x, y, z; bbox[3..5] = max x, y, z; bbox[6] = max batch index.
// Few, large workgroups and ONE set of 7 atomics per workgroup: the 7 words share a cache line, and atomics on one line
// serialise at ~10 ns each (a first version with 7 atomics per WAVE of 1024 x 4 waves took 320 us on 1.4 M rows; the
// read itself is 23 MB).
constexpr int kBboxThreads = 1024, kBboxBlocks = 128;
__global__ __launch_bounds__(kBboxThreads) void k_bbox(const int4* __restrict__ coords, int64_t n, int* __restrict__ bbox) {
  __shared__ int s_red[kBboxThreads / 64][8];
  int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[4] = {INT_MIN, INT_MIN, INT_MIN, INT_MIN};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int4 c = coords[i];
    lo[0] = min(lo[0], c.y); lo[1] = min(lo[1], c.z); lo[2] = min(lo[2], c.w);
    hi[0] = max(hi[0], c.y); hi[1] = max(hi[1], c.z); hi[2] = max(hi[2], c.w); hi[3] = max(hi[3], c.x);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
#pragma unroll
    for (int a = 0; a < 3; ++a) lo[a] = min(lo[a], __shfl_xor(lo[a], d));
#pragma unroll
    for (int a = 0; a < 4; ++a) hi[a] = max(hi[a], __shfl_xor(hi[a], d));
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) s_red[wave][a] = lo[a];
#pragma unroll
    for (int a = 0; a < 4; ++a) s_red[wave][3 + a] = hi[a];
  }
  __syncthreads();
  if (threadIdx.x < 7) {
    const int a = threadIdx.x;
    int v = s_red[0][a];
    for (int w = 1; w < kBboxThreads / 64; ++w) v = a < 3 ? min(v, s_red[w][a]) : max(v, s_red[w][a]);
    if (a < 3) atomicMin(&bbox[a], v);
    else atomicMax(&bbox[a], v);
  }
}

__global__ void k_occ_set(const int4* __restrict__ coords, int n, OccGrid g, int h, unsigned* __restrict__ bm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int4 c = coords[i];
  if (!occ_inside(g, c, h)) return;        // k_occ_conv poisons this voxel's output row
  atomicOr(&bm[occ_word(g, c.x, c.y, c.z, c.w)], 1u << ((c.y - g.minx) & 31));
}

// 64 voxels per workgroup, 4 threads per voxel.  Phase 1: the voxel's ks^2 (dy, dz) rows are dealt over its 4 threads;
// a row's ks bits (dx = -h .. h) are bits sh .. sh + ks - 1 of the 64-bit window that starts at the word of x - h, and
// land at offset index ks * ((dy + h) + ks * (dz + h)) of the voxel's occupancy mask in LDS.  Phase 2: thread (voxel,
// channel group of 8) walks the set bits in ascending k.
template <int KS>
__global__ __launch_bounds__(256) void k_occ_conv(const int4* __restrict__ coords, int n, OccGrid g,
                                                  const unsigned* __restrict__ bm, const float* __restrict__ w,
                                                  int cout, const float* __restrict__ scale,
                                                  const float* __restrict__ shift, const float* __restrict__ residual,
                                                  int64_t ldr, int relu, float* __restrict__ out, int64_t ldo) {
  constexpr int H = KS / 2, K = KS * KS * KS, MW = (K + 31) / 32;
  __shared__ unsigned s_mask[64][MW + 1];
  const int tid = threadIdx.x;
  const int r = tid >> 2, cg = tid & 3;
  const int row = blockIdx.x * 64 + r;
  for (int t = tid; t < 64 * (MW + 1); t += 256) (&s_mask[0][0])[t] = 0u;
  __syncthreads();
  bool inside = false;
  if (row < n) {
    const int4 c = coords[row];
    inside = occ_inside(g, c, H);
  }
  if (inside) {
    const int4 c = coords[row];
    const int x0 = c.y - H - g.minx;                 // >= 0: the box is padded by H
    const int sh = x0 & 31;
    const unsigned* base = bm + occ_word(g, c.x, c.y - H, c.z, c.w);
    for (int p = cg; p < KS * KS; p += 4) {
      const int oy = p % KS - H, oz = p / KS - H;
      const unsigned* q = base + ((int64_t)oz * g.dy + oy) * g.wx;
      const unsigned long long win = (unsigned long long)q[0] | ((unsigned long long)q[1] << 32);
      const unsigned bits = (unsigned)(win >> sh) & ((1u << KS) - 1u);
      if (bits) {
        const int k0 = KS * p;
        atomicOr(&s_mask[r][k0 >> 5], bits << (k0 & 31));
        if ((k0 & 31) + KS > 32) atomicOr(&s_mask[r][(k0 >> 5) + 1], bits >> (32 - (k0 & 31)));
      }
    }
  }
  __syncthreads();
  if (row >= n) return;
  if (!inside) {       // voxel outside the caller's box: a NaN row (loud downstream), never an out-of-bounds access
    for (int c = cg; c < cout; c += 4) out[(int64_t)row * ldo + c] = __int_as_float(0x7fc00000);
    return;
  }
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  for (int c0 = cg * 8; c0 < cout; c0 += 32) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mw = 0; mw < MW; ++mw) {
      unsigned m = s_mask[r][mw];
      while (m) {
        const int k = mw * 32 + __ffs((int)m) - 1;
        m &= m - 1;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + (int64_t)k * cout + c0);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(w + (int64_t)k * cout + c0 + 4);
        acc[0] += w0[0]; acc[1] += w0[1]; acc[2] += w0[2]; acc[3] += w0[3];
        acc[4] += w1[0]; acc[5] += w1[1]; acc[6] += w1[2]; acc[7] += w1[3];
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = c0 + i;
      float v = acc[i] * (scale ? scale[c] : 1.f) + (shift ? shift[c] : 0.f);
      if (residual) v += residual[(int64_t)row * ldr + c];
      if (relu) v = fmaxf(v, 0.f);
      out[(int64_t)row * ldo + c] = v;
    }
  }
}

struct OccLayout {
  OccGrid g;
  int64_t words;
  bool ok;
};

OccLayout occ_layout(const int32_t* bbox, int32_t ks) {
  OccLayout L;
  L.ok = false;
  L.words = 0;
  const int h = ks / 2;
  for (int a = 0; a < 3; ++a)
    if (bbox[3 + a] < bbox[a]) return L;
  if (bbox[6] < 0) return L;
  const int64_t dx = (int64_t)bbox[3] - bbox[0] + 1 + 2 * h, dy = (int64_t)bbox[4] - bbox[1] + 1 + 2 * h,
                dz = (int64_t)bbox[5] - bbox[2] + 1 + 2 * h;
  const int64_t wx = (dx + 31) / 32 + 1;
  const int64_t words = ((int64_t)bbox[6] + 1) * dz * dy * wx;
  if (dx > (1 << 24) || dy > (1 << 24) || dz > (1 << 24) || words > (1ll << 29)) return L;      // 2 GB of bitmap at most
  L.g.minx = bbox[0] - h;
  L.g.miny = bbox[1] - h;
  L.g.minz = bbox[2] - h;
  L.g.dy = (int)dy;
  L.g.dz = (int)dz;
  L.g.wx = (int)wx;
  L.g.nx = (int)dx;
  L.g.nb = bbox[6] + 1;
  L.words = words;
  L.ok = true;
  return L;
}

// Kernel map with the occupancy bitmap as a pre-filter: 72 % of a LiDAR map's 27 probes per voxel hit an EMPTY cell, and an
// empty cell costs the hash table its most expensive probe (a miss walks to the first free slot: ~1.5 random 128-B lines).
// PMC, 12 frames per step: k_kernel_map fetched 558 MB per step from the fabric -- the third largest consumer of the whole
// pipeline, for an index build.  The bitmap of the batch (built for conv1, 17 MB, L2 / Infinity-Cache resident, neighbouring
// voxels share its lines) answers "is there a voxel" with one 4-byte load; only the occupied cells go to the table.
__global__ void k_kernel_map_occ(const int4* __restrict__ out_coords, int64_t n_out, const int* __restrict__ n_out_dev,
                                 const unsigned long long* __restrict__ keys, const int* __restrict__ vals, uint32_t mask,
                                 int ks, int scale, OccGrid g, int pad, const unsigned* __restrict__ bm, int* __restrict__ nbr) {
  const int K = ks * ks * ks, h = ks / 2;
  int64_t total = n_out * K;
  if (n_out_dev) total = (int64_t)min((long long)n_out, (long long)*n_out_dev) * K;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = t / K;
    const int o = (int)(t - j * K);
    const int ox = o % ks - h, oy = (o / ks) % ks - h, oz = o / (ks * ks) - h;
    const int4 c = out_coords[j];
    const int x = c.y + ox * scale, y = c.z + oy * scale, z = c.w + oz * scale;
    int r = -1;
    if (apr_key_in_range(c.x, x, y, z)) {
      // inside the (unpadded) box every voxel has its bit set, so the bit decides; outside it -- the rim a strided map's
      // floored outputs reach, or voxels a wrong host-supplied box missed (they set no bit) -- the table does
      const int4 cell = make_int4(c.x, x, y, z);
      const bool maybe = !occ_inside(g, cell, pad) || ((bm[occ_word(g, c.x, x, y, z)] >> ((x - g.minx) & 31)) & 1u);
      if (maybe) r = apr_table_lookup(keys, vals, mask, apr_pack_key(c.x, x, y, z));
    }
    nbr[t] = r;
  }
}

}  // namespace

// apr_kernel_map with the occupancy bitmap that apr_occ_conv left in `occ_scratch` (same bbox_host and bitmap_kernel_size as
// that call, same stream or ordered behind it) as a pre-filter for the probes.  The INPUT map (in_keys / in_vals) must be the
// map the bitmap was built from -- the stride-1 map -- and `scale` its tensor stride (1); out_coords may be that map's own
// rows (same-level map) or a coarser map's (strided map).  Same table out as apr_kernel_map.
APR_API int apr_kernel_map_occ(const int32_t* out_coords, int64_t n_out, const int32_t* n_out_dev, const uint64_t* in_keys,
                               const int32_t* in_vals, int64_t cap, int32_t kernel_size, int32_t scale,
                               const int32_t* bbox_host, int32_t bitmap_kernel_size, const void* occ_scratch, int32_t* nbr,
                               void* stream) {
  APR_CHECK_ARG(kernel_size >= 1 && (kernel_size & 1) && kernel_size <= 7, "apr_kernel_map_occ: kernel_size=%d must be odd and <= 7",
                kernel_size);
  APR_CHECK_ARG((cap & (cap - 1)) == 0 && cap > 0 && n_out >= 0 && bbox_host && occ_scratch && nbr,
                "apr_kernel_map_occ: bad arguments");
  if (n_out == 0) return APR_OK;
  const OccLayout L = occ_layout(bbox_host, bitmap_kernel_size);
  APR_CHECK_ARG(L.ok, "apr_kernel_map_occ: empty or oversized bounding box");
  const unsigned* bm = (const unsigned*)(((uintptr_t)occ_scratch + 255) & ~(uintptr_t)255);
  const int K = kernel_size * kernel_size * kernel_size;
  int64_t nblk = cdiv64(n_out * K, 256);
  if (nblk > 65536) nblk = 65536;
  hipLaunchKernelGGL(k_kernel_map_occ, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (const int4*)out_coords, n_out,
                     n_out_dev, (const unsigned long long*)in_keys, in_vals, (uint32_t)(cap - 1), kernel_size, scale, L.g,
                     bitmap_kernel_size / 2, bm, nbr);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

// bbox_dev int32[8] <- {min x, y, z, max x, y, z, max batch index, 0} of coords int32[n, 4] (batch, x, y, z)
APR_API int apr_coords_bbox(const int32_t* coords, int64_t n, int32_t* bbox_dev, void* stream) {
  APR_CHECK_ARG(n >= 0 && bbox_dev && (n == 0 || coords), "apr_coords_bbox: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_bbox_init, dim3(1), dim3(64), 0, st, bbox_dev);
  if (n > 0) {
    int64_t nblk = cdiv64(n, kBboxThreads * 4);
    if (nblk > kBboxBlocks) nblk = kBboxBlocks;
    hipLaunchKernelGGL(k_bbox, dim3((unsigned)nblk), dim3(kBboxThreads), 0, st, (const int4*)coords, n, bbox_dev);
  }
  APR_LAUNCH_CHECK();
  return APR_OK;
}

// Bytes of the occupancy bitmap for the box bbox_host (the 8 ints of apr_coords_bbox, VOXEL units) and an odd kernel
// size; 0 = the box is empty or unreasonably large (the caller takes the kernel-map path).
APR_API size_t apr_occ_conv_scratch_bytes(const int32_t* bbox_host, int32_t kernel_size) {
  if (!bbox_host || kernel_size < 1 || !(kernel_size & 1)) return 0;
  const OccLayout L = occ_layout(bbox_host, kernel_size);
  return L.ok ? (size_t)L.words * 4 + 256 : 0;
}

// 1 if the occupancy form pays for n voxels in this box: the bitmap (cleared and probed per call) must stay small
// against the structure it replaces -- the [n, ks^3] int32 kernel map: a KITTI batch needs ~90 B of bitmap per voxel;
// one outlier voxel or two far-apart frames can blow the box up to gigabytes, and then the kernel-map path is the
// cheaper one.  Cap: 1 KB of bitmap per voxel (+ 4 MB).
APR_API int apr_occ_conv_pays(const int32_t* bbox_host, int32_t kernel_size, int64_t n) {
  const size_t b = apr_occ_conv_scratch_bytes(bbox_host, kernel_size);
  return b > 0 && n > 0 && b <= (size_t)n * 1024 + ((size_t)4 << 20);
}

// out[j, :] = act((sum_{k : cell(coords[j] + offset_k) occupied} w[k, :]) * scale + shift (+ residual[j, :])):
// a stride-1, dilation-1, ks^3 sparse convolution of the constant-1 feature over the voxels `coords` (unique rows;
// offsets x-fastest as apr_kernel_map), w f32[ks^3, cout] (the reference's [K, 1, cout] kernel), cout % 8 == 0,
// ks in {3, 5, 7}.  Every voxel must lie inside bbox_host (apr_coords_bbox of the same rows or a superset, in VOXEL
// units, batch index <= bbox_host[6]); a voxel outside it is not counted as anybody's neighbour and its own output row is
// NaN (checked per voxel: no access leaves the bitmap whatever the box).
APR_API int apr_occ_conv(const int32_t* coords, int64_t n, const int32_t* bbox_host, int32_t kernel_size,
                         const float* w, int32_t cout, const float* scale, const float* shift, const float* residual,
                         int64_t ldr, int32_t relu, float* out, int64_t ldo, void* scratch, size_t scratch_bytes,
                         void* stream) {
  APR_CHECK_ARG(n >= 0 && n < (1ll << 31) && bbox_host && w && out && scratch, "apr_occ_conv: bad arguments");
  APR_CHECK_ARG(kernel_size == 3 || kernel_size == 5 || kernel_size == 7, "apr_occ_conv: kernel_size %d, supported: 3, 5, 7",
                kernel_size);
  APR_CHECK_ARG(cout > 0 && cout % 8 == 0 && ldo >= cout && (!residual || ldr >= cout), "apr_occ_conv: cout %% 8 != 0 or short rows");
  APR_CHECK_ARG((((uintptr_t)w) & 15) == 0, "apr_occ_conv: weights must be 16-byte aligned");
  if (n == 0) return APR_OK;
  const OccLayout L = occ_layout(bbox_host, kernel_size);
  APR_CHECK_ARG(L.ok, "apr_occ_conv: empty or oversized bounding box");
  APR_CHECK_ARG(scratch_bytes >= (size_t)L.words * 4 + 256, "apr_occ_conv: scratch too small");
  hipStream_t st = (hipStream_t)stream;
  unsigned* bm = (unsigned*)(((uintptr_t)scratch + 255) & ~(uintptr_t)255);
  if (int rcf = apr_internal_fill(bm, 0, (size_t)L.words * 4, st)) return rcf;
  hipLaunchKernelGGL(k_occ_set, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, (const int4*)coords, (int)n, L.g,
                     kernel_size / 2, bm);
  const dim3 grid((unsigned)cdiv64(n, 64));
  if (kernel_size == 3)
    hipLaunchKernelGGL(k_occ_conv<3>, grid, dim3(256), 0, st, (const int4*)coords, (int)n, L.g, bm, w, cout, scale, shift,
                       residual, ldr, relu, out, ldo);
  else if (kernel_size == 5)
    hipLaunchKernelGGL(k_occ_conv<5>, grid, dim3(256), 0, st, (const int4*)coords, (int)n, L.g, bm, w, cout, scale, shift,
                       residual, ldr, relu, out, ldo);
  else
    hipLaunchKernelGGL(k_occ_conv<7>, grid, dim3(256), 0, st, (const int4*)coords, (int)n, L.g, bm, w, cout, scale, shift,
                       residual, ldr, relu, out, ldo);
  APR_LAUNCH_CHECK();
  return APR_OK;
}
