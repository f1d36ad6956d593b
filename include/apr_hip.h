/*
 * apr_hip.h -- C ABI of libapr_hip.so, the MI355X (gfx950) implementation of the
 * APR per-point feature-extraction hot path.
 *
 * Every entry point replaces one operator the reference reaches through a
 * Python binding of third-party or in-tree native code; the citation next to
 * each declaration is the reference interface it stands in for (paths relative
 * to /root/reference).  INTEGRATION.md shows the ctypes stub a maintainer of
 * the reference would add.
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch / C++ types.
 *   - every pointer is a DEVICE pointer (HBM) unless its name ends in `_host`.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and
 *     nothing synchronises unless the doc says so.
 *   - return value: 0 on success, negative APR_E* on error; apr_last_error()
 *     returns a message for the calling thread.
 *   - row-major tensors with an explicit leading dimension (`ld*`, in elements)
 *     so a conv can read from / write into a column slice of a wider buffer
 *     (fused ME.cat).
 */
#ifndef APR_HIP_H
#define APR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define APR_OK 0
#define APR_EINVAL (-1)   /* bad argument / shape */
#define APR_EHIP (-2)     /* HIP runtime error */
#define APR_ETIMEOUT (-3) /* apr_event_wait_timeout: the event was not reached before the deadline */
#define APR_ERANGE (-3)   /* coordinate outside the packed-key range */

const char* apr_last_error(void);
int apr_version(void);
/* Number of HIP devices visible; does not initialise a context. */
int apr_device_count(void);
/* sizeof of the structs of this header, in declaration order (apr_pair_desc, apr_spconv_desc, apr_resunet_layer,
 * apr_resunet_plan, apr_level_map, apr_pyramid, apr_kp_resnet_desc, apr_gcn_layer, apr_gcn_desc) -> out[0 .. n); returns how
 * many there are.  A binding compares them with its own layout at load time. */
int32_t apr_struct_sizes(int32_t* out, int32_t n);
/* Host wait for a HIP event (hipEvent_t) by hipEventQuery + nanosleep(poll_us): no spinning CPU (hipEventSynchronize spins on
 * this stack even for hipEventBlockingSync events) and, called through ctypes, no interpreter lock held while waiting.  The
 * reference's loop blocks in `.cpu()` / `.item()` at the same places (FCGF_APR/scripts/test_apr.py:137-146).
 * While it sleeps the calling thread's timer slack is 1 us (PR_SET_TIMERSLACK; the default 50 us would triple a 25 us poll);
 * the thread's previous value is restored before the call returns.
 * apr_event_wait_timeout: the same with a deadline (microseconds): APR_ETIMEOUT when the event has not been reached by then
 * (a wedged queue), so that a caller can report a hang instead of blocking for ever with the GIL released. */
int apr_event_wait(void* event, int32_t poll_us);
int apr_event_wait_timeout(void* event, int32_t poll_us, int64_t timeout_us);

/* ------------------------------------------------------------------------
 * Voxel hashing / coordinate maps
 *   replaces MinkowskiEngine's coordinate manager as used at
 *   FCGF_APR/scripts/test_apr.py:132-137 (ME.SparseTensor(F, coordinates=C)),
 *   FCGF_APR/lib/complement_data_loader.py:671-674,788-789
 *   (ME.utils.sparse_quantize(xyz / voxel_size, return_index=True)) and the
 *   implicit stride-2 maps of FCGF_APR/model/resunet.py:44,57,70.
 *
 * Key = (batch:10 | x:18 | y:18 | z:18) biased; open-addressing table of
 * `cap` (power of two, >= 2n) uint64 keys + int32 values.
 * ---------------------------------------------------------------------- */

/* Capacity (entries) the table needs for n coordinates. */
int64_t apr_hash_capacity(int64_t n);

/* Scratch bytes apr_map_build needs for n input rows. */
size_t apr_map_scratch_bytes(int64_t n);

/* xyz f32[n,3] -> coords i32[n,4] = (batch, floor(x/vs), floor(y/vs), floor(z/vs)),
 * fp32 division exactly as `xyz / voxel_size` on a float32 array. */
int apr_voxelize(const float* xyz, int64_t n, float voxel_size, int32_t batch,
                 int32_t* coords, void* stream);

/* Build one coordinate map.
 *   n_dev      nullable device row count: only min(n, *n_dev) rows are read, so a
 *              pyramid of maps can be chained without a host sync per level.
 *   coords_in  i32[n,4]; if `floor_to` > 0 the x,y,z columns are first floored to
 *              a multiple of `floor_to` (stride-2 map of a map at stride floor_to/2).
 *   keys/vals  hash table storage (cap entries each); on return the table maps
 *              key -> OUTPUT row index.
 *   out_coords i32[<=n,4]  unique coordinates in first-occurrence order.
 *   out_first  i64[<=n]    input row of each output row (nullable)
 *                          (== `index` of sparse_quantize(return_index=True)).
 *   n_out      i32[1]      device counter; read it back after a stream sync.
 *   status     i32[1]      device flag, set non-zero on key overflow (APR_ERANGE).
 */
int apr_map_build(const int32_t* coords_in, int64_t n, const int32_t* n_dev, int32_t floor_to,
                  uint64_t* keys, int32_t* vals, int64_t cap,
                  int32_t* out_coords, int64_t* out_first, int32_t* n_out,
                  int32_t* status, void* scratch, size_t scratch_bytes, void* stream);

/* Transpose of a kernel map: nbr i32[n_out, K] (rows of an n_in-row map, -1 = empty) -> nbr_t i32[n_in, K] with
 * nbr_t[i][k] = j  <=>  nbr[j][k] = i.  The table of a transposed (coarse -> fine) convolution is exactly the transpose
 * of the strided (fine -> coarse) table the encoder already built: a memset + one scatter instead of n_in * K hash
 * probes.  Requires that no (i, k) occurs twice in nbr (true for kernel maps). */
int apr_kernel_map_transpose(const int32_t* nbr, int64_t n_out, int32_t K, int64_t n_in, int32_t* nbr_t, void* stream);

/* apr_voxelize for the concatenated frames of a batch: point i gets batch index b with offsets[b] <= i < offsets[b+1]
 * (offsets i64[nseg+1] on the device, nseg <= 1023: batch index 1023 is reserved, its largest key is the empty-slot sentinel). */
int apr_voxelize_segments(const float* xyz, int64_t n, float voxel_size, const int64_t* offsets, int32_t nseg,
                          int32_t* coords, void* stream);

/* The same WITHOUT the concatenated copy (the reference voxelises frame by frame, FCGF_APR/lib/complement_data_loader.py:
 * 788-812; a batched step used to torch.cat its 2B frames first: 17 MB per 12-frame step): frames_host[b] = DEVICE pointer
 * of frame b (f32 [n_b, 3], contiguous), offsets_host[b] = points before frame b, offsets_host[nseg] = total; both arrays
 * live on the HOST and travel in the kernel arguments (nseg <= APR_MAX_FRAMES).  coords i32[total, 4] as
 * apr_voxelize_segments writes them; offsets_dev (nullable, i64[nseg + 1]) receives the offsets for apr_segment_counts.
 * apr_gather_frame_points: pts f32[*n_dev, 3] <- the input point behind every row of a map built with out_first over the
 * same frames (the representative point of each voxel, `xyz[sel]` of sparse_quantize(return_index=True)); n_max = rows
 * allocated for pts (>= *n_dev). */
#define APR_MAX_FRAMES 64
int apr_voxelize_frames(const float* const* frames_host, const int64_t* offsets_host, int32_t nseg, float voxel_size,
                        int32_t* coords, int64_t* offsets_dev, void* stream);
int apr_gather_frame_points(const float* const* frames_host, const int64_t* offsets_host, int32_t nseg,
                            const int64_t* out_first, const int32_t* n_dev, int64_t n_max, float* pts, void* stream);

/* Host-side glue of a step done by one launch instead of a chain of tensor ops: dst <- the concatenation of nsrc small
 * DEVICE int32 arrays (srcs_host[s][0 .. counts_host[s]); pointer and count arrays on the host, nsrc <= APR_MAX_PACK) --
 * the map sizes, status flags, per-frame row counts and bounding box that one device->host copy then brings over -- and
 * zero_words int32 words at zero_ptr cleared (nullable: the pair-list counters of the encode that follows).
 * apr_fill_bytes: hipMemsetAsync behind the C ABI (one fill for the three transposed tables of an encode, which
 * apr_kernel_map_transpose_prefilled then scatters into without a fill of its own). */
#define APR_MAX_PACK 96
int apr_pack_i32(const int32_t* const* srcs_host, const int32_t* counts_host, int32_t nsrc, int32_t* dst,
                 int32_t* zero_ptr, int64_t zero_words, void* stream);
int apr_fill_bytes(void* ptr, int32_t byte_value, size_t bytes, void* stream);
int apr_kernel_map_transpose_prefilled(const int32_t* nbr, int64_t n_out, int32_t K, int64_t n_in, int32_t* nbr_t,
                                       void* stream);

/* Rows per input segment of a map built over CONCATENATED point sets (a batch of frames voxelised in one build):
 * counts[b] = #{rows r < *n_dev : offsets[b] <= out_first[r] < offsets[b+1]}; out_first is ascending, the count
 * stays on the device so that it can be fetched together with the map sizes in the caller's single sync. */
int apr_segment_counts(const int64_t* out_first, const int32_t* n_dev, const int64_t* offsets, int32_t nseg,
                       int32_t* counts, void* stream);

/* Kernel map (neighbour table) of one sparse convolution:
 *   nbr[j, o] = row i of the input map with c_in[i] == c_out[j] + offset(o) * scale,
 *   or -1.  offset(o) enumerates {-h..h}^3 x fastest.  scale = +ts_in for a
 *   regular / strided conv; for a transposed conv pass the FINE map as
 *   out_coords, the COARSE table as the input table and scale = -ts_fine.
 *   n_out_dev (nullable): device row count, rows >= *n_out_dev are skipped.
 *   replaces ME's kernel-map generation behind MinkowskiConvolution /
 *   MinkowskiConvolutionTranspose (FCGF_APR/model/resunet.py:31-140).
 */
/* The same table for a SAME-LEVEL map (input map == output map, scale = its tensor stride; the maps of every
 * stride-1 convolution of the network): nbr[j][o] = i <=> nbr[i][K-1-o] = j and the centre is the voxel itself, so
 * only the offsets below the centre are probed and each hit also stores its mirror entry — half the hash probes. */
int apr_kernel_map_same(const int32_t* coords, int64_t n, const uint64_t* keys, const int32_t* vals, int64_t cap,
                        int32_t kernel_size, int32_t scale, int32_t* nbr, void* stream);
int apr_kernel_map(const int32_t* out_coords, int64_t n_out, const int32_t* n_out_dev,
                   const uint64_t* in_keys, const int32_t* in_vals, int64_t cap,
                   int32_t kernel_size, int32_t scale, int32_t* nbr, void* stream);

/* ---- first convolution on constant-1 features (occ.hip) --------------------------------------------------------
 * FCGF feeds the encoder `feats = ones((n, 1))` (FCGF_APR/lib/complement_data_loader.py:805-812) into conv1 with
 * in_channels = 1 and a 5^3 / 7^3 kernel (FCGF_APR/model/resunet.py:57-63): every gathered value is 1, so the layer is
 *   out[j, :] = act((sum over the offsets k whose cell around voxel j is occupied of w[k, :]) * scale + shift)
 * and needs occupancy, not a neighbour table.  apr_coords_bbox: bbox_dev int32[8] <- {min x, y, z, max x, y, z, max
 * batch index, 0} of coords int32[n,4] (batch, x, y, z).  apr_occ_conv: the voxels are scattered into a bitmap over
 * that box (scratch: apr_occ_conv_scratch_bytes(bbox, ks), 0 = box empty / too large -> use apr_kernel_map +
 * apr_spconv; apr_occ_conv_pays(bbox, ks, n): 1 if the bitmap is also small against the n voxels it serves, at most
 * 1 KB per voxel + 4 MB -- an outlier voxel can stretch the box to gigabytes, and then the kernel map is the cheaper
 * structure) and every voxel reads the ks x-neighbours of a (dy, dz) row with one 8-byte load.  coords: unique rows;
 * bbox_host is in VOXEL units (the coords' own units) and must contain every row -- a row outside it is skipped as a
 * neighbour and gets a NaN output row, no access ever leaves the bitmap; offsets x-fastest as apr_kernel_map;
 * w f32[ks^3, cout], cout % 8 == 0, ks in {3, 5, 7}; same bits as apr_spconv over the apr_kernel_map table on all-ones
 * features. */
int apr_coords_bbox(const int32_t* coords, int64_t n, int32_t* bbox_dev, void* stream);
size_t apr_occ_conv_scratch_bytes(const int32_t* bbox_host, int32_t kernel_size);
int apr_occ_conv_pays(const int32_t* bbox_host, int32_t kernel_size, int64_t n);
int apr_occ_conv(const int32_t* coords, int64_t n, const int32_t* bbox_host, int32_t kernel_size, const float* w,
                 int32_t cout, const float* scale, const float* shift, const float* residual, int64_t ldr,
                 int32_t relu, float* out, int64_t ldo, void* scratch, size_t scratch_bytes, void* stream);
/* apr_kernel_map with the bitmap apr_occ_conv left in `occ_scratch` as a pre-filter of the probes (72 % of a LiDAR
 * map's probes hit an empty cell, the table's most expensive case): same table out, bit for bit.  bbox_host and
 * bitmap_kernel_size: those of the apr_occ_conv call that wrote occ_scratch (same stream, or ordered behind it);
 * in_keys / in_vals: the map the bitmap was built from; out_coords: that map's rows or a coarser map's.  Cells outside
 * the bitmap's box fall back to the table.  Replaces the same MinkowskiEngine kernel-map build as apr_kernel_map. */
int apr_kernel_map_occ(const int32_t* out_coords, int64_t n_out, const int32_t* n_out_dev, const uint64_t* in_keys,
                       const int32_t* in_vals, int64_t cap, int32_t kernel_size, int32_t scale,
                       const int32_t* bbox_host, int32_t bitmap_kernel_size, const void* occ_scratch, int32_t* nbr,
                       void* stream);

/* ------------------------------------------------------------------------
 * Sparse convolution forward (gather -> MFMA -> fused epilogue), fp32.
 *   out[j, :] = act( (sum_o in[nbr[j,o], :] @ W[o]) * scale + shift + residual[j, :] )
 *   replaces MinkowskiConvolution / MinkowskiConvolutionTranspose forward
 *   (+ the MinkowskiBatchNorm eval affine, residual add, MEF.relu and ME.cat
 *   around it: FCGF_APR/model/resunet.py:142-185, residual_block.py:37-53).
 *   nbr == NULL means K == 1 with the identity map (kernel_size 1: `F @ W`).
 * ---------------------------------------------------------------------- */

/* Floats needed for the packed copy of a [K,cin,cout] weight. */
int64_t apr_spconv_packed_size(int32_t K, int32_t cin, int32_t cout);
/* w f32[K,cin,cout] -> kernel-native layout (cin padded to a multiple of 4,
 * cout to a multiple of 16). */
int apr_spconv_pack_weights(const float* w, int32_t K, int32_t cin, int32_t cout,
                            float* w_packed, void* stream);

/* wt f32[K, cout, cin] = w[K, cin, cout] transposed per offset, offsets mirrored (k -> K-1-k) when flip != 0: the kernel
 * with which apr_spconv_fwd over the reverse map computes a convolution's INPUT gradient (same-level map: the same table
 * under the mirrored offset; strided <-> transposed maps: each other's table, no flip).  Backward of
 * ME.MinkowskiConvolution / ConvolutionTranspose under FCGF_APR/lib/complement_trainer.py:484. */
int apr_weights_flip_transpose(const float* w, int32_t K, int32_t cin, int32_t cout, int32_t flip, float* wt, void* stream);

int apr_spconv_fwd(const float* in, int64_t ldi, const int32_t* nbr, int64_t n_out,
                   int32_t K, int32_t cin, int32_t cout, const float* w_packed,
                   const float* scale, const float* shift,
                   const float* residual, int64_t ldr, int32_t relu,
                   float* out, int64_t ldo, void* stream);

/* Weight-stationary variant for maps with few pairs per output tile (strided / transposed convs, deep levels):
 * the kernel map is first turned into per-offset pair lists (apr_pairlist_build, once per map), then every
 * (offset, 64 pairs, 64 channels) work unit stages its weight piece once, writes per-pair products to
 * prod_scratch, and a second kernel sums them per output row in fixed offset order
 * with the same fused epilogue as apr_spconv_fwd.  Needs cin % 64 == 0, cin <= 512, cout % 64 == 0, K <= 27.
 * counters: int32[apr_pairlist_counter_ints()] = 32 counters APR_PAIR_COUNTER_STRIDE ints apart (the pair count of
 * offset k at counters[k * APR_PAIR_COUNTER_STRIDE]: one counter per 256 B, so the build's range reservations spread
 * over the L2 channels), ZERO before apr_pairlist_build (the caller clears them: one fill can clear the counters of
 * all maps of a forward pass); plist: apr_pairlist_bytes(n_out, K) bytes.
 * prod_scratch holds n_out*K rows of cout floats (offset k owns rows [k*n_out, (k+1)*n_out), sparsely used). */
#define APR_PAIR_COUNTER_STRIDE 64
int32_t apr_pairlist_counter_ints(void);
size_t apr_pairlist_bytes(int64_t n_out, int32_t K);
int apr_pairlist_build(const int32_t* nbr, int64_t n_out, int32_t K, int32_t* counters, void* plist,
                       size_t plist_bytes, void* stream);
int apr_spconv_ws_fwd(const float* in, int64_t ldi, const int32_t* counters, const void* plist, int64_t n_out,
                      int32_t K, int32_t cin,
                      int32_t cout, const float* w_packed, const float* scale, const float* shift,
                      const float* residual, int64_t ldr, int32_t relu, float* out, int64_t ldo,
                      float* prod_scratch, void* stream);
/* The same with the per-pair contraction on the bf16 MFMA in a 3-way split (hi/mid/lo bf16 pieces of every fp32
 * operand, 6 of the 9 cross terms: fp32-equivalent accuracy, ~2e-24 relative per product) - 2.7x fewer MFMA cycles
 * than the exact-fp32 form, which runs at 1/16 of the bf16 rate on gfx950.  w_bf3: apr_spconv_pack_weights_bf3 of the
 * layer's [K, cin, cout] kernel (apr_spconv_packed_bf3_bytes bytes; cin in {64, 128, 256}); NULL or another cin falls
 * back to w_packed / the fp32 form.  Packing takes cin % 64 == 0 and cout % 64 == 0; K = 1 (the dense layers'
 * images) also cin % 32 == 0 and cout % 16 == 0. */
int64_t apr_spconv_packed_bf3_bytes(int32_t K, int32_t cin, int32_t cout);
int apr_spconv_pack_weights_bf3(const float* w, int32_t K, int32_t cin, int32_t cout, void* w_bf3, void* stream);
/* The same image for the kernel of the INPUT GRADIENT, straight from the parameter: transposed != 0 reads w as f32
 * [K, cout, cin] (the image is that of its transpose), flip != 0 mirrors the offsets (k' = K - 1 - k: a same-level map's
 * reverse map) -- apr_weights_flip_transpose + apr_spconv_pack_weights_bf3 in one pass (the training step re-packs every
 * kernel after every optimizer step, FCGF_APR/lib/trainer.py:454-527).  w_bf3 16-byte aligned. */
int apr_spconv_pack_weights_bf3_ex(const float* w, int32_t K, int32_t cin, int32_t cout, int32_t flip, int32_t transposed,
                                   void* w_bf3, void* stream);
int apr_spconv_ws_fwd_bf3(const float* in, int64_t ldi, const int32_t* counters, const void* plist, int64_t n_out,
                          int32_t K, int32_t cin, int32_t cout, const float* w_packed, const void* w_bf3,
                          const float* scale, const float* shift, const float* residual, int64_t ldr, int32_t relu,
                          float* out, int64_t ldo, float* prod_scratch, void* stream);

/* Output-stationary form for the 64-channel levels (cin 64 or 128, cout % 64 == 0; same operator as apr_spconv_fwd,
 * FCGF_APR/model/resunet.py:31-140, model/residual_block.py:23-33): the kernel map is cut into tiles of R consecutive
 * output rows (R = apr_spconv_os_tile_rows, so that the tiles fill the chip in whole rounds), apr_spconv_os_pairs_build
 * turns each tile's [R, K] slab into K compact pair lists (deterministic positions, no atomics; input rows < 2^23), and
 * apr_spconv_os_fwd keeps a tile's fp32 accumulators in LDS while it walks the K offsets with the bf16-split weight
 * slice of the offset staged by LDS-DMA: no product row ever goes through HBM (apr_spconv_ws_fwd writes and re-reads
 * one per pair).  bf16 MFMA in the exact 3-way split (fp32-equivalent), sums in ascending offset order: bitwise
 * reproducible.  w_bf3 = apr_spconv_pack_weights_bf3(w, K, cin, cout). */
int32_t apr_spconv_os_tile_rows(int64_t n_out, int32_t cin, int32_t cout);
size_t apr_spconv_os_pairs_bytes(int64_t n_out, int32_t K, int32_t R);
int apr_spconv_os_pairs_build(const int32_t* nbr, int64_t n_out, int64_t n_in, int32_t K, int32_t R, void* os_pairs,
                              size_t os_pairs_bytes, void* stream);
/* Diagnostics for apr_spconv_os_fwd (APR_OS_TRACE=1 in the environment): per-wave s_memtime stamps of one workgroup of
 * the last launch, host_out[8 * 512] (tag << 56 | cycles; entry 511 of each wave = count). */
int apr_spconv_os_trace(uint64_t* host_out, int32_t n);
/* Run-time guard of k_os_conv's hand-counted load pipeline: 0 = production kernel, 1 = the diagnostics instantiation with
 * no ablation, 2 = the same with a full drain (s_waitcnt vmcnt(0)) in front of every counted wait.  The three must give
 * identical bits (tests/test_spconv_gpu.py::test_os_conv_debug_pipeline_matches_production); process-wide switch. */
int apr_spconv_os_set_debug(int32_t mode);
int apr_spconv_os_fwd(const float* in, int64_t ldi, const void* os_pairs, int64_t n_out, int32_t K, int32_t R,
                      int32_t cin, int32_t cout, const void* w_bf3, const float* scale, const float* shift,
                      const float* residual, int64_t ldr, int32_t relu, float* out, int64_t ldo, void* stream);

/* Weight gradient of apr_spconv_fwd (training; replaces what autograd does inside
 * MinkowskiConvolution / MinkowskiConvolutionTranspose, FCGF_APR/lib/trainer.py:454-527):
 *   dw f32[K, cin, cout] (plain layout, not packed) = sum_j [nbr[j,k] >= 0] in[nbr[j,k], :]^T dout[j, :].
 * nbr NULL = identity map (K = 1).  Deterministic (per-chunk partial sums reduced in fixed order).
 * The input gradient is apr_spconv_fwd itself: dout over the reverse map with transposed weights. */
size_t apr_spconv_wgrad_scratch_bytes(int64_t n_out, int32_t K, int32_t cin, int32_t cout);
int apr_spconv_wgrad(const float* in, int64_t ldi, const float* dout, int64_t ldo, const int32_t* nbr, int64_t n_out,
                     int32_t K, int32_t cin, int32_t cout, float* dw, void* scratch, size_t scratch_bytes,
                     void* stream);
/* The same when nbr is a SAME-LEVEL map (stride 1, odd kernel: the centre column holds every row, the others about a
 * quarter): the centre offset's rows are spread over four times as many workgroups.  Identical sums, another grouping of
 * the partials (results differ from apr_spconv_wgrad in the last bits; each entry point is bit-reproducible). */
int apr_spconv_wgrad_same_level(const float* in, int64_t ldi, const float* dout, int64_t ldo, const int32_t* nbr,
                                int64_t n_out, int32_t K, int32_t cin, int32_t cout, float* dw, void* scratch,
                                size_t scratch_bytes, void* stream);

/* Triple pair lists for 27-offset maps (spconv_ws.hip): offsets are x fastest, so k = 3t + j are the three x-neighbours of
 * one (dy, dz); an ENTRY of triple t is an output row with its up to three input rows, and the gemm writes ONE product
 * row per entry (about half as many as pairs on LiDAR surfaces), accumulating the three offsets in registers -- the
 * product round trip that bounds the weight-stationary pair halves.  Same operator and epilogue as apr_spconv_ws_fwd_bf3
 * (FCGF_APR/model/resunet.py:31-140); counters: a zeroed apr_pairlist_counter_ints() block; prod_scratch f32
 * [9 * n_out, cout]; cin 64 / 128, cout % 64 == 0 (apr_spconv_ws3_supported). */
size_t apr_pairlist3_bytes(int64_t n_out);
int apr_pairlist3_build(const int32_t* nbr, int64_t n_out, int32_t K, int32_t* counters, void* plist3, size_t plist3_bytes,
                        void* stream);
int apr_spconv_ws3_supported(int32_t K, int32_t cin, int32_t cout);
int apr_spconv_ws3_fwd_bf3(const float* in, int64_t ldi, const int32_t* counters, const void* plist3, int64_t n_out,
                           int32_t cin, int32_t cout, const void* w_bf3, const float* scale, const float* shift,
                           const float* residual, int64_t ldr, int32_t relu, float* out, int64_t ldo, float* prod_scratch,
                           void* stream);

/* One launch description of apr_spconv_fwd; apr_spconv_fwd_batch enqueues n of them back to back
 * from a single call (the 23 fused conv launches of one ResUNet encode), so a host binding pays
 * one FFI transition instead of 23 and can overlap several encodes from different host threads. */
typedef struct apr_spconv_desc {
  const float* in; int64_t ldi; const int32_t* nbr; int64_t n_out;
  int32_t K, cin, cout, relu;
  const float* w_packed; const float* scale; const float* shift;
  const float* residual; int64_t ldr;
  float* out; int64_t ldo;
  int32_t* counters;      /* pair-list counters and ... */
  void* plist;            /* ... body; non-NULL: run this launch through apr_spconv_ws_fwd with ... */
  float* prod_scratch;    /* ... this product buffer (n_out*K rows); */
  int64_t plist_bytes;    /* > 0: apr_pairlist_build(nbr -> plist) first (first use of the map in the batch) */
  const void* w_bf3;      /* non-NULL (with plist): apr_spconv_ws_fwd_bf3 with these split weights */
  void* os_pairs;         /* non-NULL (with w_bf3): run this launch through apr_spconv_os_fwd over these tile lists, */
  int64_t os_rows;        /* ... of os_rows rows per tile; */
  int64_t os_build_bytes; /* > 0: apr_spconv_os_pairs_build(nbr -> os_pairs, n_in = os_n_in) first */
  int64_t os_n_in;
  int32_t l2norm;         /* != 0: out[j, :] /= |out[j, :]|_2 behind the epilogue (the encoder's normalize_feature,
                           * FCGF_APR/model/resunet.py:139-142): in the tile kernel's epilogue when it runs the layer
                           * with cout 32 or 64 (same bits as apr_l2_normalize), else as a second launch */
  int32_t ws3;            /* != 0 (with plist and w_bf3): plist is a TRIPLE pair list (apr_pairlist3_build / _bytes), the launch
                           * runs through apr_spconv_ws3_fwd_bf3 and prod_scratch holds 9 * n_out rows */
} apr_spconv_desc;
int apr_spconv_fwd_batch(const apr_spconv_desc* descs_host, int32_t n, void* stream);
/* Same, with one HIP event pair per launch recorded on `stream` (around the conv kernels only, not a pair-list
 * build); synchronises and returns each layer's elapsed milliseconds in layer_ms[n].  Measurement aid
 * (bench.py's roofline leg): back-to-back launches from C, no host gaps inside a layer. */
int apr_spconv_fwd_batch_timed(const apr_spconv_desc* descs, int32_t n, float* layer_ms, void* stream);

/* ------------------------------------------------------------------------
 * One ResUNet encode as ONE call (round 4): ResUNet2.forward in eval mode, FCGF_APR/model/resunet.py:142-191, over the
 * four coordinate maps of a batch -- conv1 on the occupancy bitmap, the 7 + 3 kernel maps, the pair lists of every routed
 * layer and the 23 conv launches are enqueued from C over one scratch arena.  The kernels, their arguments and their order
 * are those of the module-by-module plan (apr_amd/fcgf/model/resunet.py forward_fused): same bits.  Python needed ~0.8 ms
 * per encode to issue the same ~60 launches; with one pair per call (the reference loop's shape, test_apr.py:111-163) the
 * host, not the GPU, set the pace.
 *   layer[3 s + 0 / 1 / 2]: stage s = "1", "2", "3", "4", "4_tr", "3_tr", "2_tr": its (strided / transposed) conv with
 *   the folded BN, then the two convs of its residual block; layer[21] = conv1_tr, layer[22] = final.
 *   ws_conv / ws_block / os_block: bit s = stage s takes the weight-stationary / output-stationary kernels where the
 *   shape allows (else the tile kernel).  lv[l]: the map of tensor stride 2^l (coords int32[n, 4], the hash table, n).
 *   counters: apr_pairlist_counter_ints() x n_counter_slots int32, ZERO on entry (one block per pair list; NULL: a
 *   block of the arena is cleared with one fill).  bbox_host: apr_coords_bbox of lv[0] (or a superset).
 *   apr_resunet_encode_supported: 1 if the occupancy form of conv1 applies (apr_occ_conv_pays) and every map is
 *   non-empty; otherwise the caller runs the module-by-module plan. */
typedef struct apr_resunet_layer {
  int32_t K, cin, cout, relu;
  const float* w_packed;          /* apr_spconv_pack_weights image */
  const void* w_bf3;              /* apr_spconv_pack_weights_bf3 image, or NULL */
  const float* scale; const float* shift;     /* folded BatchNorm / bias, nullable */
} apr_resunet_layer;
typedef struct apr_resunet_plan {
  apr_resunet_layer layer[23];
  const float* conv1_w;           /* conv1's raw kernel f32 [ks^3, cout] (occupancy form) */
  int32_t conv1_ks, normalize;
  uint32_t ws_conv, ws_block, os_block;
  int32_t ws3, ws3_cin128, os_min_rows, occ_kernel_map;
  int64_t ws3_max_rows_128;
} apr_resunet_plan;
typedef struct apr_level_map {
  const int32_t* coords; const uint64_t* keys; const int32_t* vals; int64_t cap; int64_t n;
} apr_level_map;
int apr_resunet_encode_supported(const apr_resunet_plan* plan, const apr_level_map* lv, const int32_t* bbox_host);
size_t apr_resunet_encode_scratch_bytes(const apr_resunet_plan* plan, const apr_level_map* lv, const int32_t* bbox_host);
int apr_resunet_encode(const apr_resunet_plan* plan, const apr_level_map* lv, const int32_t* bbox_host, int32_t* counters,
                       int32_t n_counter_slots, void* scratch, size_t scratch_bytes, float* out, int64_t ldo, void* stream);

/* The front end of a step as ONE call: frames -> voxel coordinates -> the de-duplicated stride-1 map (its rows in first-
 * occurrence order = frame order, the representative input point of every row, rows per frame, bounding box) -> that map
 * re-inserted into a table of its own size -> the three coarser maps, chained on device-side row counts -> ONE int32
 * header with everything the host needs, ready for a single device -> host copy.  Replaces ME.utils.sparse_quantize +
 * batched_coordinates + the coordinate manager's strided maps (FCGF_APR/lib/complement_data_loader.py:776-812,
 * FCGF_APR/scripts/test_apr.py:120-131) for a batch of frames; the kernels and their order are those of the tensor-by-
 * tensor front end (apr_voxelize_frames, apr_map_build x 5, apr_segment_counts, apr_coords_bbox, apr_gather_frame_points,
 * apr_pack_i32): same bits.  Python needed ~45 allocations and a dozen calls for it, and with one pair per call the GPU
 * idled between its kernels.
 *   arena: apr_voxel_pyramid_scratch_bytes(total points, nseg) bytes; every pointer of *out points into it.
 *   out->lv[l]: level l's map; lv[l].n = rows ALLOCATED (the true count is in the header).  out->header (device):
 *   int32 {n, status} x 5 maps (levels 0..3, then the de-duplicating table when compact != 0, else zeros), rows per frame
 *   [nseg], bounding box [8]; header_ints of them.  status != 0: a coordinate outside the packed key range.
 *   compact != 0: level 0 is the compact table over the first compact_rows rows of the de-duplicated map; if the header
 *   says the de-duplicated map has MORE rows than that (more than a quarter of the points are distinct voxels: never on
 *   LiDAR input) the caller must rebuild through the tensor-by-tensor path.
 *   out->counters: apr_pairlist_counter_ints() x 16 int32, zeroed (for apr_resunet_encode). */
typedef struct apr_pyramid {
  apr_level_map lv[4];
  const int64_t* first;           /* [rows allocated] index of the first input point of every level-0 row */
  const float* pts;               /* [rows allocated, 3] that point */
  int32_t* header; int32_t header_ints;
  int32_t compact; int64_t compact_rows;
  int32_t* counters; int32_t n_counter_slots;
} apr_pyramid;
size_t apr_voxel_pyramid_scratch_bytes(int64_t n_points, int32_t nseg);
int apr_voxel_pyramid(const float* const* frames_host, const int64_t* offsets_host, int32_t nseg, float voxel_size,
                      void* arena, size_t arena_bytes, apr_pyramid* out, void* stream);

/* ------------------------------------------------------------------------
 * Normalisation / elementwise on feature rows [n, c]
 * ---------------------------------------------------------------------- */

/* Per-channel batch statistics over rows (training-mode MinkowskiBatchNorm,
 * FCGF_APR/model/common.py:6): mean[c], biased var[c].  fp64 accumulation. */
int apr_bn_stats(const float* x, int64_t ld, int64_t n, int32_t c,
                 float* mean, float* var, void* scratch, size_t scratch_bytes, void* stream);
size_t apr_bn_stats_scratch_bytes(int64_t n, int32_t c);

/* Backward of y = (x - mean) * rstd * gamma + beta over all n rows (training-mode MinkowskiBatchNorm,
 * FCGF_APR/model/common.py:6 under FCGF_APR/lib/trainer.py:454-527; gamma = NULL: the affine-free InstanceNorm1d of
 * KPFCNN's blocks, Predator_APR/models/blocks.py:459-468 under Predator_APR/lib/trainer.py:142-280): dx, and (non-NULL)
 * dgamma[c], dbeta[c].  mean / rstd: the forward's batch statistics (rstd = 1 / sqrt(var + eps)).  fp64 partial sums
 * combined in fixed order: deterministic. */
size_t apr_norm_backward_scratch_bytes(int64_t n, int32_t c);
int apr_norm_backward(const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t n, int32_t c,
                      const float* mean, const float* rstd, const float* gamma, float* dx, int64_t lddx,
                      float* dgamma, float* dbeta, void* scratch, size_t scratch_bytes, void* stream);

/* Training-mode BatchNorm fused with the residual add and the ReLU -- the unit a training encode repeats per convolution
 * (MinkowskiBatchNorm in train(), FCGF_APR/model/common.py:6, model/residual_block.py:37-53, under
 * lib/complement_trainer.py:350-512 / lib/trainer.py:454-527):
 *   y = act((z - mean) * rstd * gamma + beta (+ residual)),  mean / biased var over the n rows (fp64 partial sums);
 *   running_mean / running_var (nullable pair) updated in place as torch.nn.BatchNorm1d does (momentum, unbiased var);
 *   save_mean / save_rstd [c] kept for the backward; *num_batches_tracked (nullable device int64) += 1.  relu: 0 / 1.
 *   Two launches.  scratch: apr_bn_stats_scratch_bytes.
 * apr_bn_train_bwd: g = dy masked by y > 0 (relu = 1); dres (nullable) = g, the residual branch's gradient;
 *   dgamma = sum g * xhat, dbeta = sum g (nullable); dx = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat)).
 *   Partial sums in fp64 combined in a fixed order: the same bits every run.  Two launches.
 * seg_offsets_host (int64 [nseg + 1], 0 .. n; NULL / nseg <= 1: one segment): the rows of several forward calls of the
 *   reference stacked into one launch -- the two frames of a pair, FCGF_APR/lib/complement_trainer.py:386-394 -- each
 *   normalised with its own statistics (save_mean / save_rstd [nseg, c]); the running statistics take the segments'
 *   updates in order and *num_batches_tracked += nseg, exactly as consecutive calls would; dgamma / dbeta sum over all
 *   segments.  scratch: apr_bn_stats_scratch_bytes(n + 256 * nseg, c). */
int apr_bn_train_fwd(const float* z, int64_t ldz, int64_t n, int32_t c, const float* gamma, const float* beta,
                     float eps, float momentum, float* running_mean, float* running_var, const float* residual,
                     int64_t ldr, int32_t relu, float* y, int64_t ldy, float* save_mean, float* save_rstd,
                     int64_t* num_batches_tracked, const int64_t* seg_offsets_host, int32_t nseg, void* scratch,
                     size_t scratch_bytes, void* stream);
int apr_bn_train_bwd(const float* z, int64_t ldz, const float* y, int64_t ldy, const float* dy, int64_t lddy,
                     int64_t n, int32_t c, const float* mean, const float* rstd, const float* gamma, int32_t relu,
                     float* dx, int64_t lddx, float* dres, int64_t lddres, float* dgamma, float* dbeta,
                     const int64_t* seg_offsets_host, int32_t nseg, void* scratch, size_t scratch_bytes, void* stream);

/* sums[c] = sum over the n rows of x[:, c] (same scratch as apr_bn_stats; fp64 partial sums in fixed order): the bias
 * gradient of the 1x1 convolutions with bias in KPFCNN's training path (Predator_APR/models/architectures.py:92-101 under
 * lib/trainer.py:142-280). */
int apr_col_sums(const float* x, int64_t ld, int64_t n, int32_t c, float* sums, void* scratch, size_t scratch_bytes,
                 void* stream);

/* Per-channel normalisation parameters over all rows, no affine: scale = 1/sqrt(var + eps),
 * shift = -mean*scale (InstanceNorm1d over the stacked points, Predator_APR/models/blocks.py:451-466;
 * InstanceNorm2d of the edge convolutions, models/gcn.py:41-48).  Same scratch as apr_bn_stats. */
int apr_norm_params(const float* x, int64_t ld, int64_t n, int32_t c, float eps, float* scale, float* shift,
                    void* scratch, size_t scratch_bytes, void* stream);

/* apr_norm_params + apr_affine_act in one call and two launches (statistics, then rebuild-and-apply): y = act((x -
 * mean) / sqrt(var + eps) (+ residual)) per channel over all rows: the InstanceNorm1d + LeakyReLU / residual that
 * follows every Linear and KPConv of KPFCNN (Predator_APR/models/blocks.py:451-468,499-504,653-681).  51 of them per
 * forward: the pair path is bound by the host cost of its ~600 launches.  Same scratch as apr_bn_stats. */
int apr_instance_norm_act(const float* x, int64_t ldx, int64_t n, int32_t c, float eps, const float* residual,
                          int64_t ldr, int32_t relu, float negative_slope, float* y, int64_t ldy, void* scratch,
                          size_t scratch_bytes, void* stream);
/* The same per row segment [seg_offsets_host[s], seg_offsets_host[s+1]) (int64, 0 .. n): several scan pairs stacked into
 * one KPFCNN forward keep the reference's statistics (over the stacked points of ONE pair).  scratch:
 * apr_bn_stats_scratch_bytes(n + 256 * nseg, c). */
int apr_instance_norm_act_seg(const float* x, int64_t ldx, int64_t n, int32_t c, float eps, const float* residual,
                              int64_t ldr, int32_t relu, float negative_slope, float* y, int64_t ldy,
                              const int64_t* seg_offsets_host, int32_t nseg, void* scratch, size_t scratch_bytes,
                              void* stream);

/* y = act(x * scale[c] + shift[c] (+ residual)); scale/shift/residual nullable.
 * relu: 0 none, 1 ReLU, 2 LeakyReLU(negative_slope) (Predator_APR/models/blocks.py:489,574). */
int apr_affine_act(const float* x, int64_t ldx, int64_t n, int32_t c,
                   const float* scale, const float* shift,
                   const float* residual, int64_t ldr, int32_t relu, float negative_slope,
                   float* y, int64_t ldy, void* stream);
/* Backward of that activation from its OUTPUT y: dz = dy where y > 0, dy * negative_slope (mode 2) or 0 (mode 1) elsewhere;
 * mode 0 copies (training path: Predator_APR/models/blocks.py:459-468 under lib/trainer.py:142-280). */
int apr_act_backward(const float* dy, int64_t lddy, const float* y, int64_t ldy, int64_t n, int32_t c, int32_t mode,
                     float negative_slope, float* dz, int64_t lddz, void* stream);

/* Row L2 normalisation F / ||F||_2 (FCGF_APR/model/resunet.py:187-191). */
int apr_l2_normalize(const float* x, int64_t ldx, int64_t n, int32_t c,
                     float* y, int64_t ldy, void* stream);
/* Its backward, dx = (dy - y (y . dy)) / |x| per row (the normalisation at the end of ResUNet2.forward under
 * FCGF_APR/lib/complement_trainer.py:484). */
int apr_l2_normalize_backward(const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t n, int32_t c,
                              float* dx, int64_t lddx, void* stream);

/* ------------------------------------------------------------------------
 * Feature nearest neighbour (squared L2), replaces find_nn_gpu / pdist
 *   (FCGF_APR/lib/eval.py:18-48, lib/metrics.py:22-29) and the feature KD-tree
 *   search inside open3d's registration_ransac_based_on_feature_matching
 *   (FCGF_APR/scripts/test_apr.py:148-156).
 *   best[i] = (bits(d2) << 32) | j  minimising d2 = ||F0[i]-F1[j]||^2, ties ->
 *   smallest j.  d2 is accumulated in a fixed fp32 order (4 interleaved fma
 *   chains) so the C oracle reproduces it bit for bit.
 *   `best` must be u64[n0]; it is initialised by the call.
 * ---------------------------------------------------------------------- */
int apr_feature_nn(const float* f0, int64_t n0, const float* f1, int64_t n1, int32_t c,
                   uint64_t* best, void* stream);
/* Unpack: idx i64[n], d2 f32[n] (either nullable). */
/* Same result, bit for bit, by filter + exact refine: a split-bf16 (hi + lo) MFMA pass bounds every distance
 * rigorously (|approx - d| <= 1e-4 |a||b| + 5e-5 (|a|^2 + |b|^2)), a second pass evaluates the exact fp32 direct form only
 * for the pairs that can still be the arg-min.  c in {32, 64, 128}; scratch from apr_feature_nn_fast_scratch_bytes. */
size_t apr_feature_nn_fast_scratch_bytes(int64_t n0, int64_t n1, int32_t c);
int apr_feature_nn_fast(const float* f0, int64_t n0, const float* f1, int64_t n1, int32_t c, uint64_t* best,
                        void* scratch, size_t scratch_bytes, void* stream);
int apr_nn_unpack(const uint64_t* best, int64_t n, int64_t* idx, float* d2, void* stream);

/* ------------------------------------------------------------------------
 * RANSAC + Kabsch (SVD) pose from feature correspondences, replaces
 *   open3d registration_ransac_based_on_feature_matching as called at
 *   FCGF_APR/scripts/test_apr.py:148-156 and
 *   Predator_APR/lib/benchmark_utils.py:213-225.
 *   corr i64[n0]: target index of every source point (from apr_feature_nn).
 *   Hypothesis `it` samples ransac_n=4 source indices from a counter-based RNG
 *   (seed, it, slot); edge-length checker (ratio), Kabsch without scale in fp64,
 *   distance checker; every surviving hypothesis is scored by the number of
 *   correspondences within `max_dist` (then lower RMSE, then lower `it`).
 *   result_host f64[20]: T (row-major 4x4), inliers, rmse, best_it, n_valid.
 *   Synchronises the stream before returning.
 * ---------------------------------------------------------------------- */
size_t apr_ransac_scratch_bytes(int64_t n0, int64_t max_iter);
int apr_ransac_pose(const float* xyz0, int64_t n0, const float* xyz1, int64_t n1,
                    const int64_t* corr, double max_dist, double edge_ratio,
                    int64_t max_iter, uint64_t seed,
                    void* scratch, size_t scratch_bytes, double* result_host, void* stream);

/* Same sampler / checkers / Kabsch, but open3d <= 0.11 semantics as used by
 *   Predator_APR/lib/benchmark_utils.py:213-225 (`o3d.registration`, RANSACConvergenceCriteria(50000, 1000)):
 *   only the first `max_validation` surviving hypotheses (in iteration order) are validated, and a
 *   hypothesis is scored GEOMETRICALLY: a transformed source point is an inlier when its nearest
 *   target point lies within max_dist (uniform-grid NN search); fitness = inliers / n0, then RMSE.
 *   max_iter <= 2^20.  result_host as apr_ransac_pose.  Synchronises the stream. */
size_t apr_ransac_geometric_scratch_bytes(int64_t n0, int64_t n1, int64_t max_iter);
int apr_ransac_pose_geometric(const float* xyz0, int64_t n0, const float* xyz1, int64_t n1,
                              const int64_t* corr, double max_dist, double edge_ratio,
                              int64_t max_iter, int64_t max_validation, uint64_t seed,
                              void* scratch, size_t scratch_bytes, double* result_host, void* stream);
/* The same without the host synchronisation: the raw result (apr_ransac_raw_bytes() bytes) is copied to `raw_dev`
 * (device memory of the caller's); fetch the raw results of a whole batch with one copy and decode each on the host
 * with apr_ransac_decode (result_host as apr_ransac_pose). */
size_t apr_ransac_raw_bytes(void);
int apr_ransac_pose_geometric_async(const float* xyz0, int64_t n0, const float* xyz1, int64_t n1,
                                    const int64_t* corr, double max_dist, double edge_ratio, int64_t max_iter,
                                    int64_t max_validation, uint64_t seed, void* scratch, size_t scratch_bytes,
                                    void* raw_dev, void* stream);
int apr_ransac_decode(const void* raw_host, double* result_host);

/* B independent pairs, matching + pose, in ONE call and ONE host synchronisation: per pair apr_feature_nn_fast
 * (apr_feature_nn for channel counts other than 32/64/128) -> apr_nn_unpack -> single-round RANSAC, enqueued back to
 * back on `stream`; results_host f64[B][20] in the layout of apr_ransac_pose.  Identical results to the per-pair
 * entry points (a pair whose hypothesis list overflows is redone through apr_ransac_pose).  This is the body of the
 * reference's per-pair loop after the encoder (FCGF_APR/scripts/test_apr.py:139-156), batched. */
typedef struct {
  const float* f0; int64_t n0;     /* features of frame 0 (queries), row-major [n0, c], 16-byte aligned */
  const float* f1; int64_t n1;     /* features of frame 1 (targets) [n1, c] */
  const float* xyz0;               /* representative points of frame 0, f32 [n0, 3] */
  const float* xyz1;               /* ... of frame 1, f32 [n1, 3] */
  uint64_t seed;                   /* RANSAC hypothesis stream of this pair */
} apr_pair_desc;
size_t apr_match_pose_batch_scratch_bytes(int32_t B, int64_t n0_max, int64_t n1_max, int32_t c, int64_t max_iter);
int apr_match_pose_batch(const apr_pair_desc* pairs, int32_t B, int32_t c, double max_dist, double edge_ratio,
                         int64_t max_iter, void* scratch, size_t scratch_bytes, double* results_host, void* stream);
/* The same in two halves, for a host that keeps several batches in flight from ONE thread (bench.py): _enqueue
 * launches everything and ends with an asynchronous copy of the B result slots into slots_host
 * (apr_match_pose_batch_slot_bytes(B) bytes; pinned host memory makes the copy asynchronous) - no host
 * synchronisation; _finish decodes the slots into results_host f64[B][20] once the caller has waited for that copy (an
 * event recorded on `stream` behind the enqueue call).  Same pairs / parameters / scratch for both halves. */
size_t apr_match_pose_batch_slot_bytes(int32_t B);
int apr_match_pose_batch_enqueue(const apr_pair_desc* pairs, int32_t B, int32_t c, double max_dist, double edge_ratio,
                                 int64_t max_iter, void* scratch, size_t scratch_bytes, void* slots_host, void* stream);
int apr_match_pose_batch_finish(const apr_pair_desc* pairs, int32_t B, int32_t c, double max_dist, double edge_ratio,
                                int64_t max_iter, void* scratch, size_t scratch_bytes, const void* slots_host,
                                double* results_host, void* stream);

/* Which RANSAC sampling kernel the calls that follow use: 1 = first edge screened out of LDS (k_sample_screen: lowest
 * latency for a caller with ONE step in flight), 0 = the plain staged checker (k_sample_check: leaves the CUs' LDS to the
 * kernels of a pipelined caller's other streams), -1 = APR_RANSAC_SCREEN from the environment (default 1).  Same
 * candidate set either way (FCGF_APR/scripts/test_apr.py:148-156). */
int apr_ransac_set_screen(int32_t mode);
/* Launch counts of the two sampling kernels since the library was loaded: out2[0] = k_sample_check, out2[1] =
 * k_sample_screen.  The library silently takes the former when the correspondence table does not fit the LDS, a call has
 * fewer than 2^17 iterations or the 160 KB LDS opt-in failed: a measurement reports the difference of two readings, not
 * the switch it asked for (bench.py `config.ransac_sampling_kernel`). */
int apr_ransac_set_option(int32_t option, int32_t value);
/* ^ A/B and test switches of the matcher without touching the environment (the library reads APR_RANSAC_* once per process;
 * a getenv per call from GIL-free threads would race with the host application's setenv): option 0 screen, 1 count (0 = every
 * survivor scored in fp64), 2 prune (0 = counts over all correspondences), 3 force_rounds (1 = no single-round fast path);
 * value -1 restores the environment's default.  Results do not depend on any of them. */
int apr_ransac_sampling_launches(int64_t* out2);

/* Deal the pairs of a batch over `lanes` streams (1 .. 4): lane 0 is the caller's stream, the others are library-owned
 * streams forked from it by an event and joined back before the result copy, each with its own matching / RANSAC
 * scratch - ordering as seen by the caller is unchanged, results are identical.  Default 1 (or APR_MATCH_LANES): a
 * remedy for a caller with ONE step in flight; with several steps in flight on the caller's own streams the card is
 * already full and more lanes cost throughput.  Set it before apr_match_pose_batch_scratch_bytes (whose answer
 * depends on it) and not while batches are between _enqueue and _finish. */
int apr_match_pose_set_lanes(int32_t lanes);

/* Robust linearised 6-DoF pose (20 IRLS iterations), replaces
 * est_quad_linear_robust (FCGF_APR/util/transform_estimation.py:89-116).
 * pts0/pts1 f32[n,3] paired, weight f32[n] nullable; T_host f32[16]; syncs. */
int apr_irls_pose(const float* pts0, const float* pts1, const float* weight, int64_t n,
                  float* T_host, void* scratch, size_t scratch_bytes, void* stream);
size_t apr_irls_scratch_bytes(int64_t n);

/* Hardest-contrastive mining (forward), replaces the mining + masking + loss arithmetic of
 *   contrastive_hardest_negative_loss (FCGF_APR/lib/trainer.py:400-452, `_hash` util/misc.py:6-18).
 *   pos_f0/pos_f1 f32[p,c]: features of the sampled positive pairs; nn01/nn10 u64[p]: packed
 *   apr_feature_nn results of pos_f0 in F1[sel1] / pos_f1 in F0[sel0]; sorted_pos_keys: ascending
 *   int64 keys i + j*hash_seed of ALL positive pairs.  out6 f64[6] (device) =
 *   {sum pos_loss, p, sum neg_loss0, count0, sum neg_loss1, count1}. */
int apr_contrastive_reduce(const float* pos_f0, const float* pos_f1, int32_t p, int32_t c,
                           const uint64_t* nn01, const uint64_t* nn10, const int64_t* sel0, const int64_t* sel1,
                           const int64_t* pos_ind0, const int64_t* pos_ind1, const int64_t* sorted_pos_keys,
                           int32_t n_keys, int64_t hash_seed, float pos_thresh, float neg_thresh,
                           double* out6, void* stream);

/* ------------------------------------------------------------------------
 * Point-set index builds of the KPConv encoder (Predator_APR)
 * ---------------------------------------------------------------------- */

/* Barycentre grid subsampling of a batch of clouds; replaces
 *   cpp_wrappers.cpp_subsampling.grid_subsampling.subsample_batch(points, batches, sampleDl=)
 *   (Predator_APR/cpp_wrappers/cpp_subsampling/wrapper.cpp:75-82 ->
 *    grid_subsampling/grid_subsampling.cpp:5-107,109-211).
 *   pts f32[n,3] (clouds concatenated), lengths_host i32[nb] (HOST), dl = cell size.
 *   out_pts f32[<=n,3], out_feats (nullable) f32[<=n,fdim], out_lengths_host i32[nb] (HOST).
 *   Barycentres are bit-identical to the reference's; rows come in first-occurrence order of the
 *   cells instead of libstdc++ unordered_map order.  Synchronises the stream. */
size_t apr_grid_subsample_scratch_bytes(int64_t n);
int apr_grid_subsample(const float* pts, int64_t n, const int32_t* lengths_host, int32_t nb, float dl,
                       const float* feats, int32_t fdim, float* out_pts, float* out_feats,
                       int32_t* out_lengths_host, void* scratch, size_t scratch_bytes, void* stream);
/* The same without the host synchronisation: lengths_status_dev = int32[nb + 1] on the device (nb subsampled lengths,
 * then 0 or the out-of-range flag); out_pts / out_feats hold n rows, the result is their first sum(lengths) rows. */
int apr_grid_subsample_async(const float* pts, int64_t n, const int32_t* lengths_host, int32_t nb, float dl,
                             const float* feats, int32_t fdim, float* out_pts, float* out_feats,
                             int32_t* lengths_status_dev, void* scratch, size_t scratch_bytes, void* stream);

/* Batched radius neighbours, sorted by distance; replaces
 *   cpp_wrappers.cpp_neighbors.radius_neighbors.batch_query(queries, supports, q_batches, s_batches, radius=)
 *   (Predator_APR/cpp_wrappers/cpp_neighbors/wrapper.cpp:71-75 -> neighbors/neighbors.cpp:211-333)
 *   followed by the `[:, :limit]` truncation of datasets/dataloader.py:66-68.
 *   out i32[nq, out_ld]; columns [0,width) of every row are written: neighbour indices into the
 *   concatenated supports, ascending distance, padded with ns.  width = min(max_count, limit)
 *   (limit <= 0: no limit) is returned in *width_host.  out == NULL: size query only.
 *   Synchronises the stream. */
size_t apr_radius_scratch_bytes(int64_t nq, int64_t ns);
int apr_radius_neighbors(const float* queries, int64_t nq, const float* supports, int64_t ns,
                         const int32_t* q_lengths_host, const int32_t* s_lengths_host, int32_t nb, float radius,
                         int32_t limit, int32_t* out, int64_t out_ld, int32_t* width_host,
                         void* scratch, size_t scratch_bytes, void* stream);

/* The same table without a host synchronisation, for callers that know the column limit (KPConv's calibrated
 * neighbourhood limits, collate_fn_descriptor :121-180): out i32[nq, >= limit] gets the `limit` nearest neighbours per
 * query (sorted by distance, padded with ns); flags_dev i32[2] on the device: [0] = largest neighbour count of any
 * query (the reference's table width is min([0], limit); columns [that, limit) are all padding), [1] != 0 if a query
 * had more neighbours within the radius than the kernel's candidate buffer holds (error). */
int apr_radius_neighbors_async(const float* queries, int64_t nq, const float* supports, int64_t ns,
                               const int32_t* q_lengths_host, const int32_t* s_lengths_host, int32_t nb, float radius,
                               int32_t limit, int32_t* out, int64_t out_ld, int32_t* flags_dev,
                               void* scratch, size_t scratch_bytes, void* stream);
/* Same, searching the grid the PREVIOUS apr_radius_neighbors_async call left in `scratch` (same supports, lengths,
 * radius and stream): skips the grid build. */
int apr_radius_neighbors_regrid_async(const float* queries, int64_t nq, const float* supports, int64_t ns,
                               const int32_t* q_lengths_host, const int32_t* s_lengths_host, int32_t nb, float radius,
                               int32_t limit, int32_t* out, int64_t out_ld, int32_t* flags_dev,
                               void* scratch, size_t scratch_bytes, void* stream);

/* k nearest neighbours inside one cloud (brute force, k + skip_first <= 16); replaces the dense
 * square_distance + topk(k+1)[..., 1:] of Predator_APR/models/gcn.py:19-23.  out i32[n,k]. */
int apr_knn(const float* pts, int32_t n, int32_t k, int32_t skip_first, int32_t* out, void* stream);

/* ------------------------------------------------------------------------
 * KPConv encoder + overlap attention (Predator_APR/models/blocks.py, gcn.py, architectures.py)
 * ---------------------------------------------------------------------- */

/* out[i] = sum_c x[i,c]  (neighbour-count normaliser of KPConv, blocks.py:369-372). */
/* Dense out = act((in[M, cin] @ W) * scale + shift + residual) on the bf16 MFMA in the 3-way split (fp32-equivalent
 * accuracy, see apr_spconv_ws_fwd_bf3): the unary / bottleneck Linear layers and KPConv's second step
 * [N, 15 cin] x [15 cin, cout] of KPFCNN (Predator_APR/models/blocks.py:347-374, 499-504), and K = 1 convolutions.
 * w_bf3 = apr_spconv_pack_weights_bf3(w, 1, cin, cout).  cin % 64 == 0, cout % 64 == 0; rows of in / out / residual
 * and scale / shift 16-byte aligned. */
int apr_dense_gemm_bf3(const float* in, int64_t ldi, int64_t M, int32_t cin, int32_t cout, const void* w_bf3,
                       const float* scale, const float* shift, const float* residual, int64_t ldr, int32_t relu,
                       float* out, int64_t ldo, void* stream);
/* The same operator for FEW input channels over MANY rows -- the K = 1 layers at the end of the encoders
 * (FCGF_APR/model/resunet.py:126-142: conv1_tr 96 -> 64 / 160 -> 128 and `final` 64 -> 32 / 128 -> 128 on the finest level) --
 * as a row stream past weights that stay in LDS for the life of a workgroup: every row is read once, a wave owns whole rows,
 * so the row normalisation of `normalize_feature` (l2norm != 0: out[j] /= |out[j]|_2, same bits as apr_l2_normalize behind
 * the plain call) is part of the epilogue.  cin in {64, 96, 128, 160, 192}, cout in {32, 64, 128}
 * (apr_dense_rows_bf3_ok); w_bf3 = apr_spconv_pack_weights_bf3(w, 1, cin, cout), which for K = 1 takes cin % 32 == 0 and
 * cout % 16 == 0 (columns padded with zeros to a multiple of 64 inside the image).  Same bits as apr_dense_gemm_bf3 where
 * both take the shape.  apr_dense_rows_bf3_route: what apr_spconv_fwd_batch / apr_resunet_encode do with a K = 1 layer of M
 * rows -- 1: this kernel (an instantiated shape and M >= APR_DENSE_ROWS_MIN, default 32768 rows). */
int32_t apr_dense_rows_bf3_ok(int32_t cin, int32_t cout);
int32_t apr_dense_rows_bf3_route(int64_t M, int32_t cin, int32_t cout);
int apr_dense_rows_bf3(const float* in, int64_t ldi, int64_t M, int32_t cin, int32_t cout, const void* w_bf3,
                       const float* scale, const float* shift, const float* residual, int64_t ldr, int32_t relu,
                       int32_t l2norm, float* out, int64_t ldo, void* stream);
/* out = act(instance_norm(in @ W) (+ residual)) -- the Linear + InstanceNorm1d (+ shortcut) (+ LeakyReLU) every unary /
 * bottleneck layer of KPFCNN is (Predator_APR/models/blocks.py:451-468, 499-504, 653-681) -- in TWO launches: the GEMM leaves
 * the raw product in `out` and the per-tile column sums (fp64, rows in order) in scratch, the apply kernel rebuilds mean /
 * rstd from them and normalises in place.  act_mode 0 none, 1 ReLU, 2 LeakyReLU(negative_slope).  seg_offsets_host (int64
 * [nseg + 1], 0 .. M; NULL / nseg <= 1: one segment): the scan pairs stacked into one forward, each normalised on its own;
 * row tiles never straddle a segment.  Same operator as apr_dense_gemm_bf3 followed by apr_instance_norm_act[_seg]; the
 * statistics group their fp64 partial sums by GEMM tile instead of by 256 rows (results agree to the last bits, not bit for
 * bit).  `out` may not alias `in`; residual may alias neither. */
size_t apr_dense_gemm_bf3_norm_scratch_bytes(int64_t M, int32_t cout, int32_t nseg);
int apr_dense_gemm_bf3_norm_act(const float* in, int64_t ldi, int64_t M, int32_t cin, int32_t cout, const void* w_bf3,
                                float eps, const float* residual, int64_t ldr, int32_t act_mode, float negative_slope,
                                float* out, int64_t ldo, const int64_t* seg_offsets_host, int32_t nseg, void* scratch,
                                size_t scratch_bytes, void* stream);

/* One ResnetBottleneckBlock of the KPConv encoder (Predator_APR/models/blocks.py:596-681; eval mode, use_batch_norm, rigid
 * KPConv, widths multiples of 64) enqueued by ONE call: unary1 (Linear + InstanceNorm + LeakyReLU; absent when in_dim ==
 * mid), KPConv (apr_row_sums, apr_kpconv_weighted, [n_out, n_kp mid] x [n_kp mid, mid]) + InstanceNorm + LeakyReLU, the
 * shortcut (apr_gather_pool max over `nbr` when strided; Linear + InstanceNorm when in_dim != out_dim), unary2 (Linear) +
 * InstanceNorm + shortcut + LeakyReLU -> out.  The same kernels in the same order as the single calls (same bits).
 * Weights: apr_spconv_pack_weights_bf3(K = 1) images of the [cin, cout] matrices (w_kpconv: [n_kp * mid, mid]).  nbr:
 * [n_out, H] rows of the n_in-row input level (the level's `neighbors`, or `pools` of the coarser level when strided);
 * seg_in / seg_out: host row offsets (nseg + 1) of the scan pairs stacked at the input / output level (nseg <= 1: one
 * statistic over all rows).  scratch: apr_kp_resnet_scratch_bytes(desc) device bytes. */
typedef struct apr_kp_resnet_desc {
  const float* x; int64_t ldx; int64_t n_in;
  int32_t in_dim, mid, out_dim, strided;
  const float* q_pts; const float* s_pts; int64_t n_out;
  const int32_t* nbr; int32_t H; int32_t n_kp;
  const float* kernel_points; float extent; float eps; float slope; int32_t nseg;
  const void* w_unary1; const void* w_kpconv; const void* w_unary2; const void* w_shortcut;
  const int64_t* seg_in; const int64_t* seg_out;
  float* out; int64_t ldo;
  void* scratch; size_t scratch_bytes;
} apr_kp_resnet_desc;
size_t apr_kp_resnet_scratch_bytes(const apr_kp_resnet_desc* desc);
int apr_kp_resnet_block(const apr_kp_resnet_desc* desc, void* stream);

/* Host-only (no device needed): ONE round of NumPy's legacy `RandomState.choice(n, size, replace=False, p=p)` loop
 * (Predator_APR/lib/tester.py:83-92 draws its 5000 interest points per cloud with it).  The caller owns the RNG: per
 * round it passes k = size - n_uniq uniforms from `random_sample`.  p: float64 working copy (zeroed in place for found
 * indices; n_new_zeroed = how many of found[0..n_uniq) were added by the previous round); cdf: double[n] work; stamp:
 * int32[n] work, zero before round 1; round_id = 1, 2, ...  Returns the new n_uniq (loop until it reaches size), or a
 * negative error code. */
int64_t apr_weighted_choice_round(double* p, int64_t n, int64_t* found, int64_t n_uniq, int64_t n_new_zeroed,
                                  const double* x, int64_t k, double* cdf, int32_t* stamp, int32_t round_id);

int apr_row_sums(const float* x, int64_t ld, int64_t n, int32_t c, float* out, void* stream);

/* KPConv step 1 (blocks.py:269-289,326-329,347-372): kernel-point correlation
 *   wf[q, k*cin + c] = ( sum_h max(0, 1 - |s[nbr[q,h]] - q - kp[k]| / extent) * x[nbr[q,h], c] ) / num[q]
 *   num[q] = max(#{h : rowsum[nbr[q,h]] > 0}, 1); nbr entries >= ns (or < 0) are shadow neighbours.
 * Step 2 is apr_spconv_fwd(wf, nbr = NULL, K = 1, cin = 15*cin, cout) with the [15,cin,cout] weights. */
int apr_kpconv_weighted(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns, const int32_t* nbr,
                        int32_t H, const float* x, int64_t ldx, int32_t cin, const float* kernel_points,
                        int32_t n_kp, float extent, const float* rowsum, float* wf, int64_t ldwf, void* stream);
/* Backward of apr_kpconv_weighted with respect to the neighbour features (training path, SURVEY 8(f) next-3;
 * Predator_APR/models/blocks.py:326-374 through lib/trainer.py:142-280): dx[nbr[q,h], c] += (1 / num_q) * sum_k
 * w[q,k,h] * dwf[q, k*cin + c] with float atomics (dx must be zero-initialised; f32 [ns, cin]).  The weight gradient of
 * the layer is apr_spconv_wgrad on (wf, dout) with the identity map; d wf is a dense apr_spconv_fwd with the transposed
 * weights.  rowsum: apr_row_sums of the forward's x. */
int apr_kpconv_dfeat(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns, const int32_t* nbr, int32_t H,
                     const float* dwf, int64_t lddwf, int32_t cin, const float* kernel_points, int32_t n_kp,
                     float extent, const float* rowsum, float* dx, int64_t lddx, void* stream);

/* The same gradient without float atomics (deterministic): apr_kpconv_dfeat_contrib writes the contribution of every
 * (query, neighbour) entry to contrib f32 [nq * H, cin]; apr_reverse_table_build sorts the table's flat positions by the
 * support row they point at (stable: ties in ascending position; once per neighbour table) and apr_reverse_gather sums
 * every support row's contributions in that order: dx f32 [ns, cin] is WRITTEN, not accumulated. */
int apr_kpconv_dfeat_contrib(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns, const int32_t* nbr, int32_t H,
                             const float* dwf, int64_t lddwf, int32_t cin, const float* kernel_points, int32_t n_kp,
                             float extent, const float* rowsum, float* contrib, void* stream);
size_t apr_reverse_table_scratch_bytes(int64_t nq, int32_t H, int64_t ns);
int apr_reverse_table_build(const int32_t* nbr, int64_t nq, int32_t H, int64_t ns, int32_t* rev_t, int32_t* start,
                            void* scratch, size_t scratch_bytes, void* stream);
int apr_reverse_gather(const float* src, int32_t c, const int32_t* rev_t, const int32_t* start, int64_t ns, float* out,
                       int64_t ldo, void* stream);
/* The same over the flat positions [t_lo, t_hi) only (src f32 [t_hi - t_lo, c]: their rows), continuing the sums in `out`
 * when accumulate != 0: chunk after chunk in ascending t this is bit for bit one apr_reverse_gather over the whole table
 * with a contribution buffer the size of ONE chunk (the finest KPConv level of one pair needs 400 MB unchunked). */
int apr_reverse_gather_range(const float* src, int32_t c, const int32_t* rev_t, const int32_t* start, int64_t ns,
                             int64_t t_lo, int64_t t_hi, int32_t accumulate, float* out, int64_t ldo, void* stream);

/* mode 0: max_pool(x, inds) (blocks.py:86-102); mode 1: closest_pool(x, inds) (blocks.py:71-83).
 * Index ns addresses an implicit all-zero shadow row. */
int apr_gather_pool(const float* x, int64_t ldx, int64_t ns, int32_t c, const int32_t* inds, int32_t H,
                    int64_t nq, int32_t mode, float* out, int64_t ldo, void* stream);

/* Training path of the pooling gathers (Predator_APR/models/blocks.py:71-102 under lib/trainer.py:142-280): the forward of
 * max_pool that also records which neighbour gave each maximum (amax u8 [nq, c]: the first h at the maximum, H <= 255), and
 * the input gradient of max_pool (mode 0) / closest_pool (mode 1) as a gather over the reverse table of the same index
 * tensor (apr_reverse_table_build): dx[s, c] = sum of dout[q, c] over the entries (q, h) that point at s and were the
 * maximum (mode 0) / the first neighbour (mode 1), in table order -- deterministic, no float atomics. */
int apr_gather_pool_argmax(const float* x, int64_t ldx, int64_t ns, int32_t c, const int32_t* inds, int32_t H, int64_t nq,
                           float* out, int64_t ldo, uint8_t* amax, void* stream);
int apr_gather_pool_backward(const float* dout, int64_t lddo, int32_t c, const int32_t* rev_t, const int32_t* start,
                             int64_t ns, int32_t H, const uint8_t* amax, int32_t mode, float* dx, int64_t lddx, void* stream);

/* get_graph_feature (gcn.py:9-35) as rows: out[(i*k + j), :] = [f_i, f_knn(i,j) - f_i], f32 [n*k, 2c]. */
int apr_edge_features(const float* f, int64_t ldf, int32_t n, int32_t c, const int32_t* knn, int32_t k,
                      float* out, void* stream);

/* Training path (gcn.py:9-35 under Predator_APR/lib/trainer.py:142-280): the input gradient of apr_edge_features from
 * de f32 [n*k, 2c], first half -- d_center[i] = sum_j (de[(i,j), :c] - de[(i,j), c:]) and the neighbours' terms as
 * contribution rows contrib f32 [n*k, c] = de[:, c:]; apr_reverse_gather_range over the reverse table of knn (ns = n,
 * accumulate = 1) then adds every row's contributions in table order: deterministic, no float atomics. */
int apr_edge_features_backward(const float* de, int32_t n, int32_t c, int32_t k, float* d_center, int64_t ldd,
                               float* contrib, void* stream);

/* out[i,:] = max_j leaky(y[i*k+j,:] * scale + shift)  (InstanceNorm2d + LeakyReLU(0.2) + max, gcn.py:67-72). */
int apr_group_max(const float* y, int64_t ldy, int32_t n, int32_t k, int32_t c, const float* scale,
                  const float* shift, float slope, float* out, int64_t ldo, void* stream);

/* Multi-head softmax attention, channel c = d*heads + h (gcn.py:94-116). q f32[n,dim*heads], k/v f32[m,...]. */
int apr_mha(const float* q, const float* k, const float* v, int32_t n, int32_t m, int32_t dim, int32_t heads,
            float* out, void* stream);

/* The same attention for the TRAINING path (Predator_APR/lib/trainer.py:142-280): the forward keeps the probabilities
 * P f32 [heads, n, m]; the backward turns them and dout f32 [n, dim*heads] into dq, dk, dv (dS f32 [heads, n, m]: scratch).
 * Fixed summation orders, no float atomics: the same bits every run.  No limit on m. */
int apr_mha_train_forward(const float* q, const float* k, const float* v, int32_t n, int32_t m, int32_t dim, int32_t heads,
                          int32_t vdim, float* P, float* out, void* stream);
int apr_mha_train_backward(const float* q, const float* k, const float* v, const float* P, const float* dout, int32_t n,
                           int32_t m, int32_t dim, int32_t heads, int32_t vdim, float* dS, float* dq, float* dk, float* dv,
                           void* stream);
/* vdim: channels per head of v / out / dout (0: = dim).  vdim = 1 with heads = 1 is the cross-saliency
 * softmax(<a_i, b_j> / T) @ s of Predator_APR/models/architectures.py:176-181 (q = a * sqrt(dim) / T). */

/* apr_mha with q / k / v handed over HEAD-MAJOR (channel h*dim + d: the caller permutes the output channels of the
 * three projections, gcn.py:101-108) and out in apr_mha's interleaved layout; fp32 MFMA, dim = 64 only, q / k / v
 * 16-byte aligned, no limit on m. */
int apr_mha_headmajor(const float* q, const float* k, const float* v, int32_t n, int32_t m, int32_t dim,
                      int32_t heads, float* out, void* stream);

/* The overlap-attention module of one scan pair (GCN.forward, Predator_APR/models/gcn.py:171-205: 'self' = SelfAttention
 * :38-77 on each cloud, 'cross' = AttentionalPropagation :119-128 in both directions with the residual add) as ONE call.
 * The module issued 62 library calls and 6 concatenations per pair on ~1.4 k points -- the largest share of the 110 calls
 * per pair that kept KPFCNN's scheduler thread, not the GPU, at its limit.  Same kernels, arguments and order as the
 * modules' own (apr_knn, apr_edge_features, apr_dense_gemm_bf3, apr_norm_params, apr_group_max, apr_instance_norm_act,
 * apr_mha_headmajor, apr_affine_act): same bits; the concatenations become column slices of one buffer.
 *   c % 64 == 0, c / heads == 64, k + 1 <= 16; weights: apr_spconv_pack_weights_bf3(K = 1) images ([cin, cout] as the
 *   layers' own Linear / Conv1d weights transposed; wq / wk / wv with their output channels head-major, as
 *   apr_mha_headmajor reads them); biases f32, 16-byte aligned, nullable.  x0 [n0, c] / x1 [n1, c] in, out0 / out1 out. */
#define APR_GCN_MAX_LAYERS 8
typedef struct apr_gcn_layer {
  int32_t kind;                   /* 0: self attention, 1: cross attention */
  int32_t k, heads;               /* self: neighbours of the kNN graph; cross: attention heads */
  float eps1, eps2, eps3;         /* self: InstanceNorm eps of in1 / in2 / in3; cross: eps1 = mlp[1].eps */
  const void *w1, *w2, *w3;       /* self: conv1 [2c, c], conv2 [2c, 2c], conv3 [4c, c]; cross: w1 = mlp[0] [2c, 2c], w2 = mlp[3] [2c, c] */
  const float *b1, *b2;           /* cross: biases of mlp[0], mlp[3] */
  const void *wq, *wk, *wv, *wm;  /* cross: the three projections (head-major) and merge, [c, c] */
  const float *bq, *bk, *bv, *bm;
} apr_gcn_layer;
typedef struct apr_gcn_desc {
  int32_t n_layers, c;
  apr_gcn_layer layer[APR_GCN_MAX_LAYERS];
} apr_gcn_desc;
size_t apr_gcn_scratch_bytes(const apr_gcn_desc* d, int32_t n0, int32_t n1);
int apr_gcn_forward(const apr_gcn_desc* d, const float* pts0, int32_t n0, const float* pts1, int32_t n1, const float* x0,
                    int64_t ldx0, const float* x1, int64_t ldx1, float* out0, int64_t ldo0, float* out1, int64_t ldo1,
                    void* scratch, size_t scratch_bytes, void* stream);

/* out[i] = sum_j softmax_j(<a_i, b_j> / temperature) * w[j]  (cross saliency, architectures.py:176-181). */
int apr_softmax_matvec(const float* a, const float* b, const float* w, int32_t n, int32_t m, int32_t c,
                       float temperature, float* out, void* stream);
/* The same with b handed over transposed (bt [c, m] row-major): coalesced reads, several queries per workgroup; the
 * result has the same bits. */
int apr_softmax_matvec_bt(const float* a, const float* bt, const float* w, int32_t n, int32_t m, int32_t c,
                          float temperature, float* out, void* stream);
/* The same on the fp32 MFMA: a and b row-major as in apr_softmax_matvec, 16-byte aligned, c in {32, 64, 128, 256}, any
 * m; equal to fp32 summation order. */
int apr_softmax_matvec_mfma(const float* a, const float* b, const float* w, int32_t n, int32_t m, int32_t c,
                            float temperature, float* out, void* stream);

/* y[i] = clamp(sigmoid(x[i*ldx]), 0, 1), NaN/Inf -> 0  (architectures.py:131-134,203-207). */
int apr_score_head(const float* x, int64_t ldx, int64_t n, float* y, void* stream);

/* ------------------------------------------------------------------------
 * Adjacent components: APG aggregation and Chamfer 1-NN sums
 * ---------------------------------------------------------------------- */

/* out = pts @ R^T + t in fp32, T16_dev = row-major 4x4 on the device
 * (apply_transform, FCGF_APR/lib/complement_data_loader.py:65-70). */
int apr_transform_points(const float* pts, int64_t n, const float* T16_dev, float* out, void* stream);

/* Keep the points with |p|^2 < max_i |key_i|^2, order preserved
 * (crop of the aggregated complement frames, complement_data_loader.py:620-628).
 * out f32[<=n,3]; *n_out_dev device counter. */
size_t apr_crop_scratch_bytes(int64_t n);
int apr_crop_to_radius(const float* key_pts, int64_t n_key, const float* pts, int64_t n, float* out,
                       int32_t* n_out_dev, void* scratch, size_t scratch_bytes, void* stream);

/* *out_dev (f64) = sum_i min_j |a_i - b_j|^2 -- one direction of chamferdist.ChamferDistance()(a, b) as used by
 * chamfer_distance (FCGF_APR/lib/complement_trainer.py:188-196).  scratch >= 4*n bytes. */
int apr_chamfer_sum(const float* a, int64_t n, const float* b, int64_t m, double* out_dev, void* scratch,
                    size_t scratch_bytes, void* stream);

/* Exact 1-NN of every row of a [n,3] among the rows of b [m,3] in fp32, (dx^2 + dy^2) + dz^2 with every operation rounded
 * (no contraction): out_packed[i] = (bits(d^2) << 32) | j, ties -> the smallest j.  sum_dev (may be NULL): f64 sum of
 * the n minima in a fixed order (bit-reproducible).  The arg-min is what the backward of the Chamfer term needs
 * (Predator_APR/lib/trainer.py:131-140, 179-183; FCGF_APR/lib/complement_trainer.py:188-196, 446-448): the gradient
 * of sum_i min_j |a_i - b_j|^2 reaches a_i and b_argmin(i) only.
 * cell > 0: a uniform grid over b answers the queries whose neighbour lies within ~6.5 cells (ring after ring), the rest
 * fall through to the full search -- the same bits as cell = 0 (brute force, scratch unused) for every input.
 * A good `cell` is several voxel sizes (the clouds here carry one point per 0.3 m voxel; the Python layer passes 1.2 m). */
size_t apr_nn3_scratch_bytes(int64_t n, int64_t m);
int apr_nn3(const float* a, int64_t n, const float* b, int64_t m, float cell, uint64_t* out_packed, double* sum_dev,
            void* scratch, size_t scratch_bytes, void* stream);
/* nb independent searches in ONE call (the clouds of a training batch: FCGF_APR/lib/complement_trainer.py:424-449 runs the
 * Chamfer term cloud by cloud): rows a_offsets_host[s] .. [s + 1] of a are searched among rows b_offsets_host[s] .. [s + 1] of
 * b only (the grid's cells carry the cloud index).  out_packed[i] = (bits(d^2) << 32) | j with j the GLOBAL row of b;
 * sums_dev[nb]: per-cloud sums of the minima (fixed order), multiplied by sum_scale_host[s] when that is given (1 / rows: the
 * mean the Chamfer term divides by, without a host->device copy).  Same bits per cloud as apr_nn3 on that cloud alone.
 * scratch: apr_nn3_scratch_bytes(total rows of a, total rows of b); cell > 0. */
int apr_nn3_batch(const float* a, const int64_t* a_offsets_host, const float* b, const int64_t* b_offsets_host, int32_t nb,
                  float cell, uint64_t* out_packed, double* sums_dev, const double* sum_scale_host, void* scratch,
                  size_t scratch_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* APR_HIP_H */
