"""One scan pair through the FCGF_APR hot path on the GPU: voxelise -> encode -> match -> RANSAC/SVD.

This is the per-pair body of the reference's evaluation loop
(FCGF_APR/scripts/test_apr.py:111-163) with its data-loader voxelisation
(lib/complement_data_loader.py:788-812) pulled onto the device:

    sel      = sparse_quantize(xyz / voxel_size)            (GPU voxel hash)
    F0, F1   = model(SparseTensor(ones, coords))            (both frames in ONE batched
                                                             tensor; eval-mode BN makes this
                                                             identical to two separate calls)
    corr     = feature-space NN of F0 rows in F1            (fused arg-min)
    T        = RANSAC(4-pt, edge 0.9, dist = voxel_size) + Kabsch

Everything between the raw xyz upload and the 4x4 pose stays in HBM; host syncs:
one for the voxel counts of the two frames, one for the coordinate pyramid sizes,
one for the RANSAC result.
"""
from __future__ import annotations

import os

import torch

from .. import MinkowskiEngine as ME
from .. import _host, ops

# the front end of a step (voxelise -> de-duplicated map -> compact table -> coarser maps) through ONE library call
# (apr_voxel_pyramid); 0: tensor by tensor from Python (A/B switch)
FRONT_END_CALL = os.environ.get("APR_FRONT_END_CALL", "1") != "0"


class PairRegistration:
    def __init__(self, model, voxel_size=0.3, ransac_iters=4000000, edge_length=0.9, distance_factor=1.0):
        _host.limit_cpu_threads()
        self.model = model.eval()
        self.voxel_size = float(voxel_size)
        self.ransac_iters = int(ransac_iters)
        self.edge_length = float(edge_length)
        self.distance_threshold = float(voxel_size) * distance_factor  # test_apr.py:149
        # optional: called as feature_hook(F [sum n, C], rows_per_frame, pairs) -> F right after the batched encode
        # (bench.py: descriptors with a controlled share of true matches stand in for a trained checkpoint's output; the
        # encoder still runs)
        self.feature_hook = None
        self._ones_cache = {}
        # how `register_batch` waits for its two fetches (ops.wait_event): None = the process default (sleeping poll: one CPU
        # per rank with three steps in flight); "sync" = hipEventSynchronize (a spinning core, lowest wake-up latency: for a
        # caller that has ONE step in flight, like the reference's own loop)
        self.fetch_wait = None

    @torch.no_grad()
    def voxelize_pair(self, xyz0, xyz1):
        """-> (coords int32 [N0+N1,4] with batch ids 0/1, pts0, pts1, n0, n1)."""
        maps = []
        for b, xyz in enumerate((xyz0, xyz1)):
            c = ops.voxelize(xyz, self.voxel_size, b)
            maps.append(ops.build_map(c, want_first=True))
        ops.finalize_maps(maps)                      # one sync for both frames
        coords = torch.cat([maps[0].coords, maps[1].coords], 0)
        return coords, xyz0[maps[0].first], xyz1[maps[1].first], maps[0].n, maps[1].n

    @torch.no_grad()
    def encode_pair(self, coords, n0):
        feats = torch.ones((coords.shape[0], 1), dtype=torch.float32, device=coords.device)
        out = self.model(ME.SparseTensor(feats, coordinates=coords, unit_features=True)).F
        return out[:n0], out[n0:]

    @torch.no_grad()
    def voxelize_batch(self, clouds):
        """Frames -> (coordinate manager holding the whole 4-level pyramid, rows per frame, first input point of
        every row).  ALL frames are voxelised by ONE hash build over the concatenated points (batch ids 0..len-1
        in the key; rows come out in first-occurrence order = frame order), that map is adopted as the stride-1
        coordinate map, the three coarser maps are chained behind it with device-side row counts, and a single
        host sync fetches every size (12 + 4 builds and 2 syncs per step become 4 builds and 1 sync)."""
        if len(clouds) > 1023:
            raise ValueError("at most 1023 frames per batch (10-bit batch index in the voxel key; 1023 is reserved for the empty-slot key)")
        cm, counts, pts_all, first, offs = ops.drive(self._front_end_phases(clouds), wait=self.fetch_wait)
        return cm, counts, first, offs, pts_all

    def _front_end_phases(self, clouds):
        """The front end of a step as a generator (yields its ops.PendingFetch): -> (coordinate manager with the 4-level
        pyramid, rows per frame, representative point of every row, first input point of every row, frame offsets).  Up to
        ops.MAX_FRAMES frames leave through ONE library call over one arena (ops.VoxelPyramid); more frames, or a batch whose
        voxels outnumber a quarter of its points, go tensor by tensor (same kernels, same bits)."""
        if len(clouds) <= ops.MAX_FRAMES and FRONT_END_CALL:
            vp = ops.VoxelPyramid(clouds, self.voxel_size)
            yield vp.pending
            res = vp.finish()
            if res is not None:
                maps, counts, bbox, pts_all, first, counters = res
                cm = ME.CoordinateManager.from_maps(maps, counters)
                cm.set_bbox(bbox)
                return cm, counts, pts_all, first, vp.offsets
        coords_all, offs_dev, offs, gather = self._voxelize_frames(clouds)
        m = ops.build_map(coords_all, want_first=True)
        counts_dev = ops.segment_counts(m, offs_dev)
        bbox_dev = ops.coords_bbox(coords_all)         # for conv1 on occupancy (ops.occ_conv); fetched with the sizes
        pts_all = gather(m)                            # enqueued before the fetch: the host does not wait for it
        cm = ME.CoordinateManager(base_map=m)
        pending = cm.build_pyramid_async([2, 4, 8], extras=[counts_dev, bbox_dev])
        yield pending
        counts, bbox = pending.finish()
        cm.set_bbox(bbox)
        return cm, [int(c) for c in counts], pts_all[:m.n], m.first, offs

    def _voxelize_frames(self, clouds):
        """-> (coords int32 [sum n, 4] with the frame index as batch id, offsets on the device, offsets as a list, gather):
        `gather(m)` returns the input point behind every row of the map `m` built over these coords (f32 [rows, 3], rows =
        the allocation's upper bound until the map is finalised).  Up to ops.MAX_FRAMES frames go through the library's
        frame-table kernels (no concatenated copy of the points, no torch op); beyond that the frames are concatenated."""
        if len(clouds) <= ops.MAX_FRAMES:
            coords_all, offs_dev, offs = ops.voxelize_frames(clouds, self.voxel_size)
            return coords_all, offs_dev, offs, lambda m: ops.gather_frame_points(clouds, m)
        offs = [0]
        for c in clouds:
            offs.append(offs[-1] + int(c.shape[0]))
        xyz_all = torch.cat(clouds)
        offs_dev = torch.tensor(offs, dtype=torch.int64).to(clouds[0].device, non_blocking=True)
        coords_all = ops.voxelize_segments(xyz_all, self.voxel_size, offs_dev)
        return coords_all, offs_dev, offs, lambda m: xyz_all[m.first]

    @torch.no_grad()
    def register_batch_phases(self, pairs, seeds=None):
        """`register_batch` as a generator for a single-threaded scheduler (`run_pipelined`): it yields an
        `ops.PendingFetch` wherever the synchronous version blocks the host (the map sizes, the poses), enqueues
        everything else on the CURRENT stream, and returns the list of (T, info)."""
        if seeds is None:
            seeds = range(len(pairs))
        clouds = [c for p in pairs for c in p]
        if len(clouds) > 1023:
            raise ValueError("at most 1023 frames per batch (10-bit batch index in the voxel key; 1023 is reserved "
                             "for the empty-slot key)")
        cm, counts, pts_all, _, _ = yield from self._front_end_phases(clouds)
        F = self.encode_batch(cm)
        if self.feature_hook is not None:
            F = self.feature_hook(F, counts, pairs)
        o = [0]
        for n in counts:
            o.append(o[-1] + n)
        pts = [pts_all[o[b]:o[b + 1]] for b in range(len(clouds))]
        pending = ops.match_pose_batch_async([F[o[2 * i]:o[2 * i + 1]] for i in range(len(pairs))],
                                             [F[o[2 * i + 1]:o[2 * i + 2]] for i in range(len(pairs))],
                                             pts[0::2], pts[1::2], self.distance_threshold, self.edge_length,
                                             self.ransac_iters, seeds=list(seeds))
        yield pending
        out = []
        for i, (T, info) in enumerate(pending.finish()):
            info.update(n0=counts[2 * i], n1=counts[2 * i + 1])
            out.append((T, info))
        return out

    def _ones(self, n, dev):
        """[n, 1] view of a cached all-ones column (FCGF's input features, complement_data_loader.py:805-812): no fill per
        step.  Read-only by contract (SparseTensor(unit_features=True))."""
        cache = self._ones_cache.get(dev)
        if cache is None or cache.shape[0] < n:
            cache = self._ones_cache[dev] = torch.ones((max(n, 1 << 18), 1), dtype=torch.float32, device=dev)
        return cache[:n]

    @torch.no_grad()
    def encode_batch(self, cm):
        n = cm.size(1)
        feats = self._ones(n, cm.device)
        return self.model(ME.SparseTensor(feats, coordinate_map_key=ME.CoordinateMapKey(1), coordinate_manager=cm,
                                          unit_features=True)).F

    @torch.no_grad()
    def register_batch(self, pairs, seeds=None):
        """B independent pairs through ONE encoder call (a batched sparse tensor of 2B frames, batch ids 0..2B-1;
        with eval-mode BN every frame's features equal those of a separate call to fp32 summation-order noise),
        then NN + RANSAC per pair.

        Larger launches amortise the ~10-20 us latency floor of every sparse-conv / index kernel and the host cost
        of an encoder call over B pairs.  -> list of (T [4,4] float64, info)."""
        return ops.drive(self.register_batch_phases(pairs, seeds), wait=self.fetch_wait)     # one code path: the pipelined step, waited for

    @torch.no_grad()
    def __call__(self, xyz0, xyz1, seed=0):
        coords, pts0, pts1, n0, n1 = self.voxelize_pair(xyz0, xyz1)
        F0, F1 = self.encode_pair(coords, n0)
        corr = ops.feature_nn(F0.contiguous(), F1.contiguous())
        T, info = ops.ransac_pose(pts0.contiguous(), pts1.contiguous(), corr, self.distance_threshold,
                                  self.edge_length, self.ransac_iters, seed)
        info.update(n0=n0, n1=n1)
        return T, info


def run_pipelined(make_step, indices, streams):
    """ONE host thread, len(streams) steps in flight.  `make_step(i)` returns a generator that enqueues work on the
    current stream and yields `ops.PendingFetch` objects where it needs bytes back on the host (see
    `PairRegistration.register_batch_phases`).  Slot s runs its steps on streams[s]; whenever a slot's fetch has landed
    its generator is resumed, otherwise the next slot gets the host; the host only blocks when no slot can move.
    -> ({i: result}, [(i, completion time)]).

    The alternative to one Python thread per stream.  Which is faster depends on where the step's host time goes: the
    Predator batch is Python between its fetches (GIL-bound: one scheduler 272-279 pairs/s, three threads 208), the FCGF
    step is library calls, which release the GIL (three threads 2075-2116 pairs/s, one scheduler 1932-1967: a stream
    never waits for another step's 1 ms enqueue phase).  bench.py defaults accordingly."""
    import time
    it = iter(indices)
    slots = [None] * len(streams)          # (step index, generator, pending fetch)
    results, done_at = {}, []
    exhausted = False

    host_busy = 0.0                        # seconds spent inside the generators (enqueueing), not waiting

    def advance(s):
        nonlocal host_busy
        i, gen, _ = slots[s]
        t_in = time.perf_counter()
        with torch.cuda.stream(streams[s]):
            try:
                slots[s] = (i, gen, next(gen))
            except StopIteration as stop:
                results[i] = stop.value
                done_at.append((i, time.perf_counter()))
                slots[s] = None
        host_busy += time.perf_counter() - t_in

    while True:
        moved = False
        for s in range(len(streams)):
            if slots[s] is None:
                if exhausted:
                    continue
                try:
                    i = next(it)
                except StopIteration:
                    exhausted = True
                    continue
                slots[s] = (i, make_step(i), None)
                advance(s)
                moved = True
            elif slots[s][2].event.query():
                advance(s)
                moved = True
        live = [sl for sl in slots if sl is not None]
        if not live and exhausted:
            run_pipelined.last_host_busy_s = host_busy      # read by bench.py (host enqueue time per step)
            return results, done_at
        if not moved and live:      # nothing ready: sleep one poll interval (ops.wait_event: hipEventSynchronize spins)
            if ops.FETCH_WAIT == "sync":
                min(live, key=lambda sl: sl[0])[2].event.synchronize()
            else:
                ops.fine_sleep_slack()
                time.sleep(ops.FETCH_POLL_S)
