"""One scan pair through the Predator_APR hot path on the GPU: grid subsample -> KPConv index pyramid ->
KPFCNN (encoder + overlap attention + decoder) -> score-weighted sampling -> RANSAC/SVD.

This is the per-pair body of the reference's tester (Predator_APR/lib/tester.py:48-110) with its data-loader work
(datasets/kitti.py:466-471 voxel_down_sample, datasets/dataloader.py collate_fn_descriptor) pulled onto the device.
The counterpart of apr_amd.fcgf.pipeline.PairRegistration.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _host
from . import point_ops
from .datasets.dataloader import calibrate_neighbors, collate_fn_descriptor   # noqa: F401  (re-exported)
from .lib import benchmark_utils as BU


_DRAW_POOL = None


def _draw_pool():
    """One helper thread per process for the host-RNG draws (see register_batch_phases)."""
    global _DRAW_POOL
    if _DRAW_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _DRAW_POOL = ThreadPoolExecutor(max_workers=1, thread_name_prefix="apr-draw")
    return _DRAW_POOL


class HostPending:
    """A host job in flight on a helper thread, with the face of ops.PendingFetch: `.event.query()` / `.wait()` / `.finish()`
    (apr_amd.fcgf.pipeline.run_pipelined and ops.drive treat both alike)."""

    def __init__(self, future):
        self._future = future
        self.event = self

    def query(self):
        return self._future.done()

    def synchronize(self):
        self._future.result()

    def wait(self, mode=None):
        self._future.result()

    def finish(self):
        return self._future.result()


class PredatorRegistration:
    def __init__(self, model, config, neighborhood_limits, voxel_size=0.3, n_points=5000, distance_threshold=0.3,
                 max_iteration=50000, max_validation=1000):
        """`neighborhood_limits`: per-level caps of the radius neighbourhoods (the reference calibrates them once per
        dataset, datasets/dataloader.py:calibrate_neighbors); `n_points`: interest points kept per frame
        (configs/test/kitti.yaml n_points)."""
        _host.limit_cpu_threads()
        self.model = model.eval()
        self.config = config
        self.limits = list(neighborhood_limits)
        self.voxel_size = float(voxel_size)
        self.n_points = int(n_points)
        self.distance_threshold = float(distance_threshold)
        self.max_iteration, self.max_validation = int(max_iteration), int(max_validation)

    @torch.no_grad()
    def encode(self, xyz0, xyz1):
        """-> (src points, tgt points, features [n0+n1, C], overlap scores, saliency scores)."""
        dev = xyz0.device
        lens = np.array([len(xyz0), len(xyz1)], np.int32)
        pts, lens = point_ops.grid_subsample(torch.cat([xyz0, xyz1]), lens, self.voxel_size)
        src, tgt = pts[:lens[0]], pts[lens[0]:]
        ones = lambda p: torch.ones((len(p), 1), device=dev)
        batch = collate_fn_descriptor([(src, tgt, ones(src), ones(tgt))], self.config, self.limits)
        feats, overlap, saliency = self.model(batch)
        return src, tgt, feats, overlap, saliency

    @torch.no_grad()
    def encode_batch(self, pairs):
        """Several pairs through ONE collate and ONE network forward (their clouds stacked at every level; see
        `collate_fn_descriptor`).  -> per pair (src points, tgt points, features, overlap, saliency), each what
        `encode` returns for that pair alone."""
        dev = pairs[0][0].device
        clouds = [c for pair in pairs for c in pair]
        lens = np.array([len(c) for c in clouds], np.int32)
        pts, lens = point_ops.grid_subsample(torch.cat(clouds), lens, self.voxel_size)
        ends = np.cumsum(lens)
        sub = [pts[e - n:e] for e, n in zip(ends, lens)]
        ones = lambda p: torch.ones((len(p), 1), device=dev)
        items = [(sub[2 * i], sub[2 * i + 1], ones(sub[2 * i]), ones(sub[2 * i + 1])) for i in range(len(pairs))]
        feats, overlap, saliency = self.model(collate_fn_descriptor(items, self.config, self.limits))
        out = []
        for i in range(len(pairs)):
            a, b = int(ends[2 * i] - lens[2 * i]), int(ends[2 * i + 1])
            out.append((sub[2 * i], sub[2 * i + 1], feats[a:b], overlap[a:b], saliency[a:b]))
        return out

    @torch.no_grad()
    def register_batch(self, pairs, seeds=None):
        """[(xyz0, xyz1), ...] -> [(T, info), ...]: one stacked encode, then the reference's per-pair tail
        (lib/tester.py:80-100: score-weighted draws on the host RNG seeded per pair, feature NN, RANSAC)."""
        from .. import ops
        return ops.drive(self.register_batch_phases(pairs, seeds))     # one code path: the pipelined batch, waited for

    @torch.no_grad()
    def register_batch_phases(self, pairs, seeds=None):
        """`register_batch` as a generator for a single-thread scheduler (apr_amd.fcgf.pipeline.run_pipelined): every
        place where the blocking form waits for bytes from the device -- the cloud lengths after each grid subsample
        (1 + 4 per batch), the neighbour-table widths, the sampling weights, the RANSAC results -- yields an
        ops.PendingFetch instead; all device work goes to the CURRENT stream.  Returns [(T, info), ...]."""
        from .. import ops
        from .datasets.dataloader import collate_phases
        seeds = list(range(len(pairs))) if seeds is None else list(seeds)
        dev = pairs[0][0].device
        clouds = [c for pair in pairs for c in pair]
        sub0 = point_ops.grid_subsample_async(torch.cat(clouds), np.array([len(c) for c in clouds], np.int32),
                                              self.voxel_size)
        yield sub0.fetch
        pts, lens = sub0.finish()
        ends = np.cumsum(lens)
        sub = [pts[e - n:e] for e, n in zip(ends, lens)]
        ones = lambda p: torch.ones((len(p), 1), device=dev)
        items = [(sub[2 * i], sub[2 * i + 1], ones(sub[2 * i]), ones(sub[2 * i + 1])) for i in range(len(pairs))]
        batch = yield from collate_phases(items, self.config, self.limits)
        feats, overlap, saliency = self.model(batch)
        wfetch = ops.PendingFetch(overlap * saliency, lambda host: host.copy())
        yield wfetch
        w_all = torch.from_numpy(wfetch.finish())
        # the score-weighted draws of all pairs (NumPy's legacy choice on each pair's own RandomState: 0.39 ms of host time
        # per pair, most of it inside the library's host function, which releases the GIL) run on a helper thread while the
        # scheduler thread serves the other batches
        spans = [(int(ends[2 * i] - lens[2 * i]), int(lens[2 * i]), int(lens[2 * i + 1])) for i in range(len(pairs))]

        def draw_all():
            out_ = []
            for (a, n0, n1), seed in zip(spans, seeds):
                rng = np.random.RandomState(seed)
                i0 = BU.draw_by_score(n0, w_all[a:a + n0], self.n_points, rng)
                i1 = BU.draw_by_score(n1, w_all[a + n0:a + n0 + n1], self.n_points, rng)
                out_.append((i0, i1))
            return out_

        dfetch = HostPending(_draw_pool().submit(draw_all))
        yield dfetch
        drawn = dfetch.finish()
        raws, n01 = [], []
        for i, seed in enumerate(seeds):
            a, n0, n1 = spans[i]
            b = a + n0 + n1
            s_p, s_f = BU.take_drawn(sub[2 * i], feats[a:a + n0], drawn[i][0])
            t_p, t_f = BU.take_drawn(sub[2 * i + 1], feats[a + n0:b], drawn[i][1])
            corr = ops.feature_nn(s_f.contiguous(), t_f.contiguous())
            raws.append(ops.ransac_pose_geometric_async(s_p, t_p, corr, self.distance_threshold, 0.9, self.max_iteration,
                                                        self.max_validation, seed))
            n01.append((n0, n1, len(s_p)))
        rfetch = ops.PendingFetch(torch.stack(raws), lambda host: host.copy())
        yield rfetch
        out = []
        for raw, (n0, n1, ns) in zip(rfetch.finish(), n01):
            T, info = ops.ransac_decode(raw, ns)
            info.update(n0=n0, n1=n1)
            out.append((T, info))
        return out

    @torch.no_grad()
    def __call__(self, xyz0, xyz1, seed=0):
        src, tgt, feats, ov, sal = self.encode(xyz0, xyz1)
        n0 = len(src)
        rng = np.random.RandomState(seed)
        s_p, s_f, _ = BU.sample_by_score(src, feats[:n0], ov[:n0] * sal[:n0], self.n_points, rng=rng)
        t_p, t_f, _ = BU.sample_by_score(tgt, feats[n0:], ov[n0:] * sal[n0:], self.n_points, rng=rng)
        T, info = BU.ransac_pose_estimation(s_p, t_p, s_f, t_f, distance_threshold=self.distance_threshold, ransac_n=4,
                                            max_iteration=self.max_iteration, max_validation=self.max_validation,
                                            seed=seed, return_info=True)
        info.update(n0=n0, n1=len(tgt))
        return T, info
