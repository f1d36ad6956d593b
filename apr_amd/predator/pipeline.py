"""One scan pair through the Predator_APR hot path on the GPU: grid subsample -> KPConv index pyramid ->
KPFCNN (encoder + overlap attention + decoder) -> score-weighted sampling -> RANSAC/SVD.

This is the per-pair body of the reference's tester (Predator_APR/lib/tester.py:48-110) with its data-loader work
(datasets/kitti.py:466-471 voxel_down_sample, datasets/dataloader.py collate_fn_descriptor) pulled onto the device.
The counterpart of apr_amd.fcgf.pipeline.PairRegistration.
"""
from __future__ import annotations

import numpy as np
import torch

from . import point_ops
from .datasets.dataloader import calibrate_neighbors, collate_fn_descriptor   # noqa: F401  (re-exported)
from .lib import benchmark_utils as BU


class PredatorRegistration:
    def __init__(self, model, config, neighborhood_limits, voxel_size=0.3, n_points=5000, distance_threshold=0.3,
                 max_iteration=50000, max_validation=1000):
        """`neighborhood_limits`: per-level caps of the radius neighbourhoods (the reference calibrates them once per
        dataset, datasets/dataloader.py:calibrate_neighbors); `n_points`: interest points kept per frame
        (configs/test/kitti.yaml n_points)."""
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
