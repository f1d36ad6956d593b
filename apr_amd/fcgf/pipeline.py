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

import torch

from .. import MinkowskiEngine as ME
from .. import ops


class PairRegistration:
    def __init__(self, model, voxel_size=0.3, ransac_iters=4000000, edge_length=0.9, distance_factor=1.0):
        self.model = model.eval()
        self.voxel_size = float(voxel_size)
        self.ransac_iters = int(ransac_iters)
        self.edge_length = float(edge_length)
        self.distance_threshold = float(voxel_size) * distance_factor  # test_apr.py:149

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
        out = self.model(ME.SparseTensor(feats, coordinates=coords)).F
        return out[:n0], out[n0:]

    @torch.no_grad()
    def __call__(self, xyz0, xyz1, seed=0):
        coords, pts0, pts1, n0, n1 = self.voxelize_pair(xyz0, xyz1)
        F0, F1 = self.encode_pair(coords, n0)
        corr = ops.feature_nn(F0.contiguous(), F1.contiguous())
        T, info = ops.ransac_pose(pts0.contiguous(), pts1.contiguous(), corr, self.distance_threshold,
                                  self.edge_length, self.ransac_iters, seed)
        info.update(n0=n0, n1=n1)
        return T, info
