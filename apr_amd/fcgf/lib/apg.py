"""APG aggregation, GT correspondences and the NPR reconstruction loss on the GPU.

  aggregate_frames   <- FCGF_APR/lib/complement_data_loader.py:576-628,671-674 (transform the 2k complement
                        frames into the key frame, crop to the key frame's radius, voxel-quantise)
  get_matching_indices <- FCGF_APR/util/pointcloud.py:53-66 ; Predator_APR/lib/benchmark_utils.py:121-135
  chamfer_distance   <- FCGF_APR/lib/complement_trainer.py:188-196 (chamferdist 1-NN sums, both directions)
  GenerativeMLP*     <- FCGF_APR/model/mlp.py:6-37 (same module / parameter names: `mlp.0.weight` ...)
  npr_reconstruction_loss <- complement_trainer.py:424-449 (differentiable: GEMM / BN / Chamfer forward and backward on
                        the HIP kernels, apr_amd/npr.py)
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from ... import _lib, npr, ops
from ..._lib import check, ptr, stream
from ...predator import kp_ops, point_ops


def _dev():
    return torch.device('cuda', torch.cuda.current_device())


def _f32(a):
    a = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
    return a.to(device=_dev(), dtype=torch.float32).contiguous()


def apply_transform(pts, trans):
    """pts @ R.T + t in float32 (complement_data_loader.py:65-70)."""
    pts = _f32(pts)
    T = _f32(np.asarray(trans, dtype=np.float32).reshape(16) if not torch.is_tensor(trans) else trans.reshape(16))
    out = torch.empty_like(pts)
    check(_lib.load().apr_transform_points(ptr(pts), pts.shape[0], ptr(T), ptr(out), stream()))
    return out


def crop_to_radius(key_xyz, pts):
    """Rows of `pts` with |p|^2 < max |key|^2, order preserved."""
    key_xyz, pts = _f32(key_xyz), _f32(pts)
    lib = _lib.load()
    n = pts.shape[0]
    out = torch.empty_like(pts)
    cnt = torch.empty(1, dtype=torch.int32, device=pts.device)
    sb = int(lib.apr_crop_scratch_bytes(n))
    scratch = torch.empty(sb, dtype=torch.uint8, device=pts.device)
    check(lib.apr_crop_to_radius(ptr(key_xyz), key_xyz.shape[0], ptr(pts), n, ptr(out), ptr(cnt), ptr(scratch), sb,
                                 stream()))
    return out[: int(cnt.item())]


def aggregate_frames(key_xyz, complement_xyz, complement_poses, voxel_size):
    """APG: returns (xyz_nghb cropped [M,3], sel indices of its voxelised subset)."""
    moved = [apply_transform(x, M) for x, M in zip(complement_xyz, complement_poses)]
    nghb = crop_to_radius(key_xyz, torch.cat(moved, 0))
    coords = ops.voxelize(nghb, voxel_size, 0)
    m = ops.build_map(coords, want_first=True)
    ops.finalize_maps([m])
    return nghb, m.first


def get_matching_indices(source, target, trans, search_voxel_size, K=None):
    """(i, j) pairs with |T src_i - tgt_j| < search_voxel_size, ordered by i then distance -> int64 [M,2]."""
    src = apply_transform(source, trans)
    tgt = _f32(target)
    nbr = point_ops.radius_neighbors(src, tgt, [len(src)], [len(tgt)], float(search_voxel_size))
    if K is not None:
        nbr = nbr[:, :K]
    valid = nbr < len(tgt)
    i = torch.arange(len(src), device=nbr.device).unsqueeze(1).expand_as(nbr)[valid]
    return torch.stack([i, nbr[valid].long()], 1)


def chamfer_sum(a, b):
    """sum_i min_j |a_i - b_j|^2 (0-d float64 GPU tensor)."""
    a, b = _f32(a), _f32(b)
    out = torch.empty(1, dtype=torch.float64, device=a.device)
    scratch = torch.empty(a.shape[0] * 4, dtype=torch.uint8, device=a.device)
    check(_lib.load().apr_chamfer_sum(ptr(a), a.shape[0], ptr(b), b.shape[0], ptr(out), ptr(scratch), scratch.numel(),
                                      stream()))
    return out[0]


def chamfer_distance(array1, array2):
    """forward / n1 + backward / n2 (complement_trainer.py:188-196); differentiable (apr_amd/npr.py)."""
    return npr.chamfer_distance(array1, array2)


class GenerativeMLP(nn.Module):
    """Linear -> ReLU -> BatchNorm1d stacks, final Linear -> ReLU (FCGF_APR/model/mlp.py:6-29); HIP forward."""
    CHANNELS = [None, 512, 128, None]

    def __init__(self, in_channel=125, out_points=6, bn_momentum=0.1):
        super().__init__()
        CH = self.CHANNELS
        self.mlp = nn.Sequential(
            nn.Linear(in_channel, CH[1]), nn.ReLU(), nn.BatchNorm1d(CH[1], momentum=bn_momentum),
            nn.Linear(CH[1], CH[2]), nn.ReLU(), nn.BatchNorm1d(CH[2], momentum=bn_momentum),
            nn.Linear(CH[2], out_points * 3), nn.ReLU())

    def forward(self, x, segments=None):
        """`segments` (row offsets, optional): x stacks the rows of several calls (the clouds of a batch, both frames) --
        the BatchNorms keep one set of batch statistics per call, as separate calls would."""
        return npr.run_stack(self.mlp, x, segments)


class GenerativeMLP_98(GenerativeMLP):
    CHANNELS = [None, 512, 256, None]


class GenerativeMLP_54(GenerativeMLP):
    CHANNELS = [None, 32, 16, None]


def npr_regulariser(generated, reg_type='L2', alpha=0.1):
    """Length penalty on the generated offsets [N, 3*ratio] (complement_trainer.py:432-440)."""
    return npr.regulariser(generated, reg_type, alpha)


def npr_points(generated, enc_coords, voxel_size, ratio):
    """Generated offsets + their voxel's corner -> [N*ratio, 3] points (complement_trainer.py:441-442)."""
    return (generated + voxel_size * enc_coords.to(generated.dtype).repeat(1, ratio)).reshape(-1, 3)


def npr_reconstruction_loss(generator, enc_feats, enc_coords, pcd_nghb, voxel_size, ratio, reg_strength=0.01,
                            reg_type='L2', alpha=0.1):
    """chamfer(generated + voxel centres, APG cloud) + reg * regulariser for one cloud (complement_trainer.py:424-448)."""
    generated = generator(enc_feats) * voxel_size                                       # [N, 3*ratio]
    reg = npr_regulariser(generated, reg_type, alpha)
    mod = npr_points(generated, enc_coords, voxel_size, ratio)
    return chamfer_distance(mod, pcd_nghb) + reg * reg_strength
