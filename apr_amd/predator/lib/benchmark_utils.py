"""Pose estimation / metrics of the Predator_APR tester on the GPU.

`ransac_pose_estimation(src_pcd, tgt_pcd, src_feat, tgt_feat, mutual=False, distance_threshold=0.05, ransac_n=3)`
keeps the reference's signature (Predator_APR/lib/benchmark_utils.py:187-225) and returns the 4x4 transform.
The reference calls open3d 0.10's `o3d.registration.registration_ransac_based_on_feature_matching` with
RANSACConvergenceCriteria(50000, 1000): feature-NN correspondences, 4-point samples, edge-length + distance
checkers, Kabsch, and a GEOMETRIC validation of the first 1000 surviving hypotheses.  `get_angle_deviation`
follows :170-185.  open3d is absent here: parity unpinned, checked against the CPU oracle on the same
hypothesis stream and against ground truth.
"""
import numpy as np
import torch

from ... import ops


def _dev(t):
    if isinstance(t, np.ndarray):
        t = torch.from_numpy(t)
    return t.to(device=torch.device('cuda', torch.cuda.current_device()), dtype=torch.float32).contiguous()


def ransac_pose_estimation(src_pcd, tgt_pcd, src_feat, tgt_feat, mutual=False, distance_threshold=0.05, ransac_n=3,
                           max_iteration=50000, max_validation=1000, seed=0, return_info=False):
    if mutual:
        raise NotImplementedError("mutual selection is not used by the APR tester (lib/tester.py:97)")
    if ransac_n != 4:
        raise NotImplementedError("the HIP RANSAC kernel is specialised for ransac_n = 4 (KITTI / nuScenes)")
    x0, x1 = _dev(src_pcd), _dev(tgt_pcd)
    corr = ops.feature_nn(_dev(src_feat), _dev(tgt_feat))
    T, info = ops.ransac_pose_geometric(x0, x1, corr, distance_threshold, 0.9, max_iteration, max_validation, seed)
    return (T, info) if return_info else T


def get_angle_deviation(R_pred, R_gt):
    R = np.matmul(R_pred, R_gt.transpose(0, 2, 1))
    tr = np.trace(R, 0, 1, 2)
    rads = np.arccos(np.clip((tr - 1) / 2, -1, 1))
    return rads / np.pi * 180


def sample_by_score(pcd, feats, scores, n_points, rng=np.random):
    """Score-weighted sampling without replacement (lib/tester.py:80-92); the draw stays on the host RNG."""
    if pcd.shape[0] <= n_points:
        return pcd, feats, None
    s = scores.detach().cpu().double()
    probs = (s / s.sum()).numpy().flatten()
    idx = rng.choice(np.arange(pcd.shape[0]), size=n_points, replace=False, p=probs)
    idx_t = torch.from_numpy(idx).to(pcd.device) if torch.is_tensor(pcd) else idx
    return pcd[idx_t], feats[idx_t], idx
