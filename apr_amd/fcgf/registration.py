"""Feature-matching RANSAC registration: the GPU stand-in for the open3d call at
FCGF_APR/scripts/test_apr.py:148-156 (and Predator_APR/lib/benchmark_utils.py:213-225).

    T = ransac_feature_matching(xyz0, xyz1, F0, F1, distance_threshold,
                                ransac_n=4, edge_length=0.9, max_iteration=4_000_000)

returns T [4,4] float64 with xyz1 ~= xyz0 @ R.T + t, like
`ransac_result.transformation`.  Semantics follow the open3d >= 0.12 API the call
site is written against (5th positional arg `mutual_filter=False`;
`RANSACConvergenceCriteria(4000000, 10000)` -> max_iteration 4 M, confidence
clamped to 1.0 so there is no early exit): correspondences = feature-space
nearest neighbour of every source point, 4-point samples, edge-length + distance
checkers, Kabsch without scaling, hypotheses scored on the correspondence set.
open3d's RNG is unseeded / thread dependent; here the hypothesis stream is a
counter-based RNG of (seed, iteration) so runs are reproducible and the CPU
oracle can replay it.  open3d is absent from this image: parity unpinned.
`max_validation=None` (default) is that reading; an integer selects the open3d <= 0.11 reading of the same criteria
object (stop after max_validation validated hypotheses, geometric inlier scoring), as `ops.ransac_pose_geometric`.
"""
import numpy as np
import torch

from .. import ops


def _dev(t):
    if isinstance(t, np.ndarray):
        t = torch.from_numpy(t)
    return t.to(device=torch.device('cuda', torch.cuda.current_device()), dtype=torch.float32).contiguous()


def feature_correspondences(F0, F1):
    return ops.feature_nn(_dev(F0), _dev(F1))


def ransac_feature_matching(xyz0, xyz1, F0, F1, distance_threshold, ransac_n=4, edge_length=0.9,
                            max_iteration=4000000, max_validation=None, seed=0, return_info=False):
    if ransac_n != 4:
        raise NotImplementedError("the HIP RANSAC kernel is specialised for ransac_n = 4 (both call sites)")
    x0, x1 = _dev(xyz0), _dev(xyz1)
    corr = feature_correspondences(F0, F1)
    if max_validation is not None:
        # open3d <= 0.11 reading of RANSACConvergenceCriteria(max_iteration, max_validation): the loop stops after
        # `max_validation` hypotheses have passed both checkers, and a hypothesis is scored by the source points that have a
        # target point within distance_threshold after the transform (not on the correspondence set) -- the flavour
        # Predator_APR's tester pins (requirements.txt: open3d 0.10; lib/benchmark_utils.py:213-225), on the same kernels
        if int(max_validation) < 1:
            raise ValueError("ransac_feature_matching: max_validation must be >= 1 (None: open3d >= 0.12 semantics)")
        T, info = ops.ransac_pose_geometric(x0, x1, corr, distance_threshold, edge_length, max_iteration,
                                            int(max_validation), seed)
    else:
        T, info = ops.ransac_pose(x0, x1, corr, distance_threshold, edge_length, max_iteration, seed)
    return (T, info) if return_info else T


def rte_rre(T_est, T_gt):
    """Translation / rotation error as FCGF_APR/scripts/test_apr.py:162-163 (metres, degrees)."""
    T_est, T_gt = np.asarray(T_est, dtype=np.float64), np.asarray(T_gt, dtype=np.float64)
    rte = np.linalg.norm(T_est[:3, 3] - T_gt[:3, 3])
    c = (np.trace(T_est[:3, :3].T @ T_gt[:3, :3]) - 1) / 2
    rre = np.degrees(np.arccos(np.clip(c, -1.0, 1.0)))
    return float(rte), float(rre)
