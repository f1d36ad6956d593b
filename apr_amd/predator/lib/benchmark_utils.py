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


def get_correspondences(src_pcd, tgt_pcd, trans, search_voxel_size, K=None):
    """(i, j) pairs with |T src_i - tgt_j| < search_voxel_size, by i then by distance -> int64 [M, 2] (CPU tensor, as the
    reference returns it: lib/benchmark_utils.py:121-135).  The clouds are [N,3] arrays / tensors (the reference takes
    open3d point clouds; open3d is not part of this build) -- the search is the radius-neighbour kernel."""
    from ...fcgf.lib import apg
    pts = lambda p: np.asarray(p.points, dtype=np.float32) if hasattr(p, "points") else p
    return apg.get_matching_indices(pts(src_pcd), pts(tgt_pcd), trans, search_voxel_size, K).cpu()


def get_angle_deviation(R_pred, R_gt):
    R = np.matmul(R_pred, R_gt.transpose(0, 2, 1))
    tr = np.trace(R, 0, 1, 2)
    rads = np.arccos(np.clip((tr - 1) / 2, -1, 1))
    return rads / np.pi * 180


def weighted_choice(rng, n, size, p):
    """`rng.choice(np.arange(n), size=size, replace=False, p=p)` for a legacy NumPy generator (`np.random` or a
    `RandomState`), same result and same consumption of the generator's stream: the uniforms come from
    `rng.random_sample`, round by round as NumPy draws them, and each round's cumsum / searchsorted / first-occurrence
    filter runs in the library's host function `apr_weighted_choice_round` (one pass in C instead of ~2 ms of NumPy
    calls per draw of 5000 from 14 k)."""
    p = np.array(p, dtype=np.float64, copy=True).reshape(-1)
    if p.size != n:
        raise ValueError("'a' and 'p' must have same size")
    if np.logical_or.reduce(p < 0):
        raise ValueError("probabilities are not non-negative")
    atol = max(np.sqrt(np.finfo(np.float64).eps), np.sqrt(np.finfo(np.asarray(p).dtype).eps))
    if abs(float(np.sum(p)) - 1.0) > max(atol, 3.5e-4):    # NumPy's tolerance follows the caller's dtype (float32 here)
        raise ValueError("probabilities do not sum to 1")
    if size > n:
        raise ValueError("Cannot take a larger sample than population when 'replace=False'")
    if np.count_nonzero(p > 0) < size:
        raise ValueError("Fewer non-zero entries in p than size")
    lib = ops._lib_()
    found = np.zeros(size, dtype=np.int64)
    cdf = np.empty(n, dtype=np.float64)
    stamp = np.zeros(n, dtype=np.int32)
    n_uniq, added, rnd = 0, 0, 0
    while n_uniq < size:
        x = rng.random_sample(size - n_uniq)
        rnd += 1
        got = lib.apr_weighted_choice_round(p.ctypes.data, n, found.ctypes.data, n_uniq, added, x.ctypes.data, len(x),
                                            cdf.ctypes.data, stamp.ctypes.data, rnd)
        if got < 0:
            ops.check(int(got))
        added, n_uniq = got - n_uniq, got
    return found


def draw_by_score(n, scores, n_points, rng=np.random):
    """The host half of `sample_by_score`: the indices `rng.choice` draws for CPU float32 `scores` (None: keep every point)."""
    if n <= n_points:
        return None
    # lib/tester.py:85: `(scores / scores.sum()).numpy().flatten()` on the CPU float32 tensor (torch's float32 sum);
    # np.random.choice widens p to float64 itself
    s = scores.detach().cpu().float()
    probs = (s / s.sum()).numpy().flatten()
    return weighted_choice(rng, n, n_points, probs)


def take_drawn(pcd, feats, idx):
    """The device half: rows `idx` of the cloud and its features (idx None: everything)."""
    if idx is None:
        return pcd, feats
    # pinned + non_blocking: a pageable host->device copy blocks the host until the stream gets to it
    idx_t = torch.from_numpy(idx).pin_memory().to(pcd.device, non_blocking=True) if torch.is_tensor(pcd) else idx
    return pcd[idx_t], feats[idx_t]


def sample_by_score(pcd, feats, scores, n_points, rng=np.random):
    """Score-weighted sampling without replacement (lib/tester.py:80-92); the draw stays on the host RNG."""
    idx = draw_by_score(pcd.shape[0], scores, n_points, rng)
    return take_drawn(pcd, feats, idx) + (idx,)
