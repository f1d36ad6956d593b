"""Feature nearest-neighbour search with the reference's signatures (FCGF_APR/lib/eval.py:9-48).

`find_nn_gpu(F0, F1, nn_max_n=, return_distance=, dist_type=)` returns CPU
tensors exactly like the reference (indices int64 [N]; distances [N,1]), but
runs ONE fused HIP arg-min kernel instead of chunked [500,N,C] broadcasts;
`nn_max_n` is accepted and ignored (nothing is materialised).
"""
import numpy as np
import torch

from ... import ops


def find_nn_gpu(F0, F1, nn_max_n=-1, return_distance=False, dist_type='SquareL2'):
    if dist_type not in ('SquareL2', 'L2'):
        raise NotImplementedError('Not implemented')
    idx, d2 = ops.feature_nn(F0, F1, return_distance=True)
    if not return_distance:
        return idx.cpu()
    d = d2 if dist_type == 'SquareL2' else torch.sqrt(d2 + 1e-7)
    return idx.cpu(), d.unsqueeze(1).cpu()


def find_nn_device(F0, F1):
    """Same search, result left on the GPU (used by the registration pipeline)."""
    return ops.feature_nn(F0, F1)


def find_corr(xyz0, xyz1, F0, F1, subsample_size=-1, nn_max_n=500):
    """Correspondences by feature nearest neighbour, optionally on a random subsample of both clouds
    (FCGF_APR/scripts/test_apr.py:43-57, lib/trainer.py:374-390; the evaluation script calls it with 5000).
    The two `np.random.choice(len, N, replace=False)` draws happen in the reference's order (source, then target),
    so a seeded NumPy RNG selects the same points.  -> (xyz0 rows, matched xyz1 rows)."""
    subsample = len(F0) > subsample_size
    if subsample_size > 0 and subsample:
        n0, n1 = min(len(F0), subsample_size), min(len(F1), subsample_size)
        inds0 = np.random.choice(len(F0), n0, replace=False)
        inds1 = np.random.choice(len(F1), n1, replace=False)
        i0 = torch.as_tensor(inds0, device=F0.device)
        i1 = torch.as_tensor(inds1, device=F1.device)
        nn_inds = find_nn_gpu(F0[i0], F1[i1], nn_max_n=nn_max_n)
        return xyz0[inds0], xyz1[inds1[nn_inds.numpy()]]
    nn_inds = find_nn_gpu(F0, F1, nn_max_n=nn_max_n)
    return xyz0, xyz1[nn_inds.to(xyz1.device) if torch.is_tensor(xyz1) else nn_inds.numpy()]


def evaluate_hit_ratio(xyz0, xyz1, T_gth, thresh=0.1):
    """Share of correspondences closer than `thresh` once the source is moved by the ground truth
    (FCGF_APR/lib/trainer.py:392-395: sqrt(sum sq + 1e-6) < thresh)."""
    xyz0, xyz1, T_gth = (torch.as_tensor(t, dtype=torch.float32) for t in (xyz0, xyz1, T_gth))
    moved = xyz0 @ T_gth[:3, :3].t().to(xyz0.device) + T_gth[:3, 3].to(xyz0.device)
    dist = torch.sqrt(((moved - xyz1.to(xyz0.device)) ** 2).sum(1) + 1e-6)
    return (dist < thresh).float().mean().item()
