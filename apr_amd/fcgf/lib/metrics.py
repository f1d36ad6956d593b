"""`pdist` / `corr_dist` with the reference's signatures (FCGF_APR/lib/metrics.py:13-29).

`pdist` materialises the full [N,M] matrix and is kept for API parity on small
inputs only; the hot path never calls it -- nearest neighbours come from the
fused HIP arg-min (`lib.eval.find_nn_gpu`).
"""
import torch


def corr_dist(est, gth, xyz0, xyz1, weight=None, max_dist=1):
    xyz0_est = xyz0 @ est[:3, :3].t() + est[:3, 3]
    xyz0_gth = xyz0 @ gth[:3, :3].t() + gth[:3, 3]
    dists = torch.clamp(torch.sqrt(((xyz0_est - xyz0_gth).pow(2)).sum(1)), max=max_dist)
    if weight is not None:
        dists = weight * dists
    return dists.mean()


def pdist(A, B, dist_type='L2'):
    D2 = torch.sum((A.unsqueeze(1) - B.unsqueeze(0)).pow(2), 2)
    if dist_type == 'L2':
        return torch.sqrt(D2 + 1e-7)
    elif dist_type == 'SquareL2':
        return D2
    raise NotImplementedError('Not implemented')
