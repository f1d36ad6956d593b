"""`pdist` / `corr_dist` with the reference's signatures (FCGF_APR/lib/metrics.py:13-29).

`pdist` materialises the full [N,M] matrix and is kept for API parity on small
inputs only; the hot path never calls it -- nearest neighbours come from the
fused HIP arg-min (`lib.eval.find_nn_gpu`).
"""
import torch


def _rigid(T, pts):
    return pts @ T[:3, :3].t() + T[:3, 3]


def corr_dist(est, gth, xyz0, xyz1, weight=None, max_dist=1):
    """Mean (optionally weighted) distance, capped at `max_dist`, between the source points moved by the estimated
    and by the ground-truth transform (FCGF_APR/lib/metrics.py:13-19; `xyz1` is unused there too)."""
    gap = torch.linalg.vector_norm(_rigid(est, xyz0) - _rigid(gth, xyz0), dim=1).clamp(max=max_dist)
    return (gap if weight is None else weight * gap).mean()


def pdist(A, B, dist_type='L2'):
    D2 = torch.sum((A.unsqueeze(1) - B.unsqueeze(0)).pow(2), 2)
    if dist_type == 'L2':
        return torch.sqrt(D2 + 1e-7)
    elif dist_type == 'SquareL2':
        return D2
    raise NotImplementedError('Not implemented')
