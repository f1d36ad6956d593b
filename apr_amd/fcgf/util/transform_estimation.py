"""Robust linearised pose fit with the reference's signature.

`est_quad_linear_robust(pts0, pts1, weight=None) -> T [4,4] float32 (CPU)`
replaces FCGF_APR/util/transform_estimation.py:89-116 (20 IRLS iterations,
`par` halving every 5, rot_z*rot_y*rot_x composition).  The whole loop runs in
one persistent HIP workgroup (apr_irls_pose); inputs may be CPU or GPU tensors.
"""
import torch

from ... import ops


def est_quad_linear_robust(pts0, pts1, weight=None):
    dev = torch.device('cuda', torch.cuda.current_device())
    p0 = pts0.to(device=dev, dtype=torch.float32)
    p1 = pts1.to(device=dev, dtype=torch.float32)
    w = None if weight is None else weight.to(device=dev, dtype=torch.float32).reshape(-1)
    return ops.irls_pose(p0, p1, w)
