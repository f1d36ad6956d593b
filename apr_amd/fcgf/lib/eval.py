"""Feature nearest-neighbour search with the reference's signatures (FCGF_APR/lib/eval.py:9-48).

`find_nn_gpu(F0, F1, nn_max_n=, return_distance=, dist_type=)` returns CPU
tensors exactly like the reference (indices int64 [N]; distances [N,1]), but
runs ONE fused HIP arg-min kernel instead of chunked [500,N,C] broadcasts;
`nn_max_n` is accepted and ignored (nothing is materialised).
"""
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
