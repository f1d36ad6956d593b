"""NumPy-in / NumPy-out `batch_query` with the reference extension's signature
(Predator_APR/cpp_wrappers/cpp_neighbors/wrapper.cpp:71-75, 211-227), computed on the GPU.

    neighbors = batch_query(queries, supports, q_batches, s_batches, radius=1.0)   # int32 [Nq, max_count]

Rows hold support indices in ascending distance, padded with len(supports).
"""
import numpy as np
import torch

from ... import point_ops


def batch_query(queries, supports, q_batches, s_batches, radius=1.0):
    queries, supports = np.asarray(queries), np.asarray(supports)
    q_batches, s_batches = np.asarray(q_batches), np.asarray(s_batches)
    if queries.ndim != 2 or queries.shape[1] != 3:
        raise RuntimeError("Wrong dimensions : query.shape is not (N, 3)")
    if supports.ndim != 2 or supports.shape[1] != 3:
        raise RuntimeError("Wrong dimensions : support.shape is not (N, 3)")
    if q_batches.ndim != 1 or s_batches.ndim != 1 or len(q_batches) != len(s_batches):
        raise RuntimeError("Wrong number of batch elements: different for queries and supports ")
    dev = torch.device('cuda', torch.cuda.current_device())
    q = torch.from_numpy(np.ascontiguousarray(queries, dtype=np.float32)).to(dev)
    s = torch.from_numpy(np.ascontiguousarray(supports, dtype=np.float32)).to(dev)
    out = point_ops.radius_neighbors(q, s, q_batches.astype(np.int32), s_batches.astype(np.int32), float(radius), 0)
    if out.shape[1] == 0:
        raise RuntimeError("Error")   # as the reference does for an empty result (wrapper.cpp:201-205)
    return out.cpu().numpy()
