"""NumPy-in / NumPy-out `subsample_batch` with the reference extension's signature
(Predator_APR/cpp_wrappers/cpp_subsampling/wrapper.cpp:75-82, 316-322), computed on the GPU.

    s_points, s_len[, s_features] = subsample_batch(points, batches, features=None, classes=None,
                                                    sampleDl=0.1, method='barycenters', max_p=0, verbose=0)

Row order inside each cloud is the first-occurrence order of the cells (the reference's is
libstdc++ unordered_map iteration order); the barycentres themselves are bit-identical.
"""
import numpy as np
import torch

from ... import point_ops


def subsample_batch(points, batches, features=None, classes=None, sampleDl=0.1, method='barycenters', max_p=0,
                    verbose=0):
    if classes is not None:
        raise NotImplementedError("subsample_batch: class voting is not used by the APR pipeline")
    points = np.asarray(points)
    batches = np.asarray(batches)
    if points.ndim != 2 or points.shape[1] != 3:
        raise RuntimeError("Wrong dimensions : points.shape is not (N, 3)")
    if batches.ndim != 1:
        raise RuntimeError("Wrong dimensions : batches.shape is not (B,)")
    if int(batches.sum()) != points.shape[0]:
        raise RuntimeError("Wrong batch lengths : lengths do not sum to the number of points")
    if features is not None and (np.asarray(features).ndim != 2 or len(features) != len(points)):
        raise RuntimeError("Wrong dimensions : features.shape is not (N, d)")
    dev = torch.device('cuda', torch.cuda.current_device())
    p = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32)).to(dev)
    f = None if features is None else torch.from_numpy(np.ascontiguousarray(features, dtype=np.float32)).to(dev)
    res = point_ops.grid_subsample(p, batches.astype(np.int32), float(sampleDl), f)
    s_points, s_len = res[0].cpu().numpy(), res[1].copy()
    s_feat = res[2].cpu().numpy() if f is not None else None
    if max_p and max_p > 0:   # keep at most max_p points per cloud (in this implementation's row order)
        keep, start, new_len = [], 0, []
        for n in s_len:
            m = min(int(n), int(max_p))
            keep.append(np.arange(start, start + m)); new_len.append(m); start += int(n)
        keep = np.concatenate(keep)
        s_points, s_len = s_points[keep], np.asarray(new_len, np.int32)
        if s_feat is not None:
            s_feat = s_feat[keep]
    if s_feat is not None:
        return s_points, s_len, s_feat
    return s_points, s_len
