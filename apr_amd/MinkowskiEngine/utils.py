"""`MinkowskiEngine.utils` subset used by FCGF_APR.

sparse_quantize   <- FCGF_APR/lib/complement_data_loader.py:671-674,788-789; util/misc.py:80-81
sparse_collate    <- complement_data_loader.py:1254-1255,1310-1311
batched_coordinates <- util/misc.py:83
"""
from __future__ import annotations

import numpy as np
import torch

from .. import ops


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("apr_amd MinkowskiEngine shim needs a GPU (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def sparse_quantize(coordinates, features=None, labels=None, ignore_label=-100, return_index=False,
                    return_inverse=False, quantization_size=None, device=None):
    """Unique voxels of floor(coordinates); first row of each voxel in ascending row order.

    Runs the voxel hash on the GPU whatever the input container; the result comes
    back in the input's container type (numpy / CPU tensor / GPU tensor).
    """
    if return_inverse or labels is not None:
        raise NotImplementedError("sparse_quantize: labels / return_inverse are not used by APR")
    is_np = isinstance(coordinates, np.ndarray)
    c = torch.from_numpy(np.ascontiguousarray(coordinates)) if is_np else coordinates
    src_dev = c.device
    c = c.to(_device())
    if quantization_size is not None:
        c = c / quantization_size
    if c.dtype.is_floating_point:
        c = torch.floor(c)
    c = c.to(torch.int32)
    bc = torch.cat([torch.zeros((len(c), 1), dtype=torch.int32, device=c.device), c], 1)
    m = ops.build_map(bc, want_first=True)
    ops.finalize_maps([m])
    uc, idx = m.coords[:, 1:].contiguous(), m.first

    def back(t):
        t = t.to(src_dev)
        return t.numpy() if is_np else t

    outs = [back(uc)]
    if features is not None:
        f = torch.from_numpy(features) if isinstance(features, np.ndarray) else features
        f = f[idx.to(f.device)]
        outs.append(f.numpy() if isinstance(features, np.ndarray) else f)
    if return_index:
        outs.append(back(idx))
    return outs[0] if len(outs) == 1 else tuple(outs)


def batched_coordinates(coords, dtype=torch.int32, device=None):
    out = []
    for b, c in enumerate(coords):
        c = torch.from_numpy(c) if isinstance(c, np.ndarray) else c
        c = torch.floor(c) if c.dtype.is_floating_point else c
        c = c.to(dtype)
        out.append(torch.cat([torch.full((len(c), 1), b, dtype=dtype, device=c.device), c], 1))
    r = torch.cat(out, 0)
    return r.to(device) if device is not None else r


def sparse_collate(coords, feats, labels=None, dtype=torch.int32, device=None):
    bcoords = batched_coordinates(coords, dtype=dtype, device=device)
    fs = [torch.from_numpy(f) if isinstance(f, np.ndarray) else f for f in feats]
    bfeats = torch.cat(fs, 0)
    if device is not None:
        bfeats = bfeats.to(device)
    if labels is not None:
        ls = [torch.from_numpy(l) if isinstance(l, np.ndarray) else l for l in labels]
        return bcoords, bfeats, torch.cat(ls, 0)
    return bcoords, bfeats
