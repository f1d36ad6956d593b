"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the MinkowskiEngine operator subset that FCGF_APR's encoder
uses.  MinkowskiEngine itself is a third-party dependency that is NOT vendored
under /root/reference and is unpinned ("v0.5 or higher",
`/root/reference/README.md:42`, `FCGF_APR/README.md:13`); the reference holds no
tests or golden vectors at this boundary, so this restatement is
**PARITY UNPINNED** against upstream ME.  It follows ME 0.5.x's published
semantics as recorded in SURVEY.md Appendix A, anchored on the reference's call
sites:

  * `sparse_quantize`      <- `FCGF_APR/lib/complement_data_loader.py:671-674,788-789`,
                              `FCGF_APR/util/misc.py:80-81`
  * `sparse_collate`       <- `complement_data_loader.py:1310-1311`
  * `batched_coordinates`  <- `util/misc.py:83`
  * conv / conv-transpose  <- `FCGF_APR/model/resunet.py:31-140`,
                              `model/residual_block.py:23-33`
  * batch norm             <- `model/common.py:4-10`
  * relu / cat / +=        <- `resunet.py:146,168`, `residual_block.py:50-51`

Conventions (shared with the HIP kernels):
  * coordinates are int32 [N,4] = (batch, x, y, z); a map at tensor stride ts
    only holds multiples of ts.
  * kernel offsets for odd k: o in {-h..h}^3 (h = k//2) scaled by the *input*
    tensor stride, enumerated x fastest: idx = (ox+h) + k*(oy+h) + k*k*(oz+h).
  * stride-2 output map = unique(floor(c / (2 ts)) * (2 ts)) in first-occurrence
    order of the input rows.
  * regular conv pair (i,j) under offset o  <=>  c_in[i] == c_out[j] + o*ts_in.
  * transposed conv uses the forward strided map (fine rows as "in", coarse rows
    as "out", offsets scaled by the fine stride) with in/out swapped:
        out_fine[i] += in_coarse[j] @ W[o]  <=>  c_fine[i] == c_coarse[j] + o*ts_fine.
  * `sparse_quantize` keeps the first input row of every voxel, in ascending
    row order.

Everything is numpy / torch-CPU float32, vectorised so a 20 k-point frame
encodes in seconds.
"""
from __future__ import annotations

import numpy as np
import torch

# --------------------------------------------------------------------------
# coordinate helpers
# --------------------------------------------------------------------------

_B = np.int64(1) << 18  # per-axis range of the packed key (matches the HIP key layout)
_H = np.int64(1) << 17


def pack_keys(coords: np.ndarray) -> np.ndarray:
    """(b,x,y,z) int -> one int64 key; b:10 bits, x/y/z: 18 bits each (biased)."""
    c = coords.astype(np.int64)
    return ((c[:, 0] * _B + (c[:, 1] + _H)) * _B + (c[:, 2] + _H)) * _B + (c[:, 3] + _H)


def first_occurrence_unique(keys: np.ndarray):
    """Indices of the first row of every distinct key, ascending."""
    _, first = np.unique(keys, return_index=True)
    return np.sort(first)


def sparse_quantize(coords, return_index=False):
    """floor -> int32, unique voxels; first row of each voxel, ascending row order.

    `coords` is xyz / voxel_size already (the call sites divide first).
    """
    c = np.floor(np.asarray(coords)).astype(np.int32)
    z = np.zeros((len(c), 1), np.int32)
    idx = first_occurrence_unique(pack_keys(np.concatenate([z, c], 1)))
    if return_index:
        return c[idx], idx
    return c[idx]


def batched_coordinates(coords_list):
    out = []
    for b, c in enumerate(coords_list):
        c = np.asarray(c).astype(np.int32)
        out.append(np.concatenate([np.full((len(c), 1), b, np.int32), c], 1))
    return np.concatenate(out, 0)


def sparse_collate(coords_list, feats_list):
    return batched_coordinates(coords_list), np.concatenate([np.asarray(f) for f in feats_list], 0)


def stride_coords(coords: np.ndarray, ts: int):
    """Output map of a stride-2 conv on a map at tensor stride `ts`."""
    c = coords.astype(np.int64).copy()
    nts = 2 * ts
    c[:, 1:] = np.floor_divide(c[:, 1:], nts) * nts
    idx = first_occurrence_unique(pack_keys(c))
    return c[idx].astype(np.int32)


def kernel_offsets(kernel_size: int):
    h = kernel_size // 2
    r = np.arange(-h, h + 1)
    oz, oy, ox = np.meshgrid(r, r, r, indexing="ij")  # x fastest
    return np.stack([ox.ravel(), oy.ravel(), oz.ravel()], 1).astype(np.int64)


def kernel_map(in_coords, out_coords, kernel_size, ts_in):
    """Dense neighbour table nbr[j, o] = i with c_in[i] == c_out[j] + o*ts_in, else -1."""
    kin = pack_keys(in_coords)
    order = np.argsort(kin, kind="stable")
    ks = kin[order]
    offs = kernel_offsets(kernel_size) * ts_in
    nbr = np.full((len(out_coords), len(offs)), -1, np.int32)
    oc = out_coords.astype(np.int64)
    for o, off in enumerate(offs):
        q = oc.copy()
        q[:, 1:] += off
        kq = pack_keys(q)
        pos = np.searchsorted(ks, kq)
        pos = np.minimum(pos, len(ks) - 1)
        hit = ks[pos] == kq
        nbr[hit, o] = order[pos[hit]]
    return nbr


def transpose_map(nbr_fwd, n_fine):
    """Swap in/out of a forward strided map nbr_fwd[j_coarse, o] = i_fine.

    Each fine row has at most one coarse partner per offset, so the result is a
    dense table nbr_t[i_fine, o] = j_coarse.
    """
    nbr_t = np.full((n_fine, nbr_fwd.shape[1]), -1, np.int32)
    j, o = np.nonzero(nbr_fwd >= 0)
    nbr_t[nbr_fwd[j, o], o] = j
    return nbr_t


# --------------------------------------------------------------------------
# coordinate manager + sparse tensor
# --------------------------------------------------------------------------


class CoordinateManager:
    def __init__(self, coords):
        self.coords = {1: np.asarray(coords).astype(np.int32)}
        self._maps = {}

    def get_coords(self, ts):
        if ts not in self.coords:
            self.coords[ts] = stride_coords(self.get_coords(ts // 2), ts // 2)
        return self.coords[ts]

    def get_map(self, ts_in, ts_out, k):
        """nbr table for in-map ts_in -> out-map ts_out with kernel k (regular conv)."""
        key = (ts_in, ts_out, k)
        if key not in self._maps:
            self._maps[key] = kernel_map(self.get_coords(ts_in), self.get_coords(ts_out), k, ts_in)
        return self._maps[key]


class SparseTensor:
    def __init__(self, features, coordinates=None, coordinate_map_key=None, coordinate_manager=None):
        self.F = torch.as_tensor(features, dtype=torch.float32)
        if coordinates is not None:
            self.coordinate_manager = CoordinateManager(np.asarray(coordinates))
            self.coordinate_map_key = 1
        else:
            self.coordinate_manager = coordinate_manager
            self.coordinate_map_key = coordinate_map_key

    @property
    def C(self):
        return torch.from_numpy(self.coordinate_manager.get_coords(self.coordinate_map_key))

    def _like(self, F, key=None):
        return SparseTensor(F, coordinate_map_key=self.coordinate_map_key if key is None else key,
                            coordinate_manager=self.coordinate_manager)


def conv_forward(x: SparseTensor, W: torch.Tensor, kernel_size, stride, bias=None, transpose=False):
    """out[j] = sum_o sum_{i in map(o,j)} in[i] @ W[o]."""
    cm, ts = x.coordinate_manager, x.coordinate_map_key
    if kernel_size == 1 and stride == 1:
        out = x.F @ W
        if bias is not None:
            out = out + bias
        return x._like(out)
    if not transpose:
        ts_out = ts * stride
        nbr = cm.get_map(ts, ts_out, kernel_size)
    else:
        ts_out = ts // stride
        fwd = cm.get_map(ts_out, ts, kernel_size)  # fine -> coarse
        nbr = transpose_map(fwd, len(cm.get_coords(ts_out)))
    n_out = nbr.shape[0]
    out = torch.zeros(n_out, W.shape[2], dtype=torch.float32)
    for o in range(nbr.shape[1]):
        j = np.nonzero(nbr[:, o] >= 0)[0]
        if len(j) == 0:
            continue
        i = nbr[j, o]
        out.index_add_(0, torch.from_numpy(j), x.F[torch.from_numpy(i.astype(np.int64))] @ W[o])
    if bias is not None:
        out = out + bias
    return x._like(out, ts_out)


def batch_norm(x: SparseTensor, bn: torch.nn.BatchNorm1d):
    return x._like(bn(x.F))


def relu(x: SparseTensor):
    return x._like(torch.relu(x.F))


def cat(a: SparseTensor, b: SparseTensor):
    assert a.coordinate_map_key == b.coordinate_map_key
    return a._like(torch.cat([a.F, b.F], 1))
