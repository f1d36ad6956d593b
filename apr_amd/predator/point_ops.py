"""Device-tensor wrappers of the point-set kernels (grid subsample, radius neighbours, kNN)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib
from .._lib import check, ptr, stream


def _lens(lengths):
    a = np.ascontiguousarray(np.asarray(lengths, dtype=np.int32).reshape(-1))
    return a, a.ctypes.data_as(C.c_void_p)


def _pts(t, name):
    if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or t.shape[1] != 3:
        raise _lib.AprHipError(f"{name}: need a float32 [N,3] GPU tensor")
    return t.contiguous()


def grid_subsample(points, lengths, dl, features=None):
    """-> (sub_points [M,3], sub_lengths int32 numpy [B][, sub_features])."""
    lib = _lib.load()
    points = _pts(points, "grid_subsample.points")
    n = points.shape[0]
    la, lp = _lens(lengths)
    out = torch.empty_like(points)
    fdim, of = 0, None
    if features is not None:
        features = features.to(torch.float32).contiguous()
        fdim = features.shape[1]
        of = torch.empty_like(features)
    sb = int(lib.apr_grid_subsample_scratch_bytes(n))
    scratch = torch.empty(sb, dtype=torch.uint8, device=points.device)
    out_len = np.zeros(len(la), np.int32)
    check(lib.apr_grid_subsample(ptr(points), n, lp, len(la), float(dl), ptr(features), fdim, ptr(out), ptr(of),
                                 out_len.ctypes.data_as(C.c_void_p), ptr(scratch), sb, stream()))
    m = int(out_len.sum())
    if features is not None:
        return out[:m], out_len, of[:m]
    return out[:m], out_len


class PendingSubsample:
    """`grid_subsample` in flight: `fetch` (an ops.PendingFetch) completes when the cloud lengths have landed on the
    host; `finish()` then returns what `grid_subsample` returns."""
    __slots__ = ("fetch", "_out", "_feats")

    def __init__(self, fetch, out, feats):
        self.fetch, self._out, self._feats = fetch, out, feats

    def finish(self):
        out_len = self.fetch.finish()
        m = int(out_len.sum())
        if self._feats is not None:
            return self._out[:m], out_len, self._feats[:m]
        return self._out[:m], out_len


def grid_subsample_async(points, lengths, dl, features=None):
    """`grid_subsample` without the host synchronisation -> PendingSubsample (single-thread schedulers yield its
    `.fetch`; `finish()` alone is the blocking form)."""
    from .. import ops
    lib = _lib.load()
    points = _pts(points, "grid_subsample.points")
    n = points.shape[0]
    la, lp = _lens(lengths)
    out = torch.empty_like(points)
    fdim, of = 0, None
    if features is not None:
        features = features.to(torch.float32).contiguous()
        fdim = features.shape[1]
        of = torch.empty_like(features)
    sb = int(lib.apr_grid_subsample_scratch_bytes(n))
    scratch = torch.empty(sb, dtype=torch.uint8, device=points.device)
    ls = torch.empty(len(la) + 1, dtype=torch.int32, device=points.device)
    check(lib.apr_grid_subsample_async(ptr(points), n, lp, len(la), float(dl), ptr(features), fdim, ptr(out), ptr(of),
                                       ptr(ls), ptr(scratch), sb, stream()))

    def then(host):
        if host[-1] != 0:
            raise _lib.AprHipError("apr_grid_subsample: cell index outside the packed-key range")
        return host[:-1].astype(np.int32).copy()

    return PendingSubsample(ops.PendingFetch(ls, then), out, of)


def radius_neighbors(queries, supports, q_lengths, s_lengths, radius, limit=0):
    """int32 [Nq, width] neighbour table sorted by distance, padded with len(supports)."""
    lib = _lib.load()
    queries, supports = _pts(queries, "radius.queries"), _pts(supports, "radius.supports")
    nq, ns = queries.shape[0], supports.shape[0]
    qa, qp = _lens(q_lengths)
    sa, sp = _lens(s_lengths)
    if len(qa) != len(sa):
        raise _lib.AprHipError("radius_neighbors: query / support batch counts differ")
    sb = int(lib.apr_radius_scratch_bytes(nq, ns))
    scratch = torch.empty(sb, dtype=torch.uint8, device=queries.device)
    width = C.c_int32(0)
    if limit <= 0:   # unknown width: size query first
        check(lib.apr_radius_neighbors(ptr(queries), nq, ptr(supports), ns, qp, sp, len(qa), float(radius), 0, None, 0,
                                       C.byref(width), ptr(scratch), sb, stream()))
        cap = max(int(width.value), 1)
    else:
        cap = int(limit)
    out = torch.empty((nq, cap), dtype=torch.int32, device=queries.device)
    check(lib.apr_radius_neighbors(ptr(queries), nq, ptr(supports), ns, qp, sp, len(qa), float(radius), int(limit),
                                   ptr(out), cap, C.byref(width), ptr(scratch), sb, stream()))
    w = int(width.value)
    return out if w == cap else out[:, :w].contiguous()


class SearchGrid:
    """The scratch buffer of a radius_neighbors_async call, kept so that a second query set can search the same
    support grid (apr_radius_neighbors_regrid_async)."""
    __slots__ = ("scratch", "supports", "radius", "ns", "stream", "version")

    def __init__(self):
        self.scratch = self.supports = self.radius = self.ns = self.stream = self.version = None


def radius_neighbors_async(queries, supports, q_lengths, s_lengths, radius, limit, flags, keep_grid=None,
                           grid=None):
    """`radius_neighbors(..., limit)` without a host synchronisation: int32 [Nq, limit] (nearest first, padded with
    len(supports)); `flags` (int32[2] on the device) receives the largest neighbour count and the overflow flag —
    the caller reads them when convenient (`finish_radius_tables`)."""
    lib = _lib.load()
    queries, supports = _pts(queries, "radius.queries"), _pts(supports, "radius.supports")
    nq, ns = queries.shape[0], supports.shape[0]
    qa, qp = _lens(q_lengths)
    sa, sp = _lens(s_lengths)
    if len(qa) != len(sa):
        raise _lib.AprHipError("radius_neighbors: query / support batch counts differ")
    if limit <= 0 or flags.dtype != torch.int32 or flags.numel() < 2 or not flags.is_contiguous():
        raise _lib.AprHipError("radius_neighbors_async: needs limit > 0 and a contiguous int32[2] flag tensor")
    out = torch.empty((nq, int(limit)), dtype=torch.int32, device=queries.device)
    sb = int(lib.apr_radius_scratch_bytes(nq, ns))
    # the grid in the scratch is only valid for the tensor it was built from, unmodified since (in-place writes bump
    # `_version`), and for work ordered behind the build: the same stream
    if (grid is not None and grid.supports is supports and grid.radius == float(radius) and grid.ns == ns
            and grid.scratch.numel() >= sb and grid.stream == stream().value and grid.version == supports._version):
        # `keep_grid` of an earlier call on the same supports / radius: search its grid, no rebuild
        check(lib.apr_radius_neighbors_regrid_async(ptr(queries), nq, ptr(supports), ns, qp, sp, len(qa), float(radius),
                                                    int(limit), ptr(out), int(limit), ptr(flags), ptr(grid.scratch),
                                                    grid.scratch.numel(), stream()))
        return out
    if keep_grid is not None:       # room for a later, possibly larger, query set on this grid
        sb = max(sb, int(lib.apr_radius_scratch_bytes(max(nq, ns), ns)))
    scratch = torch.empty(sb, dtype=torch.uint8, device=queries.device)
    check(lib.apr_radius_neighbors_async(ptr(queries), nq, ptr(supports), ns, qp, sp, len(qa), float(radius), int(limit),
                                         ptr(out), int(limit), ptr(flags), ptr(scratch), sb, stream()))
    if keep_grid is not None:
        keep_grid.scratch, keep_grid.supports, keep_grid.radius, keep_grid.ns = scratch, supports, float(radius), ns
        keep_grid.stream, keep_grid.version = stream().value, supports._version
    return out


def finish_radius_tables(tables, flags_all):
    """One synchronisation for a whole pyramid of `radius_neighbors_async` tables: flags_all int32 [n, 2] on the
    device.  Returns the tables cut to the reference's width min(max count, limit) (columns beyond the largest
    neighbour count hold padding only); raises like `radius_neighbors` if a query overflowed the candidate buffer."""
    return finish_radius_tables_async(tables, flags_all).finish()


def finish_radius_tables_async(tables, flags_all):
    """-> ops.PendingFetch whose finish() is `finish_radius_tables`' result."""
    from .. import ops

    def then(host):
        done = []
        for t, (maxc, status) in zip(tables, host):
            if status != 0:
                raise _lib.AprHipError("apr_radius_neighbors: a query has more neighbours within the radius than the "
                                       "kernel's candidate buffer holds")
            done.append(t if maxc >= t.shape[1] else t[:, :int(maxc)].contiguous())
        return done

    return ops.PendingFetch(flags_all[:len(tables)], then)


def knn(points, k, skip_first=True):
    """int32 [N,k] nearest neighbours inside one cloud (the point itself dropped when skip_first)."""
    lib = _lib.load()
    points = _pts(points, "knn.points")
    n = points.shape[0]
    out = torch.empty((n, k), dtype=torch.int32, device=points.device)
    check(lib.apr_knn(ptr(points), n, int(k), 1 if skip_first else 0, ptr(out), stream()))
    return out
