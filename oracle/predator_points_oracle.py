"""ORACLE (test infrastructure only).  The REFERENCE's own C++ core for grid subsampling and radius
neighbours, compiled from /root/reference/Predator_APR/cpp_wrappers where it lies into
oracle/_ref/libpredator_ref.so (`make -C oracle ref`) and called through oracle/ref_shim.cpp:

  subsample_batch  <- batch_grid_subsampling   cpp_subsampling/grid_subsampling/grid_subsampling.cpp:109-211
  batch_query      <- batch_nanoflann_neighbors cpp_neighbors/neighbors/neighbors.cpp:211-333

Same call shapes as the reference's CPython extension (wrapper.cpp:75-82 / :71-75).  This is the
reference itself, so the HIP kernels' parity on this part is pinned directly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libpredator_ref.so")
_lib = None


def available():
    return os.path.exists(_SO)


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_SO)
        _lib.ref_subsample_batch.restype = C.c_int
        _lib.ref_batch_query.restype = C.c_int
    return _lib


def subsample_batch(points, batches, sampleDl=0.1, max_p=0):
    points = np.ascontiguousarray(points, dtype=np.float32)
    batches = np.ascontiguousarray(batches, dtype=np.int32)
    out = np.empty_like(points)
    out_b = np.zeros_like(batches)
    n = _load().ref_subsample_batch(points.ctypes.data_as(C.c_void_p), C.c_int(len(points)),
                                    batches.ctypes.data_as(C.c_void_p), C.c_int(len(batches)), C.c_float(sampleDl),
                                    C.c_int(max_p), out.ctypes.data_as(C.c_void_p), out_b.ctypes.data_as(C.c_void_p))
    return out[:n].copy(), out_b


def batch_query(queries, supports, q_batches, s_batches, radius=1.0):
    queries = np.ascontiguousarray(queries, dtype=np.float32)
    supports = np.ascontiguousarray(supports, dtype=np.float32)
    qb = np.ascontiguousarray(q_batches, dtype=np.int32)
    sb = np.ascontiguousarray(s_batches, dtype=np.int32)
    ptr = C.POINTER(C.c_int)()
    mc = _load().ref_batch_query(queries.ctypes.data_as(C.c_void_p), C.c_int(len(queries)),
                                 supports.ctypes.data_as(C.c_void_p), C.c_int(len(supports)),
                                 qb.ctypes.data_as(C.c_void_p), sb.ctypes.data_as(C.c_void_p), C.c_int(len(qb)),
                                 C.c_float(radius), C.byref(ptr))
    out = np.ctypeslib.as_array(ptr, shape=(len(queries) * max(mc, 1),)).copy()[: len(queries) * mc]
    _load().ref_free(ptr)
    return out.reshape(len(queries), mc).astype(np.int32)


def canonical_rows(points, lengths):
    """Sort the rows of every cloud lexicographically (the reference's row order is libstdc++-specific)."""
    out, s = [], 0
    for n in lengths:
        p = points[s:s + n]
        out.append(p[np.lexsort((p[:, 2], p[:, 1], p[:, 0]))])
        s += n
    return np.concatenate(out)
