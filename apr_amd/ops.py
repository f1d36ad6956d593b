"""Host-side operator layer: torch tensors in/out, every op a call into libapr_hip.so.

torch supplies device memory and the current HIP stream only; there is no
PyTorch fallback for any of these ops (they raise if handed CPU tensors or if the
library is missing).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream


def _lib_():
    return _lib.load()


def _f32(t, name):
    if not t.is_cuda or t.dtype != torch.float32:
        raise _lib.AprHipError(f"{name}: need a float32 GPU tensor (no CPU fallback), got {t.dtype} on {t.device}")
    return t


def _rows(t, name):
    """2-D row-major view (possibly a column slice of a wider buffer) -> (tensor, ld)."""
    _f32(t, name)
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise _lib.AprHipError(f"{name}: need a 2-D tensor with unit column stride")
    return t, (t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1]))


# ----------------------------------------------------------------------------
# voxel hashing / coordinate maps
# ----------------------------------------------------------------------------

def voxelize(xyz: torch.Tensor, voxel_size: float, batch: int = 0, out=None) -> torch.Tensor:
    """xyz f32[n,3] -> int32 [n,4] (batch, floor(xyz / voxel_size)); `out`: a contiguous [n,4] int32 row slice."""
    xyz = _f32(xyz, "voxelize.xyz").contiguous()
    n = xyz.shape[0]
    if out is None:
        coords = torch.empty((n, 4), dtype=torch.int32, device=xyz.device)
    else:
        if out.dtype != torch.int32 or tuple(out.shape) != (n, 4) or not out.is_contiguous() or out.device != xyz.device:
            raise _lib.AprHipError("voxelize: `out` must be a contiguous int32 [n,4] tensor on the same device")
        coords = out
    check(_lib_().apr_voxelize(ptr(xyz), n, float(voxel_size), int(batch), ptr(coords), stream()))
    return coords


def kernel_map_transpose(nbr: torch.Tensor, n_in: int, prefilled=None) -> torch.Tensor:
    """nbr int32 [n_out, K] over an n_in-row input map -> its transpose int32 [n_in, K] (see apr_kernel_map_transpose).
    `prefilled`: a contiguous int32 [n_in, K] tensor that already holds -1 everywhere (fill_bytes(.., 0xFF)); the scatter
    then runs without a fill of its own."""
    if nbr.dtype != torch.int32 or not nbr.is_contiguous() or nbr.dim() != 2:
        raise _lib.AprHipError("kernel_map_transpose: nbr must be a contiguous int32 [n_out, K] tensor")
    n_out, K = nbr.shape
    if prefilled is not None:
        if prefilled.dtype != torch.int32 or tuple(prefilled.shape) != (n_in, K) or not prefilled.is_contiguous():
            raise _lib.AprHipError("kernel_map_transpose: prefilled table must be contiguous int32 [n_in, K]")
        check(_lib_().apr_kernel_map_transpose_prefilled(ptr(nbr), n_out, K, n_in, ptr(prefilled), stream()))
        return prefilled
    out = torch.empty((n_in, K), dtype=torch.int32, device=nbr.device)
    check(_lib_().apr_kernel_map_transpose(ptr(nbr), n_out, K, n_in, ptr(out), stream()))
    return out


def fill_bytes(t: torch.Tensor, byte_value: int):
    """Every byte of the contiguous GPU tensor `t` <- byte_value (hipMemsetAsync on the current stream)."""
    if not t.is_cuda or not t.is_contiguous():
        raise _lib.AprHipError("fill_bytes: need a contiguous GPU tensor")
    check(_lib_().apr_fill_bytes(ptr(t), int(byte_value), t.numel() * t.element_size(), stream()))
    return t


MAX_FRAMES = 64      # APR_MAX_FRAMES


def _frame_table(clouds):
    n = len(clouds)
    ptrs = (C.c_void_p * n)()
    offs = (C.c_int64 * (n + 1))()
    keep = []
    for b, c in enumerate(clouds):
        c = _f32(c, "frames").contiguous()
        if c.dim() != 2 or c.shape[1] != 3:
            raise _lib.AprHipError("frames: every frame must be float32 [n, 3]")
        keep.append(c)
        ptrs[b] = c.data_ptr()
        offs[b + 1] = offs[b] + c.shape[0]
    return ptrs, offs, keep


def voxelize_frames(clouds, voxel_size: float):
    """Frames (list of f32 [n_b, 3] GPU tensors, at most MAX_FRAMES) -> (coords int32 [sum n_b, 4] with the frame index as
    batch id, offsets int64 GPU [len + 1], offsets as a Python list).  No concatenated copy of the points is made: the frame
    pointers travel in the kernel arguments (apr_voxelize_frames)."""
    if not 1 <= len(clouds) <= MAX_FRAMES:
        raise _lib.AprHipError(f"voxelize_frames: 1 .. {MAX_FRAMES} frames")
    ptrs, offs, keep = _frame_table(clouds)
    n = int(offs[len(clouds)])
    dev = keep[0].device
    coords = torch.empty((n, 4), dtype=torch.int32, device=dev)
    offs_dev = torch.empty(len(clouds) + 1, dtype=torch.int64, device=dev)
    check(_lib_().apr_voxelize_frames(ptrs, offs, len(clouds), float(voxel_size), ptr(coords), ptr(offs_dev), stream()))
    return coords, offs_dev, [int(o) for o in offs]


def gather_frame_points(clouds, m):
    """The input point behind every row of map `m` (built with want_first over voxelize_frames(clouds)): f32 [rows, 3], rows =
    m.n when known, else the allocation's upper bound (only the first *m.n_dev rows are written).  No sync."""
    if m.first is None:
        raise _lib.AprHipError("gather_frame_points: build the map with want_first=True")
    ptrs, offs, keep = _frame_table(clouds)
    n_max = int(m.first.shape[0])
    pts = torch.empty((n_max, 3), dtype=torch.float32, device=m.first.device)
    check(_lib_().apr_gather_frame_points(ptrs, offs, len(clouds), ptr(m.first), ptr(m.n_dev), n_max, ptr(pts), stream()))
    return pts


class VoxelPyramid:
    """The front end of a step in flight (apr_voxel_pyramid: frames -> voxel coordinates -> de-duplicated stride-1 map ->
    compact table -> three coarser maps, ONE library call over ONE arena) and the fetch of its header.  `pending`: the
    PendingFetch to wait for; `finish()` -> (maps {1, 2, 4, 8: CoordMap}, rows per frame, bounding box, representative points
    f32 [rows, 3], first-point indices int64 [rows], zeroed pair-list counters) as views of the arena, or None when the
    compact table turned out too small (more distinct voxels than a quarter of the points: the caller goes through the
    tensor-by-tensor path)."""

    def __init__(self, clouds, voxel_size):
        lib = _lib_()
        ptrs, offs, keep = _frame_table(clouds)
        self.nseg, self.n_points = len(clouds), int(offs[len(clouds)])
        sb = int(lib.apr_voxel_pyramid_scratch_bytes(self.n_points, self.nseg))
        if sb == 0:
            raise _lib.AprHipError(f"voxel_pyramid: 1 .. {MAX_FRAMES} non-empty frames")
        self.arena = torch.empty(sb, dtype=torch.uint8, device=keep[0].device)
        self.py = _lib.Pyramid()
        check(lib.apr_voxel_pyramid(ptrs, offs, self.nseg, float(voxel_size), ptr(self.arena), sb, C.byref(self.py), stream()))
        self.offsets = [int(o) for o in offs]
        self.pending = PendingFetch(self._view(self.py.header, self.py.header_ints, torch.int32), self._then, keep=keep)

    def _view(self, p, count, dtype, cols=None):
        off = p - self.arena.data_ptr()
        t = self.arena[off:off + count * dtype.itemsize].view(dtype)
        return t if cols is None else t.view(-1, cols)

    def _then(self, host):
        py, nseg = self.py, self.nseg
        if any(int(host[2 * i + 1]) != 0 for i in range(5)):
            raise _lib.AprHipError("coordinate outside the packed voxel-key range (|xyz| < 2^17 voxels, batch < 1023)")
        n0 = int(host[0])
        if py.compact and int(host[8]) > py.compact_rows:
            return None
        maps = {}
        for l in range(4):
            lv, m = py.lv[l], CoordMap()
            m.n, m.cap, m.n_in = int(host[2 * l]), int(lv.cap), int(lv.n)
            m.coords = self._view(lv.coords, m.n * 4, torch.int32, 4)
            m.keys, m.vals = self._view(lv.keys, m.cap, torch.int64), self._view(lv.vals, m.cap, torch.int32)
            m.n_dev = m.status = None
            m.first = None
            maps[1 << l] = m
        maps[1].first = first = self._view(py.first, n0, torch.int64)
        counts = [int(c) for c in host[10:10 + nseg]]
        bbox = [int(v) for v in host[10 + nseg:18 + nseg]]
        pts = self._view(py.pts, n0 * 3, torch.float32, 3)
        counters = self._view(py.counters, py.n_counter_slots * pair_counter_ints(), torch.int32, pair_counter_ints())
        return maps, counts, bbox, pts, first, counters

    def finish(self):
        return self.pending.finish()


def pack_i32(parts, zero=None):
    """Small int32 GPU tensors -> one int32 GPU vector (their concatenation), by ONE launch (apr_pack_i32); `zero`: an int32
    GPU tensor cleared by the same launch."""
    parts = [p.reshape(-1) for p in parts]
    for p in parts:
        if p.dtype != torch.int32 or not p.is_cuda or not p.is_contiguous():
            raise _lib.AprHipError("pack_i32: parts must be contiguous int32 GPU tensors")
    dev = parts[0].device if parts else zero.device
    total = sum(p.numel() for p in parts)
    dst = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
    lib = _lib_()
    zp, zw = (ptr(zero), zero.numel()) if zero is not None else (None, 0)
    if zero is not None and (zero.dtype != torch.int32 or not zero.is_contiguous()):
        raise _lib.AprHipError("pack_i32: `zero` must be a contiguous int32 GPU tensor")
    done = 0
    while True:          # APR_MAX_PACK sources per launch
        chunk = parts[done:done + 96]
        srcs = (C.c_void_p * max(len(chunk), 1))(*[p.data_ptr() for p in chunk])
        cnts = (C.c_int32 * max(len(chunk), 1))(*[p.numel() for p in chunk])
        off = sum(p.numel() for p in parts[:done])
        check(lib.apr_pack_i32(srcs, cnts, len(chunk), C.c_void_p(dst.data_ptr() + 4 * off), zp, zw, stream()))
        done += len(chunk)
        zp, zw = None, 0
        if done >= len(parts):
            break
    return dst[:total] if total else dst[:0]


def voxelize_segments(xyz_all: torch.Tensor, voxel_size: float, offsets: torch.Tensor) -> torch.Tensor:
    """Concatenated frames f32 [n,3] + int64 GPU offsets [nseg+1] -> int32 [n,4] with the frame index as batch id."""
    xyz_all = _f32(xyz_all, "voxelize_segments.xyz").contiguous()
    if offsets.dtype != torch.int64 or not offsets.is_cuda or offsets.dim() != 1 or offsets.shape[0] < 2:
        raise _lib.AprHipError("voxelize_segments: offsets must be an int64 GPU vector of nseg + 1 entries")
    n = xyz_all.shape[0]
    coords = torch.empty((n, 4), dtype=torch.int32, device=xyz_all.device)
    check(_lib_().apr_voxelize_segments(ptr(xyz_all), n, float(voxel_size), ptr(offsets.contiguous()),
                                        offsets.shape[0] - 1, ptr(coords), stream()))
    return coords


class CoordMap:
    """One coordinate map: unique int32 [n,4] rows + its hash table (key -> row)."""

    __slots__ = ("coords", "keys", "vals", "cap", "n", "n_dev", "status", "first", "n_in")

    def rows(self):
        return self.coords[: self.n]


def build_map(coords_in: torch.Tensor, floor_to: int = 0, n_in_dev=None, want_first=False) -> CoordMap:
    """Enqueue a map build.  `m.n` is None until `finalize_maps` has synced."""
    if not coords_in.is_cuda or coords_in.dtype != torch.int32 or coords_in.dim() != 2 or coords_in.shape[1] != 4:
        raise _lib.AprHipError("build_map: coordinates must be an int32 [n,4] GPU tensor")
    coords_in = coords_in.contiguous()
    lib = _lib_()
    n = coords_in.shape[0]
    dev = coords_in.device
    m = CoordMap()
    m.cap = int(lib.apr_hash_capacity(n))
    m.keys = torch.empty(m.cap, dtype=torch.int64, device=dev)
    m.vals = torch.empty(m.cap, dtype=torch.int32, device=dev)
    m.coords = torch.empty((n, 4), dtype=torch.int32, device=dev)
    m.first = torch.empty(n, dtype=torch.int64, device=dev) if want_first else None
    m.n_dev = torch.empty(1, dtype=torch.int32, device=dev)
    m.status = torch.empty(1, dtype=torch.int32, device=dev)
    m.n = None
    m.n_in = n
    sb = int(lib.apr_map_scratch_bytes(n))
    scratch = torch.empty(sb, dtype=torch.uint8, device=dev)
    check(lib.apr_map_build(ptr(coords_in), n, ptr(n_in_dev), int(floor_to), ptr(m.keys), ptr(m.vals), m.cap,
                            ptr(m.coords), ptr(m.first), ptr(m.n_dev), ptr(m.status), ptr(scratch), sb, stream()))
    return m


def segment_counts(m: CoordMap, offsets: torch.Tensor) -> torch.Tensor:
    """Rows of map `m` (built with want_first over concatenated point sets) per segment [offsets[b], offsets[b+1]).
    -> int32 [len(offsets) - 1] on the device (no sync; `m` may still be unfinalised)."""
    if m.first is None:
        raise _lib.AprHipError("segment_counts: build the map with want_first=True")
    if offsets.dtype != torch.int64 or not offsets.is_cuda or offsets.dim() != 1:
        raise _lib.AprHipError("segment_counts: offsets must be an int64 GPU vector")
    offsets = offsets.contiguous()
    nseg = offsets.shape[0] - 1
    counts = torch.empty(max(nseg, 0), dtype=torch.int32, device=offsets.device)
    check(_lib_().apr_segment_counts(ptr(m.first), ptr(m.n_dev), ptr(offsets), nseg, ptr(counts), stream()))
    return counts


def finalize_maps(maps, extras=()):
    """One host sync for any number of pending maps: fetch row counts + status flags (+ any small int32 device
    tensors in `extras`, returned as numpy arrays)."""
    pend = [m for m in maps if m.n is None]
    if not pend and not extras:
        return []
    host = pack_i32(_finalize_parts(pend, extras)).cpu().numpy()
    out, pos = [], 2 * len(pend)
    for e in extras:
        out.append(host[pos:pos + e.numel()].copy())
        pos += e.numel()
    _apply_finalize(pend, host)
    return out


# Fetch events are created with hipEventBlockingSync: a host thread waiting for a step's bytes sleeps in the driver instead
# of spinning on the event (3 waiting threads per rank x 8 ranks would otherwise burn the node's CPU quota idling;
# APR_BLOCKING_EVENTS=0 restores the spin wait for A/B runs).
BLOCKING_EVENTS = os.environ.get("APR_BLOCKING_EVENTS", "1") != "0"
# ... which turned out not to be enough on this stack: measured per worker thread (scripts/host_cpu_split.py), a thread inside
# hipEventSynchronize burns CPU for the whole wall time of the wait with the blocking flag set or not (0.254 s of CPU over
# 0.254 s of waiting; 3.06 CPUs busy per rank for 0.3 CPUs' worth of enqueue work).  The fetch wait therefore POLLS, inside
# the library (apr_event_wait: hipEventQuery + nanosleep(APR_FETCH_POLL_US, default 25 us), the thread's timer slack at
# 1 us), i.e. with the interpreter lock released for the whole wait.  A first version polled from Python (event.query() +
# time.sleep): every waiting thread then took the GIL 25 000 times a second, and on a box with a slow host the enqueueing
# threads lost 14 % of the throughput to it (2259 against 2616 pairs/s).  APR_FETCH_WAIT=sync restores hipEventSynchronize,
# APR_FETCH_WAIT=pypoll the Python loop.
FETCH_WAIT = os.environ.get("APR_FETCH_WAIT", "poll")
FETCH_POLL_S = float(os.environ.get("APR_FETCH_POLL_US", "25")) * 1e-6
FETCH_TIMEOUT_S = float(os.environ.get("APR_FETCH_TIMEOUT_S", "300"))
_SLACK = __import__("threading").local()


def fine_sleep_slack():
    """Linux rounds a thread's sleeps up by its timer slack (50 us by default): a 25 us poll would really be a 75 us one and a
    caller with ONE step in flight (two fetches per pair) pays the difference as latency (measured: 1.36 -> 1.50 ms per pair).
    PR_SET_TIMERSLACK = 1 us for the calling thread, once per thread; silently skipped where prctl is missing.  Only the
    APR_FETCH_WAIT=pypoll diagnostics path calls this (a lasting change of the caller's thread); the default wait lowers the
    slack inside the library for the duration of the wait and restores it (apr_event_wait)."""
    if getattr(_SLACK, "done", False):
        return
    _SLACK.done = True
    try:
        C.CDLL(None, use_errno=True).prctl(29, C.c_ulong(1000), 0, 0, 0)      # PR_SET_TIMERSLACK, nanoseconds
    except (OSError, AttributeError):
        pass


def fetch_event():
    return torch.cuda.Event(blocking=BLOCKING_EVENTS)


def wait_event(ev, mode=None):
    """Host wait for a fetch event without spinning on a CPU and without holding the GIL (see FETCH_WAIT above).  `mode`
    overrides the process default for this wait: "sync" = hipEventSynchronize, i.e. a spinning core and the lowest wake-up
    latency -- what a caller with ONE step in flight wants (a sleeping core of an otherwise idle host takes 50-100 us to come
    back: 775 -> 679 pairs/s one pair at a time), and what a rank sharing a node with seven others does not."""
    mode = mode or FETCH_WAIT
    if mode == "sync":
        ev.synchronize()
        return
    if mode == "pypoll":
        import time
        fine_sleep_slack()
        while not ev.query():
            time.sleep(FETCH_POLL_S)
        return
    if ev.query():      # landed already (or never recorded): nothing to wait for
        return
    # a deadline instead of waiting for ever on a wedged queue (APR_FETCH_TIMEOUT_S, default 300 s; 0 = none): the library
    # returns APR_ETIMEOUT and `check` raises, with the GIL free the whole time
    if FETCH_TIMEOUT_S > 0:
        check(_lib_().apr_event_wait_timeout(C.c_void_p(ev.cuda_event), int(FETCH_POLL_S * 1e6), int(FETCH_TIMEOUT_S * 1e6)))
    else:
        check(_lib_().apr_event_wait(C.c_void_p(ev.cuda_event), int(FETCH_POLL_S * 1e6)))


class PendingFetch:
    """A small device -> pinned-host copy in flight on the current stream: `event` completes when the bytes have
    landed, `finish()` then runs the host-side continuation and returns its value.  Lets ONE host thread keep several
    steps in flight (apr_amd.fcgf.pipeline.run_pipelined) instead of blocking in `.cpu()`."""

    __slots__ = ("event", "_host", "_then", "_keep")

    def __init__(self, dev_tensor, then, keep=()):
        self._host = torch.empty(dev_tensor.shape, dtype=dev_tensor.dtype, pin_memory=True)
        self._host.copy_(dev_tensor, non_blocking=True)
        self.event = fetch_event()
        self.event.record()
        self._then, self._keep = then, keep

    def wait(self, mode=None):
        wait_event(self.event, mode)

    def finish(self):
        wait_event(self.event)
        return self._then(self._host.numpy())


def drive(gen, wait=None):
    """Run a generator that yields PendingFetch objects (register_batch_phases, collate_phases) to completion on the
    calling thread, waiting at every fetch: the blocking form of a pipelined step.  -> the generator's return value.
    `wait`: how to wait (wait_event's `mode`; None = the process default)."""
    try:
        pending = next(gen)
        while True:
            pending.wait(wait)
            pending = gen.send(None)
    except StopIteration as stop:
        return stop.value


def _finalize_parts(pend, extras):
    parts = []
    for m in pend:
        parts += [m.n_dev, m.status]
    return parts + [e.reshape(-1) if e.dtype == torch.int32 else e.reshape(-1).to(torch.int32) for e in extras]


def finalize_maps_async(maps, extras=(), zero=None):
    """`finalize_maps` without the host synchronisation -> PendingFetch whose finish() applies the counts and returns
    the extras (numpy arrays).  The sizes, flags and extras are gathered by ONE launch (pack_i32), which also clears the
    int32 tensor `zero` if given."""
    pend = [m for m in maps if m.n is None]
    parts = _finalize_parts(pend, extras)
    if not parts:
        parts = [torch.empty(1, dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()))]

    def then(host):
        out, pos = [], 2 * len(pend)
        for e in extras:
            out.append(host[pos:pos + e.numel()].copy())
            pos += e.numel()
        _apply_finalize(pend, host)
        return out

    return PendingFetch(pack_i32(parts, zero=zero), then)


def _apply_finalize(pend, host):
    for i, m in enumerate(pend):
        if host[2 * i + 1] != 0:
            raise _lib.AprHipError(
                "coordinate outside the packed voxel-key range (|xyz| < 2^17 voxels, batch < 1023)")
        m.n = int(host[2 * i])
        m.coords = m.coords[: m.n]
        if m.first is not None:
            m.first = m.first[: m.n]


def kernel_map(out_map: CoordMap, in_map: CoordMap, kernel_size: int, scale: int) -> torch.Tensor:
    """nbr int32 [n_out, k^3]; both maps must be finalized."""
    lib = _lib_()
    K = kernel_size ** 3
    n_out = out_map.n
    nbr = torch.empty((n_out, K), dtype=torch.int32, device=out_map.coords.device)
    if out_map is in_map and kernel_size >= 5 and kernel_size % 2 == 1 and scale > 0:
        # same-level map: symmetric relation, half the probes (apr_kernel_map_same).  Pays for the 5^3 map of conv1
        # (190 k rows: 186 -> 124 us incl. the -1 fill); on 3^3 maps the extra fill launch eats the gain.
        check(lib.apr_kernel_map_same(ptr(out_map.coords), n_out, ptr(in_map.keys), ptr(in_map.vals), in_map.cap,
                                      int(kernel_size), int(scale), ptr(nbr), stream()))
        return nbr
    check(lib.apr_kernel_map(ptr(out_map.coords), n_out, None, ptr(in_map.keys), ptr(in_map.vals), in_map.cap,
                             int(kernel_size), int(scale), ptr(nbr), stream()))
    return nbr


# ----------------------------------------------------------------------------
# sparse convolution
# ----------------------------------------------------------------------------

def pack_weights(w: torch.Tensor) -> torch.Tensor:
    """[K,cin,cout] (or [cin,cout] for kernel_size 1) -> kernel-native packed layout."""
    w = _f32(w.detach(), "pack_weights.w")
    if w.dim() == 2:
        w = w.unsqueeze(0)
    w = w.contiguous()
    K, cin, cout = w.shape
    lib = _lib_()
    wp = torch.empty(int(lib.apr_spconv_packed_size(K, cin, cout)), dtype=torch.float32, device=w.device)
    check(lib.apr_spconv_pack_weights(ptr(w), K, cin, cout, ptr(wp), stream()))
    return wp


def pack_weights_bf3(w: torch.Tensor, flip=False, transposed=False):
    """[K,cin,cout] fp32 kernel -> the 3-way bf16 split image of apr_spconv_ws_fwd_bf3 (uint8 blob), or None when the
    shape is not covered (sparse kernels: cin not in 64/128/192/256/384 or cout % 64 != 0; the dense K = 1 forms take any
    64-multiples, and cin 64..192 step 32 with cout 32 / 64 / 128).  `transposed` (with `flip`: offsets mirrored): the
    image of the input gradient's kernel [K, cout, cin] straight from the parameter (apr_spconv_pack_weights_bf3_ex)."""
    w = _f32(w.detach(), "pack_weights_bf3.w")
    if w.dim() != 3:
        return None
    K, cin, cout = w.shape
    if transposed:
        cin, cout = cout, cin
    lib = _lib_()
    if K == 1:      # dense layers: 64-multiples (apr_dense_gemm_bf3) or the row-stream kernel's shapes (apr_dense_rows_bf3)
        if not ((cin % 64 == 0 and cin >= 64 and cout % 64 == 0 and cout >= 64) or lib.apr_dense_rows_bf3_ok(cin, cout)):
            return None
    elif cin not in (64, 128, 192, 256, 384) or cout % 64 != 0 or cout < 64:
        return None
    blob = torch.empty(int(lib.apr_spconv_packed_bf3_bytes(K, cin, cout)), dtype=torch.uint8, device=w.device)
    check(lib.apr_spconv_pack_weights_bf3_ex(ptr(w.contiguous()), K, cin, cout, int(bool(flip)), int(bool(transposed)), ptr(blob),
                                             stream()))
    return blob


class SpconvProfile:
    """Optional per-launch HIP-event timing of the sparse-conv kernel (bench.py roofline leg).

    Records (pairs P, cin, cout, mfma?, start, end) for every launch while active;
    events are recorded on the current stream, i.e. the stream the kernel runs on.
    """

    def __init__(self):
        self.records = []

    def pairs(self, nbr, n_out):
        if nbr is None:
            return int(n_out)
        return int((nbr >= 0).sum().item())  # not cached: map storage is recycled between pairs

    def summary(self):
        """-> algorithmic bytes / flops and summed time of the MFMA conv layers, in total and per kernel path
        ("tile" = k_spconv_pairs, "ws" = k_ws_gemm + k_ws_reduce)."""
        torch.cuda.synchronize()
        tot = dict(launches=0, bytes=0.0, flops=0.0, ms=0.0)
        by = {}
        for (P, cin, cout, mfma, e0, e1, path) in self.records:
            if not mfma:
                continue
            ms = e0 if e1 is None else e0.elapsed_time(e1)     # batch launches carry milliseconds directly
            for d in (tot, by.setdefault(path, dict(launches=0, bytes=0.0, flops=0.0, ms=0.0))):
                d["launches"] += 1
                d["bytes"] += 4.0 * P * (cin + cout) + 8.0 * P
                d["flops"] += 2.0 * P * cin * cout
                d["ms"] += ms
        tot["by_path"] = by
        return tot


PROFILE = None  # set to a SpconvProfile to time launches


class PairList:
    """Per-offset pair lists of one kernel map (apr_pairlist_build): a zeroed counter block (32 counters, one per 256 B) + a device blob.

    `built` False: only allocated; the first launch of a SpconvBatch that uses it builds it inside the same
    library call (no extra host round trip)."""

    def __init__(self, counters, blob, nbr, built):
        self.counters, self.blob, self.nbr, self.built = counters, blob, nbr, built
        self.queued = False      # its build rides in a SpconvBatch that has not been launched yet
        self.n_out, self.K = nbr.shape

    def prod_scratch(self, cout):
        """Product rows [K * n_out, cout]; offset k uses the head of its own n_out-row region."""
        return torch.empty(self.n_out * self.K * cout, dtype=torch.float32, device=self.blob.device)

    def build(self):
        if self.queued:      # a second build would add to the (already reserved) range counters
            raise _lib.AprHipError("PairList: its build is queued in a SpconvBatch that has not been launched yet")
        if not self.built:
            check(_lib_().apr_pairlist_build(ptr(self.nbr), self.n_out, self.K, ptr(self.counters), ptr(self.blob),
                                             self.blob.numel(), stream()))
            self.built = True
        return self

    def counts(self):
        """Pairs per offset (host sync; tests / diagnostics)."""
        return self.build().counters[::pair_counter_stride()][:self.K].cpu().numpy()


class PairList3(PairList):
    """Triple pair lists of a 27-offset kernel map (apr_pairlist3_build): an entry = an output row with its up to three
    x-neighbours of one (dy, dz); the weight-stationary gemm writes ONE product row per entry -- about half as many as
    pairs (apr_spconv_ws3_fwd_bf3).  Same life cycle as PairList (lazy build inside a SpconvBatch)."""

    def prod_scratch(self, cout):
        """Product rows [9 * n_out, cout]; triple t uses the head of its own n_out-row region."""
        return torch.empty(self.n_out * 9 * cout, dtype=torch.float32, device=self.blob.device)

    def build(self):
        if self.queued:
            raise _lib.AprHipError("PairList3: its build is queued in a SpconvBatch that has not been launched yet")
        if not self.built:
            check(_lib_().apr_pairlist3_build(ptr(self.nbr), self.n_out, self.K, ptr(self.counters), ptr(self.blob),
                                              self.blob.numel(), stream()))
            self.built = True
        return self

    def counts(self):
        """Entries per triple (host sync; tests / diagnostics)."""
        return self.build().counters[::pair_counter_stride()][:9].cpu().numpy()


def ws3_supported(K, cin, cout):
    return bool(_lib_().apr_spconv_ws3_supported(int(K), int(cin), int(cout)))


def build_pairlist3(nbr, lazy=False, counters=None):
    """nbr int32 [n_out, 27] -> PairList3 (see build_pairlist for `counters`)."""
    if nbr.dtype != torch.int32 or not nbr.is_contiguous() or nbr.dim() != 2 or nbr.shape[1] != 27:
        raise _lib.AprHipError("build_pairlist3: nbr must be a contiguous int32 [n_out, 27] tensor")
    nb = int(_lib_().apr_pairlist3_bytes(nbr.shape[0]))
    if counters is None:
        counters = torch.zeros(pair_counter_ints(), dtype=torch.int32, device=nbr.device)
    pl = PairList3(counters, torch.empty(nb, dtype=torch.uint8, device=nbr.device), nbr, False)
    return pl if lazy else pl.build()


def pair_counter_ints() -> int:
    """Length of the int32 counter block of one pair list (32 counters, one per 256 B: apr_pairlist_counter_ints)."""
    return int(_lib_().apr_pairlist_counter_ints())


def pair_counter_stride() -> int:
    return pair_counter_ints() // 32


def build_pairlist(nbr, lazy=False, counters=None):
    """nbr int32 [n_out, K] -> PairList.  `counters`: an all-zero int32[pair_counter_ints()] block to use (a coordinate
    manager clears the counters of all its maps with one fill); default: a fresh zero block."""
    if nbr.dtype != torch.int32 or not nbr.is_contiguous() or nbr.dim() != 2:
        raise _lib.AprHipError("build_pairlist: nbr must be a contiguous int32 [n_out, K] tensor")
    n_out, K = nbr.shape
    nb = int(_lib_().apr_pairlist_bytes(n_out, K))
    if counters is None:
        counters = torch.zeros(pair_counter_ints(), dtype=torch.int32, device=nbr.device)
    pl = PairList(counters, torch.empty(nb, dtype=torch.uint8, device=nbr.device), nbr, False)
    return pl if lazy else pl.build()


class OsPairs:
    """Per-tile pair lists of one kernel map for the output-stationary conv (apr_spconv_os_pairs_build): tiles of `R`
    consecutive output rows, K compact lists each.  `built` False: allocated only; the first launch of a SpconvBatch that
    uses it builds it inside the same library call."""

    def __init__(self, nbr, n_in, R):
        self.nbr, self.n_in, self.R = nbr, int(n_in), int(R)
        self.n_out, self.K = nbr.shape
        nb = int(_lib_().apr_spconv_os_pairs_bytes(self.n_out, self.K, self.R))
        self.blob = torch.empty(nb, dtype=torch.uint8, device=nbr.device)
        self.built = False
        self.queued = False      # its build rides in a SpconvBatch that has not been launched yet

    def build(self):
        if not self.built and not self.queued:
            check(_lib_().apr_spconv_os_pairs_build(ptr(self.nbr), self.n_out, self.n_in, self.K, self.R, ptr(self.blob),
                                                    self.blob.numel(), stream()))
            self.built = True
        return self


def os_tile_rows(n_out, cin, cout):
    """Rows per tile of the output-stationary conv for this layer shape; 0 = shape not covered."""
    return int(_lib_().apr_spconv_os_tile_rows(n_out, cin, cout))


def build_os_pairs(nbr, n_in, R, lazy=False):
    if nbr.dtype != torch.int32 or not nbr.is_contiguous() or nbr.dim() != 2:
        raise _lib.AprHipError("build_os_pairs: nbr must be a contiguous int32 [n_out, K] tensor")
    p = OsPairs(nbr, n_in, R)
    return p if lazy else p.build()


def spconv_os(x, os_pairs, cin, cout, w_bf3, scale=None, shift=None, residual=None, relu=False, out=None):
    """The sparse conv through apr_spconv_os_fwd (output-stationary, accumulators in LDS); os_pairs from build_os_pairs."""
    x, ldi = _rows(x, "spconv_os.x")
    if x.shape[1] != cin:
        raise _lib.AprHipError(f"spconv_os: input has {x.shape[1]} channels, weight expects {cin}")
    if x.shape[0] > os_pairs.n_in:
        pass
    n_out, K = os_pairs.n_out, os_pairs.K
    if out is None:
        out = torch.empty((n_out, cout), dtype=torch.float32, device=x.device)
    out, ldo = _rows(out, "spconv_os.out")
    if out.shape[0] != n_out or out.shape[1] != cout:
        raise _lib.AprHipError("spconv_os: output shape mismatch")
    ldr = 0
    if residual is not None:
        residual, ldr = _rows(residual, "spconv_os.residual")
        if residual.shape[0] != n_out or residual.shape[1] != cout:
            raise _lib.AprHipError("spconv_os: residual shape mismatch")
    if w_bf3 is None:
        raise _lib.AprHipError("spconv_os: needs the bf16-split weights (pack_weights_bf3)")
    if os_pairs.queued:
        raise _lib.AprHipError("spconv_os: the tile pair lists are queued for building in a SpconvBatch that has not been launched")
    os_pairs.build()
    check(_lib_().apr_spconv_os_fwd(ptr(x), ldi, ptr(os_pairs.blob), n_out, K, os_pairs.R, cin, cout, ptr(w_bf3),
                                    ptr(scale), ptr(shift), ptr(residual), ldr, int(bool(relu)), ptr(out), ldo, stream()))
    return out


def ws_supported(K, cin, cout):
    return K <= 27 and cin % 64 == 0 and cin <= 512 and cout % 64 == 0


def spconv(x, nbr, K, cin, cout, wp, scale=None, shift=None, residual=None, relu=False, out=None, n_out=None,
           plist=None, w_bf3=None, os_pairs=None):
    """out[j] = act((sum_o x[nbr[j,o]] @ W[o]) * scale + shift + residual[j]).

    x / residual / out may be column slices of wider row-major buffers.  With `plist` (the PairList of `nbr`)
    the launch takes the weight-stationary path (apr_spconv_ws_fwd); with `os_pairs` (the OsPairs of `nbr`) and the
    split weights the output-stationary one (apr_spconv_os_fwd).
    """
    if os_pairs is not None and w_bf3 is not None and nbr is not None:
        return spconv_os(x, os_pairs, cin, cout, w_bf3, scale=scale, shift=shift, residual=residual, relu=relu, out=out)
    x, ldi = _rows(x, "spconv.x")
    if x.shape[1] != cin:
        raise _lib.AprHipError(f"spconv: input has {x.shape[1]} channels, weight expects {cin}")
    if nbr is not None:
        if nbr.dtype != torch.int32 or not nbr.is_contiguous() or nbr.shape[1] != K:
            raise _lib.AprHipError("spconv: nbr must be a contiguous int32 [n_out, K] tensor")
        n_out = nbr.shape[0]
    elif n_out is None:
        n_out = x.shape[0]
    if out is None:
        out = torch.empty((n_out, cout), dtype=torch.float32, device=x.device)
    out, ldo = _rows(out, "spconv.out")
    if out.shape[0] != n_out or out.shape[1] != cout:
        raise _lib.AprHipError("spconv: output shape mismatch")
    ldr = 0
    if residual is not None:
        residual, ldr = _rows(residual, "spconv.residual")
        if residual.shape[0] != n_out or residual.shape[1] != cout:
            raise _lib.AprHipError("spconv: residual shape mismatch")
    if (nbr is None and K == 1 and w_bf3 is not None and PROFILE is None and n_out > 0 and ldi % 4 == 0
            and ldo % 4 == 0 and x.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0
            and (residual is None or (ldr % 4 == 0 and residual.data_ptr() % 16 == 0))
            and (scale is None or scale.data_ptr() % 16 == 0) and (shift is None or shift.data_ptr() % 16 == 0)):
        # identity map: the dense GEMMs on the bf16 split (what a batched launch picks as well) -- few input channels over
        # many rows stream past LDS-resident weights, the other 64-multiple widths take the tiled kernel
        lib = _lib_()
        if lib.apr_dense_rows_bf3_route(n_out, cin, cout):
            check(lib.apr_dense_rows_bf3(ptr(x), ldi, n_out, cin, cout, ptr(w_bf3), ptr(scale), ptr(shift), ptr(residual), ldr,
                                         int(bool(relu)), 0, ptr(out), ldo, stream()))
            return out
        if cin % 64 == 0 and cout % 64 == 0:
            check(lib.apr_dense_gemm_bf3(ptr(x), ldi, n_out, cin, cout, ptr(w_bf3), ptr(scale), ptr(shift), ptr(residual), ldr,
                                         int(bool(relu)), ptr(out), ldo, stream()))
            return out
    use_ws = plist is not None and nbr is not None and ws_supported(K, cin, cout)
    use_ws3 = use_ws and isinstance(plist, PairList3)
    if use_ws3 and (w_bf3 is None or not ws3_supported(K, cin, cout)):
        raise _lib.AprHipError("spconv: triple pair lists need the bf16-split weights and cin 64 / 128, K = 27")
    if use_ws:
        if plist.n_out != n_out or plist.K != K:
            raise _lib.AprHipError("spconv: pair list does not belong to this kernel map")
        prod = plist.build().prod_scratch(cout)
    prof = PROFILE
    if prof is not None:
        P = prof.pairs(nbr, n_out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if use_ws3:
        check(_lib_().apr_spconv_ws3_fwd_bf3(ptr(x), ldi, ptr(plist.counters), ptr(plist.blob), n_out, cin, cout, ptr(w_bf3),
                                             ptr(scale), ptr(shift), ptr(residual), ldr, int(bool(relu)), ptr(out), ldo,
                                             ptr(prod), stream()))
    elif use_ws:
        check(_lib_().apr_spconv_ws_fwd_bf3(ptr(x), ldi, ptr(plist.counters), ptr(plist.blob), n_out, K, cin, cout, ptr(wp),
                                            ptr(w_bf3), ptr(scale), ptr(shift), ptr(residual), ldr, int(bool(relu)),
                                            ptr(out), ldo, ptr(prod), stream()))
    else:
        check(_lib_().apr_spconv_fwd(ptr(x), ldi, ptr(nbr), n_out, K, cin, cout, ptr(wp), ptr(scale), ptr(shift),
                                     ptr(residual), ldr, int(bool(relu)), ptr(out), ldo, stream()))
    if prof is not None:
        e1.record()
        prof.records.append((P, cin, cout, K <= 32 and cin % 32 == 0 and cout % 32 == 0, e0, e1,
                             "ws" if use_ws else "tile"))
    return out


def dense_rows_bf3(x, w_bf3, cin, cout, scale=None, shift=None, residual=None, relu=False, l2norm=False, out=None):
    """act((x @ W) * scale + shift + residual) [then rows / |row|_2] through apr_dense_rows_bf3 (weights resident in LDS, the
    rows streamed once): cin 64..192 step 32, cout 32 / 64 / 128; `w_bf3` from pack_weights_bf3(W[None])."""
    x, ldi = _rows(x, "dense_rows_bf3.x")
    if x.shape[1] != cin:
        raise _lib.AprHipError(f"dense_rows_bf3: input has {x.shape[1]} channels, weight expects {cin}")
    n = x.shape[0]
    if out is None:
        out = torch.empty((n, cout), dtype=torch.float32, device=x.device)
    out, ldo = _rows(out, "dense_rows_bf3.out")
    if out.shape[0] != n or out.shape[1] != cout:
        raise _lib.AprHipError("dense_rows_bf3: output shape mismatch")
    ldr = 0
    if residual is not None:
        residual, ldr = _rows(residual, "dense_rows_bf3.residual")
        if residual.shape[0] != n or residual.shape[1] != cout:
            raise _lib.AprHipError("dense_rows_bf3: residual shape mismatch")
    if n == 0:
        return out
    check(_lib_().apr_dense_rows_bf3(ptr(x), ldi, n, cin, cout, ptr(w_bf3), ptr(scale), ptr(shift), ptr(residual), ldr,
                                     int(bool(relu)), int(bool(l2norm)), ptr(out), ldo, stream()))
    return out


def dense_gemm_bf3(x, w_bf3, cin, cout, scale=None, shift=None, residual=None, relu=False, out=None):
    """act((x @ W) * scale + shift + residual) through apr_dense_gemm_bf3; `w_bf3` from pack_weights_bf3(W[None])."""
    x, ldi = _rows(x, "dense_gemm_bf3.x")
    if x.shape[1] != cin:
        raise _lib.AprHipError(f"dense_gemm_bf3: input has {x.shape[1]} channels, weight expects {cin}")
    n = x.shape[0]
    if out is None:
        out = torch.empty((n, cout), dtype=torch.float32, device=x.device)
    out, ldo = _rows(out, "dense_gemm_bf3.out")
    ldr = 0
    if residual is not None:
        residual, ldr = _rows(residual, "dense_gemm_bf3.residual")
    check(_lib_().apr_dense_gemm_bf3(ptr(x), ldi, n, cin, cout, ptr(w_bf3), ptr(scale), ptr(shift), ptr(residual), ldr,
                                     int(bool(relu)), ptr(out), ldo, stream()))
    return out


WGRAD_PAD = os.environ.get("APR_WGRAD_PAD", "1") != "0"      # A/B switch: 0 = odd widths on the fp32 scalar-gather kernel


def weights_flip_transpose(w, flip):
    """[K, cin, cout] -> [K, cout, cin] with the offsets mirrored when `flip` (apr_weights_flip_transpose)."""
    w = _f32(w.detach(), "weights_flip_transpose.w").contiguous()
    K, cin, cout = w.shape
    wt = torch.empty((K, cout, cin), dtype=torch.float32, device=w.device)
    check(_lib_().apr_weights_flip_transpose(ptr(w), K, cin, cout, int(bool(flip)), ptr(wt), stream()))
    return wt


def spconv_wgrad(x, dout, nbr, K, cin, cout, same_level=False):
    """dW f32 [K, cin, cout] = sum_j [nbr[j,k] >= 0] x[nbr[j,k]]^T dout[j]  (nbr None: identity map, K = 1).
    `same_level`: nbr is a stride-1 map of an odd kernel (its centre column is full): apr_spconv_wgrad_same_level."""
    if x.shape[1] != cin or dout.shape[1] != cout:
        raise _lib.AprHipError("spconv_wgrad: channel mismatch")
    if (cin % 32 or cout % 32) and WGRAD_PAD and x.shape[0] * (-cin % 32) + dout.shape[0] * (-cout % 32) < (1 << 26):
        # odd widths (conv1's single input channel, a decoder's 12 outputs) ride on the bf16-split MFMA kernel zero-padded to
        # its 32-channel granule instead of the scalar-gather fallback (180 us per call at a frame's row count)
        cp, op = (cin + 31) // 32 * 32, (cout + 31) // 32 * 32
        xp = torch.nn.functional.pad(x, (0, cp - cin)) if cp != cin else x
        dp = torch.nn.functional.pad(dout, (0, op - cout)) if op != cout else dout
        return spconv_wgrad(xp, dp, nbr, K, cp, op, same_level=same_level)[:, :cin, :cout]
    x, ldi = _rows(x, "spconv_wgrad.x")
    dout, ldo = _rows(dout, "spconv_wgrad.dout")
    n_out = dout.shape[0]
    if nbr is not None and (nbr.dtype != torch.int32 or not nbr.is_contiguous() or tuple(nbr.shape) != (n_out, K)):
        raise _lib.AprHipError("spconv_wgrad: nbr must be a contiguous int32 [n_out, K] tensor")
    lib = _lib_()
    dw = torch.empty((K, cin, cout), dtype=torch.float32, device=x.device)
    sb = int(lib.apr_spconv_wgrad_scratch_bytes(n_out, K, cin, cout))
    scratch = torch.empty(sb, dtype=torch.uint8, device=x.device)
    fn = lib.apr_spconv_wgrad_same_level if (same_level and nbr is not None) else lib.apr_spconv_wgrad
    check(fn(ptr(x), ldi, ptr(dout), ldo, ptr(nbr), n_out, K, cin, cout, ptr(dw), ptr(scratch), sb, stream()))
    return dw


def _seg_array(segments, n):
    """(ctypes int64 array or None, nseg) for the BN-train entry points; `segments`: row offsets [0, ..., n] or None."""
    if segments is None or len(segments) <= 2:
        return None, 0
    if int(segments[0]) != 0 or int(segments[-1]) != n:
        raise _lib.AprHipError("bn_train: segment offsets must run 0 .. n")
    return (C.c_int64 * len(segments))(*[int(v) for v in segments]), len(segments) - 1


def bn_train_fwd(z, bn, residual=None, relu=False, segments=None):
    """y = act(batch_norm(z) (+ residual)) with the batch statistics of z's rows, `bn`'s running statistics updated in place
    (apr_bn_train_fwd: two launches) -> (y, save_mean, save_rstd).  `segments` (row offsets): the rows of several
    forward calls stacked into one launch, each normalised on its own (statistics [nseg, c])."""
    z, ldz = _rows(z, "bn_train_fwd.z")
    n, c = z.shape
    lib = _lib_()
    segs, nseg = _seg_array(segments, n)
    y = torch.empty((n, c), dtype=torch.float32, device=z.device)
    stats = torch.empty((2, max(nseg, 1), c), dtype=torch.float32, device=z.device)
    ldr = 0
    if residual is not None:
        residual, ldr = _rows(residual, "bn_train_fwd.residual")
    sb = int(lib.apr_bn_stats_scratch_bytes(n + 256 * max(nseg, 1), c))
    scratch = torch.empty(sb, dtype=torch.uint8, device=z.device)
    track = bn.track_running_stats and bn.running_mean is not None
    mom = 0.0
    if track:
        # momentum None (cumulative average) needs the counter's value on the host: one sync per call, not used by APR
        mom = bn.momentum if bn.momentum is not None else 1.0 / float(int(bn.num_batches_tracked) + 1)
    g = bn.weight.detach() if bn.weight is not None else None
    b = bn.bias.detach() if bn.bias is not None else None
    check(lib.apr_bn_train_fwd(ptr(z), ldz, n, c, ptr(g), ptr(b), float(bn.eps), float(mom),
                               ptr(bn.running_mean) if track else None, ptr(bn.running_var) if track else None,
                               ptr(residual), ldr, int(bool(relu)), ptr(y), c, ptr(stats[0]), ptr(stats[1]),
                               ptr(bn.num_batches_tracked) if track else None, segs, nseg, ptr(scratch), sb, stream()))
    if track:
        # the library wrote into the buffers behind torch's back: bump their version counters (folded-BN caches key on them)
        torch.autograd.graph.increment_version((bn.running_mean, bn.running_var, bn.num_batches_tracked))
    return y, stats[0], stats[1]


def bn_train_bwd(z, y, dy, mean, rstd, gamma, relu, want_dres, segments=None):
    """-> (dz, dres or None, dgamma, dbeta) of bn_train_fwd (apr_bn_train_bwd: two launches, deterministic)."""
    z, ldz = _rows(z, "bn_train_bwd.z")
    dy, lddy = _rows(dy, "bn_train_bwd.dy")
    n, c = z.shape
    lib = _lib_()
    segs, nseg = _seg_array(segments, n)
    ldy = 0
    if relu:
        y, ldy = _rows(y, "bn_train_bwd.y")
    dz = torch.empty((n, c), dtype=torch.float32, device=z.device)
    dres = torch.empty((n, c), dtype=torch.float32, device=z.device) if want_dres else None
    dgb = torch.empty((2, c), dtype=torch.float32, device=z.device)
    sb = int(lib.apr_bn_stats_scratch_bytes(n + 256 * max(nseg, 1), c))
    scratch = torch.empty(sb, dtype=torch.uint8, device=z.device)
    check(lib.apr_bn_train_bwd(ptr(z), ldz, ptr(y) if relu else None, ldy, ptr(dy), lddy, n, c, ptr(mean), ptr(rstd),
                               ptr(gamma), int(bool(relu)), ptr(dz), c, ptr(dres), c, ptr(dgb[0]), ptr(dgb[1]), segs, nseg,
                               ptr(scratch), sb, stream()))
    return dz, dres, dgb[0], dgb[1]


class BnTrainFunction(torch.autograd.Function):
    """Training-mode BatchNorm1d on rows (apr_bn_train_fwd / _bwd), optionally per row SEGMENT: the rows of several module
    calls stacked into one (each call's own statistics; running statistics updated call after call).  The NPR decoder's
    norms when its per-cloud calls (FCGF_APR/lib/complement_trainer.py:424-431) ride in one launch."""

    @staticmethod
    def forward(ctx, x, weight, bias, bn, segments):
        x = x.contiguous()
        y, mean, rstd = bn_train_fwd(x, bn, segments=segments)
        ctx.save_for_backward(x, mean, rstd, weight)
        ctx.segs = segments
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, weight = ctx.saved_tensors
        dz, _, dg, db = bn_train_bwd(x, None, dy.contiguous(), mean, rstd, weight.detach() if weight is not None else None, False,
                                     False, segments=ctx.segs)
        return dz, (dg.reshape(weight.shape) if weight is not None else None), db, None, None


class ConvBnActFunction(torch.autograd.Function):
    """One unit of a training encode -- sparse convolution -> training-mode BatchNorm -> (+ residual) -> ReLU
    (FCGF_APR/model/resunet.py:142-193, model/residual_block.py:37-53 under lib/complement_trainer.py:350-512) -- as ONE
    autograd node on the HIP kernels:
      forward   z = conv(x) through the inference routing (weight-stationary / triple lists / tile kernel, no epilogue),
                y = apr_bn_train_fwd(z) (statistics + apply; running statistics updated by the kernel);
      backward  apr_bn_train_bwd (ReLU mask, dgamma, dbeta, dz, the residual's gradient), the input gradient = the SAME
                routed convolution over the reverse map with the flipped-transposed kernel (packed once per optimizer step),
                the kernel gradient = apr_spconv_wgrad on the bf16-split MFMA.
    Without a norm (`bn` None: the two K = 1 layers behind the decoder) the bias add / ReLU ride in the conv epilogue.
    `cfg`: a dict (conv, bn, nbr, plist, nbr_bwd, plist_bwd, flip, relu, n_out[, segs: the row offsets of the stacked forward
    calls, each with its own batch statistics])."""

    @staticmethod
    def forward(ctx, x, kernel, gamma, beta, bias, residual, cfg):
        conv, bn = cfg["conv"], cfg["bn"]
        x = x.contiguous()
        n_out = cfg["n_out"]
        if bn is not None:
            z = conv.run(x, cfg["nbr"], n_out, plist=cfg["plist"], raw=True)
            y, mean, rstd = bn_train_fwd(z, bn.bn, residual=residual, relu=cfg["relu"], segments=cfg.get("segs"))
            ctx.save_for_backward(x, kernel, z, y, mean, rstd, gamma)
        else:
            y = conv.run(x, cfg["nbr"], n_out, plist=cfg["plist"], relu=cfg["relu"], residual=residual)
            ctx.save_for_backward(x, kernel, y)
        ctx.cfg = cfg
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        cfg = ctx.cfg
        conv, bn = cfg["conv"], cfg["bn"]
        dy = dy.contiguous()
        dgamma = dbeta = dbias = dres = None
        if bn is not None:
            x, kernel, z, y, mean, rstd, gamma = ctx.saved_tensors
            dz, dres, dgamma, dbeta = bn_train_bwd(z, y, dy, mean, rstd, gamma.detach() if gamma is not None else None,
                                                   cfg["relu"], ctx.has_res and ctx.needs_input_grad[5], segments=cfg.get("segs"))
            if gamma is not None:
                dgamma = dgamma.reshape(gamma.shape)
        else:
            x, kernel, y = ctx.saved_tensors
            dz = dy
            if cfg["relu"]:
                dz = torch.empty_like(dy)
                c = dy.shape[1]
                check(_lib_().apr_act_backward(ptr(dy), c, ptr(y), c, dy.shape[0], c, 1, 0.0, ptr(dz), c, stream()))
            if ctx.has_res and ctx.needs_input_grad[5]:
                dres = dz
            if conv.bias is not None and ctx.needs_input_grad[4]:
                dbias = col_sums(dz).reshape(conv.bias.shape)
        K, cin, cout = conv.kernel_volume if cfg["nbr"] is not None else 1, conv.in_channels, conv.out_channels
        din = dw = None
        if ctx.needs_input_grad[0]:
            din = conv.run_T(dz, cfg["nbr_bwd"], x.shape[0], cfg["flip"], plist=cfg["plist_bwd"])
        if ctx.needs_input_grad[1]:
            dw = spconv_wgrad(x, dz, cfg["nbr"], K, cin, cout, same_level=cfg["flip"]).reshape(kernel.shape)
        return din, dw, dgamma, dbeta, dbias, dres, None


class SparseConvFunction(torch.autograd.Function):
    """out = spconv(feats, nbr_fwd, kernel) with its two gradients on the HIP kernels:
    d feats = spconv(dout, nbr_bwd, kernel[mirror]^T)  — the same operator over the reverse map;
    d kernel = apr_spconv_wgrad.  `flip`: same-level map, reverse pairs sit under the mirrored offset K-1-k."""

    @staticmethod
    def forward(ctx, feats, kernel, nbr_fwd, nbr_bwd, flip):
        K, cin, cout = kernel.shape
        feats = feats.contiguous()
        out = spconv(feats, nbr_fwd, K, cin, cout, pack_weights(kernel.detach()))
        ctx.save_for_backward(feats, kernel)
        ctx.maps = (nbr_fwd, nbr_bwd, bool(flip))
        return out

    @staticmethod
    def backward(ctx, dout):
        feats, kernel = ctx.saved_tensors
        nbr_fwd, nbr_bwd, flip = ctx.maps
        K, cin, cout = kernel.shape
        dout = dout.contiguous()
        din = dw = None
        if ctx.needs_input_grad[0]:
            wb = (kernel.flip(0) if flip else kernel).detach().transpose(1, 2).contiguous()
            din = spconv(dout, nbr_bwd, K, cout, cin, pack_weights(wb))
        if ctx.needs_input_grad[1]:
            dw = spconv_wgrad(feats, dout, nbr_fwd, K, cin, cout)
        return din, dw, None, None, None


class SpconvBatch:
    """Collects sparse-conv launches and enqueues them with ONE library call (apr_spconv_fwd_batch)."""

    def __init__(self):
        self.descs = []
        self.keep = []      # tensors referenced by raw pointers stay alive until the launch call returns
        self.prod = {}      # weight-stationary product buffers by size
        self.meta = []      # (P, cin, cout, mfma?, path) per launch while a SpconvProfile is active
        self.pending = []   # pair lists whose build rides in this batch: marked built when the batch has been launched

    def add(self, x, nbr, K, cin, cout, wp, scale=None, shift=None, residual=None, relu=False, out=None, n_out=None,
            plist=None, w_bf3=None, os_pairs=None, l2norm=False):
        x, ldi = _rows(x, "spconv.x")
        if nbr is not None:
            n_out = nbr.shape[0]
        elif n_out is None:
            n_out = x.shape[0]
        if out is None:
            out = torch.empty((n_out, cout), dtype=torch.float32, device=x.device)
        out, ldo = _rows(out, "spconv.out")
        ldr = 0
        if residual is not None:
            residual, ldr = _rows(residual, "spconv.residual")
        d = _lib.SpconvDesc()
        d.inp, d.ldi, d.nbr, d.n_out = x.data_ptr(), ldi, (nbr.data_ptr() if nbr is not None else None), n_out
        d.K, d.cin, d.cout, d.relu = K, cin, cout, int(bool(relu))
        d.l2norm = int(bool(l2norm))        # rows of the result divided by their 2-norm (fused where the kernel allows)
        d.w_packed = wp.data_ptr()
        d.scale = scale.data_ptr() if scale is not None else None
        d.shift = shift.data_ptr() if shift is not None else None
        d.residual = residual.data_ptr() if residual is not None else None
        d.ldr, d.out, d.ldo = ldr, out.data_ptr(), ldo
        prod = None
        if os_pairs is not None and w_bf3 is not None and nbr is not None:
            if os_pairs.n_out != n_out or os_pairs.K != K:
                raise _lib.AprHipError("spconv: tile pair lists do not belong to this kernel map")
            d.os_pairs, d.os_rows, d.os_n_in, d.w_bf3 = os_pairs.blob.data_ptr(), os_pairs.R, os_pairs.n_in, w_bf3.data_ptr()
            if not os_pairs.built and not os_pairs.queued:      # built by this batch's launch; `built` is set there
                d.os_build_bytes, os_pairs.queued = os_pairs.blob.numel(), True
                self.pending.append(os_pairs)
            plist = None
        if plist is not None and nbr is not None and ws_supported(K, cin, cout):
            if plist.n_out != n_out or plist.K != K:
                raise _lib.AprHipError("spconv: pair list does not belong to this kernel map")
            is3 = isinstance(plist, PairList3)
            if is3 and (w_bf3 is None or not ws3_supported(K, cin, cout)):
                raise _lib.AprHipError("spconv: triple pair lists need the bf16-split weights and cin 64 / 128, K = 27")
            need = plist.n_out * (9 if is3 else plist.K) * cout      # launches run in order on one stream: share the scratch
            prod = self.prod.get(need)
            if prod is None:
                prod = self.prod[need] = plist.prod_scratch(cout)
            d.counters, d.plist, d.prod_scratch = plist.counters.data_ptr(), plist.blob.data_ptr(), prod.data_ptr()
            d.ws3 = int(is3)
            if w_bf3 is not None:
                d.w_bf3 = w_bf3.data_ptr()
            if not plist.built and not plist.queued:
                d.plist_bytes, plist.queued = plist.blob.numel(), True
                self.pending.append(plist)
        elif nbr is None and K == 1 and w_bf3 is not None and ldi % 4 == 0 \
                and ((cin % 64 == 0 and cout % 64 == 0) or _lib_().apr_dense_rows_bf3_ok(cin, cout)) \
                and ldo % 4 == 0 and x.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0 \
                and (residual is None or (ldr % 4 == 0 and residual.data_ptr() % 16 == 0)) \
                and (scale is None or scale.data_ptr() % 16 == 0) and (shift is None or shift.data_ptr() % 16 == 0):
            d.w_bf3 = w_bf3.data_ptr()          # identity map: the dense GEMMs on the bf16 split (the library routes by shape)
        self.descs.append(d)
        if PROFILE is not None:
            self.meta.append((PROFILE.pairs(nbr, n_out), cin, cout, K <= 32 and cin % 32 == 0 and cout % 32 == 0,
                              "os" if d.os_pairs else "ws" if prod is not None else "tile"))
        self.keep += [x, nbr, wp, scale, shift, residual, out, plist, prod, w_bf3, os_pairs]
        return out

    def launch(self):
        if not self.descs:
            return
        arr = (_lib.SpconvDesc * len(self.descs))(*self.descs)
        if PROFILE is not None and len(self.meta) == len(self.descs):
            ms = (C.c_float * len(self.descs))()
            check(_lib_().apr_spconv_fwd_batch_timed(arr, len(self.descs), ms, stream()))
            PROFILE.records += [m[:4] + (float(t), None, m[4]) for m, t in zip(self.meta, ms)]
        else:
            check(_lib_().apr_spconv_fwd_batch(arr, len(self.descs), stream()))
        for pl in self.pending:
            pl.built, pl.queued = True, False
        self.descs, self.keep, self.prod, self.meta, self.pending = [], [], {}, [], []


# ----------------------------------------------------------------------------
# normalisation / elementwise
# ----------------------------------------------------------------------------

def bn_stats(x):
    x, ld = _rows(x, "bn_stats.x")
    n, c = x.shape
    lib = _lib_()
    mean = torch.empty(c, dtype=torch.float32, device=x.device)
    var = torch.empty(c, dtype=torch.float32, device=x.device)
    sb = int(lib.apr_bn_stats_scratch_bytes(n, c))
    scratch = torch.empty(sb, dtype=torch.uint8, device=x.device)
    check(lib.apr_bn_stats(ptr(x), ld, n, c, ptr(mean), ptr(var), ptr(scratch), sb, stream()))
    return mean, var


def col_sums(x):
    """sum over the rows of every column of x [n, c] -> f32 [c] (apr_col_sums: fp64 partials in fixed order)."""
    x, ld = _rows(x, "col_sums.x")
    n, c = x.shape
    lib = _lib_()
    out = torch.empty(c, dtype=torch.float32, device=x.device)
    sb = int(lib.apr_bn_stats_scratch_bytes(n, c))
    scratch = torch.empty(sb, dtype=torch.uint8, device=x.device)
    check(lib.apr_col_sums(ptr(x), ld, n, c, ptr(out), ptr(scratch), sb, stream()))
    return out


def norm_params(x, eps):
    """(scale, shift) of a no-affine per-channel normalisation over all rows (one fused stats pass)."""
    x, ld = _rows(x, "norm_params.x")
    n, c = x.shape
    lib = _lib_()
    ss = torch.empty((2, c), dtype=torch.float32, device=x.device)
    sb = int(lib.apr_bn_stats_scratch_bytes(n, c))
    scratch = torch.empty(sb, dtype=torch.uint8, device=x.device)
    check(lib.apr_norm_params(ptr(x), ld, n, c, float(eps), ptr(ss[0]), ptr(ss[1]), ptr(scratch), sb, stream()))
    return ss[0], ss[1]


def norm_backward(x, dy, mean, rstd, gamma=None, want_affine_grads=True, out=None):
    """dx (and dgamma, dbeta) of y = (x - mean) * rstd * gamma + beta over all rows (apr_norm_backward).  `out`: a contiguous
    [n, c] tensor (e.g. a row slice) that receives dx."""
    x, ldx = _rows(x, "norm_backward.x")
    dy, lddy = _rows(dy, "norm_backward.dy")
    n, c = x.shape
    lib = _lib_()
    if out is not None and (tuple(out.shape) != (n, c) or not out.is_contiguous() or out.dtype != torch.float32):
        raise _lib.AprHipError("norm_backward: `out` must be a contiguous float32 [n, c] tensor")
    dx = out if out is not None else torch.empty((n, c), dtype=torch.float32, device=x.device)
    dg = torch.empty(c, dtype=torch.float32, device=x.device) if want_affine_grads else None
    db = torch.empty(c, dtype=torch.float32, device=x.device) if want_affine_grads else None
    sb = int(lib.apr_norm_backward_scratch_bytes(n, c))
    scratch = torch.empty(sb, dtype=torch.uint8, device=x.device)
    check(lib.apr_norm_backward(ptr(x), ldx, ptr(dy), lddy, n, c, ptr(mean), ptr(rstd), ptr(gamma), ptr(dx), c, ptr(dg),
                                ptr(db), ptr(scratch), sb, stream()))
    return dx, dg, db


class NormFunction(torch.autograd.Function):
    """y = (x - mean) / sqrt(var + eps) * weight + bias with the batch statistics of the rows of x, forward and backward on
    the HIP kernels (apr_bn_stats + apr_affine_act; apr_norm_backward).  weight / bias None: no affine (InstanceNorm1d of
    KPFCNN's blocks).  Returns (y, mean, var) -- mean / var detached, for the caller's running statistics."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        x = x.contiguous()
        mean, var = bn_stats(x)
        rstd = torch.rsqrt(var + eps)
        # (x - mean) first, then the scale: x * scale - mean * scale cancels two large terms when |mean| >> std
        scale = rstd if weight is None else rstd * weight.reshape(-1)
        y = affine_act(x - mean, scale=scale.contiguous(), shift=None if bias is None else bias.reshape(-1).contiguous())
        ctx.save_for_backward(x, mean, rstd, weight)
        ctx.has_bias = bias is not None
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, _dm, _dv):
        x, mean, rstd, weight = ctx.saved_tensors
        g = None if weight is None else weight.reshape(-1).contiguous()
        dx, dg, db = norm_backward(x, dy.contiguous(), mean, rstd, g, want_affine_grads=weight is not None or ctx.has_bias)
        dw = dg.reshape(weight.shape) if weight is not None else None
        dbias = db if ctx.has_bias else None
        return dx, dw, dbias, None


def affine_act(x, scale=None, shift=None, residual=None, relu=False, out=None, leaky=None):
    """y = act(x*scale + shift + residual); relu=True -> ReLU, leaky=slope -> LeakyReLU(slope)."""
    x, ldx = _rows(x, "affine_act.x")
    n, c = x.shape
    if out is None:
        out = torch.empty((n, c), dtype=torch.float32, device=x.device)
    out, ldy = _rows(out, "affine_act.out")
    ldr = 0
    if residual is not None:
        residual, ldr = _rows(residual, "affine_act.residual")
    mode = 2 if leaky is not None else int(bool(relu))
    check(_lib_().apr_affine_act(ptr(x), ldx, n, c, ptr(scale), ptr(shift), ptr(residual), ldr, mode,
                                 float(leaky or 0.0), ptr(out), ldy, stream()))
    return out


def l2_normalize(x, out=None):
    x, ldx = _rows(x, "l2_normalize.x")
    n, c = x.shape
    if out is None:
        out = torch.empty((n, c), dtype=torch.float32, device=x.device)
    out, ldy = _rows(out, "l2_normalize.out")
    check(_lib_().apr_l2_normalize(ptr(x), ldx, n, c, ptr(out), ldy, stream()))
    return out


class L2NormalizeFunction(torch.autograd.Function):
    """Row normalisation F / |F|_2 (FCGF_APR/model/resunet.py:187-190) with its backward on the HIP kernels."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return l2_normalize(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        n, c = x.shape
        dx = torch.empty_like(x)
        check(_lib_().apr_l2_normalize_backward(ptr(x), c, ptr(dy), c, n, c, ptr(dx), c, stream()))
        return dx


# ----------------------------------------------------------------------------
# matching / pose
# ----------------------------------------------------------------------------

def feature_nn(f0, f1, return_distance=False, impl=None):
    """Squared-L2 nearest neighbour of every row of f0 in f1 -> int64 [n0] (and d2 f32 [n0]).

    impl "fast" (default for C in 32/64/128; APR_NN_IMPL overrides): split-bf16 MFMA bound + exact fp32 refine of the
    few pairs that can still be the arg-min — the same bits out as "brute" for ANY input (a predicated brute-force
    kernel takes over if the candidate list overflows).  Measured per 14 k x 14 k x 32 KITTI pair: 145 us on
    discriminative features, ~255 us on the collapsed features of a random-init encoder (58 candidates / query).
    impl "brute": every distance in exact fp32, 290-350 us independent of the data (at the fp32 VALU roof).
    """
    f0 = _f32(f0, "feature_nn.f0").contiguous()
    f1 = _f32(f1, "feature_nn.f1").contiguous()
    if f0.shape[1] != f1.shape[1]:
        raise _lib.AprHipError("feature_nn: channel mismatch")
    n0, c = f0.shape
    n1 = f1.shape[0]
    lib = _lib_()
    best = torch.empty(n0, dtype=torch.int64, device=f0.device)
    if impl is None:
        impl = os.environ.get("APR_NN_IMPL", "fast")
    if impl not in ("brute", "fast"):
        raise _lib.AprHipError(f"feature_nn: unknown impl {impl!r}")
    if impl == "fast" and c in (32, 64, 128):
        sb = int(lib.apr_feature_nn_fast_scratch_bytes(n0, n1, c))     # bf16 MFMA filter + exact refine
        scratch = torch.empty(sb, dtype=torch.uint8, device=f0.device)
        check(lib.apr_feature_nn_fast(ptr(f0), n0, ptr(f1), n1, c, ptr(best), ptr(scratch), sb, stream()))
    else:
        check(lib.apr_feature_nn(ptr(f0), n0, ptr(f1), n1, c, ptr(best), stream()))
    idx = torch.empty(n0, dtype=torch.int64, device=f0.device)
    d2 = torch.empty(n0, dtype=torch.float32, device=f0.device) if return_distance else None
    check(lib.apr_nn_unpack(ptr(best), n0, ptr(idx), ptr(d2), stream()))
    return (idx, d2) if return_distance else idx


def ransac_pose(xyz0, xyz1, corr, max_dist, edge_ratio=0.9, max_iter=4000000, seed=0):
    """GPU RANSAC + Kabsch.  Returns (T float64 [4,4] numpy, info dict).  Synchronises."""
    xyz0 = _f32(xyz0, "ransac.xyz0").contiguous()
    xyz1 = _f32(xyz1, "ransac.xyz1").contiguous()
    if corr.dtype != torch.int64 or not corr.is_cuda:
        raise _lib.AprHipError("ransac_pose: corr must be an int64 GPU tensor")
    corr = corr.contiguous()
    n0, n1 = xyz0.shape[0], xyz1.shape[0]
    if corr.shape[0] != n0:
        raise _lib.AprHipError("ransac_pose: one correspondence per source point expected")
    lib = _lib_()
    sb = int(lib.apr_ransac_scratch_bytes(n0, int(max_iter)))
    scratch = torch.empty(sb, dtype=torch.uint8, device=xyz0.device)
    res = (C.c_double * 20)()
    check(lib.apr_ransac_pose(ptr(xyz0), n0, ptr(xyz1), n1, ptr(corr), float(max_dist), float(edge_ratio),
                              int(max_iter), int(seed) & 0xFFFFFFFFFFFFFFFF, ptr(scratch), sb, res, stream()))
    r = np.array(list(res), dtype=np.float64)
    T = r[:16].reshape(4, 4).copy()
    info = dict(inliers=int(r[16]), rmse=float(r[17]), best_iteration=int(r[18]), n_valid=int(r[19]),
                fitness=float(r[16]) / max(n0, 1))
    return T, info


def match_pose_batch(feats0, feats1, pts0, pts1, max_dist, edge_ratio=0.9, max_iter=4000000, seeds=None):
    """B pairs: feature NN + RANSAC/Kabsch each, ONE library call and ONE host synchronisation.

    feats0[i] [n0_i, C] / feats1[i] [n1_i, C] (row slices of the encoder output are fine), pts0[i] / pts1[i] f32 [n, 3].
    -> list of (T float64 [4,4] numpy, info dict), identical to feature_nn + ransac_pose per pair."""
    B = len(feats0)
    if not (B == len(feats1) == len(pts0) == len(pts1)) or B == 0:
        raise _lib.AprHipError("match_pose_batch: need the same (non-zero) number of entries in every list")
    if seeds is None:
        seeds = range(B)
    lib = _lib_()
    descs = (_lib.PairDesc * B)()
    keep = []
    c = feats0[0].shape[1]
    n0m = n1m = 0
    for i in range(B):
        f0 = _f32(feats0[i], "match_pose_batch.feats0").contiguous()
        f1 = _f32(feats1[i], "match_pose_batch.feats1").contiguous()
        p0 = _f32(pts0[i], "match_pose_batch.pts0").contiguous()
        p1 = _f32(pts1[i], "match_pose_batch.pts1").contiguous()
        if f0.shape[1] != c or f1.shape[1] != c or p0.shape != (f0.shape[0], 3) or p1.shape != (f1.shape[0], 3):
            raise _lib.AprHipError("match_pose_batch: inconsistent shapes in pair %d" % i)
        keep += [f0, f1, p0, p1]
        d = descs[i]
        d.f0, d.n0, d.f1, d.n1 = f0.data_ptr(), f0.shape[0], f1.data_ptr(), f1.shape[0]
        d.xyz0, d.xyz1, d.seed = p0.data_ptr(), p1.data_ptr(), int(seeds[i]) & 0xFFFFFFFFFFFFFFFF
        n0m, n1m = max(n0m, f0.shape[0]), max(n1m, f1.shape[0])
    sb = int(lib.apr_match_pose_batch_scratch_bytes(B, n0m, n1m, c, int(max_iter)))
    scratch = torch.empty(sb, dtype=torch.uint8, device=keep[0].device)
    res = (C.c_double * (20 * B))()
    check(lib.apr_match_pose_batch(descs, B, c, float(max_dist), float(edge_ratio), int(max_iter), ptr(scratch), sb, res,
                                   stream()))
    r = np.array(list(res), dtype=np.float64).reshape(B, 20)
    out = []
    for i in range(B):
        n0 = descs[i].n0
        out.append((r[i, :16].reshape(4, 4).copy(),
                    dict(inliers=int(r[i, 16]), rmse=float(r[i, 17]), best_iteration=int(r[i, 18]),
                         n_valid=int(r[i, 19]), fitness=float(r[i, 16]) / max(n0, 1))))
    return out


def coords_bbox(coords):
    """int32 [n, 4] (batch, x, y, z) -> device int32 [8]: min x, y, z, max x, y, z, max batch index, 0 (no sync)."""
    if coords.dtype != torch.int32 or coords.dim() != 2 or coords.shape[1] != 4 or not coords.is_contiguous():
        raise _lib.AprHipError("coords_bbox: coords must be contiguous int32 [n, 4]")
    bbox = torch.empty(8, dtype=torch.int32, device=coords.device)
    check(_lib_().apr_coords_bbox(ptr(coords), coords.shape[0], ptr(bbox), stream()))
    return bbox


def occ_conv_supported(bbox, kernel_size, cout, n=None):
    """True if `occ_conv` takes this case (odd kernel 3 / 5 / 7, cout % 8 == 0, a box whose bitmap stays under 2 GB and --
    with the voxel count `n` given -- under 1 KB per voxel + 4 MB: one outlier voxel can stretch the box until the bitmap
    costs more than the kernel map it replaces)."""
    if bbox is None or kernel_size not in (3, 5, 7) or cout % 8 != 0:
        return False
    box = (C.c_int32 * 8)(*[int(v) for v in bbox])
    if n is not None:
        return bool(_lib_().apr_occ_conv_pays(box, int(kernel_size), int(n)))
    return int(_lib_().apr_occ_conv_scratch_bytes(box, int(kernel_size))) > 0


def occ_conv(coords, n, bbox, kernel_size, w, scale=None, shift=None, relu=False, residual=None, out=None, keep=None):
    """Stride-1 ks^3 convolution of the constant-1 feature over the voxels coords[:n] (apr_occ_conv): w f32 [ks^3, cout];
    bbox: the 8 host ints of `coords_bbox` for these rows (or a superset; voxel units).  Same bits as `spconv` over the
    kernel map on all-ones features; a row outside the box comes out as NaN.  keep: a list that receives
    (scratch, bbox tuple, kernel_size) -- the occupancy bitmap, for `kernel_map_occ`."""
    lib = _lib_()
    w = _f32(w, "occ_conv.w").contiguous()
    K, cout = w.shape
    if K != kernel_size ** 3:
        raise _lib.AprHipError("occ_conv: w must be [kernel_size^3, cout]")
    box = (C.c_int32 * 8)(*[int(v) for v in bbox])
    sb = int(lib.apr_occ_conv_scratch_bytes(box, int(kernel_size)))
    if sb == 0:
        raise _lib.AprHipError("occ_conv: empty or oversized bounding box (use occ_conv_supported)")
    if out is None:
        out = torch.empty((n, cout), dtype=torch.float32, device=coords.device)
    out, ldo = _rows(out, "occ_conv.out")
    ldr = 0
    if residual is not None:
        residual, ldr = _rows(residual, "occ_conv.residual")
    scratch = torch.empty(sb, dtype=torch.uint8, device=coords.device)
    check(lib.apr_occ_conv(ptr(coords), int(n), box, int(kernel_size), ptr(w), cout, ptr(scale), ptr(shift), ptr(residual),
                           ldr, int(bool(relu)), ptr(out), ldo, ptr(scratch), sb, stream()))
    if keep is not None:
        # with the rows it covers: a bitmap built from a SUBSET of a map must not answer probes for the whole map (a clear bit
        # inside the box reads as "no voxel"): kernel_map_occ checks (row count, coordinate storage) against the map it is given
        keep.append((scratch, tuple(int(v) for v in bbox), int(kernel_size), int(n), coords.data_ptr()))
    return out


def kernel_map_occ(out_map: CoordMap, in_map: CoordMap, kernel_size: int, scale: int, occ) -> torch.Tensor:
    """`kernel_map` with the occupancy bitmap `occ` = (scratch, bbox, bitmap kernel size, rows, coordinate storage) that
    `occ_conv(keep=...)` left for `in_map` as a pre-filter of the probes (apr_kernel_map_occ): same table.  A bitmap that was
    not built from EVERY row of `in_map` is not used: the plain hash-table map is returned instead."""
    scratch, bbox, bks = occ[:3]
    if len(occ) < 5 or in_map.n is None or occ[3] != in_map.n or occ[4] != in_map.coords.data_ptr():
        return kernel_map(out_map, in_map, kernel_size, scale)
    nbr = torch.empty((out_map.n, kernel_size ** 3), dtype=torch.int32, device=out_map.coords.device)
    box = (C.c_int32 * 8)(*bbox)
    check(_lib_().apr_kernel_map_occ(ptr(out_map.coords), out_map.n, None, ptr(in_map.keys), ptr(in_map.vals), in_map.cap,
                                     int(kernel_size), int(scale), box, int(bks), ptr(scratch), ptr(nbr), stream()))
    return nbr


def set_ransac_screen(mode):
    """Which RANSAC sampling kernel the calls that follow use (apr_ransac_set_screen): True = the LDS-screened one (lowest
    latency with ONE step in flight), False = the plain one (friendlier to the kernels of other streams: a pipelined caller),
    None = the environment's APR_RANSAC_SCREEN (default on).  Same candidates either way."""
    check(_lib_().apr_ransac_set_screen(-1 if mode is None else int(bool(mode))))


_RANSAC_OPTIONS = {"screen": 0, "count": 1, "prune": 2, "force_rounds": 3}


def set_ransac_option(name, value):
    """A/B / test switch of the matcher (apr_ransac_set_option): name in screen / count / prune / force_rounds, value 0, 1 or
    None (= the environment's default, which the library reads once per process).  Results never depend on them."""
    check(_lib_().apr_ransac_set_option(_RANSAC_OPTIONS[name], -1 if value is None else int(bool(value))))


class ransac_options:
    """with ops.ransac_options(count=0, prune=0): ... -- set, then restore the defaults."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        for k, v in self.kw.items():
            set_ransac_option(k, v)

    def __exit__(self, *exc):
        for k in self.kw:
            set_ransac_option(k, None)


def ransac_sampling_launches():
    """(k_sample_check launches, k_sample_screen launches) since the library was loaded."""
    out = (C.c_int64 * 2)()
    check(_lib_().apr_ransac_sampling_launches(out))
    return int(out[0]), int(out[1])


def set_match_lanes(lanes):
    """Streams the pairs of a `match_pose_batch` are dealt over inside libapr_hip (apr_match_pose_set_lanes; 1 .. 4)."""
    check(_lib_().apr_match_pose_set_lanes(int(lanes)))


def match_pose_batch_async(feats0, feats1, pts0, pts1, max_dist, edge_ratio=0.9, max_iter=4000000, seeds=None):
    """`match_pose_batch` in two halves (apr_match_pose_batch_enqueue / _finish): everything is enqueued on the current
    stream, the B result slots travel to pinned host memory asynchronously -> PendingFetch whose finish() returns the
    list of (T, info)."""
    B = len(feats0)
    if not (B == len(feats1) == len(pts0) == len(pts1)) or B == 0:
        raise _lib.AprHipError("match_pose_batch: need the same (non-zero) number of entries in every list")
    if seeds is None:
        seeds = range(B)
    lib = _lib_()
    descs = (_lib.PairDesc * B)()
    keep = []
    c = feats0[0].shape[1]
    n0m = n1m = 0
    for i in range(B):
        f0 = _f32(feats0[i], "match_pose_batch.feats0").contiguous()
        f1 = _f32(feats1[i], "match_pose_batch.feats1").contiguous()
        p0 = _f32(pts0[i], "match_pose_batch.pts0").contiguous()
        p1 = _f32(pts1[i], "match_pose_batch.pts1").contiguous()
        if f0.shape[1] != c or f1.shape[1] != c or p0.shape != (f0.shape[0], 3) or p1.shape != (f1.shape[0], 3):
            raise _lib.AprHipError("match_pose_batch: inconsistent shapes in pair %d" % i)
        keep += [f0, f1, p0, p1]
        d = descs[i]
        d.f0, d.n0, d.f1, d.n1 = f0.data_ptr(), f0.shape[0], f1.data_ptr(), f1.shape[0]
        d.xyz0, d.xyz1, d.seed = p0.data_ptr(), p1.data_ptr(), int(seeds[i]) & 0xFFFFFFFFFFFFFFFF
        n0m, n1m = max(n0m, f0.shape[0]), max(n1m, f1.shape[0])
    sb = int(lib.apr_match_pose_batch_scratch_bytes(B, n0m, n1m, c, int(max_iter)))
    scratch = torch.empty(sb, dtype=torch.uint8, device=keep[0].device)
    slots = torch.empty(int(lib.apr_match_pose_batch_slot_bytes(B)), dtype=torch.uint8, pin_memory=True)
    st = stream()
    check(lib.apr_match_pose_batch_enqueue(descs, B, c, float(max_dist), float(edge_ratio), int(max_iter), ptr(scratch),
                                           sb, C.c_void_p(slots.data_ptr()), st))
    pf = PendingFetch.__new__(PendingFetch)
    pf._host = slots
    pf.event = fetch_event()
    pf.event.record()
    pf._keep = (keep, scratch, descs)

    def then(_):
        res = (C.c_double * (20 * B))()
        check(lib.apr_match_pose_batch_finish(descs, B, c, float(max_dist), float(edge_ratio), int(max_iter),
                                              ptr(scratch), sb, C.c_void_p(slots.data_ptr()), res, st))
        r = np.array(list(res), dtype=np.float64).reshape(B, 20)
        return [(r[i, :16].reshape(4, 4).copy(),
                 dict(inliers=int(r[i, 16]), rmse=float(r[i, 17]), best_iteration=int(r[i, 18]), n_valid=int(r[i, 19]),
                      fitness=float(r[i, 16]) / max(descs[i].n0, 1))) for i in range(B)]

    pf._then = then
    return pf


def ransac_pose_geometric(xyz0, xyz1, corr, max_dist, edge_ratio=0.9, max_iter=50000, max_validation=1000, seed=0):
    """open3d <= 0.11 flavour (Predator_APR): first `max_validation` survivors, geometric inlier count."""
    xyz0 = _f32(xyz0, "ransac.xyz0").contiguous()
    xyz1 = _f32(xyz1, "ransac.xyz1").contiguous()
    if corr.dtype != torch.int64 or not corr.is_cuda or corr.shape[0] != xyz0.shape[0]:
        raise _lib.AprHipError("ransac_pose_geometric: corr must be an int64 GPU tensor, one entry per source point")
    corr = corr.contiguous()
    n0, n1 = xyz0.shape[0], xyz1.shape[0]
    lib = _lib_()
    sb = int(lib.apr_ransac_geometric_scratch_bytes(n0, n1, int(max_iter)))
    scratch = torch.empty(sb, dtype=torch.uint8, device=xyz0.device)
    res = (C.c_double * 20)()
    check(lib.apr_ransac_pose_geometric(ptr(xyz0), n0, ptr(xyz1), n1, ptr(corr), float(max_dist), float(edge_ratio),
                                        int(max_iter), int(max_validation), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                        ptr(scratch), sb, res, stream()))
    r = np.array(list(res), dtype=np.float64)
    info = dict(inliers=int(r[16]), rmse=float(r[17]), best_iteration=int(r[18]), n_valid=int(r[19]),
                fitness=float(r[16]) / max(n0, 1))
    return r[:16].reshape(4, 4).copy(), info


def ransac_pose_geometric_async(xyz0, xyz1, corr, max_dist, edge_ratio=0.9, max_iter=50000, max_validation=1000, seed=0):
    """`ransac_pose_geometric` enqueued without a host synchronisation -> uint8 device tensor holding the raw result;
    fetch the raw results of a batch with one copy and decode each with `ransac_decode`."""
    xyz0 = _f32(xyz0, "ransac.xyz0").contiguous()
    xyz1 = _f32(xyz1, "ransac.xyz1").contiguous()
    if corr.dtype != torch.int64 or not corr.is_cuda or corr.shape[0] != xyz0.shape[0]:
        raise _lib.AprHipError("ransac_pose_geometric: corr must be an int64 GPU tensor, one entry per source point")
    corr = corr.contiguous()
    n0, n1 = xyz0.shape[0], xyz1.shape[0]
    lib = _lib_()
    sb = int(lib.apr_ransac_geometric_scratch_bytes(n0, n1, int(max_iter)))
    scratch = torch.empty(sb, dtype=torch.uint8, device=xyz0.device)
    raw = torch.empty(int(lib.apr_ransac_raw_bytes()), dtype=torch.uint8, device=xyz0.device)
    check(lib.apr_ransac_pose_geometric_async(ptr(xyz0), n0, ptr(xyz1), n1, ptr(corr), float(max_dist),
                                              float(edge_ratio), int(max_iter), int(max_validation),
                                              int(seed) & 0xFFFFFFFFFFFFFFFF, ptr(scratch), sb, ptr(raw), stream()))
    return raw


def ransac_decode(raw_host, n0):
    """numpy uint8 raw result -> (T [4,4] float64, info) as `ransac_pose_geometric` returns them."""
    res = (C.c_double * 20)()
    raw_host = np.ascontiguousarray(raw_host)
    check(_lib_().apr_ransac_decode(raw_host.ctypes.data, res))
    r = np.array(list(res), dtype=np.float64)
    info = dict(inliers=int(r[16]), rmse=float(r[17]), best_iteration=int(r[18]), n_valid=int(r[19]),
                fitness=float(r[16]) / max(n0, 1))
    return r[:16].reshape(4, 4).copy(), info


def irls_pose(pts0, pts1, weight=None):
    """est_quad_linear_robust on the GPU -> float32 [4,4] CPU tensor.  Synchronises."""
    pts0 = _f32(pts0, "irls.pts0").contiguous()
    pts1 = _f32(pts1, "irls.pts1").contiguous()
    n = pts0.shape[0]
    if weight is not None:
        weight = _f32(weight, "irls.weight").contiguous().view(-1)
    lib = _lib_()
    sb = int(lib.apr_irls_scratch_bytes(n))
    scratch = torch.empty(sb, dtype=torch.uint8, device=pts0.device)
    T = (C.c_float * 16)()
    check(lib.apr_irls_pose(ptr(pts0), ptr(pts1), ptr(weight), n, T, ptr(scratch), sb, stream()))
    return torch.tensor(list(T), dtype=torch.float32).view(4, 4)
