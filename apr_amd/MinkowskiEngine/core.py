"""Sparse tensor, coordinate manager and layers of the MinkowskiEngine-compatible subset.

Semantics = SURVEY.md Appendix A / oracle/me_oracle.py; every numeric op is a
libapr_hip.so call (apr_amd.ops).  The modules keep MinkowskiEngine's parameter
names (`kernel`, `bias`, `bn.*`) so reference checkpoints' state_dict keys load.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

import os

from .. import ops
from .._lib import AprHipError

# kernel maps over the stride-1 map probe conv1's occupancy bitmap before the hash table (ops.kernel_map_occ); 0: table only
OCC_KERNEL_MAP = os.environ.get("APR_OCC_KERNEL_MAP", "1") != "0"


class CoordinateMapKey:
    """Identifies a coordinate map of a manager by its tensor stride."""

    __slots__ = ("stride",)

    def __init__(self, stride: int):
        self.stride = int(stride)

    def get_tensor_stride(self):
        return [self.stride] * 3

    def __eq__(self, other):
        return isinstance(other, CoordinateMapKey) and other.stride == self.stride

    def __hash__(self):
        return hash(self.stride)

    def __repr__(self):
        return f"CoordinateMapKey(tensor_stride={self.stride})"


class CoordinateManager:
    """Voxel-hash coordinate maps + cached kernel maps of one batched point set.

    `maps[ts]` is an `ops.CoordMap`; `kernel_map(ts_in, ts_out, k, transpose)` is
    the dense neighbour table the sparse-conv kernel consumes.
    """

    def __init__(self, coordinates: torch.Tensor = None, base_map=None):
        """`coordinates`: unique int32 [n,4] rows.  `base_map`: adopt an `ops.CoordMap` that already de-duplicated
        raw voxel coordinates (its rows ARE the stride-1 map: no second hash build; may still be unfinalised)."""
        if (coordinates is None) == (base_map is None):
            raise AprHipError("CoordinateManager: give either coordinates or base_map")
        self._adopted = base_map is not None
        if self._adopted and base_map.n_in > 4 * 65536:
            # the de-duplicating table was sized for the RAW points (2 x 1.4 M slots, ~50 MB for a 12-frame step) and
            # holds ~10x fewer voxels: every kernel-map probe into it would miss L2.  Re-insert the unique rows into
            # a table of their own size, chained on the device-side row count (no sync).
            # Its size is a GUESS (a scan puts >= ~8 points into a voxel): rows / 4; `_finalize` compares it with the
            # true count and falls back to the big table if the guess was too small.
            self._compact_rows = max(65536, base_map.n_in // 4)
            compact = ops.build_map(base_map.coords[:self._compact_rows], n_in_dev=base_map.n_dev)
            self.maps = {1: compact}
            self._dedup = base_map
        else:
            self.maps = {1: base_map if self._adopted else ops.build_map(coordinates)}
            self._dedup = None
        self._kmaps = {}
        self._plists = {}
        self._plist_counters = None
        self._tpool = None
        self._bbox = None
        self._occ = []           # [(scratch, bbox, kernel size)]: the occupancy bitmap conv1 left over the stride-1 map
        self.device = self.maps[1].keys.device

    @classmethod
    def from_maps(cls, maps, counters=None):
        """A manager over FINALISED maps {tensor stride: ops.CoordMap} that something else built (ops.VoxelPyramid: the
        whole pyramid by one library call); `counters`: zeroed pair-list counter blocks [16, ops.pair_counter_ints()]."""
        self = cls.__new__(cls)
        self._adopted, self._dedup = True, None
        self.maps = dict(maps)
        self._kmaps, self._plists = {}, {}
        self._plist_counters = counters
        self._tpool, self._bbox, self._occ = None, None, []
        self.device = self.maps[1].keys.device
        return self

    # -- coordinate maps -----------------------------------------------------
    def _finalize(self, extras=()):
        pend = [m for m in self.maps.values() if m.n is None]
        if self._dedup is not None and self._dedup.n is None:
            pend.append(self._dedup)
        fetched = []
        if pend or extras:
            fetched = ops.finalize_maps(pend, extras)
            self._after_finalize()
        return fetched

    def _after_finalize(self):
        if self._dedup is not None and self._dedup.n > self._compact_rows:
            # more voxels than the compact table was sized for: adopt the big table after all and rebuild
            # whatever was chained behind the truncated one (second sync; never on LiDAR-density input)
            strides = [ts for ts in self.maps if ts != 1]
            self.maps = {1: self._dedup}
            self._dedup = None
            self._kmaps, self._plists, self._tpool = {}, {}, None
            self.build_pyramid(strides)
        m1 = self.maps[1]
        if not self._adopted and m1.n != m1.n_in:
            raise AprHipError(
                f"SparseTensor: {m1.n_in - m1.n} duplicate coordinates; quantize first "
                "(ME.utils.sparse_quantize)")

    def build_pyramid_async(self, strides, extras=()):
        """`build_pyramid` without the host synchronisation -> ops.PendingFetch; finish() returns the extras and leaves
        the manager finalised (the rare oversize fallback inside synchronises)."""
        for ts in sorted(strides):
            if ts not in self.maps:
                src = self.maps[ts // 2]
                self.maps[ts] = ops.build_map(src.coords, floor_to=ts, n_in_dev=src.n_dev if src.n is None else None)
        pend = [m for m in self.maps.values() if m.n is None]
        if self._dedup is not None and self._dedup.n is None:
            pend.append(self._dedup)
        if self._plist_counters is None:      # cleared by the launch that gathers the sizes (no fill of their own)
            self._plist_counters = torch.empty((16, ops.pair_counter_ints()), dtype=torch.int32, device=self.device)
            pf = ops.finalize_maps_async(pend, extras, zero=self._plist_counters)
        else:
            pf = ops.finalize_maps_async(pend, extras)
        inner = pf._then

        def then(host):
            fetched = inner(host)
            self._after_finalize()
            return fetched

        pf._then = then
        return pf

    def build_pyramid(self, strides, extras=()):
        """Enqueue all missing strided maps back to back and sync ONCE (`extras`: small device int tensors fetched
        in the same sync, returned as numpy arrays)."""
        for ts in sorted(strides):
            if ts not in self.maps:
                src = self.maps[ts // 2]
                self.maps[ts] = ops.build_map(src.coords, floor_to=ts, n_in_dev=src.n_dev if src.n is None else None)
        return self._finalize(extras)

    def get_map(self, ts):
        if ts not in self.maps:
            self.build_pyramid([ts] if ts // 2 in self.maps else
                               [s for s in (2 ** i for i in range(1, 12)) if s <= ts])
        self._finalize()
        return self.maps[ts]

    def set_bbox(self, bbox):
        """The 8 host ints of `ops.coords_bbox` -- min x, y, z, max x, y, z, max batch index, 0, in VOXEL units (integer
        coordinates as stored in the stride-1 map, not metres) -- over the stride-1 coordinates or over the voxelised input
        rows they were made unique from (a superset is fine): handed over by a caller that fetched them with the map
        sizes, so `get_bbox` costs no sync.  A box that misses a voxel turns that voxel's conv1 output row into NaN
        (ops.occ_conv); APR_CHECK_BBOX=1 verifies the box against the map here (synchronises)."""
        self._bbox = tuple(int(v) for v in bbox)
        import os
        if os.environ.get("APR_CHECK_BBOX") == "1":
            m = self.get_map(1)
            true = ops.coords_bbox(m.coords[:m.n].contiguous()).tolist()
            b = self._bbox
            if any(b[a] > true[a] for a in range(3)) or any(b[a] < true[a] for a in range(3, 7)):
                raise AprHipError(f"set_bbox: box {b[:7]} does not contain the stride-1 coordinates {tuple(true[:7])}")

    def get_bbox(self):
        if self._bbox is None:
            m = self.get_map(1)
            self._bbox = tuple(ops.coords_bbox(m.coords[:m.n].contiguous()).tolist())      # synchronises
        return self._bbox

    def get_coordinates(self, key):
        return self.get_map(key.stride if isinstance(key, CoordinateMapKey) else key).coords

    def size(self, ts):
        return self.get_map(ts).n

    # -- kernel maps ---------------------------------------------------------
    def kernel_map(self, ts_in, ts_out, kernel_size, transpose=False):
        key = (ts_in, ts_out, kernel_size, transpose)
        nbr = self._kmaps.get(key)
        if nbr is None:
            in_map, out_map = self.get_map(ts_in), self.get_map(ts_out)
            if transpose and ts_in == 2 * ts_out:
                # coarse -> fine: the transpose of the strided fine -> coarse table (same (fine, coarse, offset)
                # triples), which the encoder builds anyway: a scatter instead of n_fine * K hash probes
                fwd = self.kernel_map(ts_out, ts_in, kernel_size, False)
                nbr = ops.kernel_map_transpose(fwd, out_map.n, prefilled=self._transposed_table(ts_out, kernel_size))
            else:
                # regular: c_in = c_out + o*ts_in ; transposed (coarse->fine): c_coarse = c_fine - o*ts_fine
                scale = -ts_out if transpose else ts_in
                if self._occ and ts_in == 1 and not transpose and kernel_size == 3 and OCC_KERNEL_MAP:
                    # the bitmap conv1 built over this map answers most probes (empty cells) without the hash table
                    nbr = ops.kernel_map_occ(out_map, in_map, kernel_size, scale, self._occ[0])
                else:
                    nbr = ops.kernel_map(out_map, in_map, kernel_size, scale)
            self._kmaps[key] = nbr
        return nbr

    def _transposed_table(self, ts_fine, kernel_size):
        """A -1-filled int32 [n(ts_fine), kernel_size^3] table for the transposed map onto level ts_fine.  The tables of
        ALL levels that have a coarser level above them come out of one allocation cleared by ONE fill (an encoder asks for
        three of them, one per decoder level); None when the pool does not cover the request."""
        if kernel_size != 3:
            return None
        if self._tpool is None:
            self._finalize()
            levels = [ts for ts in sorted(self.maps) if 2 * ts in self.maps]
            sizes = [self.maps[ts].n * 27 for ts in levels]
            buf = torch.empty((sum(sizes) + 63) // 64 * 64 + 64, dtype=torch.int32, device=self.device)   # whole 256-B lines: one fill kernel
            ops.fill_bytes(buf, 0xFF)
            self._tpool, pos = {}, 0
            for ts, sz in zip(levels, sizes):
                self._tpool[ts] = buf[pos:pos + sz].view(-1, 27)
                pos += sz
        return self._tpool.pop(ts_fine, None)      # handed out once: the table is written into

    def pair_list(self, ts_in, ts_out, kernel_size, transpose=False, triples=False):
        """Per-offset pair lists of kernel_map(...) for the weight-stationary conv path (cached per map); `triples`: the
        x-triple entry lists of a 27-offset map instead (ops.PairList3: half the product rows)."""
        key = ("ws3" if triples else "ws", ts_in, ts_out, kernel_size, transpose)
        pl = self._plists.get(key)
        if pl is None:
            nbr = self.kernel_map(ts_in, ts_out, kernel_size, transpose)
            if self._plist_counters is None:      # one fill clears the counters of every map of this manager
                self._plist_counters = torch.zeros((16, ops.pair_counter_ints()), dtype=torch.int32, device=nbr.device)
            slot = sum(1 for k in self._plists if k[0] in ("ws", "ws3"))
            build = ops.build_pairlist3 if triples else ops.build_pairlist
            pl = build(nbr, lazy=True, counters=self._plist_counters[slot] if slot < 16 else None)
            self._plists[key] = pl
        return pl


    def os_pair_list(self, ts_in, ts_out, kernel_size, transpose, cin, cout):
        """Per-tile pair lists of kernel_map(...) for the output-stationary conv path (ops.OsPairs, cached per map and
        tile height), or None when the layer shape is not covered (apr_spconv_os_tile_rows)."""
        nbr = self.kernel_map(ts_in, ts_out, kernel_size, transpose)
        n_in = self.size(ts_in)
        rows = ops.os_tile_rows(nbr.shape[0], cin, cout) if n_in < (1 << 23) and nbr.shape[1] <= 27 else 0
        if rows <= 0:
            return None
        key = ("os", ts_in, ts_out, kernel_size, transpose, rows)
        pl = self._plists.get(key)
        if pl is None:
            pl = self._plists[key] = ops.build_os_pairs(nbr, n_in, rows, lazy=True)
        return pl


class SparseTensor:
    def __init__(self, features, coordinates=None, coordinate_map_key=None, coordinate_manager=None,
                 device=None, **unused):
        if device is None:
            device = features.device if features.is_cuda else None
        if device is None and coordinates is not None and coordinates.is_cuda:
            device = coordinates.device
        if device is None:
            raise AprHipError("SparseTensor: features/coordinates must live on the GPU (no CPU fallback)")
        device = torch.device(device)
        if device.type != "cuda":
            raise AprHipError("SparseTensor: device must be a GPU")
        # the caller's promise that every feature is exactly 1.0 (FCGF's input, complement_data_loader.py:805-812):
        # a first convolution with in_channels = 1 then runs on occupancy alone (ops.occ_conv); never inherited
        self.unit_features = bool(unused.get("unit_features", False))
        self.F = features.to(device=device, dtype=torch.float32)
        if self.F.dim() != 2:
            raise AprHipError("SparseTensor: features must be [N, C]")
        if coordinates is not None:
            if coordinates.dim() != 2 or coordinates.shape[1] != 4:
                raise AprHipError("SparseTensor: coordinates must be [N, 1+3] (batch, x, y, z)")
            if coordinates.shape[0] != self.F.shape[0]:
                raise AprHipError("SparseTensor: coordinates / features row mismatch")
            c = coordinates.to(device=device)
            if c.dtype.is_floating_point:
                c = torch.floor(c)
            c = c.to(torch.int32).contiguous()
            self._input_coords = c                 # rows keep the input order: .C of a fresh tensor without finalising its map
            self.coordinate_manager = CoordinateManager(c)
            self.coordinate_map_key = CoordinateMapKey(1)
        else:
            if coordinate_manager is None or coordinate_map_key is None:
                raise AprHipError("SparseTensor: need coordinates or (coordinate_map_key, coordinate_manager)")
            self.coordinate_manager = coordinate_manager
            self.coordinate_map_key = coordinate_map_key

    # -- attributes the reference reads -----------------------------------------
    @property
    def C(self):
        return self.coordinate_manager.get_coordinates(self.coordinate_map_key)

    @property
    def coordinates(self):
        return self.C

    @property
    def features(self):
        return self.F

    @property
    def tensor_stride(self):
        return self.coordinate_map_key.get_tensor_stride()

    @property
    def device(self):
        return self.F.device

    @property
    def D(self):
        return 3

    @property
    def shape(self):
        return self.F.shape

    def __len__(self):
        return self.F.shape[0]

    def size(self):
        return self.F.size()

    @property
    def decomposed_coordinates_and_features(self):
        C = self.C
        b = C[:, 0]
        nb = int(b.max().item()) + 1 if len(C) else 0
        coords, feats = [], []
        for i in range(nb):
            m = b == i
            coords.append(C[m][:, 1:])
            feats.append(self.F[m])
        return coords, feats

    @property
    def decomposed_features(self):
        return self.decomposed_coordinates_and_features[1]

    @property
    def decomposed_coordinates(self):
        return self.decomposed_coordinates_and_features[0]

    def _like(self, F, key=None):
        return SparseTensor(F, coordinate_map_key=self.coordinate_map_key if key is None else key,
                            coordinate_manager=self.coordinate_manager)

    def _check_same_map(self, other):
        if (other.coordinate_manager is not self.coordinate_manager
                or other.coordinate_map_key != self.coordinate_map_key):
            raise AprHipError("sparse tensors live on different coordinate maps")

    def __iadd__(self, other):
        self._check_same_map(other)
        if _tracking(self.F, other.F):
            self.F = self.F + other.F          # autograd: out of place
        else:
            ops.affine_act(self.F, residual=other.F, out=self.F)
        return self

    def __add__(self, other):
        self._check_same_map(other)
        if _tracking(self.F, other.F):
            return self._like(self.F + other.F)
        return self._like(ops.affine_act(self.F, residual=other.F))

    def __repr__(self):
        return f"SparseTensor(N={len(self)}, C={self.F.shape[1]}, {self.coordinate_map_key})"


def _tracking(*tensors):
    """Autograd is recording and at least one of the tensors / parameters takes part in it."""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def _no_grad_guard(module):
    if torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters(recurse=False)):
        raise AprHipError("this apr_amd op is forward-only: call it under torch.no_grad()")


class MinkowskiNetwork(nn.Module):
    def __init__(self, D):
        super().__init__()
        self.D = D


class _ConvBase(nn.Module):
    TRANSPOSE = False

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator=None, expand_coordinates=False, convolution_mode=None, dimension=None):
        super().__init__()
        if dimension not in (None, 3):
            raise NotImplementedError("only 3-D sparse convolution (dimension=3)")
        if dilation != 1:
            raise NotImplementedError("dilation != 1 is not used by APR")
        if kernel_size is None or kernel_size < 1 or kernel_size % 2 == 0:
            raise ValueError(f"kernel_size must be odd and positive, got {kernel_size}")
        if stride not in (1, 2):
            raise NotImplementedError("stride must be 1 or 2")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.dilation = kernel_size, stride, dilation
        kv = kernel_size ** 3
        self.kernel_volume = kv
        self.use_mm = kv == 1 and stride == 1
        shape = (in_channels, out_channels) if self.use_mm else (kv, in_channels, out_channels)
        self.kernel = nn.Parameter(torch.empty(shape))
        self.bias = nn.Parameter(torch.empty(1, out_channels)) if bias else None
        self._packed = None
        self._packed_key = None
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1.0 / math.sqrt(self.in_channels * self.kernel_volume)
        with torch.no_grad():
            self.kernel.uniform_(-stdv, stdv)
            if self.bias is not None:
                self.bias.uniform_(-stdv, stdv)

    def packed_weight(self):
        k = (self.kernel.data_ptr(), self.kernel._version, self.kernel.device)
        if self._packed_key != k:
            self._packed = ops.pack_weights(self.kernel)
            self._packed_key = k
        return self._packed

    def packed_weight_bf3(self):
        """3-way bf16 split of the kernel for the weight-stationary path (None if the shape is not covered or the
        split path is switched off with APR_WS_BF3=0)."""
        import os
        if os.environ.get("APR_WS_BF3", "1") == "0":
            return None
        k = (self.kernel.data_ptr(), self.kernel._version, self.kernel.device)
        if getattr(self, "_bf3_key", None) != k:      # a kernel-size-1 layer stores [Cin, Cout]: the dense K = 1 image
            self._bf3 = ops.pack_weights_bf3(self.kernel if self.kernel.dim() == 3 else self.kernel.unsqueeze(0))
            self._bf3_key = k
        return self._bf3

    def packed_weight_T(self, flip, tile=True, bf3=True):
        """(tile pack or None, bf16-split pack or None) of the input gradient's kernel [K, cout, cin] (offsets mirrored when
        `flip`: ops.weights_flip_transpose), cached per parameter version like the forward packs -- one pack per optimizer
        step, not one per call -- and built only for the route that asks (`tile` / `bf3`)."""
        k = (self.kernel.data_ptr(), self.kernel._version, self.kernel.device, bool(flip))
        if getattr(self, "_packT_key", None) != k:
            self._packT = {}
            self._packT_key = k
        c = self._packT
        if (tile and "tile" not in c) or (bf3 and "bf3" not in c):
            w = self.kernel.detach()
            w = w if w.dim() == 3 else w.unsqueeze(0)
            if tile and "tile" not in c:
                if "wt" not in c:
                    c["wt"] = ops.weights_flip_transpose(w, flip)
                c["tile"] = ops.pack_weights(c["wt"])
            if bf3 and "bf3" not in c:      # split image straight from the parameter: no [K, cout, cin] copy in between
                import os
                c["bf3"] = None if os.environ.get("APR_WS_BF3", "1") == "0" else ops.pack_weights_bf3(w, flip=flip, transposed=True)
        return c.get("tile"), c.get("bf3")

    def _maps(self, x: SparseTensor):
        """-> (nbr or None, out tensor stride)."""
        ts = x.coordinate_map_key.stride
        cm = x.coordinate_manager
        if self.use_mm:
            return None, ts
        if not self.TRANSPOSE:
            ts_out = ts * self.stride
            return cm.kernel_map(ts, ts_out, self.kernel_size, False), ts_out
        if ts % self.stride != 0 or ts // self.stride < 1:
            raise AprHipError("transposed convolution below tensor stride 1")
        ts_out = ts // self.stride
        if ts_out not in cm.maps:
            raise AprHipError("transposed convolution needs the encoder's coordinate map of that stride")
        return cm.kernel_map(ts, ts_out, self.kernel_size, self.stride != 1), ts_out

    def _bf3_only(self, plist, cin, w_bf3):
        """The launch reads the bf16-split image alone (weight-stationary / triple-list / output-stationary kernels at the
        channel counts they cover): the tile pack need not exist."""
        return plist is not None and w_bf3 is not None and cin in (64, 128, 192, 256, 384)

    def run_T(self, dout, nbr_bwd, n_in, flip, plist=None):
        """The input gradient: the same routed launch over the reverse map with the flipped-transposed kernel (channels
        swapped); `plist`: the reverse map's pair lists."""
        want_bf3 = plist is not None or nbr_bwd is None
        w_bf3 = self.packed_weight_T(flip, tile=False, bf3=True)[1] if want_bf3 else None
        wp = None if self._bf3_only(plist, self.out_channels, w_bf3) else self.packed_weight_T(flip, tile=True, bf3=False)[0]
        os_pairs = plist if isinstance(plist, ops.OsPairs) else None
        return ops.spconv(dout, nbr_bwd, self.kernel_volume if nbr_bwd is not None else 1, self.out_channels,
                          self.in_channels, wp, n_out=n_in, plist=None if os_pairs is not None else plist,
                          w_bf3=w_bf3, os_pairs=os_pairs)

    def run(self, feats, nbr, n_out, scale=None, shift=None, residual=None, relu=False, out=None, batch=None,
            plist=None, l2norm=False, raw=False):
        """Raw fused launch on feature rows (used by the fused encoder plan); `batch` defers the launch; `l2norm`
        (batched launches only): rows of the result divided by their 2-norm in the same launch; `raw`: no bias either (the
        training path's norm follows)."""
        if shift is None and self.bias is not None and not raw:
            shift = self.bias.detach().view(-1)
        fn = ops.spconv if batch is None else batch.add
        os_pairs = plist if isinstance(plist, ops.OsPairs) else None     # output-stationary tile lists
        kw = {"l2norm": True} if l2norm else {}
        if l2norm and batch is None:
            raise AprHipError("conv.run(l2norm=True) needs a SpconvBatch")
        w_bf3 = self.packed_weight_bf3() if (plist is not None or nbr is None) else None
        # a direct (un-batched) launch on a route that reads the split image alone does not need the tile pack: a training
        # step re-packs every kernel once per optimizer step, so what is not read is not built
        wp = None if (batch is None and self._bf3_only(plist, self.in_channels, w_bf3)) else self.packed_weight()
        return fn(feats, nbr, self.kernel_volume if nbr is not None else 1, self.in_channels,
                  self.out_channels, wp, scale=scale, shift=shift, residual=residual,
                  relu=relu, out=out, n_out=n_out, plist=None if os_pairs is not None else plist,
                  w_bf3=w_bf3, os_pairs=os_pairs, **kw)

    def occ_ready(self, x: SparseTensor):
        """True if this layer on this input is the occupancy special case (ops.occ_conv): constant-1 features, one input
        channel, stride-1 same-level map.  APR_OCC_CONV=0 keeps the kernel-map path."""
        import os
        if not (getattr(x, "unit_features", False) and self.in_channels == 1 and not self.TRANSPOSE and self.stride == 1
                and not self.use_mm and x.coordinate_map_key.stride == 1 and x.F.shape[1] == 1):
            return False
        if os.environ.get("APR_OCC_CONV", "1") == "0" or _tracking(x.F, self.kernel, self.bias):
            return False
        if os.environ.get("APR_CHECK_UNIT_FEATURES") == "1" and not bool((x.F == 1.0).all()):     # debugging aid: synchronises
            raise AprHipError("SparseTensor(unit_features=True): the features are not all 1.0")
        return ops.occ_conv_supported(x.coordinate_manager.get_bbox(), self.kernel_size, self.out_channels, n=x.F.shape[0])

    def run_occ(self, cm, n_out, scale=None, shift=None, relu=False, out=None):
        if shift is None and self.bias is not None:
            shift = self.bias.view(-1)
        w = self.kernel.detach().reshape(self.kernel_volume, self.out_channels)
        return ops.occ_conv(cm.get_map(1).coords, n_out, cm.get_bbox(), self.kernel_size, w, scale=scale, shift=shift,
                            relu=relu, out=out, keep=cm._occ if not cm._occ else None)

    def _reverse_map(self, x: SparseTensor, nbr_fwd, ts_out):
        """Map of the input gradient: (table, mirrored offsets?)  (backward)."""
        ts = x.coordinate_map_key.stride
        cm = x.coordinate_manager
        if ts == ts_out:                       # same level: the same table under the mirrored offset
            return nbr_fwd, True
        if not self.TRANSPOSE:                 # fine -> coarse: its transposed-conv table (coarse rows per fine row)
            return cm.kernel_map(ts_out, ts, self.kernel_size, True), False
        return cm.kernel_map(ts_out, ts, self.kernel_size, False), False   # coarse -> fine: the strided table

    def forward(self, x: SparseTensor):
        if x.F.shape[1] != self.in_channels:
            raise AprHipError(f"conv expects {self.in_channels} input channels, got {x.F.shape[1]}")
        if self.occ_ready(x):
            cm = x.coordinate_manager
            return x._like(self.run_occ(cm, cm.size(1)), CoordinateMapKey(1))
        nbr, ts_out = self._maps(x)
        if _tracking(x.F, self.kernel, self.bias):
            # training: forward + both gradients on the HIP kernels through autograd
            if nbr is None:
                F = x.F @ self.kernel
            else:
                nbr_bwd, flip = self._reverse_map(x, nbr, ts_out)
                F = ops.SparseConvFunction.apply(x.F, self.kernel, nbr, nbr_bwd, flip)
            if self.bias is not None:
                F = F + self.bias
            return x._like(F, CoordinateMapKey(ts_out))
        n_out = x.coordinate_manager.size(ts_out)
        F = self.run(x.F, nbr, n_out)
        return x._like(F, CoordinateMapKey(ts_out))

    def extra_repr(self):
        return (f"in={self.in_channels}, out={self.out_channels}, kernel_size={self.kernel_size}, "
                f"stride={self.stride}")


class MinkowskiConvolution(_ConvBase):
    TRANSPOSE = False


class MinkowskiConvolutionTranspose(_ConvBase):
    TRANSPOSE = True


def fold_bn(bn: nn.BatchNorm1d, mean=None, var=None):
    """BatchNorm -> per-channel (scale, shift)."""
    mean = bn.running_mean if mean is None else mean
    var = bn.running_var if var is None else var
    scale = torch.rsqrt(var + bn.eps)
    if bn.weight is not None:
        scale = scale * bn.weight
    shift = -mean * scale
    if bn.bias is not None:
        shift = shift + bn.bias
    return scale.contiguous(), shift.contiguous()


class MinkowskiBatchNorm(nn.Module):
    """BatchNorm1d over all rows of F; state_dict prefix `.bn.`."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine,
                                 track_running_stats=track_running_stats)
        self._folded = None
        self._folded_key = None

    def folded(self):
        """Eval-mode (scale, shift), cached until a parameter / buffer changes."""
        bn = self.bn
        key = tuple((t.data_ptr(), t._version) for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)
                    if t is not None)
        if self._folded_key != key:
            with torch.no_grad():
                self._folded = fold_bn(bn)
            self._folded_key = key
        return self._folded

    def forward(self, x: SparseTensor):
        bn = self.bn
        if _tracking(x.F, bn.weight, bn.bias):
            if not (self.training or not bn.track_running_stats) or bn.weight is None or x.F.shape[0] < 2:
                return x._like(bn(x.F))        # eval-mode statistics under autograd: torch's BatchNorm1d
            # training: batch statistics, forward AND backward on the HIP kernels (ops.NormFunction); running statistics
            # as torch's BatchNorm1d keeps them (unbiased variance, momentum)
            y, mean, var = ops.NormFunction.apply(x.F, bn.weight, bn.bias, bn.eps)
            if bn.track_running_stats:
                with torch.no_grad():
                    n = x.F.shape[0]
                    mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked + 1)
                    bn.running_mean.mul_(1 - mom).add_(mean, alpha=mom)
                    bn.running_var.mul_(1 - mom).add_(var * (n / max(n - 1, 1)), alpha=mom)
                    bn.num_batches_tracked += 1
            return x._like(y)
        if self.training or not bn.track_running_stats:
            mean, var = ops.bn_stats(x.F)
            if bn.track_running_stats:
                with torch.no_grad():
                    n = x.F.shape[0]
                    mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked + 1)
                    bn.running_mean.mul_(1 - mom).add_(mean, alpha=mom)
                    bn.running_var.mul_(1 - mom).add_(var * (n / max(n - 1, 1)), alpha=mom)
                    bn.num_batches_tracked += 1
            scale, shift = fold_bn(bn, mean, var)
        else:
            scale, shift = self.folded()
        return x._like(ops.affine_act(x.F, scale=scale, shift=shift))


class MinkowskiInstanceNorm(nn.Module):
    """Per-cloud (batch index) mean/var over rows, learnable [1,C] affine, eps 1e-8."""

    def __init__(self, num_features, dimension=None):
        super().__init__()
        self.num_features = num_features
        self.eps = 1e-8
        self.weight = nn.Parameter(torch.ones(1, num_features))
        self.bias = nn.Parameter(torch.zeros(1, num_features))

    def forward(self, x: SparseTensor):
        b = x.C[:, 0]
        nb = int(b.max().item()) + 1
        # rows of one cloud are contiguous (collate order is preserved by first-occurrence maps)
        counts = torch.bincount(b, minlength=nb).cpu().tolist()
        if _tracking(x.F, self.weight, self.bias):
            segs, r0 = [], 0                   # training: plain torch ops per cloud
            for n in counts:
                if n == 0:
                    continue
                segs.append(ops.NormFunction.apply(x.F[r0:r0 + n], self.weight, self.bias, self.eps)[0])   # HIP fwd + bwd
                r0 += n
            return x._like(torch.cat(segs, 0))
        out = torch.empty_like(x.F)
        r0 = 0
        for n in counts:
            if n == 0:
                continue
            seg = x.F[r0:r0 + n]
            mean, var = ops.bn_stats(seg)
            scale = torch.rsqrt(var + self.eps) * self.weight.view(-1)
            shift = self.bias.view(-1) - mean * scale
            ops.affine_act(seg, scale=scale.contiguous(), shift=shift.contiguous(), out=out[r0:r0 + n])
            r0 += n
        return x._like(out)


class MinkowskiReLU(nn.Module):
    def forward(self, x):
        return relu(x)


def relu(x: SparseTensor):
    if _tracking(x.F):
        return x._like(torch.relu(x.F))
    return x._like(ops.affine_act(x.F, relu=True))


def cat(*tensors):
    if len(tensors) == 1 and isinstance(tensors[0], (list, tuple)):
        tensors = tuple(tensors[0])
    a = tensors[0]
    for t in tensors[1:]:
        a._check_same_map(t)
    if _tracking(*[t.F for t in tensors]):
        return a._like(torch.cat([t.F for t in tensors], 1))
    n = len(a)
    ctot = sum(t.F.shape[1] for t in tensors)
    out = torch.empty((n, ctot), dtype=torch.float32, device=a.F.device)
    c0 = 0
    for t in tensors:
        c = t.F.shape[1]
        ops.affine_act(t.F, out=out[:, c0:c0 + c])
        c0 += c
    return a._like(out)
