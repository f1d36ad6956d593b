"""Sparse-voxel ResUNet encoder of FCGF_APR on the HIP operator library.

Constructor arguments, sub-module names (= state_dict keys), channel tables and
forward semantics follow /root/reference/FCGF_APR/model/resunet.py:10-251
(ResUNet2 and its BN2/BN2B/BN2C/BN2D/BN2E/FatBN/IN2* variants), so
`Model(in_channels, out_channels, bn_momentum=, normalize_feature=,
conv1_kernel_size=, D=3)` and `model(x: SparseTensor) -> SparseTensor` are
drop-in (call sites FCGF_APR/lib/trainer.py:41-47, scripts/test_apr.py:78-83).

Two execution paths produce the same numbers:
  * `forward_modular` walks the ME-style modules one op per launch (any mode);
  * `forward_fused` (eval mode, BN everywhere) runs the whole encoder as 23
    fused sparse-conv launches: eval BatchNorm folded into the conv epilogue,
    residual add + ReLU fused, ME.cat removed by writing decoder/skip outputs
    straight into column slices of the concat buffers, and the coordinate
    pyramid (strides 2/4/8) built with a single host sync.
"""
import os

import numpy as np
import torch

from ... import MinkowskiEngine as ME
from ... import ops
from ...MinkowskiEngine import MinkowskiFunctional as MEF
from ...MinkowskiEngine.core import CoordinateMapKey
from .common import get_norm
from .residual_block import get_block


# stages of the fused plan that take the weight-stationary conv path (few pairs per tile, large Cin*Cout); chosen from
# per-layer timings (scripts/layer_bench.py: tile vs gemm + reduce + pair-list build) and the single-stream step time
# (scripts/host_vs_gpu.py).  With 12 frames per call: conv3 79 -> 53 + 13 us, conv4 94 -> 56 + 13 us, conv2_tr
# 204 -> 119 + 19 us; block2_tr (64 channels x 190 k rows, two convs on one pair list) was a draw on the fp32 MFMA
# (2 x 221 vs 2 x 218 + 21 us: its product rows make a round trip through HBM); the 32-channel level stays on the
# tile kernel.  Override with APR_WS_STAGES="conv3,block4,..." / "none".
# Round 2: with the contraction on the bf16 MFMA (3-way split, spconv_ws.hip) the weight-stationary pair also wins on
# block2_tr (191 + 10 us of the shared pair-list build against 218 us on the tile kernel): 1906 -> 1964 pairs/s.
WS_STAGES = ("block2", "block3", "block4", "conv4_tr", "block4_tr", "conv3_tr", "block3_tr", "conv3", "conv4",
             "conv2_tr", "block2_tr")


# Round 3: the residual blocks of the 64-channel levels (two 64 -> 64 convolutions on one same-level map each) run
# output-stationary (spconv_os.hip: tile accumulators in LDS, no product rows): block2_tr 2 x 189 -> 2 x 126 us, block2 /
# block3_tr 2 x 79 -> 2 x 57 us at 12 frames per call.  Only taken where the layer has 64 input channels
# (apr_spconv_os_tile_rows); elsewhere the stage keeps its entry in WS_STAGES.  APR_OS_STAGES overrides ("none": off).
OS_STAGES = ("block2", "block3_tr", "block2_tr")
# ... and only on maps large enough: the tile kernel's per-offset barrier and its streamed weight slices are a fixed cost per
# tile, the triple-list gemm's units are not.  scripts/layer_bench.py, 64 -> 64, os vs ws3 (us): 189 k rows 126 / 141,
# 71 k rows 55 / 54.5, 27.7 k rows (ONE pair per call) 38.4 / 29.1, 10.2 k rows 31.1 / 16.2.
OS_MIN_ROWS = int(os.environ.get("APR_OS_MIN_ROWS", "50000"))
# the fused plan issued by ONE library call (apr_resunet_encode) instead of stage by stage from Python; 0: A/B switch
ENCODE_PLAN = os.environ.get("APR_ENCODE_PLAN", "1") != "0"


# train() + autograd: the fused-node path (forward_train); 0 = the module-by-module autograd path (A/B switch)
TRAIN_FUSED = os.environ.get("APR_TRAIN_FUSED", "1") != "0"
WS3_MAX_ROWS_128 = int(os.environ.get("APR_WS3_MAX_ROWS_128", "100000"))
WS3_CIN128 = os.environ.get("APR_WS3_CIN128", "1") != "0"      # A/B switch: 0 = triple lists for 64 input channels only


def _os_stages():
    env = os.environ.get("APR_OS_STAGES")
    if env is None:
        return set(OS_STAGES)
    return set() if env in ("", "none") else set(env.split(","))


def _ws_stages():
    env = os.environ.get("APR_WS_STAGES")
    if env is None:
        return set(WS_STAGES)
    return set() if env in ("", "none") else set(env.split(","))


class ResUNet2(ME.MinkowskiNetwork):
    NORM_TYPE = None
    BLOCK_NORM_TYPE = 'BN'
    CHANNELS = [None, 32, 64, 128, 256]
    TR_CHANNELS = [None, 32, 64, 64, 128]

    def __init__(self, in_channels=3, out_channels=32, bn_momentum=0.1, normalize_feature=None,
                 conv1_kernel_size=None, D=3):
        ME.MinkowskiNetwork.__init__(self, D)
        NORM_TYPE, BLOCK_NORM_TYPE = self.NORM_TYPE, self.BLOCK_NORM_TYPE
        CH, TR = self.CHANNELS, self.TR_CHANNELS
        self.normalize_feature = normalize_feature

        def conv(cin, cout, k, s, bias=False):
            return ME.MinkowskiConvolution(in_channels=cin, out_channels=cout, kernel_size=k, stride=s,
                                           dilation=1, bias=bias, dimension=D)

        def conv_tr(cin, cout):
            return ME.MinkowskiConvolutionTranspose(in_channels=cin, out_channels=cout, kernel_size=3, stride=2,
                                                    dilation=1, bias=False, dimension=D)

        def norm(c):
            return get_norm(NORM_TYPE, c, bn_momentum=bn_momentum, D=D)

        def block(c):
            return get_block(BLOCK_NORM_TYPE, c, c, bn_momentum=bn_momentum, D=D)

        self.conv1 = conv(in_channels, CH[1], conv1_kernel_size, 1)
        self.norm1 = norm(CH[1])
        self.block1 = block(CH[1])
        self.conv2 = conv(CH[1], CH[2], 3, 2)
        self.norm2 = norm(CH[2])
        self.block2 = block(CH[2])
        self.conv3 = conv(CH[2], CH[3], 3, 2)
        self.norm3 = norm(CH[3])
        self.block3 = block(CH[3])
        self.conv4 = conv(CH[3], CH[4], 3, 2)
        self.norm4 = norm(CH[4])
        self.block4 = block(CH[4])
        self.conv4_tr = conv_tr(CH[4], TR[4])
        self.norm4_tr = norm(TR[4])
        self.block4_tr = block(TR[4])
        self.conv3_tr = conv_tr(CH[3] + TR[4], TR[3])
        self.norm3_tr = norm(TR[3])
        self.block3_tr = block(TR[3])
        self.conv2_tr = conv_tr(CH[2] + TR[3], TR[2])
        self.norm2_tr = norm(TR[2])
        self.block2_tr = block(TR[2])
        self.conv1_tr = conv(CH[1] + TR[2], TR[1], 1, 1)
        self.final = conv(TR[1], out_channels, 1, 1, bias=True)
        self.use_fused = True

    # ------------------------------------------------------------------ modular
    def forward_modular(self, x):
        out_s1 = self.block1(self.norm1(self.conv1(x)))
        out = MEF.relu(out_s1)
        out_s2 = self.block2(self.norm2(self.conv2(out)))
        out = MEF.relu(out_s2)
        out_s4 = self.block3(self.norm3(self.conv3(out)))
        out = MEF.relu(out_s4)
        out_s8 = self.block4(self.norm4(self.conv4(out)))
        out = MEF.relu(out_s8)

        out = self.block4_tr(self.norm4_tr(self.conv4_tr(out)))
        out_s4_tr = MEF.relu(out)
        out = ME.cat(out_s4_tr, out_s4)

        out = self.block3_tr(self.norm3_tr(self.conv3_tr(out)))
        out_s2_tr = MEF.relu(out)
        out = ME.cat(out_s2_tr, out_s2)

        out = self.block2_tr(self.norm2_tr(self.conv2_tr(out)))
        out_s1_tr = MEF.relu(out)
        out = ME.cat(out_s1_tr, out_s1)

        out = MEF.relu(self.conv1_tr(out))
        out = self.final(out)
        if self.normalize_feature:
            if torch.is_grad_enabled() and out.F.requires_grad:     # resunet.py:187-190, through autograd
                F = out.F / torch.norm(out.F, p=2, dim=1, keepdim=True)
            else:
                F = ops.l2_normalize(out.F)
            return ME.SparseTensor(F, coordinate_map_key=out.coordinate_map_key,
                                   coordinate_manager=out.coordinate_manager)
        return out

    # ------------------------------------------------------------------ fused
    def _can_fuse(self):
        return (self.use_fused and not self.training and not torch.is_grad_enabled() and self.NORM_TYPE == 'BN'
                and self.BLOCK_NORM_TYPE == 'BN')

    # ---- the fused plan as ONE library call (apr_resunet_encode, csrc/encoder.hip): same launches, issued from C
    _STAGES = ("1", "2", "3", "4", "4_tr", "3_tr", "2_tr")

    def _encode_plan(self):
        """The model as an apr_resunet_plan (packed weights, folded BatchNorm, routing), rebuilt when a parameter, a buffer
        or a routing switch changes."""
        import ctypes as C
        from ... import _lib
        tensors = getattr(self, "_plan_tensors", None)
        if tensors is None:
            tensors = self._plan_tensors = list(self.parameters()) + list(self.buffers())
        ws, osn = _ws_stages(), _os_stages()
        env = (tuple(sorted(ws)), tuple(sorted(osn)), os.environ.get("APR_WS3", "1"), os.environ.get("APR_WS_BF3", "1"),
               WS3_CIN128, WS3_MAX_ROWS_128, OS_MIN_ROWS, ME.core.OCC_KERNEL_MAP, bool(self.normalize_feature))
        key = (tuple(t._version for t in tensors), tuple(t.data_ptr() for t in tensors), env)
        if getattr(self, "_plan_key", None) == key:
            return self._plan
        plan, keep = _lib.ResunetPlan(), []

        def fill(i, conv, norm, relu):
            L = plan.layer[i]
            L.K, L.cin, L.cout, L.relu = conv.kernel_volume if conv.kernel.dim() == 3 else 1, conv.in_channels, conv.out_channels, int(relu)
            wp, w3 = conv.packed_weight(), conv.packed_weight_bf3()
            sc, sh = norm.folded() if norm is not None else (None, None)
            if sh is None and conv.bias is not None:
                sh = conv.bias.detach().view(-1)
            keep.extend([wp, w3, sc, sh])
            L.w_packed, L.w_bf3 = wp.data_ptr(), (w3.data_ptr() if w3 is not None else None)
            L.scale, L.shift = (sc.data_ptr() if sc is not None else None), (sh.data_ptr() if sh is not None else None)

        for s, name in enumerate(self._STAGES):
            blk = getattr(self, "block" + name)
            fill(3 * s, getattr(self, "conv" + name), getattr(self, "norm" + name), False)
            fill(3 * s + 1, blk.conv1, blk.norm1, True)
            fill(3 * s + 2, blk.conv2, blk.norm2, True)
            if "conv" + name in ws:
                plan.ws_conv |= 1 << s
            if "block" + name in ws:
                plan.ws_block |= 1 << s
            if "block" + name in osn:
                plan.os_block |= 1 << s
        fill(21, self.conv1_tr, None, True)
        fill(22, self.final, None, False)
        w1 = self.conv1.kernel.detach().reshape(self.conv1.kernel_volume, self.conv1.out_channels).float().contiguous()
        keep.append(w1)
        plan.conv1_w, plan.conv1_ks, plan.normalize = w1.data_ptr(), int(self.conv1.kernel_size), int(bool(self.normalize_feature))
        plan.ws3, plan.ws3_cin128 = int(os.environ.get("APR_WS3", "1") != "0"), int(WS3_CIN128)
        plan.os_min_rows, plan.occ_kernel_map, plan.ws3_max_rows_128 = OS_MIN_ROWS, int(ME.core.OCC_KERNEL_MAP), WS3_MAX_ROWS_128
        self._plan, self._plan_keep, self._plan_key = plan, keep, key
        return plan

    def _forward_plan(self, x, cm):
        """The encode through apr_resunet_encode, or None when the case is not covered (conv1 not on occupancy, a profile
        being recorded, APR_ENCODE_PLAN=0): the caller then walks the stages itself."""
        import ctypes as C
        from ... import _lib
        if not ENCODE_PLAN or ops.PROFILE is not None or self.conv1.bias is not None or not self.conv1.occ_ready(x):
            return None
        lib = _lib.load()
        plan = self._encode_plan()
        lv = (_lib.LevelMap * 4)()
        for l, ts in enumerate((1, 2, 4, 8)):
            m = cm.get_map(ts)
            lv[l].coords, lv[l].keys, lv[l].vals, lv[l].cap, lv[l].n = m.coords.data_ptr(), m.keys.data_ptr(), m.vals.data_ptr(), m.cap, m.n
        box = (C.c_int32 * 8)(*[int(v) for v in cm.get_bbox()])
        pp = C.byref(plan)
        sb = int(lib.apr_resunet_encode_scratch_bytes(pp, lv, box))
        if sb == 0:
            return None
        dev = x.F.device
        scratch = torch.empty(sb, dtype=torch.uint8, device=dev)
        out = torch.empty((cm.size(1), self.final.out_channels), dtype=torch.float32, device=dev)
        counters, slots = None, 0
        if cm._plist_counters is not None and not cm._plists:      # zeroed with the map sizes' fetch and not yet used: hand them over
            counters, slots, cm._plist_counters = cm._plist_counters, cm._plist_counters.shape[0], None
        ops.check(lib.apr_resunet_encode(pp, lv, box, ops.ptr(counters), slots, ops.ptr(scratch), sb, ops.ptr(out), out.shape[1],
                                         ops.stream()))
        return out

    def forward_fused(self, x):
        cm = x.coordinate_manager
        if x.coordinate_map_key.stride != 1:
            raise ValueError("encoder input must be at tensor stride 1")
        cm.build_pyramid([2, 4, 8])
        planned = self._forward_plan(x, cm)
        if planned is not None:
            return ME.SparseTensor(planned, coordinate_map_key=CoordinateMapKey(1), coordinate_manager=cm)
        N1, N2, N3, N4 = (cm.size(s) for s in (1, 2, 4, 8))
        CH, TR = self.CHANNELS, self.TR_CHANNELS
        dev = x.F.device
        k1 = self.conv1.kernel_size

        def buf(n, c):
            return torch.empty((n, c), dtype=torch.float32, device=dev)

        # concat buffers: [decoder output | encoder skip]
        cat1, cat2, cat3 = buf(N1, TR[2] + CH[1]), buf(N2, TR[3] + CH[2]), buf(N3, TR[4] + CH[3])
        s1, s2, s4 = cat1[:, TR[2]:], cat2[:, TR[3]:], cat3[:, TR[4]:]

        batch = ops.SpconvBatch()   # the 23 conv launches leave through ONE library call

        ws, osn = _ws_stages(), _os_stages()
        # One library call per stage (9 per encode) instead of one per encode: with a single step in flight the GPU idled
        # for the ~0.6 ms Python needs to plan all 23 launches before the first one left (APR_FUSED_EAGER=0: one call)
        eager = os.environ.get("APR_FUSED_EAGER", "1") != "0"

        ws3 = os.environ.get("APR_WS3", "1") != "0"      # A/B switch: 0 = one product row per pair for every ws layer

        def ws_list(layer, m):
            """The weight-stationary pair lists of map m for `layer`: x-triple entry lists (half the product rows) where
            the gemm takes the shape (27 offsets, 64 / 128 input channels), per-offset lists elsewhere."""
            tri = ws3 and layer.packed_weight_bf3() is not None and ops.ws3_supported(layer.kernel_volume, layer.in_channels,
                                                                                      layer.out_channels)
            # 128 input channels: the three slices take 144 KB of LDS (one 8-wave workgroup per CU instead of three 4-wave
            # ones) and the zero-padded MFMA share grows with the pairs per row; on the largest same-level maps that costs
            # more than the halved product rows save (FatBN's block2_tr, 189 k rows: 470 vs 413 us; 71 k rows: 161 vs 166)
            if tri and layer.in_channels == 128 and ((m[0] == m[1] and cm.size(m[1]) > WS3_MAX_ROWS_128) or not WS3_CIN128):
                tri = False
            return cm.pair_list(*m, triples=tri)

        def stage(name, feats, cmap, n_out, bmap, out):
            """conv -> folded BN -> residual block; cmap / bmap = (ts_in, ts_out, kernel, transpose)."""
            conv, norm, blk = getattr(self, "conv" + name), getattr(self, "norm" + name), getattr(self, "block" + name)
            sc, sh = norm.folded()
            if name == "1" and conv.occ_ready(x):
                # constant-1 input: conv1 on occupancy alone -- its 5^3 kernel map is never built (ops.occ_conv)
                a = conv.run_occ(cm, n_out, scale=sc, shift=sh)
            else:
                a = conv.run(feats, cm.kernel_map(*cmap), n_out, scale=sc, shift=sh, batch=batch,
                             plist=ws_list(conv, cmap) if "conv" + name in ws else None)
            bl = None
            # 64 input channels only: at 128 the kernel exists and is tested but loses to the weight-stationary pair
            # on every level (12 frames per call: 494 vs 444 us at 189 k rows, 79 vs 69 us at 26 k)
            if ("block" + name in osn and blk.conv1.in_channels == 64 and blk.conv1.packed_weight_bf3() is not None
                    and n_out >= OS_MIN_ROWS):
                bl = cm.os_pair_list(*bmap, blk.conv1.in_channels, blk.conv1.out_channels)
            if bl is None and "block" + name in ws:
                bl = ws_list(blk.conv1, bmap)
            r = blk.fused_eval(a, cm.kernel_map(*bmap), out, batch=batch, plist=bl)
            if eager:
                batch.launch()      # the stage's three convolutions leave now: the GPU works while Python plans the next stage
            return r

        stage("1", x.F, (1, 1, k1, False), N1, (1, 1, 3, False), s1)
        stage("2", s1, (1, 2, 3, False), N2, (2, 2, 3, False), s2)
        stage("3", s2, (2, 4, 3, False), N3, (4, 4, 3, False), s4)
        s8 = stage("4", s4, (4, 8, 3, False), N4, (8, 8, 3, False), buf(N4, CH[4]))
        stage("4_tr", s8, (8, 4, 3, True), N3, (4, 4, 3, False), cat3[:, :TR[4]])
        stage("3_tr", cat3, (4, 2, 3, True), N2, (2, 2, 3, False), cat2[:, :TR[3]])
        stage("2_tr", cat2, (2, 1, 3, True), N1, (1, 1, 3, False), cat1[:, :TR[2]])
        h = self.conv1_tr.run(cat1, None, N1, relu=True, batch=batch)
        # the row normalisation rides in the last layer's launch (same bits as ops.l2_normalize behind it)
        out = self.final.run(h, None, N1, batch=batch, l2norm=self.normalize_feature)
        batch.launch()
        return ME.SparseTensor(out, coordinate_map_key=CoordinateMapKey(1), coordinate_manager=cm)

    # ------------------------------------------------------------------ training (round 5)
    def _can_train_fused(self, x):
        return (TRAIN_FUSED and self.training and torch.is_grad_enabled() and self.NORM_TYPE == 'BN'
                and self.BLOCK_NORM_TYPE == 'BN' and x.coordinate_map_key.stride == 1 and x.F.shape[0] >= 2
                and all(m.bn.weight is not None and m.bn.track_running_stats for m in self.modules()
                        if isinstance(m, ME.MinkowskiBatchNorm)))

    def forward_frames(self, tensors):
        """Several forward calls of the reference in ONE walk of the network: `tensors` = the SparseTensors the trainer would
        pass one after the other (the two frames of a pair, FCGF_APR/lib/complement_trainer.py:386-394; lib/trainer.py:
        478-489) -> their outputs, in order.  The frames ride in one batched tensor (batch indices shifted), every launch
        covers all of them, and each BatchNorm keeps one set of batch statistics PER CALL and feeds them to the running
        statistics in call order (apr_bn_train_fwd's segments) -- the numbers of separate calls, half the launches.
        Outside train() + autograd (or for the IN variants) it simply loops."""
        tensors = list(tensors)
        if len(tensors) < 2 or not all(self._can_train_fused(t) for t in tensors):
            return [self(t) for t in tensors]
        coords, shift = [], None                      # batch indices of call i shifted past those of calls 0 .. i-1 (on the device)
        for t in tensors:
            c = getattr(t, "_input_coords", None)
            c = t.C if c is None else c
            coords.append(c if shift is None else torch.cat((c[:, :1] + shift, c[:, 1:]), 1))
            top = coords[-1][:, 0].max() + 1
            shift = top if shift is None else torch.maximum(shift, top)
        rows = [int(t.F.shape[0]) for t in tensors]
        x = ME.SparseTensor(torch.cat([t.F for t in tensors], 0), coordinates=torch.cat(coords, 0))
        out = self.forward_train(x, frame_rows=rows)
        res, r0 = [], 0
        for t, n in zip(tensors, rows):
            res.append(ME.SparseTensor(out.F[r0:r0 + n], coordinate_map_key=t.coordinate_map_key,
                                       coordinate_manager=t.coordinate_manager))
            r0 += n
        return res

    def forward_train(self, x, frame_rows=None):
        """train() + autograd: the same network as `forward_modular`, walked as 23 fused autograd nodes
        (ops.ConvBnActFunction: routed sparse conv -> training-mode BatchNorm -> (+ residual) -> ReLU, forward and backward
        on the HIP kernels) instead of ~110 module-level ones with torch elementwise ops between them.  The ReLU the reference
        applies to a block's output a second time (resunet.py:146-166) is the identity on it, forward and backward.
        `frame_rows`: x stacks several forward calls (forward_frames): rows per call at stride 1."""
        cm = x.coordinate_manager
        cm.build_pyramid([2, 4, 8])
        segs = {}
        if frame_rows is not None and len(frame_rows) > 1:
            # rows of every pyramid level per call: a level's rows keep the order of the first voxel they cover, so a call's
            # rows are contiguous and start where the batch index first reaches the call's first cloud
            C1 = cm.get_map(1).coords
            bounds = torch.tensor([0] + list(np.cumsum(frame_rows)[:-1]), device=C1.device)
            first_batch = C1[bounds, 0].contiguous()                     # batch index of each call's first cloud
            per = [torch.searchsorted(cm.get_map(ts).coords[:cm.size(ts), 0].contiguous(), first_batch) for ts in (2, 4, 8)]
            host = torch.stack(per).cpu().tolist()                       # one small fetch per encode
            segs[1] = [0] + [int(v) for v in np.cumsum(frame_rows)]
            for ts, b in zip((2, 4, 8), host):
                segs[ts] = [int(v) for v in b] + [cm.size(ts)]
        ws = _ws_stages()
        ws3 = os.environ.get("APR_WS3", "1") != "0"
        k1 = self.conv1.kernel_size

        def lists(name, conv, m, cin, has_bf3):
            """Pair lists of map m for a layer with `cin` input channels, or None (tile kernel)."""
            if name not in ws or m is None or not ops.ws_supported(27, cin, 64) or not has_bf3:
                return None
            tri = ws3 and ops.ws3_supported(27, cin, 64)
            if tri and cin == 128 and ((m[0] == m[1] and cm.size(m[1]) > WS3_MAX_ROWS_128) or not WS3_CIN128):
                tri = False
            return cm.pair_list(*m, triples=tri)

        def reverse(m):
            ts_in, ts_out, k, tr = m
            if ts_in == ts_out:
                return m, True
            return (ts_out, ts_in, k, not tr), False

        def unit(stage_name, conv, norm, feats, m, relu, residual=None):
            if m is None:                      # kernel size 1: the identity map
                nbr = nbr_b = pl = pl_b = None
                flip, n_out = False, feats.shape[0]
            else:
                nbr = cm.kernel_map(*m)
                mb, flip = reverse(m)
                nbr_b = nbr if flip else cm.kernel_map(*mb)
                n_out = cm.size(m[1])
                ok3 = conv.kernel.dim() == 3 and conv.kernel_volume == 27
                pl = lists(stage_name, conv, m, conv.in_channels, ok3 and conv.packed_weight_bf3() is not None) \
                    if conv.out_channels % 64 == 0 else None
                pl_b = None
                if feats.requires_grad and ok3 and conv.in_channels % 64 == 0:
                    pl_b = lists(stage_name, conv, mb, conv.out_channels, conv.packed_weight_T(flip)[1] is not None)
            cfg = dict(conv=conv, bn=norm, nbr=nbr, plist=pl, nbr_bwd=nbr_b, plist_bwd=pl_b, flip=flip, relu=relu, n_out=n_out,
                       segs=segs.get(m[1] if m is not None else 1))
            bn = norm.bn if norm is not None else None
            return ops.ConvBnActFunction.apply(feats, conv.kernel, bn.weight if bn is not None else None,
                                               bn.bias if bn is not None else None, conv.bias, residual, cfg)

        def stage(name, feats, cmap, bmap):
            conv, norm, blk = getattr(self, "conv" + name), getattr(self, "norm" + name), getattr(self, "block" + name)
            a = unit("conv" + name, conv, norm, feats, cmap, False)
            h = unit("block" + name, blk.conv1, blk.norm1, a, bmap, True)
            return unit("block" + name, blk.conv2, blk.norm2, h, bmap, True, residual=a)

        s1 = stage("1", x.F, (1, 1, k1, False), (1, 1, 3, False))
        s2 = stage("2", s1, (1, 2, 3, False), (2, 2, 3, False))
        s4 = stage("3", s2, (2, 4, 3, False), (4, 4, 3, False))
        s8 = stage("4", s4, (4, 8, 3, False), (8, 8, 3, False))
        t4 = stage("4_tr", s8, (8, 4, 3, True), (4, 4, 3, False))
        t2 = stage("3_tr", torch.cat((t4, s4), 1), (4, 2, 3, True), (2, 2, 3, False))
        t1 = stage("2_tr", torch.cat((t2, s2), 1), (2, 1, 3, True), (1, 1, 3, False))
        h = unit("conv1_tr", self.conv1_tr, None, torch.cat((t1, s1), 1), None, True)
        out = unit("final", self.final, None, h, None, False)
        if self.normalize_feature:
            out = ops.L2NormalizeFunction.apply(out)
        return ME.SparseTensor(out, coordinate_map_key=CoordinateMapKey(1), coordinate_manager=cm)

    def forward(self, x):
        if self._can_fuse():
            return self.forward_fused(x)
        if self._can_train_fused(x):
            return self.forward_train(x)
        return self.forward_modular(x)


class ResUNetBN2(ResUNet2):
    NORM_TYPE = 'BN'


class ResUNetBN2B(ResUNet2):
    NORM_TYPE = 'BN'
    CHANNELS = [None, 32, 64, 128, 256]
    TR_CHANNELS = [None, 64, 64, 64, 64]


class ResUNetBN2C(ResUNet2):
    NORM_TYPE = 'BN'
    CHANNELS = [None, 32, 64, 128, 256]
    TR_CHANNELS = [None, 64, 64, 64, 128]


class ResUNetBN2D(ResUNet2):
    NORM_TYPE = 'BN'
    CHANNELS = [None, 32, 64, 128, 256]
    TR_CHANNELS = [None, 64, 64, 128, 128]


class ResUNetBN2E(ResUNet2):
    NORM_TYPE = 'BN'
    CHANNELS = [None, 128, 128, 128, 256]
    TR_CHANNELS = [None, 64, 128, 128, 128]


class ResUNetFatBN(ResUNet2):
    NORM_TYPE = 'BN'
    CHANNELS = [None, 32, 64, 128, 256]
    TR_CHANNELS = [None, 128, 128, 128, 256]


class ResUNetIN2(ResUNet2):
    NORM_TYPE = 'BN'
    BLOCK_NORM_TYPE = 'IN'


class ResUNetIN2B(ResUNetBN2B):
    NORM_TYPE = 'BN'
    BLOCK_NORM_TYPE = 'IN'


class ResUNetIN2C(ResUNetBN2C):
    NORM_TYPE = 'BN'
    BLOCK_NORM_TYPE = 'IN'


class ResUNetIN2D(ResUNetBN2D):
    NORM_TYPE = 'BN'
    BLOCK_NORM_TYPE = 'IN'


class ResUNetIN2E(ResUNetBN2E):
    NORM_TYPE = 'BN'
    BLOCK_NORM_TYPE = 'IN'
