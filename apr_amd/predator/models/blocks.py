"""KPConv network blocks on the HIP operator library.

Module names, constructor arguments and parameter names follow
/root/reference/Predator_APR/models/blocks.py (KPConv :134-379, BatchNormBlock :436-474,
UnaryBlock :477-510, LastUnaryBlock :513-536, SimpleBlock :539-593, ResnetBottleneckBlock :596-681,
NearestUpsampleBlock :697-712, MaxPoolBlock :715-726, block_decider :385-433), so a reference
state_dict loads unchanged.  Only the rigid KPConv the APR configs use is implemented
(KP_influence 'linear', aggregation 'sum', not deformable); other settings raise.
Inference runs the HIP kernels; when autograd is recording (training) every block switches
to differentiable torch ops with the reference's formulation — KPConv backward kernels are a later round.
"""
import ctypes as C
import math
import os

import torch
import torch.nn as nn
from torch.nn.init import kaiming_uniform_
from torch.nn.parameter import Parameter

from ... import _lib, ops
from .. import kp_ops
from ..kernels.kernel_points import load_kernels

FUSED_BLOCK = os.environ.get("APR_KP_FUSED_BLOCK", "1") != "0"     # A/B switch: 0 = one library call per op of a block
KPCONV_TRAIN_HIP = os.environ.get("APR_KPCONV_TRAIN_HIP", "1") != "0"   # A/B switch: 0 = KPConv's training path on torch ops
_SEG_ARRAYS = {}


def _seg_array(seg):
    """ctypes int64 array of a pair-offset list, cached by value (the same few lists recur at every block of a batch)."""
    key = tuple(seg)
    a = _SEG_ARRAYS.get(key)
    if a is None:
        if len(_SEG_ARRAYS) > 4096:
            _SEG_ARRAYS.clear()
        a = _SEG_ARRAYS[key] = (C.c_int64 * len(key))(*key)
    return a


def _param_key(*ts):
    return tuple((t.data_ptr(), t._version) for t in ts)


def max_pool(x, inds):
    if kp_ops.tracking(x):
        return kp_ops.pool_train(x, inds, "max")
    return kp_ops.gather_pool(x, inds, "max")


def closest_pool(x, inds, out=None):
    if kp_ops.tracking(x):
        return kp_ops.pool_train(x, inds, "closest")
    return kp_ops.gather_pool(x, inds, "closest", out=out)


class KPConv(nn.Module):
    def __init__(self, kernel_size, p_dim, in_channels, out_channels, KP_extent, radius, fixed_kernel_points='center',
                 KP_influence='linear', aggregation_mode='sum', deformable=False, modulated=False):
        super().__init__()
        if deformable or modulated or KP_influence != 'linear' or aggregation_mode != 'sum':
            raise NotImplementedError("HIP KPConv: rigid / linear influence / sum aggregation only (the APR configs)")
        self.K, self.p_dim = kernel_size, p_dim
        self.in_channels, self.out_channels = in_channels, out_channels
        self.radius, self.KP_extent = radius, KP_extent
        self.fixed_kernel_points = fixed_kernel_points
        self.weights = Parameter(torch.zeros((self.K, in_channels, out_channels), dtype=torch.float32),
                                 requires_grad=True)
        kaiming_uniform_(self.weights, a=math.sqrt(5))
        self.kernel_points = Parameter(torch.tensor(load_kernels(self.radius, self.K, dimension=self.p_dim,
                                                                 fixed=self.fixed_kernel_points),
                                                    dtype=torch.float32), requires_grad=False)
        self._packed, self._key = None, None

    def _weight(self):
        key = _param_key(self.weights)
        if key != self._key:
            kk = self.K * self.in_channels
            ld = (kk + 31) // 32 * 32
            w = self.weights.detach().reshape(kk, self.out_channels)
            if ld != kk:   # zero rows for the padded columns of the weighted-feature matrix
                w = torch.cat([w, torch.zeros((ld - kk, self.out_channels), dtype=w.dtype, device=w.device)], 0)
            self._packed = kp_ops.pack_linear(w)
            self._key = key
        return self._packed

    def forward(self, q_pts, s_pts, neighb_inds, x):
        if kp_ops.tracking(x, self.weights):
            # training: forward and both gradients on the HIP kernels.  The input gradient needs
            # 64-channel multiples (apr_kpconv_dfeat); the first layer (cin = 1) reads the constant input features, which
            # take no gradient: its forward (the generic correlation kernel) and d W (apr_spconv_wgrad over the recomputed
            # [N, 15] weighted features) run on HIP as well
            if (self.in_channels % 64 == 0 or not x.requires_grad) and neighb_inds.shape[1] <= 128 \
                    and self.out_channels % 32 == 0 and KPCONV_TRAIN_HIP:
                return kp_ops.KPConvFunction.apply(q_pts, s_pts, neighb_inds, x, self.weights, self.kernel_points,
                                                   self.KP_extent)
            return self._forward_autograd(q_pts, s_pts, neighb_inds, x)       # other shapes: torch ops
        prof = kp_ops.PROFILE
        if prof is not None:      # bench.py roofline leg: HIP events on the launch stream around the layer's kernels
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        wf = kp_ops.kpconv_weighted(q_pts, s_pts, neighb_inds, x, self.kernel_points, self.KP_extent)
        out = kp_ops.linear(wf, self._weight())
        if prof is not None:
            e1.record()
            prof.append((q_pts.shape[0], neighb_inds.shape[1], self.in_channels, self.out_channels, self.K, e0, e1))
        return out

    def _forward_autograd(self, q_pts, s_pts, inds, x):
        """blocks.py:229-374, rigid / linear influence / sum aggregation, with plain torch ops."""
        s_pad = torch.cat((s_pts, torch.zeros_like(s_pts[:1]) + 1e6), 0)
        neighbors = s_pad[inds.long()] - q_pts.unsqueeze(1)                          # [n, h, 3]
        sq = torch.sum((neighbors.unsqueeze(2) - self.kernel_points) ** 2, dim=3)      # [n, h, k]
        w = torch.clamp(1 - torch.sqrt(sq) / self.KP_extent, min=0.0).transpose(1, 2)  # [n, k, h]
        nx = kp_ops.gather_pad(x, inds)                                              # [n, h, cin]
        weighted = torch.matmul(w, nx).permute(1, 0, 2)                              # [k, n, cin]
        out = torch.sum(torch.matmul(weighted, self.weights), dim=0)
        num = torch.sum(torch.gt(torch.sum(nx, dim=-1), 0.0), dim=-1)
        return out / torch.max(num, torch.ones_like(num)).unsqueeze(1)

    def __repr__(self):
        return 'KPConv(radius: {:.2f}, extent: {:.2f}, in_feat: {:d}, out_feat: {:d})'.format(
            self.radius, self.KP_extent, self.in_channels, self.out_channels)


class BatchNormBlock(nn.Module):
    """`use_bn` -> InstanceNorm1d over ALL stacked points (no affine, no running stats), else a bias."""

    def __init__(self, in_dim, use_bn, bn_momentum):
        super().__init__()
        self.bn_momentum, self.use_bn, self.in_dim = bn_momentum, use_bn, in_dim
        if self.use_bn:
            self.batch_norm = nn.InstanceNorm1d(in_dim, momentum=bn_momentum)
        else:
            self.bias = Parameter(torch.zeros(in_dim, dtype=torch.float32), requires_grad=True)

    def forward(self, x, leaky=None, residual=None, segments=None, out=None):
        """`out` (inference only): a column slice that receives the result (a decoder concat buffer)."""
        if self.use_bn:
            return kp_ops.instance_norm_act(x, eps=self.batch_norm.eps, leaky=leaky, residual=residual,
                                            segments=segments, out=None if kp_ops.tracking(x, residual) else out)
        if kp_ops.tracking(x, self.bias, residual):
            y = x + self.bias
            return kp_ops._act(y if residual is None else y + residual, leaky, False)
        return ops.affine_act(x, shift=self.bias, leaky=leaky, residual=residual, out=out)


class UnaryBlock(nn.Module):
    def __init__(self, in_dim, out_dim, use_bn, bn_momentum, no_relu=False):
        super().__init__()
        self.bn_momentum, self.use_bn, self.no_relu = bn_momentum, use_bn, no_relu
        self.in_dim, self.out_dim = in_dim, out_dim
        self.mlp = nn.Linear(in_dim, out_dim, bias=False)
        self.batch_norm = BatchNormBlock(out_dim, self.use_bn, self.bn_momentum)
        if not no_relu:
            self.leaky_relu = nn.LeakyReLU(0.1)
        self._packed, self._key = None, None

    def _weight(self):
        key = _param_key(self.mlp.weight)
        if key != self._key:
            self._packed = kp_ops.pack_linear(self.mlp.weight.detach().t())
            self._key = key
        return self._packed

    def _weight_cat(self, cx, cs, width):
        """The weight for a `[skip (cs) | x (cx) | zero pad]` concat buffer of `width` columns: the reference concatenates
        [x, skip] (architectures.py:188-190), so the skip rows of W move to the front -- then both parts of the buffer start
        on a 16-byte boundary and neither the torch.cat nor the padding copy is needed."""
        key = (_param_key(self.mlp.weight), cx, cs, width)
        if key != getattr(self, "_cat_key", None):
            w = self.mlp.weight.detach().t()                                   # [cx + cs, out]
            wp = torch.zeros((width, w.shape[1]), dtype=w.dtype, device=w.device)
            wp[:cs], wp[cs:cs + cx] = w[cx:cx + cs], w[:cx]
            self._cat_packed, self._cat_key = kp_ops.pack_linear(wp), key
        return self._cat_packed

    def forward(self, x, batch=None, cat=None):
        """`cat = (cx, cs)`: x is a concat buffer laid out [skip | x | pad] (see _weight_cat)."""
        if cat is not None:
            y = kp_ops.linear(x, self._weight_cat(cat[0], cat[1], x.shape[1]))
        else:
            y = (kp_ops.linear_train(x, self.mlp.weight, self._weight()) if kp_ops.tracking(x, self.mlp.weight)
                 else kp_ops.linear(x, self._weight()))
        return self.batch_norm(y, leaky=None if self.no_relu else 0.1, segments=pair_segments(batch, y))


class LastUnaryBlock(nn.Module):
    def __init__(self, in_dim, out_dim, use_bn, bn_momentum, no_relu=False):
        super().__init__()
        self.in_dim, self.out_dim = in_dim, out_dim
        self.mlp = nn.Linear(in_dim, out_dim, bias=False)
        self._packed, self._key = None, None

    _weight_cat = UnaryBlock._weight_cat

    def forward(self, x, batch=None, cat=None):
        if cat is not None:
            return kp_ops.linear(x, self._weight_cat(cat[0], cat[1], x.shape[1]))
        key = _param_key(self.mlp.weight)
        if key != self._key:
            self._packed = kp_ops.pack_linear(self.mlp.weight.detach().t())
            self._key = key
        if kp_ops.tracking(x, self.mlp.weight):
            return kp_ops.linear_train(x, self.mlp.weight, self._packed)
        return kp_ops.linear(x, self._packed)


def pair_segments(batch, x):
    """Row offsets of the scan pairs stacked in `x`, or None for the reference's one pair per batch.  A collate of
    several pairs (`collate_fn_descriptor` with len(list_data) > 1) records them per level under 'pair_rows', keyed by
    the level's row count (two levels with the same count hold the same clouds row for row)."""
    if batch is None:
        return None
    table = batch.get('pair_rows')
    return None if not table else table[int(x.shape[0])]


def _layer_inputs(block_name, layer_ind, batch):
    if 'strided' in block_name:
        return batch['points'][layer_ind + 1], batch['points'][layer_ind], batch['pools'][layer_ind]
    return batch['points'][layer_ind], batch['points'][layer_ind], batch['neighbors'][layer_ind]


class SimpleBlock(nn.Module):
    def __init__(self, block_name, in_dim, out_dim, radius, layer_ind, config):
        super().__init__()
        current_extent = radius * config.KP_extent / config.conv_radius
        self.bn_momentum, self.use_bn = config.batch_norm_momentum, config.use_batch_norm
        self.layer_ind, self.block_name, self.in_dim, self.out_dim = layer_ind, block_name, in_dim, out_dim
        self.KPConv = KPConv(config.num_kernel_points, config.in_points_dim, in_dim, out_dim // 2, current_extent, radius,
                             fixed_kernel_points=config.fixed_kernel_points, KP_influence=config.KP_influence,
                             aggregation_mode=config.aggregation_mode, deformable='deform' in block_name,
                             modulated=config.modulated)
        self.batch_norm = BatchNormBlock(out_dim // 2, self.use_bn, self.bn_momentum)
        self.leaky_relu = nn.LeakyReLU(0.1)

    def forward(self, x, batch):
        q_pts, s_pts, inds = _layer_inputs(self.block_name, self.layer_ind, batch)
        y = self.KPConv(q_pts, s_pts, inds, x)
        return self.batch_norm(y, leaky=0.1, segments=pair_segments(batch, y))


class ResnetBottleneckBlock(nn.Module):
    def __init__(self, block_name, in_dim, out_dim, radius, layer_ind, config):
        super().__init__()
        current_extent = radius * config.KP_extent / config.conv_radius
        self.bn_momentum, self.use_bn = config.batch_norm_momentum, config.use_batch_norm
        self.block_name, self.layer_ind, self.in_dim, self.out_dim = block_name, layer_ind, in_dim, out_dim
        self.unary1 = UnaryBlock(in_dim, out_dim // 4, self.use_bn, self.bn_momentum) if in_dim != out_dim // 4 \
            else nn.Identity()
        self.KPConv = KPConv(config.num_kernel_points, config.in_points_dim, out_dim // 4, out_dim // 4, current_extent,
                             radius, fixed_kernel_points=config.fixed_kernel_points, KP_influence=config.KP_influence,
                             aggregation_mode=config.aggregation_mode, deformable='deform' in block_name,
                             modulated=config.modulated)
        self.batch_norm_conv = BatchNormBlock(out_dim // 4, self.use_bn, self.bn_momentum)
        self.unary2 = UnaryBlock(out_dim // 4, out_dim, self.use_bn, self.bn_momentum, no_relu=True)
        self.unary_shortcut = UnaryBlock(in_dim, out_dim, self.use_bn, self.bn_momentum, no_relu=True) \
            if in_dim != out_dim else nn.Identity()
        self.leaky_relu = nn.LeakyReLU(0.1)

    def _fused_ok(self, features, inds):
        """One library call for the whole block (apr_kp_resnet_block): inference, InstanceNorm blocks, widths that the
        bf16-split dense GEMM takes, rows 16-byte aligned.  APR_KP_FUSED_BLOCK=0 keeps the module-by-module path."""
        return (FUSED_BLOCK and self.use_bn and not kp_ops.tracking(features, self.KPConv.weights) and kp_ops.PROFILE is None
                and kp_ops.DENSE_BF3 and self.in_dim % 64 == 0 and self.out_dim % 256 == 0 and features.is_cuda
                and features.dtype == torch.float32 and features.dim() == 2 and features.stride(1) == 1
                and features.stride(0) % 4 == 0 and features.data_ptr() % 16 == 0 and inds.dtype == torch.int32
                and inds.is_contiguous() and features.shape[0] > 0 and inds.shape[0] > 0)

    def _forward_fused(self, features, batch, q_pts, s_pts, inds, out=None):
        """-> the block's output, or None when a weight has no bf16-split image (the caller takes the modular path)."""
        w1 = self.unary1._weight()[5] if isinstance(self.unary1, UnaryBlock) else None
        wk, w2 = self.KPConv._weight()[5], self.unary2._weight()[5]
        ws = self.unary_shortcut._weight()[5] if isinstance(self.unary_shortcut, UnaryBlock) else None
        if wk is None or w2 is None or (w1 is None) != isinstance(self.unary1, nn.Identity) \
                or (ws is None) != isinstance(self.unary_shortcut, nn.Identity):
            return None
        lib = _lib.load()
        strided = 'strided' in self.block_name
        n_in, n_out = features.shape[0], q_pts.shape[0]
        mid = self.out_dim // 4
        d = _lib.KpResnetDesc()
        d.x, d.ldx, d.n_in = features.data_ptr(), features.stride(0), n_in
        d.in_dim, d.mid, d.out_dim, d.strided = self.in_dim, mid, self.out_dim, int(strided)
        q_pts, s_pts = q_pts.contiguous(), s_pts.contiguous()
        d.q_pts, d.s_pts, d.n_out = q_pts.data_ptr(), s_pts.data_ptr(), n_out
        d.nbr, d.H, d.n_kp = inds.data_ptr(), inds.shape[1], self.KPConv.K
        kpts = self.KPConv.kernel_points
        d.kernel_points, d.extent = kpts.data_ptr(), float(self.KPConv.KP_extent)
        d.eps, d.slope = float(self.unary2.batch_norm.batch_norm.eps), 0.1
        d.w_unary1 = w1.data_ptr() if w1 is not None else None
        d.w_kpconv, d.w_unary2 = wk.data_ptr(), w2.data_ptr()
        d.w_shortcut = ws.data_ptr() if ws is not None else None
        table = batch.get('pair_rows') if batch is not None else None
        keep = [q_pts, s_pts, w1, wk, w2, ws]
        if table:
            seg_in, seg_out = table[n_in], table[n_out]
            a_in, a_out = _seg_array(seg_in), _seg_array(seg_out)
            d.seg_in, d.seg_out, d.nseg = C.addressof(a_in), C.addressof(a_out), len(seg_out) - 1
            keep += [a_in, a_out]
        else:
            d.nseg = 1
        if out is None:
            out = torch.empty((n_out, self.out_dim), dtype=torch.float32, device=features.device)
        d.out, d.ldo = out.data_ptr(), out.stride(0)
        sb = int(lib.apr_kp_resnet_scratch_bytes(C.byref(d)))
        scratch = torch.empty(sb, dtype=torch.uint8, device=features.device)
        d.scratch, d.scratch_bytes = scratch.data_ptr(), sb
        _lib.check(lib.apr_kp_resnet_block(C.byref(d), _lib.stream()))
        return out

    def forward(self, features, batch, out=None):
        """`out` (inference only): a [rows, out_dim] column slice of a decoder concat buffer that receives the result."""
        q_pts, s_pts, inds = _layer_inputs(self.block_name, self.layer_ind, batch)
        if out is not None and (kp_ops.tracking(features, self.KPConv.weights) or tuple(out.shape) != (q_pts.shape[0], self.out_dim)
                                or out.stride(1) != 1 or out.stride(0) % 4 != 0 or out.data_ptr() % 16 != 0):
            raise ValueError("ResnetBottleneckBlock: `out` must be a 16-byte aligned [rows, out_dim] float32 slice (inference)")
        if self._fused_ok(features, inds):
            res = self._forward_fused(features, batch, q_pts, s_pts, inds, out)
            if res is not None:
                return res
        x = self.unary1(features, batch) if isinstance(self.unary1, UnaryBlock) else features
        x = self.KPConv(q_pts, s_pts, inds, x)
        seg = pair_segments(batch, x)
        x = self.batch_norm_conv(x, leaky=0.1, segments=seg)
        shortcut = max_pool(features, inds) if 'strided' in self.block_name else features
        if isinstance(self.unary_shortcut, UnaryBlock):
            shortcut = self.unary_shortcut(shortcut, batch)
        # unary2 (no ReLU) + shortcut + LeakyReLU fused into the normalisation epilogue
        y = self.unary2.mlp(x) if kp_ops.tracking(x, self.unary2.mlp.weight) else kp_ops.linear(x, self.unary2._weight())
        return self.unary2.batch_norm(y, leaky=0.1, residual=shortcut, segments=seg, out=out)


class NearestUpsampleBlock(nn.Module):
    def __init__(self, layer_ind):
        super().__init__()
        self.layer_ind = layer_ind

    def forward(self, x, batch, out=None):
        return closest_pool(x, batch['upsamples'][self.layer_ind - 1], out=out)

    def __repr__(self):
        return 'NearestUpsampleBlock(layer: {:d} -> {:d})'.format(self.layer_ind, self.layer_ind - 1)


class MaxPoolBlock(nn.Module):
    def __init__(self, layer_ind):
        super().__init__()
        self.layer_ind = layer_ind

    def forward(self, x, batch):
        return max_pool(x, batch['pools'][self.layer_ind + 1])


def block_decider(block_name, radius, in_dim, out_dim, layer_ind, config):
    if block_name == 'unary':
        return UnaryBlock(in_dim, out_dim, config.use_batch_norm, config.batch_norm_momentum)
    if block_name == 'last_unary':
        if config.switch_to_decoder and config.symmetric:
            return LastUnaryBlock(in_dim, config.point_generation_ratio * 3, config.use_batch_norm,
                                  config.batch_norm_momentum)
        return LastUnaryBlock(in_dim, config.final_feats_dim + 2, config.use_batch_norm, config.batch_norm_momentum)
    if block_name in ['simple', 'simple_strided']:
        return SimpleBlock(block_name, in_dim, out_dim, radius, layer_ind, config)
    if block_name in ['resnetb', 'resnetb_strided']:
        return ResnetBottleneckBlock(block_name, in_dim, out_dim, radius, layer_ind, config)
    if block_name in ('max_pool', 'max_pool_wide'):
        return MaxPoolBlock(layer_ind)
    if block_name == 'nearest_upsample':
        return NearestUpsampleBlock(layer_ind)
    if any(t in block_name for t in ('deformable', 'invariant', 'equivariant', 'global')):
        raise NotImplementedError(f"block '{block_name}' is not used by the APR configs")
    raise ValueError('Unknown block name in the architecture definition : ' + block_name)
