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
import math

import torch
import torch.nn as nn
from torch.nn.init import kaiming_uniform_
from torch.nn.parameter import Parameter

from ... import ops
from .. import kp_ops
from ..kernels.kernel_points import load_kernels


def _param_key(*ts):
    return tuple((t.data_ptr(), t._version) for t in ts)


def max_pool(x, inds):
    if kp_ops.tracking(x):
        return kp_ops.gather_pad(x, inds).max(1)[0]
    return kp_ops.gather_pool(x, inds, "max")


def closest_pool(x, inds):
    if kp_ops.tracking(x):
        return kp_ops.gather_pad(x, inds[:, 0])
    return kp_ops.gather_pool(x, inds, "closest")


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
            if self.in_channels % 64 == 0 and neighb_inds.shape[1] <= 128 and self.out_channels % 32 == 0:
                # training: forward and both gradients on the HIP kernels
                return kp_ops.KPConvFunction.apply(q_pts, s_pts, neighb_inds, x, self.weights, self.kernel_points,
                                                   self.KP_extent)
            return self._forward_autograd(q_pts, s_pts, neighb_inds, x)       # first layer (cin = 1): torch ops
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

    def forward(self, x, leaky=None, residual=None, segments=None):
        if self.use_bn:
            return kp_ops.instance_norm_act(x, eps=self.batch_norm.eps, leaky=leaky, residual=residual,
                                            segments=segments)
        if kp_ops.tracking(x, self.bias, residual):
            y = x + self.bias
            return kp_ops._act(y if residual is None else y + residual, leaky, False)
        return ops.affine_act(x, shift=self.bias, leaky=leaky, residual=residual)


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

    def forward(self, x, batch=None):
        y = (kp_ops.linear_train(x, self.mlp.weight, self._weight()) if kp_ops.tracking(x, self.mlp.weight)
             else kp_ops.linear(x, self._weight()))
        return self.batch_norm(y, leaky=None if self.no_relu else 0.1, segments=pair_segments(batch, y))


class LastUnaryBlock(nn.Module):
    def __init__(self, in_dim, out_dim, use_bn, bn_momentum, no_relu=False):
        super().__init__()
        self.in_dim, self.out_dim = in_dim, out_dim
        self.mlp = nn.Linear(in_dim, out_dim, bias=False)
        self._packed, self._key = None, None

    def forward(self, x, batch=None):
        if kp_ops.tracking(x, self.mlp.weight):
            return self.mlp(x)
        key = _param_key(self.mlp.weight)
        if key != self._key:
            self._packed = kp_ops.pack_linear(self.mlp.weight.detach().t())
            self._key = key
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

    def forward(self, features, batch):
        q_pts, s_pts, inds = _layer_inputs(self.block_name, self.layer_ind, batch)
        x = self.unary1(features, batch) if isinstance(self.unary1, UnaryBlock) else features
        x = self.KPConv(q_pts, s_pts, inds, x)
        seg = pair_segments(batch, x)
        x = self.batch_norm_conv(x, leaky=0.1, segments=seg)
        shortcut = max_pool(features, inds) if 'strided' in self.block_name else features
        if isinstance(self.unary_shortcut, UnaryBlock):
            shortcut = self.unary_shortcut(shortcut, batch)
        # unary2 (no ReLU) + shortcut + LeakyReLU fused into the normalisation epilogue
        y = self.unary2.mlp(x) if kp_ops.tracking(x, self.unary2.mlp.weight) else kp_ops.linear(x, self.unary2._weight())
        return self.unary2.batch_norm(y, leaky=0.1, residual=shortcut, segments=seg)


class NearestUpsampleBlock(nn.Module):
    def __init__(self, layer_ind):
        super().__init__()
        self.layer_ind = layer_ind

    def forward(self, x, batch):
        return closest_pool(x, batch['upsamples'][self.layer_ind - 1])

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
