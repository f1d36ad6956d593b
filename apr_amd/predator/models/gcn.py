"""Overlap-attention module (self / cross / self) on the HIP operator library.

Parameter names and module structure follow /root/reference/Predator_APR/models/gcn.py
(SelfAttention :38-77, MLP :80-92, MultiHeadedAttention :101-116, AttentionalPropagation
:119-128, GCN :171-205).  Features travel as row-major [N, C] tensors (the reference's [1, C, N]
transposed); nothing of size N x N is materialised: the kNN graph comes from a brute-force kNN
kernel, edge features are built per (point, neighbour) row, attention is a fused softmax kernel.
"""
from copy import deepcopy

import torch
import torch.nn as nn

from .. import kp_ops, point_ops
from .blocks import _param_key


class _Packed:
    """Caches the packed [cin, cout] form of a Conv1d/Conv2d(k=1) weight."""

    def __init__(self):
        self.key, self.val = None, None

    def get(self, conv):
        key = _param_key(conv.weight)
        if key != self.key:
            w = conv.weight.detach().reshape(conv.weight.shape[0], conv.weight.shape[1]).t()
            self.val = kp_ops.pack_linear(w)
            self.key = key
        return self.val


class _PackedHeadMajor:
    """The packed weight and the bias of a projection with its OUTPUT channels reordered from the reference's
    interleaved head layout (c = d*heads + h, gcn.py:105-107) to head-major (c = h*dim + d): every output channel is
    its own dot product, so the values are the same bits, only where they land changes -- and a head's 64 channels
    become one contiguous 256-byte run for the attention kernel."""

    def __init__(self):
        self.key, self.val = None, None

    def get(self, conv, heads):
        key = (_param_key(conv.weight), None if conv.bias is None else _param_key(conv.bias), heads)
        if key != self.key:
            cout = conv.weight.shape[0]
            dim = cout // heads
            perm = torch.arange(cout, device=conv.weight.device).view(dim, heads).t().reshape(-1)   # new h*dim+d <- d*heads+h
            w = conv.weight.detach().reshape(cout, conv.weight.shape[1])[perm].t().contiguous()
            bias = None if conv.bias is None else conv.bias.detach()[perm].contiguous()
            self.val = (kp_ops.pack_linear(w), bias)
            self.key = key
        return self.val


def conv1x1(x, conv, cache, relu=False):
    """Conv1d / Conv2d with kernel size 1 on rows: x [N,cin] -> [N,cout] (+ bias)."""
    if kp_ops.tracking(x, conv.weight, conv.bias):
        y = torch.nn.functional.linear(x, conv.weight.reshape(conv.weight.shape[0], conv.weight.shape[1]), conv.bias)
        return torch.relu(y) if relu else y
    bias = None if conv.bias is None else conv.bias.detach()
    return kp_ops.linear(x, cache.get(conv), shift=bias, relu=relu)


class SelfAttention(nn.Module):
    def __init__(self, feature_dim, k=10):
        super().__init__()
        self.conv1 = nn.Conv2d(feature_dim * 2, feature_dim, kernel_size=1, bias=False)
        self.in1 = nn.InstanceNorm2d(feature_dim)
        self.conv2 = nn.Conv2d(feature_dim * 2, feature_dim * 2, kernel_size=1, bias=False)
        self.in2 = nn.InstanceNorm2d(feature_dim * 2)
        self.conv3 = nn.Conv2d(feature_dim * 4, feature_dim, kernel_size=1, bias=False)
        self.in3 = nn.InstanceNorm2d(feature_dim)
        self.k = k
        self._c = [_Packed(), _Packed(), _Packed()]

    def _edge_conv(self, feats, knn, conv, cache, eps):
        n = feats.shape[0]
        if kp_ops.tracking(feats, conv.weight):
            ctr = feats.unsqueeze(1).expand(-1, knn.shape[1], -1)
            e = torch.cat((ctr, feats[knn.long()] - ctr), dim=2).reshape(n * knn.shape[1], -1)
            y = kp_ops.instance_norm_rows(conv1x1(e, conv, cache), eps)          # InstanceNorm2d over n*k
            return torch.nn.functional.leaky_relu(y, 0.2).reshape(n, knn.shape[1], -1).max(1)[0]
        e = kp_ops.edge_features(feats, knn)                   # [n*k, 2c]
        y = conv1x1(e, conv, cache)                            # [n*k, c']
        scale, shift = kp_ops.ops.norm_params(y, eps)          # InstanceNorm2d: per channel over n*k
        return kp_ops.group_max(y, n, knn.shape[1], scale, shift, 0.2)

    def forward(self, coords, features):
        """coords [N,3], features [N,C] -> [N,C]."""
        knn = point_ops.knn(coords, self.k, skip_first=True)
        x0 = features
        x1 = self._edge_conv(x0, knn, self.conv1, self._c[0], self.in1.eps)
        x2 = self._edge_conv(x1, knn, self.conv2, self._c[1], self.in2.eps)
        x3 = conv1x1(torch.cat((x0, x1, x2), dim=1), self.conv3, self._c[2])
        return kp_ops.instance_norm_act(x3, eps=self.in3.eps, leaky=0.2)


def MLP(channels: list, do_bn=True):
    n = len(channels)
    layers = []
    for i in range(1, n):
        layers.append(nn.Conv1d(channels[i - 1], channels[i], kernel_size=1, bias=True))
        if i < (n - 1):
            if do_bn:
                layers.append(nn.InstanceNorm1d(channels[i]))
            layers.append(nn.ReLU())
    return nn.Sequential(*layers)


class MultiHeadedAttention(nn.Module):
    def __init__(self, num_heads: int, d_model: int):
        super().__init__()
        assert d_model % num_heads == 0
        self.dim = d_model // num_heads
        self.num_heads = num_heads
        self.merge = nn.Conv1d(d_model, d_model, kernel_size=1)
        self.proj = nn.ModuleList([deepcopy(self.merge) for _ in range(3)])
        self._c = [_Packed() for _ in range(4)]
        self._hm = [_PackedHeadMajor() for _ in range(3)]

    def forward(self, query, key, value):
        if self.dim == 64 and kp_ops.MHA_MFMA and not kp_ops.tracking(query, key, value, *self.parameters()):
            # projections written head-major, attention on the fp32 MFMA, output back in the interleaved layout
            packed = [c.get(l, self.num_heads) for l, c in zip(self.proj, self._hm)]
            q, k, v = [kp_ops.linear(x, wp, shift=b) for x, (wp, b) in zip((query, key, value), packed)]
            x = kp_ops.mha_headmajor(q, k, v, self.num_heads)
            return conv1x1(x, self.merge, self._c[3])
        q, k, v = [conv1x1(x, l, c) for l, x, c in zip(self.proj, (query, key, value), self._c[:3])]
        if kp_ops.tracking(q, k, v):
            qh, kh, vh = (t.view(-1, self.dim, self.num_heads) for t in (q, k, v))      # channel c = d*heads + h
            prob = torch.softmax(torch.einsum('ndh,mdh->hnm', qh, kh) / self.dim ** .5, dim=-1)
            x = torch.einsum('hnm,mdh->ndh', prob, vh).reshape(q.shape[0], -1)
            return conv1x1(x, self.merge, self._c[3])
        x = kp_ops.mha(q, k, v, self.num_heads)
        return conv1x1(x, self.merge, self._c[3])


class AttentionalPropagation(nn.Module):
    def __init__(self, feature_dim: int, num_heads: int):
        super().__init__()
        self.attn = MultiHeadedAttention(num_heads, feature_dim)
        self.mlp = MLP([feature_dim * 2, feature_dim * 2, feature_dim])
        nn.init.constant_(self.mlp[-1].bias, 0.0)
        self._c = [_Packed(), _Packed()]

    def forward(self, x, source):
        message = self.attn(x, source, source)
        h = conv1x1(torch.cat([x, message], dim=1), self.mlp[0], self._c[0])
        h = kp_ops.instance_norm_act(h, eps=self.mlp[1].eps, relu=True)
        return conv1x1(h, self.mlp[3], self._c[1])


class GCN(nn.Module):
    def __init__(self, num_head: int, feature_dim: int, k: int, layer_names: list):
        super().__init__()
        layers = []
        for atten_type in layer_names:
            if atten_type == 'cross':
                layers.append(AttentionalPropagation(feature_dim, num_head))
            elif atten_type == 'self':
                layers.append(SelfAttention(feature_dim, k))
            else:
                raise NotImplementedError(f"attention type '{atten_type}' is not used by the APR configs")
        self.layers = nn.ModuleList(layers)
        self.names = layer_names

    def forward(self, coords0, coords1, desc0, desc1):
        """coords [N,3], desc [N,C] (row-major) -> updated descriptors."""
        for layer, name in zip(self.layers, self.names):
            if name == 'cross':
                if kp_ops.tracking(desc0, desc1, *layer.parameters()):
                    desc0 = desc0 + layer(desc0, desc1)
                    desc1 = desc1 + layer(desc1, desc0)
                else:
                    desc0 = kp_ops.ops.affine_act(layer(desc0, desc1), residual=desc0)
                    desc1 = kp_ops.ops.affine_act(layer(desc1, desc0), residual=desc1)
            elif name == 'self':
                desc0 = layer(coords0, desc0)
                desc1 = layer(coords1, desc1)
        return desc0, desc1
