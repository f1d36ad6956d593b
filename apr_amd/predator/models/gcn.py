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

import os

from .. import kp_ops, point_ops
from .blocks import _param_key

# the overlap-attention module of a pair through ONE library call (apr_gcn_forward); 0: layer by layer (A/B switch)
GCN_CALL = os.environ.get("APR_GCN_CALL", "1") != "0"
# the self-attention layers' edge convolutions in training on the HIP kernels; 0: torch ops (A/B switch)
EDGE_CONV_TRAIN_HIP = os.environ.get("APR_EDGE_CONV_TRAIN_HIP", "1") != "0"


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
        # training: forward, d x, d W and d bias on the HIP kernels (kp_ops.LinearFunction; any width -- proj_score is 256 -> 1)
        y = kp_ops.linear_train(x, conv.weight.reshape(conv.weight.shape[0], conv.weight.shape[1]), cache.get(conv), conv.bias)
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
            k = knn.shape[1]
            if EDGE_CONV_TRAIN_HIP and feats.dim() == 2 and k <= 255:
                # training on the HIP kernels: edge features (gradient over the reverse table of the kNN graph), the 1x1
                # convolution (LinearFunction), InstanceNorm2d + LeakyReLU as one Function over the n*k rows, and the max
                # over a point's k edges as max_pool over the rows i*k .. i*k + k - 1 (arg-max kept, gather backward)
                e = kp_ops.EdgeFeaturesFunction.apply(feats, knn)
                y = kp_ops.instance_norm_act(conv1x1(e, conv, cache), eps=eps, leaky=0.2)
                return kp_ops.pool_train(y, kp_ops.group_rows(n, k, feats.device), "max")
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
        if kp_ops.tracking(q, k, v) and kp_ops.HIP_TRAIN_MHA:
            # training: softmax attention forward (probabilities kept) and backward on the HIP kernels
            return conv1x1(kp_ops.MHAFunction.apply(q, k, v, self.num_heads), self.merge, self._c[3])
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

    def _desc(self):
        """The module as an apr_gcn_desc (packed weights, biases, eps), or None when a layer is outside the one-call form
        (channels not a multiple of 64, heads of another width than 64, split weights switched off); rebuilt when a
        parameter changes."""
        from ... import _lib
        params = getattr(self, "_desc_params", None)
        if params is None:
            params = self._desc_params = list(self.parameters())
        key = (tuple(p._version for p in params), tuple(p.data_ptr() for p in params), kp_ops.DENSE_BF3, kp_ops.MHA_MFMA)
        if getattr(self, "_desc_key", None) == key:
            return self._desc_val
        d, keep, ok = _lib.GcnDesc(), [], kp_ops.DENSE_BF3 and kp_ops.MHA_MFMA and len(self.layers) <= 8
        d.n_layers = len(self.layers)

        def w3(conv, cache, heads=None):
            wp = cache.get(conv) if heads is None else cache.get(conv, heads)
            info, bias = (wp, None if conv.bias is None else conv.bias.detach()) if heads is None else wp
            keep.extend([info, bias])
            split = info[5]
            return (split.data_ptr() if split is not None else None), (bias.data_ptr() if bias is not None else None)

        for i, (layer, name) in enumerate(zip(self.layers, self.names)):
            if not ok:
                break
            L = d.layer[i]
            if name == 'self':
                c = layer.conv1.weight.shape[0]
                L.kind, L.k = 0, int(layer.k)
                L.eps1, L.eps2, L.eps3 = layer.in1.eps, layer.in2.eps, layer.in3.eps
                (L.w1, _), (L.w2, _), (L.w3, _) = (w3(cv, ch) for cv, ch in zip((layer.conv1, layer.conv2, layer.conv3), layer._c))
                ok = ok and None not in (L.w1, L.w2, L.w3) and layer.k + 1 <= 16
            else:
                at = layer.attn
                c = at.merge.weight.shape[0]
                L.kind, L.heads, L.eps1 = 1, int(at.num_heads), layer.mlp[1].eps
                (L.wq, L.bq), (L.wk, L.bk), (L.wv, L.bv) = (w3(cv, ch, at.num_heads) for cv, ch in zip(at.proj, at._hm))
                L.wm, L.bm = w3(at.merge, at._c[3])
                (L.w1, L.b1), (L.w2, L.b2) = w3(layer.mlp[0], layer._c[0]), w3(layer.mlp[3], layer._c[1])
                ok = ok and None not in (L.wq, L.wk, L.wv, L.wm, L.w1, L.w2) and at.dim == 64
            d.c = c
            ok = ok and c % 64 == 0
        self._desc_val, self._desc_keep, self._desc_key = (d if ok else None), keep, key
        return self._desc_val

    def _forward_call(self, coords0, coords1, desc0, desc1, out0, out1):
        """The module through apr_gcn_forward (one library call per pair), or None when not covered."""
        import ctypes as C
        from ... import _lib
        d = self._desc()
        n0, n1 = desc0.shape[0], desc1.shape[0]
        if d is None or any(t.stride(0) % 4 or t.data_ptr() % 16 or t.stride(1) != 1 for t in (desc0, desc1)):
            return None
        lib = _lib.load()
        sb = int(lib.apr_gcn_scratch_bytes(C.byref(d), n0, n1))
        if sb == 0:
            return None
        dev = desc0.device
        if out0 is None:
            out0 = torch.empty((n0, d.c), dtype=torch.float32, device=dev)
        if out1 is None:
            out1 = torch.empty((n1, d.c), dtype=torch.float32, device=dev)
        scratch = torch.empty(sb, dtype=torch.uint8, device=dev)
        p0, p1 = (point_ops._pts(p, "gcn.coords") for p in (coords0, coords1))
        kp_ops.check(lib.apr_gcn_forward(C.byref(d), kp_ops.ptr(p0), n0, kp_ops.ptr(p1), n1, kp_ops.ptr(desc0), desc0.stride(0),
                                         kp_ops.ptr(desc1), desc1.stride(0), kp_ops.ptr(out0), out0.stride(0), kp_ops.ptr(out1),
                                         out1.stride(0), kp_ops.ptr(scratch), sb, kp_ops.stream()))
        return out0, out1

    def forward(self, coords0, coords1, desc0, desc1, out0=None, out1=None):
        """coords [N,3], desc [N,C] (row-major) -> updated descriptors (written into out0 / out1 when given)."""
        if GCN_CALL and not kp_ops.tracking(desc0, desc1, *self.parameters()):
            done = self._forward_call(coords0, coords1, desc0, desc1, out0, out1)
            if done is not None:
                return done
        res = self._forward_layers(coords0, coords1, desc0, desc1)
        if out0 is not None:
            out0.copy_(res[0]); out1.copy_(res[1])
            return out0, out1
        return res

    def _forward_layers(self, coords0, coords1, desc0, desc1):
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
