"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of FCGF_APR's sparse-voxel ResUNet on top of `me_oracle`.
Structure follows `/root/reference/FCGF_APR/model/resunet.py:10-193`
(ResUNet2 ctor + forward), `model/residual_block.py:9-53` (BasicBlockBase),
`model/common.py:4-10` (get_norm) and the channel tables at
`resunet.py:196-251`.  Parameter names equal the reference's state_dict keys
(`conv1.kernel`, `norm1.bn.weight`, `block1.conv1.kernel`, `final.bias` ...),
so one state_dict drives both this oracle and the HIP model.

PARITY UNPINNED against upstream MinkowskiEngine (see me_oracle.py header).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import me_oracle as ME


class OConv(nn.Module):
    def __init__(self, cin, cout, kernel_size, stride=1, bias=False, transpose=False):
        super().__init__()
        self.kernel_size, self.stride, self.transpose = kernel_size, stride, transpose
        kv = kernel_size ** 3
        shape = (cin, cout) if (kv == 1 and stride == 1) else (kv, cin, cout)
        self.kernel = nn.Parameter(torch.empty(shape))
        self.bias = nn.Parameter(torch.empty(1, cout)) if bias else None
        stdv = 1.0 / math.sqrt(cin * kv)
        with torch.no_grad():
            self.kernel.uniform_(-stdv, stdv)
            if bias:
                self.bias.uniform_(-stdv, stdv)

    def forward(self, x):
        return ME.conv_forward(x, self.kernel, self.kernel_size, self.stride, self.bias, self.transpose)


class OBatchNorm(nn.Module):
    def __init__(self, c, momentum=0.1):
        super().__init__()
        self.bn = nn.BatchNorm1d(c, momentum=momentum)

    def forward(self, x):
        return ME.batch_norm(x, self.bn)


class OInstanceNorm(nn.Module):
    """Per-cloud mean/var over rows with learnable [1,C] affine."""

    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(1, c))
        self.bias = nn.Parameter(torch.zeros(1, c))

    def forward(self, x):
        C = x.C[:, 0]
        out = torch.empty_like(x.F)
        for b in torch.unique(C):
            m = C == b
            f = x.F[m]
            mu = f.mean(0, keepdim=True)
            var = f.var(0, unbiased=False, keepdim=True)
            out[m] = (f - mu) / torch.sqrt(var + 1e-8)
        return x._like(out * self.weight + self.bias)


def get_norm(norm_type, c, bn_momentum):
    if norm_type == 'BN':
        return OBatchNorm(c, bn_momentum)
    if norm_type == 'IN':
        return OInstanceNorm(c)
    raise ValueError(norm_type)


class OBasicBlock(nn.Module):
    def __init__(self, norm_type, c, bn_momentum):
        super().__init__()
        self.conv1 = OConv(c, c, 3)
        self.norm1 = get_norm(norm_type, c, bn_momentum)
        self.conv2 = OConv(c, c, 3)
        self.norm2 = get_norm(norm_type, c, bn_momentum)

    def forward(self, x):
        out = ME.relu(self.norm1(self.conv1(x)))
        out = self.norm2(self.conv2(out))
        out = out._like(out.F + x.F)
        return ME.relu(out)


class ResUNet2(nn.Module):
    NORM_TYPE = None
    BLOCK_NORM_TYPE = 'BN'
    CHANNELS = [None, 32, 64, 128, 256]
    TR_CHANNELS = [None, 32, 64, 64, 128]

    def __init__(self, in_channels=3, out_channels=32, bn_momentum=0.1, normalize_feature=None,
                 conv1_kernel_size=None, D=3):
        super().__init__()
        N, B, C, T = self.NORM_TYPE, self.BLOCK_NORM_TYPE, self.CHANNELS, self.TR_CHANNELS
        self.normalize_feature = normalize_feature
        self.conv1 = OConv(in_channels, C[1], conv1_kernel_size)
        self.norm1 = get_norm(N, C[1], bn_momentum)
        self.block1 = OBasicBlock(B, C[1], bn_momentum)
        self.conv2 = OConv(C[1], C[2], 3, 2)
        self.norm2 = get_norm(N, C[2], bn_momentum)
        self.block2 = OBasicBlock(B, C[2], bn_momentum)
        self.conv3 = OConv(C[2], C[3], 3, 2)
        self.norm3 = get_norm(N, C[3], bn_momentum)
        self.block3 = OBasicBlock(B, C[3], bn_momentum)
        self.conv4 = OConv(C[3], C[4], 3, 2)
        self.norm4 = get_norm(N, C[4], bn_momentum)
        self.block4 = OBasicBlock(B, C[4], bn_momentum)
        self.conv4_tr = OConv(C[4], T[4], 3, 2, transpose=True)
        self.norm4_tr = get_norm(N, T[4], bn_momentum)
        self.block4_tr = OBasicBlock(B, T[4], bn_momentum)
        self.conv3_tr = OConv(C[3] + T[4], T[3], 3, 2, transpose=True)
        self.norm3_tr = get_norm(N, T[3], bn_momentum)
        self.block3_tr = OBasicBlock(B, T[3], bn_momentum)
        self.conv2_tr = OConv(C[2] + T[3], T[2], 3, 2, transpose=True)
        self.norm2_tr = get_norm(N, T[2], bn_momentum)
        self.block2_tr = OBasicBlock(B, T[2], bn_momentum)
        self.conv1_tr = OConv(C[1] + T[2], T[1], 1)
        self.final = OConv(T[1], out_channels, 1, bias=True)

    def forward(self, x):
        out_s1 = self.block1(self.norm1(self.conv1(x)))
        out = ME.relu(out_s1)
        out_s2 = self.block2(self.norm2(self.conv2(out)))
        out = ME.relu(out_s2)
        out_s4 = self.block3(self.norm3(self.conv3(out)))
        out = ME.relu(out_s4)
        out_s8 = self.block4(self.norm4(self.conv4(out)))
        out = ME.relu(out_s8)

        out = self.block4_tr(self.norm4_tr(self.conv4_tr(out)))
        out = ME.cat(ME.relu(out), out_s4)
        out = self.block3_tr(self.norm3_tr(self.conv3_tr(out)))
        out = ME.cat(ME.relu(out), out_s2)
        out = self.block2_tr(self.norm2_tr(self.conv2_tr(out)))
        out = ME.cat(ME.relu(out), out_s1)
        out = ME.relu(self.conv1_tr(out))
        out = self.final(out)
        if self.normalize_feature:
            return out._like(out.F / torch.norm(out.F, p=2, dim=1, keepdim=True))
        return out


def _variant(name, norm, block_norm, ch, tr):
    return type(name, (ResUNet2,), dict(NORM_TYPE=norm, BLOCK_NORM_TYPE=block_norm, CHANNELS=ch, TR_CHANNELS=tr))


_C = [None, 32, 64, 128, 256]
ResUNetBN2 = _variant('ResUNetBN2', 'BN', 'BN', _C, [None, 32, 64, 64, 128])
ResUNetBN2B = _variant('ResUNetBN2B', 'BN', 'BN', _C, [None, 64, 64, 64, 64])
ResUNetBN2C = _variant('ResUNetBN2C', 'BN', 'BN', _C, [None, 64, 64, 64, 128])
ResUNetBN2D = _variant('ResUNetBN2D', 'BN', 'BN', _C, [None, 64, 64, 128, 128])
ResUNetBN2E = _variant('ResUNetBN2E', 'BN', 'BN', [None, 128, 128, 128, 256], [None, 64, 128, 128, 128])
ResUNetFatBN = _variant('ResUNetFatBN', 'BN', 'BN', _C, [None, 128, 128, 128, 256])
ResUNetIN2 = _variant('ResUNetIN2', 'BN', 'IN', _C, [None, 32, 64, 64, 128])
ResUNetIN2B = _variant('ResUNetIN2B', 'BN', 'IN', _C, [None, 64, 64, 64, 64])
ResUNetIN2C = _variant('ResUNetIN2C', 'BN', 'IN', _C, [None, 64, 64, 64, 128])
ResUNetIN2D = _variant('ResUNetIN2D', 'BN', 'IN', _C, [None, 64, 64, 128, 128])
ResUNetIN2E = _variant('ResUNetIN2E', 'BN', 'IN', [None, 128, 128, 128, 256], [None, 64, 128, 128, 128])

MODELS = {m.__name__: m for m in [ResUNetBN2, ResUNetBN2B, ResUNetBN2C, ResUNetBN2D, ResUNetBN2E, ResUNetFatBN,
                                   ResUNetIN2, ResUNetIN2B, ResUNetIN2C, ResUNetIN2D, ResUNetIN2E]}


def randomize_bn_stats(model, seed=0):
    """Non-trivial eval-mode BN: running mean N(0,0.1), var U[0.5,1.5], affine perturbed."""
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, nn.BatchNorm1d):
            with torch.no_grad():
                m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
                m.weight.copy_(1.0 + 0.2 * torch.randn(m.num_features, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.num_features, generator=g))
