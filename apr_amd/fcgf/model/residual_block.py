"""Residual block of the sparse ResUNet (FCGF_APR/model/residual_block.py:9-77).

Same constructor arguments, sub-module names (`conv1, norm1, conv2, norm2,
downsample`) and forward semantics as the reference's BasicBlockBase:
conv3 -> norm -> relu -> conv3 -> norm -> (+x) -> relu.
"""
import torch.nn as nn

from ... import MinkowskiEngine as ME
from ...MinkowskiEngine import MinkowskiFunctional as MEF
from .common import get_norm


class BasicBlockBase(nn.Module):
    expansion = 1
    NORM_TYPE = 'BN'

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, D=3):
        super().__init__()
        self.conv1 = ME.MinkowskiConvolution(inplanes, planes, kernel_size=3, stride=stride, dimension=D)
        self.norm1 = get_norm(self.NORM_TYPE, planes, bn_momentum=bn_momentum, D=D)
        self.conv2 = ME.MinkowskiConvolution(planes, planes, kernel_size=3, stride=1, dilation=dilation,
                                             bias=False, dimension=D)
        self.norm2 = get_norm(self.NORM_TYPE, planes, bn_momentum=bn_momentum, D=D)
        self.downsample = downsample

    def forward(self, x):
        residual = x
        out = MEF.relu(self.norm1(self.conv1(x)))
        out = self.norm2(self.conv2(out))
        if self.downsample is not None:
            residual = self.downsample(x)
        out += residual
        return MEF.relu(out)

    def fused_eval(self, feats, nbr, out, batch=None, plist=None):
        """Eval-mode BN block as two fused sparse-conv launches.

        feats [N,C] rows in, `out` (maybe a column slice of a concat buffer)
        receives relu(norm2(conv2(relu(norm1(conv1(x))))) + x).
        """
        n = feats.shape[0]
        s1, b1 = self.norm1.folded()
        s2, b2 = self.norm2.folded()
        h = self.conv1.run(feats, nbr, n, scale=s1, shift=b1, relu=True, batch=batch, plist=plist)
        return self.conv2.run(h, nbr, n, scale=s2, shift=b2, residual=feats, relu=True, out=out, batch=batch,
                              plist=plist)


class BasicBlockBN(BasicBlockBase):
    NORM_TYPE = 'BN'


class BasicBlockIN(BasicBlockBase):
    NORM_TYPE = 'IN'


def get_block(norm_type, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, D=3):
    if norm_type == 'BN':
        return BasicBlockBN(inplanes, planes, stride, dilation, downsample, bn_momentum, D)
    elif norm_type == 'IN':
        return BasicBlockIN(inplanes, planes, stride, dilation, downsample, bn_momentum, D)
    else:
        raise ValueError(f'Type {norm_type}, not defined')
