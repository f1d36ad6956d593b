"""NPR decoders of Predator_APR on the HIP kernels.

Mirrors /root/reference/Predator_APR/models/mlp.py:103-179: the `GenerativeMLP*` family built as `list_modules`, an
nn.ModuleList of nn.Sequential(Linear, ReLU, BatchNorm1d) -- the reference's `if layer_idx < len(CHANNELS) - 1` is true
for every layer, so the LAST block carries a BatchNorm1d too (:118-127) -- with the same constructor arguments, the same
`state_dict` keys (`list_modules.N.{0,2}.*`: a checkpoint's `generative_model_state_dict`, lib/trainer.py:98-99, loads
strictly), the `(x, radius)` return when a radius was given (:140-143) and `get_GenerativeMLP(config, radius,
in_channels)` (:166-175).  forward() runs apr_amd/npr.run_stack: GEMM + bias + ReLU and the batch-norm forward /
backward are libapr_hip.so calls in train and eval mode alike.
"""
import torch.nn as nn

from ... import npr


class GenerativeMLP(nn.Module):
    CHANNELS = [None, 512, 128, None]

    def __init__(self, in_channel=125, out_points=6, radius=1, bn_momentum=0.1):
        super().__init__()
        channels = list(self.CHANNELS)            # the reference writes into the class attribute; the values are the same
        channels[0], channels[-1] = in_channel, out_points * 3
        self.channels = channels
        self.radius = radius
        self.list_modules = nn.ModuleList(
            nn.Sequential(nn.Linear(channels[i], channels[i + 1]), nn.ReLU(),
                          nn.BatchNorm1d(channels[i + 1], momentum=bn_momentum))
            for i in range(len(channels) - 1))

    def forward(self, x, segments=None):
        x = npr.run_stack([m for block in self.list_modules for m in block], x, segments)
        if self.radius is None:
            return x
        return x, self.radius


class GenerativeMLP_99(GenerativeMLP):
    CHANNELS = [None, 512, 512, None]


class GenerativeMLP_98(GenerativeMLP):
    CHANNELS = [None, 512, 256, None]


class GenerativeMLP_54(GenerativeMLP):
    CHANNELS = [None, 32, 16, None]


class GenerativeMLP_4(GenerativeMLP):
    CHANNELS = [None, 16, None]


class GenerativeMLP_11_10_9(GenerativeMLP):
    CHANNELS = [None, 2048, 1024, 512, None]


def get_GenerativeMLP(config, radius=None, in_channels=None):
    models = [GenerativeMLP_4, GenerativeMLP_98, GenerativeMLP_99, GenerativeMLP_54, GenerativeMLP_11_10_9]
    mdict = {model.__name__: model for model in models}
    if in_channels is None:
        in_channels = config.final_feats_dim
    return mdict[config.generative_model](in_channel=in_channels, out_points=config.point_generation_ratio,
                                          radius=radius, bn_momentum=config.batch_norm_momentum)
