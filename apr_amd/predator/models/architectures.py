"""KPFCNN: KPConv encoder + overlap attention + nearest-upsample decoder on the HIP operator library.

Constructor (`KPFCNN(config)` reading the attributes listed in SURVEY 5.6), sub-module / parameter
names and `forward(batch) -> (feats_f [N0,32] unit norm, scores_overlap [N0], scores_saliency [N0])`
follow /root/reference/Predator_APR/models/architectures.py:9-212; a reference `state_dict` loads
unchanged.  `batch` is the dict built by `collate_fn_descriptor` (points / neighbors / pools /
upsamples / stack_lengths / features), index tensors int32 or int64, everything on the GPU.
"""
import os
import threading

import numpy as np
import torch
import torch.nn as nn

from ... import ops
from .. import kp_ops
from .blocks import NearestUpsampleBlock, ResnetBottleneckBlock, _layer_inputs, block_decider
from .gcn import GCN, _Packed, conv1x1

CAT_BUFFERS = os.environ.get("APR_KP_CAT_BUFFERS", "1") != "0"     # A/B switch: 0 = torch.cat in front of the decoder's unary blocks


def _is_level_change(name):
    return 'pool' in name or 'strided' in name


def plan_architecture(config):
    """Walk `config.architecture` once -> (encoder rows, decoder rows, skip block ids, skip widths, bottleneck width).

    A row is (block name, radius, in width, out width, pyramid level) = the arguments of `block_decider`.  The schedule
    is the reference's (Predator_APR/models/architectures.py:11-123): the radius starts at first_subsampling_dl *
    conv_radius and doubles at every pooling / strided block together with the width, a 'simple' block hands on half
    of its width, the encoder stops at the first 'upsample' block; the decoder then undoes the doubling level by level
    and a block that FOLLOWS an upsampling one takes that level's skip features as extra input channels."""
    names = list(config.architecture)
    first_up = next((i for i, n in enumerate(names) if 'upsample' in n), len(names))
    radius = config.first_subsampling_dl * config.conv_radius
    width_in, width_out, level = config.in_feats_dim, config.first_feats_dim, 0
    enc, skips, skip_widths = [], [], []
    for i, name in enumerate(names[:first_up + 1]):
        if 'equivariant' in name and width_out % 3 != 0:
            raise ValueError('Equivariant block but features dimension is not a factor of 3')
        if _is_level_change(name) or 'upsample' in name or 'global' in name:
            skips.append(i)
            skip_widths.append(width_in)
        if i == first_up:
            break
        enc.append((name, radius, width_in, width_out, level))
        width_in = width_out // 2 if 'simple' in name else width_out
        if _is_level_change(name):
            level, radius, width_out = level + 1, radius * 2, width_out * 2
    bottleneck = width_in
    # decoder: its first block is the parameter-free upsampling one, so the width handed to it (the encoder's last,
    # as in the reference) does not matter; from there on every block emits the conditioned width
    width_out = config.gnn_feats_dim + (2 if config.add_cross_score else 1)
    dec, concats = [], []
    tail = names[first_up:]
    for j, name in enumerate(tail):
        if j > 0 and 'upsample' in tail[j - 1]:
            width_in += skip_widths[level]
            concats.append(j)
        dec.append((name, radius, width_in, width_out, level))
        width_in = width_out
        if 'upsample' in name:
            level, radius, width_out = level - 1, radius * 0.5, width_out // 2
    return enc, dec, skips, skip_widths, concats, bottleneck


_SIDE = threading.local()


def _side_streams(device, n):
    """Per host thread and device: n streams for the per-pair attention blocks of a stacked batch."""
    pool = getattr(_SIDE, "pool", None)
    if pool is None:
        pool = _SIDE.pool = {}
    lst = pool.setdefault(device, [])
    while len(lst) < n:
        lst.append(torch.cuda.Stream(device=device))
    return lst


class KPFCNN(nn.Module):
    def __init__(self, config):
        super().__init__()
        enc, dec, skips, skip_widths, concats, bottleneck = plan_architecture(config)
        self.final_feats_dim = config.final_feats_dim
        self.K = config.num_kernel_points
        self.epsilon = torch.nn.Parameter(torch.tensor(-5.0))
        self.condition = config.condition_feature
        self.add_cross_overlap = config.add_cross_score
        # attribute order = the reference's, so that module registration (and with it RNG consumption at
        # construction and state_dict order) is the same
        self.encoder_blocks = nn.ModuleList(block_decider(*row, config) for row in enc)
        self.encoder_skip_dims, self.encoder_skips = skip_widths, skips
        g = config.gnn_feats_dim
        self.bottle = nn.Conv1d(bottleneck, g, kernel_size=1, bias=True)
        self.gnn = GCN(config.num_head, g, config.dgcnn_k, config.nets)
        self.proj_gnn = nn.Conv1d(g, g, kernel_size=1, bias=True)
        self.proj_score = nn.Conv1d(g, 1, kernel_size=1, bias=True)
        self.decoder_blocks = nn.ModuleList(block_decider(*row, config) for row in dec)
        self.decoder_concats = concats
        self._c = [_Packed(), _Packed(), _Packed()]

    def _host_temperature(self):
        """exp(epsilon) + 0.03 as a host float, fetched once per parameter version: reading it inside every forward is a
        device->host synchronisation in the middle of the network (the host then waits for the whole encoder before it
        can enqueue the attention block, and a single-thread scheduler serves no other batch meanwhile)."""
        key = (self.epsilon._version, self.epsilon.data_ptr())
        if getattr(self, "_temp_key", None) != key:
            self._temp = float(torch.exp(self.epsilon.detach()) + 0.03)
            self._temp_key = key
        return self._temp

    def _drop_temperature(self):
        self._temp_key = None

    def train(self, mode=True):                       # writes through `.data` do not bump `_version`: every entry point
        self._drop_temperature()                      # that can rewrite epsilon behind the cache's back drops it
        return super().train(mode)

    def load_state_dict(self, *args, **kwargs):
        self._drop_temperature()
        return super().load_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self._drop_temperature()
        return super()._apply(fn, *args, **kwargs)

    def _cat_plan(self):
        """{encoder block index: (skip width, upsampled width, buffer width)} for the decoder's concats: the k-th concat from
        the END of the decoder consumes the k-th recorded skip, i.e. the output of the encoder block in front of it."""
        plan = getattr(self, "_cat_plan_cache", None)
        if plan is None:
            plan = {}
            skips = [i for i in self.encoder_skips if i < len(self.encoder_blocks)]      # inputs of these blocks are kept
            for k, dec_i in enumerate(self.decoder_concats):
                if k >= len(skips):
                    break
                src = skips[len(skips) - 1 - k] - 1                                      # the block that PRODUCES that input
                dec = self.decoder_blocks[dec_i]
                up = self.decoder_blocks[dec_i - 1] if dec_i > 0 else None
                if src < 0 or not isinstance(up, NearestUpsampleBlock) or not hasattr(dec, "mlp"):
                    continue
                cs = self.encoder_blocks[src].out_dim
                cx = dec.mlp.in_features - cs
                if cx <= 0 or cs % 4 != 0:
                    continue
                plan[src] = (cs, cx, (cs + cx + 31) // 32 * 32)
            self._cat_plan_cache = plan
        return plan

    def regular_score(self, score):
        score = torch.where(torch.isnan(score), torch.zeros_like(score), score)
        return torch.where(torch.isinf(score), torch.zeros_like(score), score)

    def forward(self, batch):
        """`model.eval()` (or grad disabled): HIP kernels under no_grad.  `model.train()` with grad enabled and
        parameters requiring grad: the same graph on differentiable torch ops."""
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self._forward(batch, True)
        with torch.no_grad():
            return self._forward(batch, False)

    def _forward(self, batch, grad):
        x = batch['features'].clone().detach()
        # rows of the coarsest level: [src, tgt] of one pair (the reference's batch of 1), or of several stacked pairs
        lens_c = [int(v) for v in batch['stack_lengths'][-1]]
        if len(lens_c) % 2:
            raise ValueError("KPFCNN: the batch must hold whole (source, target) pairs")
        pcd_c = batch['points'][-1]

        # 1. joint encoder.  Inference: an encoder output that the decoder later concatenates with an upsampled tensor
        # (architectures.py:188-190) is written straight into the front columns of that concat buffer [skip | x | pad]
        # (blocks.UnaryBlock._weight_cat); the three torch.cat copies and two padding copies of the decoder go away
        skip_x, cat_bufs = [], []
        plan = self._cat_plan() if (not grad and CAT_BUFFERS) else {}
        for block_i, block_op in enumerate(self.encoder_blocks):
            if block_i in self.encoder_skips:
                skip_x.append(x)
            if block_i in plan and isinstance(block_op, ResnetBottleneckBlock):
                cs, cx, width = plan[block_i]
                rows = _layer_inputs(block_op.block_name, block_op.layer_ind, batch)[0].shape[0]
                buf = torch.empty((rows, width), dtype=torch.float32, device=x.device)
                if width > cs + cx:
                    buf[:, cs + cx:].zero_()
                cat_bufs.append((buf, cs, cx))
                x = block_op(x, batch, out=buf[:, :cs])
            else:
                x = block_op(x, batch)

        # 2. bottleneck projection (rows: [N_c, C])
        unconditioned_feats = conv1x1(x, self.bottle, self._c[0])

        # 3./4. per pair: overlap attention between its two clouds, then cross saliency.  The coarsest level holds
        # ~1.4 k points per pair: every kernel of the attention block is a few workgroups running serial loops, so the
        # pairs of a stacked batch go to side streams and run side by side (forked from / joined to the caller's stream)
        gnn_rows, raw_rows, sal_rows, row0 = [], [], [], 0
        temperature = (torch.exp(self.epsilon) + 0.03) if grad else self._host_temperature()
        npairs = len(lens_c) // 2
        side = _side_streams(pcd_c.device, npairs) if (npairs > 1 and not grad) else None
        main = torch.cuda.current_stream() if side else None
        # inference: every pair's two results land in their rows of ONE buffer (allocated on the caller's stream, which joins
        # the side streams below before anything reads it): no concatenation
        gnn_out = None if grad else torch.empty_like(unconditioned_feats)
        for pi, p in enumerate(range(0, len(lens_c), 2)):
            a, b = row0 + lens_c[p], row0 + lens_c[p] + lens_c[p + 1]
            outs = {} if grad else {"out0": gnn_out[row0:a], "out1": gnn_out[a:b]}
            if side:
                side[pi].wait_stream(main)
                with torch.cuda.stream(side[pi]):
                    src_feats_c, tgt_feats_c = self.gnn(pcd_c[row0:a].contiguous(), pcd_c[a:b].contiguous(),
                                                        unconditioned_feats[row0:a], unconditioned_feats[a:b], **outs)
            else:
                src_feats_c, tgt_feats_c = self.gnn(pcd_c[row0:a].contiguous(), pcd_c[a:b].contiguous(),
                                                    unconditioned_feats[row0:a], unconditioned_feats[a:b], **outs)
            gnn_rows += [src_feats_c, tgt_feats_c]
            row0 = b
        if side:
            for st in side[:npairs]:
                main.wait_stream(st)
        feats_c = conv1x1(torch.cat(gnn_rows, dim=0) if grad else gnn_out, self.proj_gnn, self._c[1])
        scores_c_raw = conv1x1(feats_c, self.proj_score, self._c[2])          # [N_c, 1]
        feats_gnn_norm = (torch.nn.functional.normalize(feats_c, p=2, dim=1) if grad else ops.l2_normalize(feats_c))
        feats_gnn_raw = feats_c
        # softmax(<src, tgt> / T) @ scores, both directions, never forming N x N outside autograd
        row0 = 0
        for p in range(0, len(lens_c), 2):
            a, b = row0 + lens_c[p], row0 + lens_c[p] + lens_c[p + 1]
            src_n, tgt_n = feats_gnn_norm[row0:a], feats_gnn_norm[a:b]
            src_s, tgt_s = scores_c_raw[row0:a], scores_c_raw[a:b]
            if grad and kp_ops.HIP_TRAIN_MHA:    # architectures.py:176-181 on the attention kernels of the training path
                sal_rows += [kp_ops.softmax_matvec_train(src_n, tgt_n, tgt_s, temperature).unsqueeze(1),
                             kp_ops.softmax_matvec_train(tgt_n, src_n, src_s, temperature).unsqueeze(1)]
            elif grad:    # the same with torch ops (the N_c x N_c product is ~1 k x 1 k at the coarsest level)
                inner = torch.matmul(src_n, tgt_n.t())
                sal_rows += [torch.matmul(torch.softmax(inner / temperature, dim=1), tgt_s),
                             torch.matmul(torch.softmax(inner.t() / temperature, dim=1), src_s)]
            elif side:
                pi = p // 2
                side[pi].wait_stream(main)
                with torch.cuda.stream(side[pi]):
                    s12 = [kp_ops.softmax_matvec(src_n, tgt_n, tgt_s, temperature),
                           kp_ops.softmax_matvec(tgt_n, src_n, src_s, temperature)]
                for t_ in s12:
                    t_.record_stream(main)
                sal_rows += s12
            else:
                sal_rows += [kp_ops.softmax_matvec(src_n, tgt_n, tgt_s, temperature),
                             kp_ops.softmax_matvec(tgt_n, src_n, src_s, temperature)]
            row0 = b
        if side:
            for st in side[:npairs]:
                main.wait_stream(st)
        scores_saliency = torch.cat(sal_rows, dim=0)
        if not grad:
            scores_saliency = scores_saliency.unsqueeze(1)

        if self.condition and self.add_cross_overlap:
            x = torch.cat([scores_c_raw, scores_saliency, feats_gnn_raw], dim=1)
        elif self.condition and not self.add_cross_overlap:
            x = torch.cat([scores_c_raw, feats_gnn_raw], dim=1)
        elif not self.condition and self.add_cross_overlap:
            x = torch.cat([scores_c_raw, scores_saliency, unconditioned_feats], dim=1)
        else:
            x = torch.cat([scores_c_raw, unconditioned_feats], dim=1)

        # 5. decoder
        for block_i, block_op in enumerate(self.decoder_blocks):
            if cat_bufs and block_i + 1 in self.decoder_concats and isinstance(block_op, NearestUpsampleBlock):
                buf, cs, cx = cat_bufs[-1]                       # the upsampled rows land behind the skip columns
                block_op(x, batch, out=buf[:, cs:cs + cx])
                continue
            if block_i in self.decoder_concats:
                if cat_bufs:
                    buf, cs, cx = cat_bufs.pop()
                    skip_x.pop()
                    x = block_op(buf, batch, cat=(cx, cs))
                    continue
                x = torch.cat([x, skip_x.pop()], dim=1)
            x = block_op(x, batch)
        x = x.contiguous()
        if grad:    # architectures.py:197-212
            feats_f = torch.nn.functional.normalize(x[:, :self.final_feats_dim], p=2, dim=1)
            scores_overlap = self.regular_score(torch.clamp(torch.sigmoid(x[:, self.final_feats_dim]), min=0, max=1))
            scores_saliency = self.regular_score(torch.clamp(torch.sigmoid(x[:, self.final_feats_dim + 1]), min=0, max=1))
            return feats_f, scores_overlap, scores_saliency
        feats_f = ops.l2_normalize(x[:, :self.final_feats_dim])
        scores_overlap = kp_ops.score_head(x[:, self.final_feats_dim])
        scores_saliency = kp_ops.score_head(x[:, self.final_feats_dim + 1])
        return feats_f, scores_overlap, scores_saliency
