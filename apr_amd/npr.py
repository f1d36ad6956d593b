"""Shared pieces of the NPR (neighbourhood point reconstruction) branch of APR, used by both trees:

  FCGF_APR/model/mlp.py:6-37 + lib/complement_trainer.py:188-196,424-449      -> apr_amd/fcgf/lib/apg.py
  Predator_APR/models/mlp.py:103-179 + lib/trainer.py:131-140,175-207          -> apr_amd/predator/models/mlp.py, lib/trainer.py

Both decoders are stacks of Linear -> ReLU -> BatchNorm1d on rows; both losses are a two-way Chamfer term between the
generated points and the aggregated (APG) cloud plus a length regulariser.  Everything that touches more than a scalar
runs in libapr_hip.so: the GEMMs with bias + ReLU in the epilogue (`kp_ops.LinearReluFunction`), the batch statistics,
normalisation and its backward (`ops.NormFunction`), the exact 1-NN searches of the Chamfer term (`apr_nn3`) and the
ordered scatter of its gradient (reverse table + `apr_reverse_gather`: no float atomics, same bits every run).
"""
import ctypes as C
import weakref

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import check, ptr, stream
from .predator import kp_ops


def _f32_dev(a):
    if not torch.is_tensor(a):
        a = torch.as_tensor(a)
    return a.to(device=torch.device('cuda', torch.cuda.current_device()), dtype=torch.float32)


NN3_CELL = 1.2      # grid cell of the accelerated 1-NN (four voxels of the 0.3 m clouds); 0 = brute force (same bits)


def nn3(a, b, want_sum=True, cell=None):
    """Exact 1-NN of the rows of a [n,3] in b [m,3] -> (index int64 [n], d2 float32 [n], f64 sum or None)."""
    a, b = a.contiguous(), b.contiguous()
    n, m = a.shape[0], b.shape[0]
    cell = NN3_CELL if cell is None else float(cell)
    lib = _lib.load()
    packed = torch.empty(n, dtype=torch.int64, device=a.device)
    total = torch.empty(1, dtype=torch.float64, device=a.device) if want_sum else None
    sb = int(lib.apr_nn3_scratch_bytes(n, m)) if cell > 0 else 0
    scratch = torch.empty(sb, dtype=torch.uint8, device=a.device) if sb else None
    check(lib.apr_nn3(ptr(a), n, ptr(b), m, cell, ptr(packed), ptr(total), ptr(scratch), sb, stream()))
    idx = packed & 0xFFFFFFFF
    d2 = (packed >> 32).to(torch.int32).view(torch.float32)
    return idx, d2, (total[0] if want_sum else None)


def nn3_batch(a, a_offsets, b, b_offsets, cell=None, mean=False):
    """nb independent exact 1-NN searches in one call (apr_nn3_batch): cloud s of a (rows a_offsets[s] .. [s+1]) among cloud s
    of b -> (GLOBAL index into b int64 [n], d2 float32 [n], per-cloud f64 sums [nb]; `mean`: divided by the cloud's rows)."""
    a, b = a.contiguous(), b.contiguous()
    n, m, nb = a.shape[0], b.shape[0], len(a_offsets) - 1
    cell = NN3_CELL if cell is None else float(cell)
    lib = _lib.load()
    packed = torch.empty(n, dtype=torch.int64, device=a.device)
    sums = torch.empty(nb, dtype=torch.float64, device=a.device)
    sb = int(lib.apr_nn3_scratch_bytes(n, m))
    scratch = torch.empty(sb, dtype=torch.uint8, device=a.device)
    ao = (C.c_int64 * (nb + 1))(*[int(v) for v in a_offsets])
    bo = (C.c_int64 * (nb + 1))(*[int(v) for v in b_offsets])
    sc = (C.c_double * nb)(*[1.0 / (int(a_offsets[s + 1]) - int(a_offsets[s])) for s in range(nb)]) if mean else None
    check(lib.apr_nn3_batch(ptr(a), ao, ptr(b), bo, nb, cell, ptr(packed), ptr(sums), sc, ptr(scratch), sb, stream()))
    return packed & 0xFFFFFFFF, (packed >> 32).to(torch.int32).view(torch.float32), sums


def _scatter_rows(rows4, idx, n_out):
    """out[k] = sum of rows4[t] over the t with idx[t] == k, in ascending t (reverse table: deterministic)."""
    rev_t, start = kp_ops.reverse_table(idx.to(torch.int32).view(-1, 1).contiguous(), n_out)
    out = torch.empty((n_out, 4), dtype=torch.float32, device=rows4.device)
    check(_lib.load().apr_reverse_gather(ptr(rows4), 4, ptr(rev_t), ptr(start), n_out, ptr(out), 4, stream()))
    return out


class ChamferFunction(torch.autograd.Function):
    """forward_cd / n1 + backward_cd / n2 with cd(a, b) = sum_i min_j |a_i - b_j|^2 (the value `chamfer_distance` of both
    trainers returns).  `min` passes its gradient to the arg-min pair only, so the backward is: every point pulls on its own
    nearest neighbour (a direct term) and is pulled by the points of the other cloud that chose it (a scatter over the
    arg-min indices, summed per row in a fixed order)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        i_ab, _, s_ab = nn3(a, b)
        i_ba, _, s_ba = nn3(b, a)
        ctx.save_for_backward(a, b, i_ab, i_ba)
        return (s_ab / a.shape[0] + s_ba / b.shape[0]).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        a, b, i_ab, i_ba = ctx.saved_tensors
        n, m = a.shape[0], b.shape[0]
        g = g.to(torch.float32)
        ga = gb = None
        d_ab = (a - b[i_ab]) * (2.0 / n)            # d cd(a,b)/n  / d a_i ; minus this lands on b[i_ab]
        d_ba = (b - a[i_ba]) * (2.0 / m)
        pad = lambda t: torch.nn.functional.pad(t, (0, 1)).contiguous()
        if ctx.needs_input_grad[0]:
            ga = (d_ab - _scatter_rows(pad(d_ba), i_ba, n)[:, :3]) * g
        if ctx.needs_input_grad[1]:
            gb = (d_ba - _scatter_rows(pad(d_ab), i_ab, m)[:, :3]) * g
        return ga, gb


class ChamferBatchFunction(torch.autograd.Function):
    """ChamferFunction for nb cloud pairs at once -> the nb Chamfer values [nb] (float32).  a / b stack the clouds
    (a_offsets / b_offsets: host row offsets); a point's neighbours are searched in its own cloud's partner only."""

    @staticmethod
    def forward(ctx, a, b, a_offsets, b_offsets):
        a, b = a.contiguous(), b.contiguous()
        i_ab, _, m_ab = nn3_batch(a, a_offsets, b, b_offsets, mean=True)
        i_ba, _, m_ba = nn3_batch(b, b_offsets, a, a_offsets, mean=True)
        ctx.save_for_backward(a, b, i_ab, i_ba)
        ctx.offs = (list(a_offsets), list(b_offsets))
        return (m_ab + m_ba).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        a, b, i_ab, i_ba = ctx.saved_tensors
        ao, bo = ctx.offs
        n, m = a.shape[0], b.shape[0]
        g = g.to(torch.float32)
        d_ab = a - b[i_ab]
        d_ba = b - a[i_ba]
        for s in range(len(ao) - 1):        # per cloud: 2 g_s / rows (sliced in place: no host -> device copy in the backward)
            d_ab[ao[s]:ao[s + 1]] *= g[s] * (2.0 / (ao[s + 1] - ao[s]))
            d_ba[bo[s]:bo[s + 1]] *= g[s] * (2.0 / (bo[s + 1] - bo[s]))
        pad = lambda t: torch.nn.functional.pad(t, (0, 1)).contiguous()
        ga = gb = None
        if ctx.needs_input_grad[0]:
            ga = d_ab - _scatter_rows(pad(d_ba), i_ba, n)[:, :3]
        if ctx.needs_input_grad[1]:
            gb = d_ba - _scatter_rows(pad(d_ab), i_ab, m)[:, :3]
        return ga, gb, None, None


def chamfer_distance_batch(a, a_offsets, b, b_offsets):
    """The Chamfer value of `chamfer_distance` for every cloud pair of a batch in one call -> float32 [nb], differentiable."""
    return ChamferBatchFunction.apply(_f32_dev(a), _f32_dev(b), [int(v) for v in a_offsets], [int(v) for v in b_offsets])


def chamfer_distance(array1, array2):
    """`chamfer_distance(array1, array2)` of both trainers: forward / n1 + backward / n2, 0-d float32 GPU tensor,
    differentiable in both clouds."""
    a, b = _f32_dev(array1), _f32_dev(array2)
    if a.shape[0] == 0 or b.shape[0] == 0:
        return torch.full((), float('nan'), device=a.device)        # chamferdist on an empty cloud: the caller's NaN check
    return ChamferFunction.apply(a, b)


_PACK_CACHE = {}


def _packed(weight):
    """The GEMM image of an nn.Linear weight, rebuilt only when the parameter changed (optimizer step / load)."""
    key = id(weight)
    ver = (weight.data_ptr(), weight._version)
    hit = _PACK_CACHE.get(key)
    if hit is None or hit[0]() is not weight or hit[1] != ver:
        if len(_PACK_CACHE) > 64:
            _PACK_CACHE.clear()
        hit = (weakref.ref(weight), ver, kp_ops.pack_linear(weight.detach().t(), bf3=True))
        _PACK_CACHE[key] = hit
    return hit[2]


def linear_relu(x, lin: nn.Linear, relu=True):
    wp = _packed(lin.weight)
    if kp_ops.tracking(x, lin.weight, lin.bias):
        if relu:
            return kp_ops.LinearReluFunction.apply(x, lin.weight, wp, lin.bias)
        return kp_ops.LinearFunction.apply(x, lin.weight, wp, lin.bias)
    return kp_ops.linear(x.contiguous(), wp, shift=None if lin.bias is None else lin.bias.detach(), relu=relu)


def batch_norm_rows(x, bn: nn.BatchNorm1d, segments=None):
    """nn.BatchNorm1d on rows [n, c]: training = batch statistics + running-statistics update, eval = running statistics.
    `segments` (row offsets, training only): the rows of several module calls stacked -- statistics per call."""
    use_batch = bn.training or not bn.track_running_stats
    if segments is not None and len(segments) > 2 and use_batch:
        if bn.weight is None or not bn.track_running_stats or bn.momentum is None:
            raise NotImplementedError("stacked calls need an affine BatchNorm1d with running statistics and a fixed momentum")
        if kp_ops.tracking(x, bn.weight, bn.bias):
            return ops.BnTrainFunction.apply(x, bn.weight, bn.bias, bn, list(segments))
        return ops.bn_train_fwd(x.contiguous(), bn, segments=list(segments))[0]
    if use_batch and x.shape[0] < 2:
        raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(x.shape)}")
    if kp_ops.tracking(x, bn.weight, bn.bias):
        if not use_batch or bn.weight is None:
            return bn(x)
        y, mean, var = ops.NormFunction.apply(x, bn.weight, bn.bias, bn.eps)
    elif use_batch:
        mean, var = ops.bn_stats(x)
        y = None
    else:
        mean, var, y = bn.running_mean, bn.running_var, None
    if use_batch and bn.track_running_stats:
        with torch.no_grad():
            n = x.shape[0]
            mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked + 1)
            bn.running_mean.mul_(1 - mom).add_(mean, alpha=mom)
            bn.running_var.mul_(1 - mom).add_(var * (n / max(n - 1, 1)), alpha=mom)
            bn.num_batches_tracked += 1
    if y is not None:
        return y
    with torch.no_grad():
        scale = torch.rsqrt(var + bn.eps)
        if bn.weight is not None:
            scale = scale * bn.weight.detach()
        shift = -mean * scale
        if bn.bias is not None:
            shift = shift + bn.bias.detach()
    return ops.affine_act(x, scale=scale.contiguous(), shift=shift.contiguous())


def run_stack(mods, x, segments=None):
    """A list of nn.Linear / nn.ReLU / nn.BatchNorm1d modules in the order the reference stacks them, on the HIP kernels.
    `segments`: x stacks the rows of several calls of the module (row offsets): every BatchNorm keeps per-call statistics."""
    x = _f32_dev(x)
    if x.dim() != 2:
        raise ValueError("NPR decoder: expected rows [n, c]")
    mods = list(mods)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Linear):
            relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            x = linear_relu(x, m, relu)
            i += 2 if relu else 1
        elif isinstance(m, nn.BatchNorm1d):
            x = batch_norm_rows(x, m, segments)
            i += 1
        elif isinstance(m, nn.ReLU):
            x = torch.relu(x)
            i += 1
        else:
            raise NotImplementedError(type(m))
    return x


def regulariser(generated, reg_type='L2', alpha=0.1):
    """Length penalty on the generated offsets [N, 3*ratio] (complement_trainer.py:432-440; Predator lib/trainer.py:176)."""
    sq = (generated.reshape(-1, 3) ** 2).sum(-1)
    if reg_type == 'L2':
        return sq.mean()
    if reg_type == 'RepelL2':
        return sq.mean() + (1.0 / (sq + alpha)).mean()
    if reg_type == 'RepelL1':
        return ((torch.pow(sq + 1e-5, 0.25) - 1) ** 2).mean()
    raise ValueError(reg_type)
