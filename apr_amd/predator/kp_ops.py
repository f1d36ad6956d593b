"""Tensor-level wrappers of the KPConv / overlap-attention kernels (all libapr_hip.so calls)."""
from __future__ import annotations

import ctypes as C
import math
import os

import torch

from .. import _lib, ops
from .._lib import check, ptr, stream


PROFILE = None   # set to a list to collect (N, H, cin, cout, K, start event, end event) per KPConv layer


def kpconv_profile_summary(records):
    """Algorithmic bytes / FLOP (bytes = 4*N*H*(3+cin) + 8*N*H + 4*N*cout; FLOP = 2*N*H*K*(3+cin) +
    2*N*K*cin*cout) and summed HIP-event time of the recorded KPConv layers."""
    torch.cuda.synchronize()
    tot = dict(launches=0, bytes=0.0, flops=0.0, ms=0.0)
    for (n, h, cin, cout, k, e0, e1) in records:
        tot["launches"] += 1
        tot["bytes"] += 4.0 * n * h * (3 + cin) + 8.0 * n * h + 4.0 * n * cout
        tot["flops"] += 2.0 * n * h * k * (3 + cin) + 2.0 * n * k * cin * cout
        tot["ms"] += e0.elapsed_time(e1)
    return tot


def tracking(*tensors):
    """Autograd is recording and one of the tensors / parameters takes part: the modules then run differentiable torch
    ops (the reference's own formulation) instead of the forward-only HIP kernels."""
    return torch.is_grad_enabled() and any(t is not None and torch.is_tensor(t) and t.requires_grad for t in tensors)


def gather_pad(x, inds):
    """x[inds] where index len(x) addresses an appended zero row (blocks.py:71-102 `gather` on the padded tensor)."""
    return torch.cat((x, torch.zeros_like(x[:1])), 0)[inds.long()]


def instance_norm_rows(x, eps):
    mu = x.mean(0, keepdim=True)
    var = x.var(0, unbiased=False, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps)


def _act(y, leaky, relu):
    if leaky is not None:
        return torch.nn.functional.leaky_relu(y, leaky)
    return torch.relu(y) if relu else y


def _i32(t, name):
    if t.dtype != torch.int32:
        t = t.to(torch.int32)
    if not t.is_cuda:
        raise _lib.AprHipError(f"{name}: need a GPU index tensor")
    return t.contiguous()


def linear(x, wp_info, shift=None, relu=False, out=None):
    """x [N,cin] @ W (packed by `pack_linear`) (+ shift) -> [N,cout] on the MFMA GEMM path.

    Channel counts that are not multiples of 32 (the decoder's 1282->129, 641->64, 320->34, the 256->1 score
    head) run zero-padded: W is padded once at pack time, x / shift per call; the result is a column slice.
    """
    wp, cin, cout, cin_p, cout_p, w_bf3 = wp_info
    if cin_p != cin or x.stride(0) % 4 != 0 or x.data_ptr() % 16 != 0:
        x = torch.nn.functional.pad(x, (0, cin_p - cin))
    if cout_p == cout:
        if w_bf3 is not None and DENSE_BF3 and (out is None or (out.stride(0) % 4 == 0 and out.data_ptr() % 16 == 0)) \
                and (shift is None or shift.data_ptr() % 16 == 0):
            # fp32-equivalent contraction on the bf16 MFMA (apr_dense_gemm_bf3): 1.5-1.7x the exact-fp32 kernel
            return ops.dense_gemm_bf3(x, w_bf3, cin_p, cout_p, shift=shift, relu=relu, out=out)
        return ops.spconv(x, None, 1, cin_p, cout_p, wp, shift=shift, relu=relu, out=out, n_out=x.shape[0])
    if shift is not None:
        shift = torch.nn.functional.pad(shift.view(-1), (0, cout_p - cout))
    y = ops.spconv(x, None, 1, cin_p, cout_p, wp, shift=shift, relu=relu, n_out=x.shape[0])[:, :cout]
    if out is not None:
        out.copy_(y)
        return out
    return y


def linear_norm_act(x, wp_info, eps=1e-5, leaky=None, relu=False, residual=None, segments=None, out=None):
    """act(instance_norm(x @ W) (+ residual)) in two launches (apr_dense_gemm_bf3_norm_act: the GEMM leaves its tiles'
    column sums behind, the apply kernel normalises in place) when both widths are multiples of 64; `linear` +
    `instance_norm_act` (three launches) otherwise.  `segments`: row offsets of the stacked scan pairs."""
    wp, cin, cout, cin_p, cout_p, w_bf3 = wp_info
    ok = (w_bf3 is not None and DENSE_BF3 and cin_p == cin and cout_p == cout and x.dim() == 2 and x.is_contiguous()
          and x.data_ptr() % 16 == 0 and not tracking(x, residual))
    if not ok:
        y = linear(x, wp_info)
        return instance_norm_act(y, eps=eps, leaky=leaky, relu=relu, residual=residual, out=out, segments=segments)
    n = x.shape[0]
    if out is None:
        out = torch.empty((n, cout), dtype=torch.float32, device=x.device)
    out, ldo = ops._rows(out, "linear_norm_act.out")
    ldr = 0
    if residual is not None:
        residual, ldr = ops._rows(residual, "linear_norm_act.residual")
    nseg = 0 if segments is None or len(segments) <= 2 else len(segments) - 1
    offs = (C.c_int64 * (nseg + 1))(*[int(v) for v in segments]) if nseg else None
    lib = _lib.load()
    sb = int(lib.apr_dense_gemm_bf3_norm_scratch_bytes(n, cout, max(nseg, 1)))
    scratch = torch.empty(sb, dtype=torch.uint8, device=x.device)
    mode = 2 if leaky is not None else int(bool(relu))
    check(lib.apr_dense_gemm_bf3_norm_act(ptr(x), x.stride(0), n, cin, cout, ptr(w_bf3), float(eps), ptr(residual), ldr, mode,
                                          float(leaky or 0.0), ptr(out), ldo, offs, nseg, ptr(scratch), sb, stream()))
    return out


class LinearFunction(torch.autograd.Function):
    """y = x @ W^T (nn.Linear without bias; KPFCNN's unary / bottleneck layers, Predator_APR/models/blocks.py:499-504) with
    forward, d x and d W on the HIP kernels: y and d x = dy @ W through the dense GEMM (`linear`), d W = dy^T x through
    apr_spconv_wgrad with the identity map (deterministic chunked reduction)."""

    @staticmethod
    def forward(ctx, x, weight, wp_info, bias=None):
        x = x.contiguous()
        ctx.save_for_backward(x, weight)
        return linear(x, wp_info, shift=None if bias is None else bias.detach())

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        cout, cin = weight.shape
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = linear(dy, pack_linear(weight.detach()))          # [N, cout] @ [cout, cin]
        if ctx.needs_input_grad[1]:
            dw = ops.spconv_wgrad(x, dy, None, 1, cin, cout)[0].t()  # (x^T dy)^T
        if len(ctx.needs_input_grad) > 3 and ctx.needs_input_grad[3]:
            db = ops.col_sums(dy)                                  # the bias gradient: column sums, fp64 partials in fixed order
        return dx, dw, None, db


class LinearReluFunction(torch.autograd.Function):
    """relu(x @ W^T + b): nn.Linear followed by nn.ReLU as the NPR decoders stack them (Predator_APR/models/mlp.py:122-127,
    FCGF_APR/model/mlp.py:16-24).  Forward = the dense GEMM with the bias and the ReLU in its epilogue; backward =
    apr_act_backward on the saved output, then LinearFunction's three products."""

    @staticmethod
    def forward(ctx, x, weight, wp_info, bias):
        x = x.contiguous()
        y = linear(x, wp_info, shift=None if bias is None else bias.detach(), relu=True).contiguous()   # padded widths: a slice
        ctx.save_for_backward(x, weight, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        dy = dy.contiguous()
        n, c = y.shape
        dz = torch.empty_like(dy)
        check(_lib.load().apr_act_backward(ptr(dy), c, ptr(y), c, n, c, 1, 0.0, ptr(dz), c, stream()))
        cout, cin = weight.shape
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = linear(dz, pack_linear(weight.detach()))
        if ctx.needs_input_grad[1]:
            dw = ops.spconv_wgrad(x, dz, None, 1, cin, cout)[0].t()
        if ctx.needs_input_grad[3]:
            db = ops.col_sums(dz)
        return dx, dw, None, db


def linear_train(x, weight, wp_info, bias=None):
    """The tracked (training) form of a Linear / 1x1 convolution on rows: forward, d x, d W (and d bias) on the HIP kernels
    (LinearFunction)."""
    if HIP_TRAIN_LINEAR and x.shape[0] > 0 and x.dim() == 2:      # any width: `linear` pads to the GEMM's granule
        return LinearFunction.apply(x, weight, wp_info, bias)
    return torch.nn.functional.linear(x, weight, bias)


HIP_TRAIN_LINEAR = os.environ.get("APR_HIP_TRAIN_LINEAR", "1") != "0"   # A/B switch: 0 = torch's Linear in training
HIP_TRAIN_NORM = os.environ.get("APR_HIP_TRAIN_NORM", "1") != "0"   # A/B switch: 0 = torch ops for the training-mode norms
DENSE_BF3 = os.environ.get("APR_DENSE_BF3", "1") != "0"     # A/B switch: 0 keeps every Linear on the exact-fp32 MFMA


def pack_linear(w_in_out, bf3=True):
    """[cin, cout] weight -> (packed, cin, cout, cin_padded, cout_padded, bf16 3-way split image or None).
    The split image exists when both padded channel counts are multiples of 64 (apr_dense_gemm_bf3)."""
    w = w_in_out.detach().to(torch.float32)
    cin, cout = w.shape
    cin_p, cout_p = (cin + 31) // 32 * 32, (cout + 31) // 32 * 32
    if (cin_p, cout_p) != (cin, cout):
        w = torch.nn.functional.pad(w, (0, cout_p - cout, 0, cin_p - cin))
    w = w.contiguous()
    w_bf3 = ops.pack_weights_bf3(w[None]) if (bf3 and DENSE_BF3 and cin_p % 64 == 0 and cout_p % 64 == 0) else None
    return ops.pack_weights(w), cin, cout, cin_p, cout_p, w_bf3


def row_sums(x):
    x, ld = ops._rows(x, "row_sums.x")
    out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    check(_lib.load().apr_row_sums(ptr(x), ld, x.shape[0], x.shape[1], ptr(out), stream()))
    return out


def kpconv_weighted(q_pts, s_pts, nbr, x, kernel_points, extent):
    """KPConv step 1 -> weighted features [Nq, ld] (ld = 15*cin rounded up to 32), already / neighbour count."""
    nbr = _i32(nbr, "kpconv.nbr")
    x, ldx = ops._rows(x, "kpconv.x")
    nq, ns, cin = q_pts.shape[0], s_pts.shape[0], x.shape[1]
    kk = kernel_points.shape[0] * cin
    ld = (kk + 31) // 32 * 32
    wf = torch.zeros((nq, ld), dtype=torch.float32, device=x.device) if ld != kk else \
        torch.empty((nq, ld), dtype=torch.float32, device=x.device)
    rs = row_sums(x)
    check(_lib.load().apr_kpconv_weighted(ptr(q_pts.contiguous()), nq, ptr(s_pts.contiguous()), ns, ptr(nbr), nbr.shape[1],
                                          ptr(x), ldx, cin, ptr(kernel_points.contiguous()), kernel_points.shape[0],
                                          float(extent), ptr(rs), ptr(wf), ld, stream()))
    return wf


_REV_CACHE = {}     # (data_ptr, shape, version, ns) -> (nbr tensor kept alive, rev_t, start): one build per neighbour table


def reverse_table(nbr, ns):
    """Flat positions of `nbr` [nq, H] sorted by the support row they point at + the start of every row's run
    (apr_reverse_table_build; stable, deterministic).  Cached per table: the layers of a pyramid level share theirs."""
    key = (nbr.data_ptr(), tuple(nbr.shape), nbr._version, int(ns))
    hit = _REV_CACHE.get(key)
    if hit is not None:
        return hit[1], hit[2]
    if len(_REV_CACHE) > 16:
        _REV_CACHE.clear()
    lib = _lib.load()
    nq, H = nbr.shape
    rev_t = torch.empty(nq * H, dtype=torch.int32, device=nbr.device)
    start = torch.empty(int(ns) + 1, dtype=torch.int32, device=nbr.device)
    sb = int(lib.apr_reverse_table_scratch_bytes(nq, H, int(ns)))
    scratch = torch.empty(sb, dtype=torch.uint8, device=nbr.device)
    check(lib.apr_reverse_table_build(ptr(nbr), nq, H, int(ns), ptr(rev_t), ptr(start), ptr(scratch), sb, stream()))
    _REV_CACHE[key] = (nbr, rev_t, start)
    return rev_t, start


DET_DX = os.environ.get("APR_KPCONV_DX", "det") != "atomic"     # A/B switch: "atomic" = round 2's float-atomic scatter
DX_CHUNK_BYTES = int(os.environ.get("APR_KPCONV_DX_CHUNK_MB", "64")) << 20     # contribution rows in flight per chunk


class KPConvFunction(torch.autograd.Function):
    """Rigid KPConv (blocks.py:229-374) with forward AND both gradients on the HIP kernels:
    forward   wf = step 1 (apr_kpconv_weighted), out = wf @ W (dense MFMA GEMM);
    d W       = wf^T @ dout                       (apr_spconv_wgrad, identity map; wf is recomputed, not stored);
    d x       = scatter of (dout @ W^T) through the influences (apr_kpconv_dfeat on the forward's neighbour table).
    The points, the kernel points and the neighbour table get no gradient (rigid kernel)."""

    @staticmethod
    def forward(ctx, q_pts, s_pts, inds, x, weights, kernel_points, extent):
        inds = _i32(inds, "kpconv.nbr")
        x = x.contiguous()
        K, cin, cout = weights.shape
        wf = kpconv_weighted(q_pts, s_pts, inds, x, kernel_points, extent)
        w2 = weights.detach().reshape(K * cin, cout)
        if wf.shape[1] != K * cin:
            w2 = torch.cat([w2, torch.zeros((wf.shape[1] - K * cin, cout), dtype=w2.dtype, device=w2.device)], 0)
        out = linear(wf, pack_linear(w2, bf3=False))
        ctx.save_for_backward(q_pts, s_pts, inds, x, weights, kernel_points)
        ctx.extent = float(extent)
        return out

    @staticmethod
    def backward(ctx, dout):
        q_pts, s_pts, inds, x, weights, kernel_points = ctx.saved_tensors
        K, cin, cout = weights.shape
        dout = dout.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[4]:
            wf = kpconv_weighted(q_pts, s_pts, inds, x, kernel_points, ctx.extent)
            dw = ops.spconv_wgrad(wf, dout, None, 1, wf.shape[1], cout)[0, :K * cin].reshape(K, cin, cout)
        if ctx.needs_input_grad[3]:
            dwf = linear(dout, pack_linear(weights.detach().reshape(K * cin, cout).t().contiguous(), bf3=False))   # [nq, K*cin]
            dwf, lddwf = ops._rows(dwf, "kpconv.dwf")
            rs = row_sums(x)
            lib = _lib.load()
            nq, H, ns = q_pts.shape[0], inds.shape[1], s_pts.shape[0]
            if DET_DX and cin % 4 == 0 and nq * H < (1 << 31):
                # deterministic: one contribution row per (query, neighbour), summed per support row in the fixed order of the
                # reverse table (no float atomics: the same bits every run).  The queries go through in chunks whose
                # contribution rows fit DX_CHUNK_BYTES (64 MB; unchunked the finest level of one pair needs 400 MB): a
                # support row's run is ascending in the flat position q * H + h, so chunk after chunk the gather continues
                # the same sequence of additions -- the bits do not depend on the chunk size (tested).
                rev_t, start = reverse_table(inds, ns)
                qc = max(1, min(nq, DX_CHUNK_BYTES // max(1, H * cin * 4)))
                contrib = torch.empty((min(nq, qc) * H, cin), dtype=torch.float32, device=x.device)
                dx = torch.empty_like(x)
                qp, sp, kp = q_pts.contiguous(), s_pts.contiguous(), kernel_points.contiguous()
                for q0 in range(0, nq, qc):
                    q1 = min(nq, q0 + qc)
                    check(lib.apr_kpconv_dfeat_contrib(ptr(qp[q0:q1]), q1 - q0, ptr(sp), ns, ptr(inds[q0:q1]), H,
                                                       ptr(dwf[q0:q1]), lddwf, cin, ptr(kp), kp.shape[0], ctx.extent, ptr(rs),
                                                       ptr(contrib), stream()))
                    check(lib.apr_reverse_gather_range(ptr(contrib), cin, ptr(rev_t), ptr(start), ns, q0 * H, q1 * H,
                                                       int(q0 > 0), ptr(dx), dx.stride(0), stream()))
            else:
                dx = torch.zeros_like(x)
                check(lib.apr_kpconv_dfeat(ptr(q_pts.contiguous()), nq, ptr(s_pts.contiguous()), ns, ptr(inds), H, ptr(dwf),
                                           lddwf, cin, ptr(kernel_points.contiguous()), kernel_points.shape[0], ctx.extent,
                                           ptr(rs), ptr(dx), dx.stride(0), stream()))
        return None, None, None, dx, dw, None, None


def gather_pool(x, inds, mode, out=None):
    """mode 'max' -> max_pool(x, inds); 'closest' -> closest_pool(x, inds).  `out`: a [nq, c] column slice to write into."""
    inds = _i32(inds, "gather_pool.inds")
    x, ldx = ops._rows(x, "gather_pool.x")
    nq, c = inds.shape[0], x.shape[1]
    if out is None:
        out = torch.empty((nq, c), dtype=torch.float32, device=x.device)
    out, ldo = ops._rows(out, "gather_pool.out")
    if tuple(out.shape) != (nq, c):
        raise _lib.AprHipError("gather_pool: output shape mismatch")
    check(_lib.load().apr_gather_pool(ptr(x), ldx, x.shape[0], c, ptr(inds), inds.shape[1], nq,
                                      1 if mode == "closest" else 0, ptr(out), ldo, stream()))
    return out


class PoolFunction(torch.autograd.Function):
    """max_pool / closest_pool (blocks.py:71-102) with forward and backward on the HIP kernels: the forward records which
    neighbour gave each maximum (apr_gather_pool_argmax), the backward gathers dout over the reverse table of the index tensor
    (apr_gather_pool_backward: deterministic, the same bits every run).  Training path; inference keeps gather_pool."""

    @staticmethod
    def forward(ctx, x, inds, mode):
        inds = _i32(inds, "pool.inds")
        x = x.contiguous()
        ns, c = x.shape
        nq, H = inds.shape
        lib = _lib.load()
        amax = None
        if mode == "max":
            out = torch.empty((nq, c), dtype=torch.float32, device=x.device)
            amax = torch.empty((nq, c), dtype=torch.uint8, device=x.device)
            check(lib.apr_gather_pool_argmax(ptr(x), c, ns, c, ptr(inds), H, nq, ptr(out), c, ptr(amax), stream()))
        else:
            out = gather_pool(x, inds, "closest")
        ctx.save_for_backward(inds, amax if amax is not None else inds)
        ctx.cfg = (mode, int(ns), int(c))
        return out

    @staticmethod
    def backward(ctx, dout):
        inds, amax = ctx.saved_tensors
        mode, ns, c = ctx.cfg
        if not ctx.needs_input_grad[0]:
            return None, None, None
        dout = dout.contiguous()
        rev_t, start = reverse_table(inds, ns)
        dx = torch.empty((ns, c), dtype=torch.float32, device=dout.device)
        check(_lib.load().apr_gather_pool_backward(ptr(dout), c, c, ptr(rev_t), ptr(start), ns, inds.shape[1],
                                                   ptr(amax) if mode == "max" else None, 0 if mode == "max" else 1, ptr(dx), c,
                                                   stream()))
        return dx, None, None


HIP_TRAIN_POOL = os.environ.get("APR_HIP_TRAIN_POOL", "1") != "0"     # A/B switch: 0 = torch indexing for the pools in training


def pool_train(x, inds, mode):
    """The tracked (training) form of max_pool / closest_pool: HIP kernels when the table is narrow enough for the u8 arg-max."""
    if HIP_TRAIN_POOL and x.dim() == 2 and x.shape[0] > 0 and inds.shape[0] > 0 and inds.shape[1] <= 255:
        return PoolFunction.apply(x, inds, mode)
    g = gather_pad(x, inds if mode == "max" else inds[:, 0])
    return g.max(1)[0] if mode == "max" else g


def edge_features(f, knn):
    knn = _i32(knn, "edge_features.knn")
    f, ldf = ops._rows(f, "edge_features.f")
    n, c = f.shape
    out = torch.empty((n * knn.shape[1], 2 * c), dtype=torch.float32, device=f.device)
    check(_lib.load().apr_edge_features(ptr(f), ldf, n, c, ptr(knn), knn.shape[1], ptr(out), stream()))
    return out


class EdgeFeaturesFunction(torch.autograd.Function):
    """get_graph_feature as rows (gcn.py:9-35): e[(i, j)] = [f_i, f_knn(i, j) - f_i], with the input gradient on the HIP
    kernels: the centre's own terms by apr_edge_features_backward, the neighbours' terms gathered over the reverse table of
    the kNN graph in table order (apr_reverse_gather_range): deterministic, no float atomics."""

    @staticmethod
    def forward(ctx, f, knn):
        knn = _i32(knn, "edge_features.knn").contiguous()
        ctx.save_for_backward(knn)
        ctx.n, ctx.c = f.shape
        return edge_features(f.detach(), knn)

    @staticmethod
    def backward(ctx, de):
        knn, = ctx.saved_tensors
        n, c, k = ctx.n, ctx.c, knn.shape[1]
        de = de.contiguous()
        lib = _lib.load()
        dx = torch.empty((n, c), dtype=torch.float32, device=de.device)
        contrib = torch.empty((n * k, c), dtype=torch.float32, device=de.device)
        check(lib.apr_edge_features_backward(ptr(de), n, c, k, ptr(dx), c, ptr(contrib), stream()))
        rev_t, start = reverse_table(knn, n)
        check(lib.apr_reverse_gather_range(ptr(contrib), c, ptr(rev_t), ptr(start), n, 0, n * k, 1, ptr(dx), c, stream()))
        return dx, None


_GROUP_INDS = {}


def group_rows(n, k, device):
    """int32 [n, k] = the rows i*k + j of an [n*k, c] edge tensor (cached): max over a point's k edges is max_pool over it."""
    key = (n, k, device)
    t = _GROUP_INDS.get(key)
    if t is None:
        if len(_GROUP_INDS) > 8:
            _GROUP_INDS.clear()
        t = _GROUP_INDS[key] = torch.arange(n * k, dtype=torch.int32, device=device).view(n, k)
    return t


def group_max(y, n, k, scale, shift, slope):
    y, ldy = ops._rows(y, "group_max.y")
    c = y.shape[1]
    out = torch.empty((n, c), dtype=torch.float32, device=y.device)
    check(_lib.load().apr_group_max(ptr(y), ldy, n, k, c, ptr(scale), ptr(shift), float(slope), ptr(out), c, stream()))
    return out


def mha(q, k, v, heads):
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    n, c = q.shape
    out = torch.empty_like(q)
    check(_lib.load().apr_mha(ptr(q), ptr(k), ptr(v), n, k.shape[0], c // heads, heads, ptr(out), stream()))
    return out


class MHAFunction(torch.autograd.Function):
    """Multi-head softmax attention (gcn.py:94-116, channel c = d*heads + h) with forward and backward on the HIP kernels
    (apr_mha_train_forward / _backward): the forward keeps the probabilities, every sum has a fixed order."""

    @staticmethod
    def forward(ctx, q, k, v, heads):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        n, c = q.shape
        m, cv = k.shape[0], v.shape[1]          # v may be narrower than q / k (the cross saliency: one column)
        P = torch.empty((heads, n, m), dtype=torch.float32, device=q.device)
        out = torch.empty((n, cv), dtype=torch.float32, device=q.device)
        check(_lib.load().apr_mha_train_forward(ptr(q), ptr(k), ptr(v), n, m, c // heads, heads, cv // heads, ptr(P), ptr(out),
                                                stream()))
        ctx.save_for_backward(q, k, v, P)
        ctx.heads = heads
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, P = ctx.saved_tensors
        heads = ctx.heads
        n, c = q.shape
        m, cv = k.shape[0], v.shape[1]
        dout = dout.contiguous()
        dS = torch.empty_like(P)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        check(_lib.load().apr_mha_train_backward(ptr(q), ptr(k), ptr(v), ptr(P), ptr(dout), n, m, c // heads, heads, cv // heads,
                                                 ptr(dS), ptr(dq), ptr(dk), ptr(dv), stream()))
        return dq, dk, dv, None


def softmax_matvec_train(a, b, w, temperature):
    """softmax_j(<a_i, b_j> / temperature) @ w under autograd on the HIP attention kernels (one head, one value column):
    the scores are <q_i, b_j> / sqrt(c) with q = a * (sqrt(c) / temperature) -- that scaling is the only torch op, and it
    carries the gradient of the learned temperature."""
    c = a.shape[1]
    q = a * (math.sqrt(c) / temperature)
    return MHAFunction.apply(q, b, w.reshape(-1, 1), 1).reshape(-1)


HIP_TRAIN_MHA = os.environ.get("APR_HIP_TRAIN_MHA", "1") != "0"     # A/B switch: 0 = the attention of the training path on torch ops
MHA_MFMA = os.environ.get("APR_MHA", "mfma") != "scalar"      # A/B switch: the scalar kernel on interleaved channels


def mha_headmajor(q, k, v, heads):
    """`mha` with q / k / v head-major (channel h*dim + d); the result is in `mha`'s interleaved layout."""
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    n, c = q.shape
    out = torch.empty_like(q)
    check(_lib.load().apr_mha_headmajor(ptr(q), ptr(k), ptr(v), n, k.shape[0], c // heads, heads, ptr(out), stream()))
    return out


def softmax_matvec(a, b, w, temperature):
    a, w = a.contiguous(), w.contiguous().view(-1)
    out = torch.empty(a.shape[0], dtype=torch.float32, device=a.device)
    n, m, c = a.shape[0], b.shape[0], a.shape[1]
    if c in (32, 64, 128, 256) and a.data_ptr() % 16 == 0:
        b = b.contiguous()
        if b.data_ptr() % 16 == 0:
            check(_lib.load().apr_softmax_matvec_mfma(ptr(a), ptr(b), ptr(w), n, m, c, float(temperature), ptr(out),
                                                      stream()))
            return out
    if 4 * (m + c) * 4 <= 60 * 1024:
        bt = b.t().contiguous()          # [c, m]: the kernel's lanes run along the keys
        check(_lib.load().apr_softmax_matvec_bt(ptr(a), ptr(bt), ptr(w), n, m, c, float(temperature), ptr(out), stream()))
    else:
        check(_lib.load().apr_softmax_matvec(ptr(a), ptr(b.contiguous()), ptr(w), n, m, c, float(temperature), ptr(out),
                                             stream()))
    return out


def score_head(x_col):
    """x_col: a [N] column view (stride allowed) -> clamp(sigmoid(x)) with NaN/Inf -> 0."""
    n = x_col.shape[0]
    out = torch.empty(n, dtype=torch.float32, device=x_col.device)
    check(_lib.load().apr_score_head(ptr(x_col), x_col.stride(0), n, ptr(out), stream()))
    return out


class NormActFunction(torch.autograd.Function):
    """act(instance_norm(x) (+ residual)) -- KPFCNN's BatchNormBlock + LeakyReLU (+ shortcut), blocks.py:459-468, 489, 574,
    653-681 -- with forward AND backward on the HIP kernels: the forward is the fused inference launch pair
    (apr_instance_norm_act[_seg]); the backward turns dy into dz through the activation's own output (apr_act_backward), hands
    dz to the residual branch as it is, and runs apr_norm_backward per pair segment with the segment's statistics
    (apr_bn_stats, recomputed: nothing but x and the output is kept).  No torch op between them."""

    @staticmethod
    def forward(ctx, x, residual, eps, leaky, relu, segments):
        x = x.contiguous()
        res = residual.contiguous() if residual is not None else None
        with torch.no_grad():
            y = instance_norm_act(x.detach(), eps=eps, leaky=leaky, relu=relu,
                                  residual=None if res is None else res.detach(), segments=segments)
        ctx.save_for_backward(x, y)
        ctx.cfg = (float(eps), leaky, bool(relu), None if segments is None else list(segments), res is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        eps, leaky, relu, segments, has_res = ctx.cfg
        dy = dy.contiguous()
        n, c = x.shape
        mode = 2 if leaky is not None else int(relu)
        dz = dy
        if mode:
            dz = torch.empty_like(dy)
            check(_lib.load().apr_act_backward(ptr(dy), c, ptr(y), c, n, c, mode, float(leaky or 0.0), ptr(dz), c, stream()))
        dx = None
        if ctx.needs_input_grad[0]:
            segs = segments if (segments is not None and len(segments) > 2) else [0, n]
            dx = torch.empty_like(x)
            for a, b in zip(segs[:-1], segs[1:]):
                mean, var = ops.bn_stats(x[a:b])
                rstd = torch.rsqrt(var + eps)
                ops.norm_backward(x[a:b], dz[a:b], mean, rstd, None, want_affine_grads=False, out=dx[a:b])
        return dx, (dz if has_res and ctx.needs_input_grad[1] else None), None, None, None, None


HIP_TRAIN_NORMACT = os.environ.get("APR_HIP_TRAIN_NORMACT", "1") != "0"   # A/B switch: 0 = norm on HIP, activation / residual add in torch


def instance_norm_act(x, eps=1e-5, leaky=None, relu=False, residual=None, out=None, segments=None):
    """Per-channel normalisation over all rows, no affine (InstanceNorm1d on [1,C,N]) + activation.
    `segments`: row offsets [0, ..., n] of the scan pairs stacked in x (one statistic per pair and channel)."""
    if segments is not None and len(segments) <= 2:
        segments = None
    if tracking(x, residual):
        if HIP_TRAIN_NORM and HIP_TRAIN_NORMACT and x.dim() == 2 and x.shape[0] > 0:
            # training: the fused forward launches, and a backward of apr_act_backward + apr_norm_backward (no torch op)
            return NormActFunction.apply(x, residual, eps, leaky, relu, segments)
        # statistics, normalisation and its backward on the HIP kernels (ops.NormFunction, no affine); the
        # activation / residual add as elementwise torch ops
        nf = lambda t: ops.NormFunction.apply(t, None, None, eps)[0] if HIP_TRAIN_NORM else instance_norm_rows(t, eps)
        if segments is None:
            y = nf(x)
        else:
            y = torch.cat([nf(x[a:b]) for a, b in zip(segments[:-1], segments[1:])], 0)
        return _act(y if residual is None else y + residual, leaky, relu)
    if segments is not None:
        x, ldx = ops._rows(x, "instance_norm_act.x")
        n, c = x.shape
        if out is None:
            out = torch.empty((n, c), dtype=torch.float32, device=x.device)
        out, ldy = ops._rows(out, "instance_norm_act.out")
        ldr = 0
        if residual is not None:
            residual, ldr = ops._rows(residual, "instance_norm_act.residual")
        lib = _lib.load()
        nseg = len(segments) - 1
        sb = int(lib.apr_bn_stats_scratch_bytes(n + 256 * nseg, c))
        scratch = torch.empty(sb, dtype=torch.uint8, device=x.device)
        offs = (C.c_int64 * (nseg + 1))(*[int(v) for v in segments])
        mode = 2 if leaky is not None else int(bool(relu))
        check(lib.apr_instance_norm_act_seg(ptr(x), ldx, n, c, float(eps), ptr(residual), ldr, mode, float(leaky or 0.0),
                                            ptr(out), ldy, offs, nseg, ptr(scratch), sb, stream()))
        return out
    x, ldx = ops._rows(x, "instance_norm_act.x")
    n, c = x.shape
    if out is None:
        out = torch.empty((n, c), dtype=torch.float32, device=x.device)
    out, ldy = ops._rows(out, "instance_norm_act.out")
    ldr = 0
    if residual is not None:
        residual, ldr = ops._rows(residual, "instance_norm_act.residual")
    lib = _lib.load()
    sb = int(lib.apr_bn_stats_scratch_bytes(n, c))
    scratch = torch.empty(sb, dtype=torch.uint8, device=x.device)
    mode = 2 if leaky is not None else int(bool(relu))
    check(lib.apr_instance_norm_act(ptr(x), ldx, n, c, float(eps), ptr(residual), ldr, mode, float(leaky or 0.0),
                                    ptr(out), ldy, ptr(scratch), sb, stream()))
    return out
