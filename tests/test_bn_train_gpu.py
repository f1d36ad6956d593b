"""apr_bn_train_fwd / apr_bn_train_bwd (training-mode BatchNorm fused with residual add and ReLU, per row segment: the unit of
ResUNet2.forward_train) against torch.nn.BatchNorm1d + autograd in float64: value, running statistics, every gradient; odd
widths (no 16-byte rows), ragged segments, tiny inputs."""
import numpy as np
import pytest
import torch

from apr_amd import ops
from tests.helpers import rel_l2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,c,segs,relu,with_res", [
    (5000, 64, None, True, True), (3001, 34, None, False, False), (2, 32, None, True, False), (777, 128, [0, 300, 777], True, True),
    (9000, 5, [0, 2, 4000, 9000], True, True), (14000, 256, [0, 7100, 14000], False, True), (257, 1, None, True, False)])
def test_bn_train_forward_backward_match_torch(dev, n, c, segs, relu, with_res):
    rng = np.random.default_rng(n * 7 + c)
    z0 = torch.from_numpy((rng.standard_normal((n, c)) * rng.uniform(0.5, 3, c) + rng.uniform(-2, 2, c)).astype(np.float32))
    r0 = torch.from_numpy(rng.standard_normal((n, c)).astype(np.float32)) if with_res else None
    gy = torch.from_numpy(rng.standard_normal((n, c)).astype(np.float32))
    bn = torch.nn.BatchNorm1d(c, momentum=0.05).to(dev)
    with torch.no_grad():
        bn.weight.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32)))
        bn.bias.copy_(torch.from_numpy(rng.uniform(-0.5, 0.5, c).astype(np.float32)))
        bn.running_mean.copy_(torch.from_numpy(rng.standard_normal(c).astype(np.float32)))
        bn.running_var.copy_(torch.from_numpy(rng.uniform(0.5, 2, c).astype(np.float32)))
    ref = torch.nn.BatchNorm1d(c, momentum=0.05).double()
    ref.load_state_dict({k: v.detach().cpu().double() if v.dtype.is_floating_point else v.cpu() for k, v in bn.state_dict().items()})
    ref.train()
    # reference: one BatchNorm1d call per segment, in order (what separate forward calls of the trainer do)
    zr = z0.double().requires_grad_(True)
    rr = r0.double().requires_grad_(True) if with_res else None
    bounds = segs or [0, n]
    parts = [ref(zr[a:b]) for a, b in zip(bounds[:-1], bounds[1:])]
    yr = torch.cat(parts, 0)
    if with_res:
        yr = yr + rr
    if relu:
        yr = torch.relu(yr)
    (yr * gy.double()).sum().backward()
    # HIP
    z = z0.to(dev)
    res = r0.to(dev) if with_res else None
    y, mean, rstd = ops.bn_train_fwd(z, bn, residual=res, relu=relu, segments=segs)
    assert rel_l2(y.cpu(), yr.detach()) < 2e-6
    assert torch.allclose(bn.running_mean.cpu().double(), ref.running_mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(bn.running_var.cpu().double(), ref.running_var, rtol=1e-5, atol=1e-6)
    assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked) == len(bounds) - 1
    dz, dres, dg, db = ops.bn_train_bwd(z, y, gy.to(dev), mean, rstd, bn.weight.detach(), relu, with_res, segments=segs)
    tol = 5e-5 if n > 2 else 5e-3          # two rows: xhat = +-1 exactly, the gradient is a difference of equal terms
    assert rel_l2(dz.cpu(), zr.grad) < tol or float(zr.grad.norm()) < 1e-6
    assert rel_l2(dg.cpu(), ref.weight.grad) < 1e-5 and rel_l2(db.cpu(), ref.bias.grad) < 1e-5
    if with_res:
        assert rel_l2(dres.cpu(), rr.grad) < 1e-6
    # the same bits from a second run
    bn2_state = {k: v.clone() for k, v in bn.state_dict().items()}
    y2, _, _ = ops.bn_train_fwd(z, bn, residual=res, relu=relu, segments=segs)
    dz2, _, dg2, db2 = ops.bn_train_bwd(z, y2, gy.to(dev), mean, rstd, bn.weight.detach(), relu, with_res, segments=segs)
    assert torch.equal(y, y2) and torch.equal(dz, dz2) and torch.equal(dg, dg2) and torch.equal(db, db2)
    del bn2_state
