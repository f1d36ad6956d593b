"""Backward passes: sparse conv / transposed conv gradients on the HIP kernels, the encoder in
training mode and the hardest-contrastive loss, against autograd through the CPU oracle (torch ops)."""
import numpy as np
import pytest
import torch

from apr_amd import MinkowskiEngine as ME
from apr_amd import ops, synth
from apr_amd.fcgf.lib.trainer import HardestContrastiveLoss
from oracle import match_pose_oracle as MO
from oracle import me_oracle as OME
from tests.helpers import model_pair, rel_l2

pytestmark = pytest.mark.gpu


def _ref_conv(x, nbr, W):
    """Differentiable float64 reference: out[j] = sum_k x[nbr[j,k]] @ W[k] over the non-empty entries."""
    out = torch.zeros(nbr.shape[0], W.shape[2], dtype=torch.float64)
    for k in range(nbr.shape[1]):
        j = np.nonzero(nbr[:, k] >= 0)[0]
        if len(j):
            out = out.index_add(0, torch.from_numpy(j), x[torch.from_numpy(nbr[j, k].astype(np.int64))] @ W[k])
    return out


@pytest.mark.parametrize("cin,cout,K,n_in,n_out", [(32, 32, 27, 900, 900), (64, 128, 27, 700, 250), (128, 64, 27, 200, 900),
                                                    (1, 32, 125, 1500, 1500), (96, 64, 27, 513, 1031), (3, 5, 27, 100, 77)])
def test_wgrad_kernel_matches_autograd(dev, cin, cout, K, n_in, n_out):
    rng = np.random.default_rng(cin + cout + K)
    x = torch.from_numpy(rng.standard_normal((n_in, cin)).astype(np.float32))
    g = torch.from_numpy(rng.standard_normal((n_out, cout)).astype(np.float32))
    nbr = rng.integers(0, n_in, size=(n_out, K)).astype(np.int32)
    nbr[rng.random((n_out, K)) > 0.3] = -1
    W = torch.zeros(K, cin, cout, dtype=torch.float64, requires_grad=True)
    (_ref_conv(x.double(), nbr, W) * g.double()).sum().backward()
    dw = ops.spconv_wgrad(x.to(dev), g.to(dev), torch.from_numpy(nbr).to(dev), K, cin, cout)
    assert rel_l2(dw.cpu(), W.grad) < 2e-6
    again = ops.spconv_wgrad(x.to(dev), g.to(dev), torch.from_numpy(nbr).to(dev), K, cin, cout)
    assert torch.equal(dw, again)                      # deterministic
    # the same-level entry point (centre offset spread over more workgroups): same sums, another grouping of the partials
    same = ops.spconv_wgrad(x.to(dev), g.to(dev), torch.from_numpy(nbr).to(dev), K, cin, cout, same_level=True)
    assert rel_l2(same.cpu(), W.grad) < 2e-6
    assert torch.equal(same, ops.spconv_wgrad(x.to(dev), g.to(dev), torch.from_numpy(nbr).to(dev), K, cin, cout, same_level=True))


def test_wgrad_identity_map_and_many_row_blocks(dev):
    """K = 1 with nbr = None (Linear layers, 1x1 convolutions) and a map long enough for several 512-row blocks per work
    item of the bf16-split kernel, centre column full as on a real stride-1 map."""
    rng = np.random.default_rng(11)
    n, cin, cout = 9000, 160, 128
    x = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32)).to(dev)
    g = torch.from_numpy(rng.standard_normal((n, cout)).astype(np.float32)).to(dev)
    dw = ops.spconv_wgrad(x, g, None, 1, cin, cout)
    ref = x.double().t() @ g.double()
    assert rel_l2(dw[0].cpu(), ref.cpu()) < 2e-6
    K, cin, cout = 27, 64, 64
    nbr = rng.integers(0, n, size=(n, K)).astype(np.int32)
    nbr[rng.random((n, K)) > 0.25] = -1
    nbr[:, 13] = np.arange(n)
    x = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32)).to(dev)
    g = torch.from_numpy(rng.standard_normal((n, cout)).astype(np.float32)).to(dev)
    nb = torch.from_numpy(nbr).to(dev)
    dw = ops.spconv_wgrad(x, g, nb, K, cin, cout, same_level=True)
    ref = torch.zeros(K, cin, cout, dtype=torch.float64, device=dev)
    for k in range(K):
        j = torch.nonzero(nb[:, k] >= 0).squeeze(1)
        ref[k] = x[nb[j, k].long()].double().t() @ g[j].double()
    assert rel_l2(dw.cpu(), ref.cpu()) < 2e-6


@pytest.mark.parametrize("kind", ["same", "strided", "transposed"])
def test_conv_module_gradients(dev, kind):
    """MinkowskiConvolution / Transpose under autograd: d input and d kernel vs autograd through the oracle conv."""
    rng = np.random.default_rng(3)
    xyz = rng.uniform(-6, 6, (3000, 3)).astype(np.float32)
    c, _ = OME.sparse_quantize(xyz / np.float32(0.3), return_index=True)
    C = OME.batched_coordinates([c])
    cin, cout = 32, 64
    F = rng.standard_normal((len(C), cin)).astype(np.float32)
    # oracle chain: optional stride-2 conv first so that a coarse level exists
    ox = OME.SparseTensor(torch.from_numpy(F).requires_grad_(True), coordinates=C)
    hx = ME.SparseTensor(torch.from_numpy(F).to(dev).requires_grad_(True), coordinates=torch.from_numpy(C).to(dev))
    Wd = torch.from_numpy((rng.standard_normal((27, cin, cin)) * 0.1).astype(np.float32))
    W = torch.from_numpy((rng.standard_normal((27, cin, cout)) * 0.1).astype(np.float32))
    down_h = ME.MinkowskiConvolution(cin, cin, kernel_size=3, stride=2, dimension=3).to(dev)
    with torch.no_grad():
        down_h.kernel.copy_(Wd)
    if kind == "same":
        conv_h = ME.MinkowskiConvolution(cin, cout, kernel_size=3, stride=1, dimension=3).to(dev)
        oy = OME.conv_forward(ox, W.clone().requires_grad_(True), 3, 1)
        Wo = oy  # placeholder, replaced below
    elif kind == "strided":
        conv_h = ME.MinkowskiConvolution(cin, cout, kernel_size=3, stride=2, dimension=3).to(dev)
    else:
        conv_h = ME.MinkowskiConvolutionTranspose(cin, cout, kernel_size=3, stride=2, dimension=3).to(dev)
    with torch.no_grad():
        conv_h.kernel.copy_(W)
    Wo = W.clone().requires_grad_(True)
    Wdo = Wd.clone().requires_grad_(True)
    if kind == "same":
        oy = OME.conv_forward(ox, Wo, 3, 1)
        hy = conv_h(hx)
    elif kind == "strided":
        oy = OME.conv_forward(ox, Wo, 3, 2)
        hy = conv_h(hx)
    else:
        oy = OME.conv_forward(OME.conv_forward(ox, Wdo, 3, 2), Wo, 3, 2, transpose=True)
        hy = conv_h(down_h(hx))
    proj = torch.from_numpy(rng.standard_normal(tuple(oy.F.shape)).astype(np.float32))
    assert rel_l2(hy.F.detach().cpu(), oy.F.detach()) < 1e-5
    (oy.F * proj).sum().backward()
    (hy.F * proj.to(dev)).sum().backward()
    assert rel_l2(conv_h.kernel.grad.cpu(), Wo.grad) < 1e-5
    assert rel_l2(hx.F.grad.cpu(), ox.F.grad) < 1e-5
    if kind == "transposed":
        assert rel_l2(down_h.kernel.grad.cpu(), Wdo.grad) < 1e-5


@pytest.mark.parametrize("routing", ["tile", "default"])
def test_encoder_training_step_gradients(dev, routing, monkeypatch):
    """ResUNetBN2C in train mode: forward, a scalar loss, backward — every parameter gradient and the BN running
    statistics against the oracle network (same state_dict) on the CPU.

    Round 5: train() now walks the fused nodes (ResUNet2.forward_train).  "tile": every convolution on the tile kernel, the
    summation order of rounds 1-4 -- the original 2e-3 bar (measured 1e-6).  "default": the inference routing
    (weight-stationary / triple-list kernels) -- another summation order, features still equal to 5e-7, but ONE output of
    the last residual block lands on the other side of its ReLU's kink than in the oracle (scripts/dbg_train_units.py:
    1 mask flip in 266 k entries; with the oracle's mask the same gradients agree to 3e-7).  A single flipped entry is
    1 / sqrt(N) = 2e-3 of that layer's gradient norm and rides upstream from there, so this leg's bar is 1e-2."""
    if routing == "tile":
        monkeypatch.setenv("APR_WS_STAGES", "none")
    om, hm = model_pair("ResUNetBN2C", 32)
    om.train(); hm.train()
    xyz, _, _ = synth.make_pair(5, n_beams=16, n_azimuth=700)
    c, _ = OME.sparse_quantize(xyz / np.float32(0.3), return_index=True)
    C = OME.batched_coordinates([c])
    F = np.ones((len(C), 1), np.float32)
    oy = om(OME.SparseTensor(torch.from_numpy(F), coordinates=C)).F
    hy = hm(ME.SparseTensor(torch.from_numpy(F).to(dev), coordinates=torch.from_numpy(C).to(dev))).F
    assert rel_l2(hy.detach().cpu(), oy.detach()) < 1e-4
    proj = torch.from_numpy(np.random.default_rng(0).standard_normal(tuple(oy.shape)).astype(np.float32))
    (oy * proj).sum().backward()
    (hy * proj.to(dev)).sum().backward()
    og = dict(om.named_parameters())
    worst = 0.0
    for name, p in hm.named_parameters():
        assert p.grad is not None, name
        worst = max(worst, rel_l2(p.grad.cpu(), og[name].grad))
    assert worst < (2e-3 if routing == "tile" else 1e-2), worst   # fp32 through 23 conv + 22 BN layers
    ob = dict(om.named_buffers())
    for name, b in hm.named_buffers():
        if name.endswith("running_mean") or name.endswith("running_var"):
            assert torch.allclose(b.cpu(), ob[name], rtol=1e-3, atol=1e-5), name
        if name.endswith("num_batches_tracked"):
            assert int(b) == int(ob[name]) == 1, name


def test_hardest_contrastive_loss_backward(dev):
    rng = np.random.default_rng(2)
    n0, n1, c = 3000, 2800, 32
    F0 = rng.standard_normal((n0, c)).astype(np.float32); F0 /= np.linalg.norm(F0, axis=1, keepdims=True)
    F1 = rng.standard_normal((n1, c)).astype(np.float32); F1 /= np.linalg.norm(F1, axis=1, keepdims=True)
    pairs = np.stack([rng.permutation(n0)[:1500], rng.permutation(n1)[:1500]], 1).astype(np.int64)
    F1[pairs[:, 1]] = F0[pairs[:, 0]] + 0.2 * rng.standard_normal((1500, c)).astype(np.float32)
    sel0 = rng.choice(n0, 1024, replace=False); sel1 = rng.choice(n1, 1024, replace=False)
    pos_sel = rng.choice(1500, 600, replace=False)
    a = torch.from_numpy(F0).requires_grad_(True); b = torch.from_numpy(F1).requires_grad_(True)
    pl_o, nl_o = MO.hardest_contrastive(a, b, pairs, sel0, sel1, pos_sel)
    (pl_o + nl_o).backward()
    ga = torch.from_numpy(F0).to(dev).requires_grad_(True); gb = torch.from_numpy(F1).to(dev).requires_grad_(True)
    crit = HardestContrastiveLoss()
    pl, nl = crit.contrastive_hardest_negative_loss(ga, gb, pairs, num_pos=600, num_hn_samples=1024,
                                                    draws=(sel0, sel1, pos_sel))
    assert abs(pl.item() - pl_o.item()) < 1e-5 and abs(nl.item() - nl_o.item()) < 1e-5
    (pl + nl).backward()
    assert rel_l2(ga.grad.cpu(), a.grad) < 1e-5 and rel_l2(gb.grad.cpu(), b.grad) < 1e-5


def test_two_sgd_steps_follow_the_oracle(dev):
    """The reference's training iteration (lib/trainer.py:454-527: encode both frames, hardest-contrastive loss,
    backward, optimiser step) run twice on the HIP path and on the CPU oracle from the same weights, pairs and
    random draws: losses and the updated weights stay together."""
    om, hm = model_pair("ResUNetBN2C", 32, seed=4)
    om.train(); hm.train()
    xyz0, xyz1, T = synth.make_pair(9, n_beams=16, n_azimuth=600)
    cs, pts = [], []
    for xyz in (xyz0, xyz1):
        c, sel = OME.sparse_quantize(xyz / np.float32(0.3), return_index=True)
        cs.append(c); pts.append(xyz[sel])
    # ground-truth positive pairs: voxels of frame 0 whose transformed point has a frame-1 voxel point within 0.3 m
    p0 = pts[0].astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    d = ((p0[:, None, :] - pts[1][None, :, :].astype(np.float64)) ** 2).sum(-1)
    j = d.argmin(1)
    keep = d[np.arange(len(j)), j] < 0.3 ** 2
    pairs = np.stack([np.nonzero(keep)[0], j[keep]], 1).astype(np.int64)
    assert len(pairs) > 200
    C0, C1 = OME.batched_coordinates([cs[0]]), OME.batched_coordinates([cs[1]])
    rng = np.random.default_rng(0)
    opt_o = torch.optim.SGD(om.parameters(), lr=0.05, momentum=0.8)
    opt_h = torch.optim.SGD(hm.parameters(), lr=0.05, momentum=0.8)
    crit = HardestContrastiveLoss()
    for step in range(2):
        sel0 = rng.choice(len(C0), 512, replace=False); sel1 = rng.choice(len(C1), 512, replace=False)
        pos_sel = rng.choice(len(pairs), 200, replace=False)
        opt_o.zero_grad(); opt_h.zero_grad()
        of0 = om(OME.SparseTensor(torch.ones(len(C0), 1), coordinates=C0)).F
        of1 = om(OME.SparseTensor(torch.ones(len(C1), 1), coordinates=C1)).F
        pl_o, nl_o = MO.hardest_contrastive(of0, of1, pairs, sel0, sel1, pos_sel)
        (pl_o + nl_o).backward(); opt_o.step()
        hf0 = hm(ME.SparseTensor(torch.ones(len(C0), 1, device=dev), coordinates=torch.from_numpy(C0).to(dev))).F
        hf1 = hm(ME.SparseTensor(torch.ones(len(C1), 1, device=dev), coordinates=torch.from_numpy(C1).to(dev))).F
        pl, nl = crit.contrastive_hardest_negative_loss(hf0, hf1, pairs, num_pos=200, num_hn_samples=512,
                                                        draws=(sel0, sel1, pos_sel))
        (pl + nl).backward(); opt_h.step()
        assert abs(pl.item() - pl_o.item()) < 2e-3 * max(1.0, abs(pl_o.item())), (step, pl.item(), pl_o.item())
        assert abs(nl.item() - nl_o.item()) < 2e-3 * max(1.0, abs(nl_o.item())), (step, nl.item(), nl_o.item())
    op = dict(om.named_parameters())
    for name, p in hm.named_parameters():
        assert rel_l2(p.detach().cpu(), op[name].detach()) < 1e-3, name


def test_fused_training_path_matches_module_path_and_is_reproducible(dev, monkeypatch):
    """Round 5: ResUNet2.forward_train (23 fused autograd nodes: routed conv -> apr_bn_train_fwd / _bwd, bf16-split wgrad)
    against the module-by-module autograd path on APR's own encoder (ResUNetFatBN, 128 features): features, every gradient,
    the running statistics; and the same bits from two runs of the fused path."""
    from apr_amd.fcgf.model import resunet as RU
    _, hm = model_pair("ResUNetFatBN", 128, seed=2)
    hm.train()
    state0 = {k: v.clone() for k, v in hm.state_dict().items()}
    xyz, _, _ = synth.make_pair(6, n_beams=32, n_azimuth=900)
    c, _ = OME.sparse_quantize(xyz / np.float32(0.3), return_index=True)
    C = torch.from_numpy(OME.batched_coordinates([c])).to(dev)
    proj = None
    runs = []
    for fused in (True, True, False):
        monkeypatch.setattr(RU, "TRAIN_FUSED", fused)
        hm.load_state_dict(state0)
        hm.zero_grad()
        y = hm(ME.SparseTensor(torch.ones(len(C), 1, device=dev), coordinates=C)).F
        if proj is None:
            proj = torch.from_numpy(np.random.default_rng(1).standard_normal(tuple(y.shape)).astype(np.float32)).to(dev)
        (y * proj).sum().backward()
        runs.append((y.detach().clone(), {n: p.grad.clone() for n, p in hm.named_parameters()},
                     {n: b.clone() for n, b in hm.named_buffers()}))
    (ya, ga, ba), (yb, gb, bb), (ym, gm, bm) = runs
    assert torch.equal(ya, yb) and all(torch.equal(ga[n], gb[n]) for n in ga) and all(torch.equal(ba[n], bb[n]) for n in ba)
    assert rel_l2(ya.cpu(), ym.cpu()) < 1e-4
    worst = max(rel_l2(ga[n].cpu(), gm[n].cpu()) for n in ga)
    assert worst < 1e-2, worst          # other summation order in the routed kernels: ReLU kink flips (see the test above)
    for n in ba:
        assert torch.allclose(ba[n].float(), bm[n].float(), rtol=1e-4, atol=1e-6), n


def test_stacked_frames_equal_separate_calls(dev):
    """ResUNet2.forward_frames: the two encoder calls of a training iteration in one walk of the network (per-call BatchNorm
    statistics through apr_bn_train_fwd's segments) against two separate calls: features, every gradient, and the running
    statistics / num_batches_tracked after both calls."""
    _, hm = model_pair("ResUNetFatBN", 128, seed=5)
    hm.train()
    state0 = {k: v.clone() for k, v in hm.state_dict().items()}
    Cs = []
    for seed, beams in ((3, 32), (4, 16)):
        xyz, _, _ = synth.make_pair(seed, n_beams=beams, n_azimuth=800)
        c, _ = OME.sparse_quantize(xyz / np.float32(0.3), return_index=True)
        Cs.append(torch.from_numpy(OME.batched_coordinates([c])).to(dev))
    projs, runs = None, []
    for stacked in (False, True, True):
        hm.load_state_dict(state0)
        hm.zero_grad()
        xs = [ME.SparseTensor(torch.ones(len(C), 1, device=dev), coordinates=C) for C in Cs]
        ys = hm.forward_frames(xs) if stacked else [hm(x) for x in xs]
        assert [tuple(y.F.shape) for y in ys] == [(len(C), 128) for C in Cs]
        if projs is None:
            g = torch.Generator().manual_seed(0)
            projs = [torch.randn(tuple(y.F.shape), generator=g).to(dev) for y in ys]
        sum((y.F * p).sum() for y, p in zip(ys, projs)).backward()
        runs.append(([y.F.detach().clone() for y in ys], {n: p.grad.clone() for n, p in hm.named_parameters()},
                     {n: b.clone() for n, b in hm.named_buffers()}))
    (ya, ga, ba), (yb, gb, bb), (yc, gc, bc) = runs
    assert all(torch.equal(u, v) for u, v in zip(yb, yc)) and all(torch.equal(gb[n], gc[n]) for n in gb)      # reproducible
    for u, v in zip(ya, yb):
        assert rel_l2(v.cpu(), u.cpu()) < 1e-5
    worst = max(rel_l2(gb[n].cpu(), ga[n].cpu()) for n in ga)
    assert worst < 1e-2, worst                       # other summation order: ReLU kink flips, see the routing test above
    for n in ba:
        if n.endswith("num_batches_tracked"):
            assert int(ba[n]) == int(bb[n]) == 2, n
        else:
            assert torch.allclose(ba[n], bb[n], rtol=1e-4, atol=1e-6), n


@pytest.mark.parametrize("K,cin,cout", [(27, 64, 128), (27, 128, 128), (27, 256, 64), (8, 128, 64), (1, 128, 64), (1, 64, 64)])
@pytest.mark.parametrize("flip", [False, True])
def test_input_gradient_kernel_image_packs_straight_from_the_parameter(K, cin, cout, flip):
    """apr_spconv_pack_weights_bf3_ex(flip, transposed) writes the same bytes as apr_weights_flip_transpose followed by the
    plain pack: the training step's re-pack after every optimizer step needs no [K, cout, cin] copy."""
    from apr_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(K * 1000 + cin + cout)
    w = torch.randn(K, cin, cout, generator=g).to(dev)
    direct = ops.pack_weights_bf3(w, flip=flip, transposed=True)
    two_step = ops.pack_weights_bf3(ops.weights_flip_transpose(w, flip))
    assert direct is not None and two_step is not None
    assert torch.equal(direct, two_step)
