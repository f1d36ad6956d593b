"""APG aggregation, GT correspondences, NPR MLP + Chamfer vs CPU restatements."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from apr_amd import synth
from apr_amd.fcgf.lib import apg
from oracle import me_oracle as OME

pytestmark = pytest.mark.gpu


def test_apg_aggregation_matches_numpy(dev):
    rng = np.random.default_rng(0)
    key = synth.make_small_frame(0)
    frames, poses = [], []
    for k in range(4):
        f = synth.make_small_frame(10 + k)
        a = np.deg2rad(rng.uniform(-10, 10))
        M = np.eye(4); M[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
        M[:3, 3] = [6.0 * (k - 1.5), rng.uniform(-1, 1), 0.0]
        frames.append(f); poses.append(M)
    # numpy restatement of complement_data_loader.py:65-70, 576-579, 620-628, 671-674
    moved = [f @ M.astype(np.float32)[:3, :3].T + M.astype(np.float32)[:3, 3] for f, M in zip(frames, poses)]
    cat = np.concatenate(moved, 0)
    lim = np.max((key ** 2).sum(-1))
    ref = cat[np.where((cat ** 2).sum(-1) < lim)[0]]
    _, sel_ref = OME.sparse_quantize(ref / np.float32(0.3), return_index=True)
    nghb, sel = apg.aggregate_frames(key, frames, poses, 0.3)
    got = nghb.cpu().numpy()
    assert abs(len(got) - len(ref)) <= 3                      # fp32 rounding at the crop boundary
    if len(got) == len(ref):
        assert np.allclose(got, ref, atol=1e-4)
        assert (np.isin(sel.cpu().numpy(), sel_ref).mean() > 0.999)


def test_get_matching_indices_matches_bruteforce(dev):
    a, b, T = synth.make_pair(2, n_beams=8, n_azimuth=400)
    a, b = a[::2], b[::2]
    pairs = apg.get_matching_indices(a, b, T, 0.45).cpu().numpy()
    src = (a @ T[:3, :3].T.astype(np.float32) + T[:3, 3].astype(np.float32)).astype(np.float32)
    d2 = ((src[:, None, :] - b[None]) ** 2).sum(-1)
    ref = np.argwhere(d2 < 0.45 ** 2)
    assert abs(len(pairs) - len(ref)) <= 4
    assert len(set(map(tuple, pairs)) ^ set(map(tuple, ref))) <= 8
    p1 = apg.get_matching_indices(a, b, T, 0.45, K=1).cpu().numpy()
    assert len(np.unique(p1[:, 0])) == len(p1)                 # at most one (the nearest) match per source point


def test_chamfer_matches_bruteforce(dev):
    rng = np.random.default_rng(1)
    a = rng.uniform(-5, 5, (3000, 3)).astype(np.float32)
    b = rng.uniform(-5, 5, (4500, 3)).astype(np.float32)
    d2 = ((a[:, None].astype(np.float64) - b[None]) ** 2).sum(-1)
    ref = d2.min(1).sum() / len(a) + d2.min(0).sum() / len(b)
    got = float(apg.chamfer_distance(a, b))
    assert abs(got - ref) < 1e-5 * ref


@pytest.mark.parametrize("train", [False, True])
def test_generative_mlp_matches_torch(dev, train):
    torch.manual_seed(0)
    m = apg.GenerativeMLP_98(in_channel=32, out_points=4)
    ref = GenerativeMLPRef(32, 4)
    ref.load_state_dict(m.state_dict())
    with torch.no_grad():
        for mod in ref.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.normal_(0, 0.1); mod.running_var.uniform_(0.5, 1.5)
    m.load_state_dict(ref.state_dict())
    m.train(train); ref.train(train)
    x = torch.randn(2000, 32)
    with torch.no_grad():
        y_ref = ref(x)
    y = m.to(dev)(x.to(dev))
    assert torch.allclose(y.cpu(), y_ref, rtol=1e-4, atol=1e-5)
    if train:
        assert torch.allclose(m.mlp[2].running_mean.cpu(), ref.mlp[2].running_mean, atol=1e-5)


class GenerativeMLPRef(torch.nn.Module):
    """torch-CPU twin with the reference layout (FCGF_APR/model/mlp.py:6-37, GenerativeMLP_98)."""

    def __init__(self, cin, out_points):
        super().__init__()
        nn = torch.nn
        self.mlp = nn.Sequential(nn.Linear(cin, 512), nn.ReLU(), nn.BatchNorm1d(512, momentum=0.1),
                                 nn.Linear(512, 256), nn.ReLU(), nn.BatchNorm1d(256, momentum=0.1),
                                 nn.Linear(256, out_points * 3), nn.ReLU())

    def forward(self, x):
        return self.mlp(x)


def test_npr_loss_matches_restatement(dev):
    torch.manual_seed(1)
    gen = apg.GenerativeMLP_98(in_channel=32, out_points=4).eval()
    feats = torch.randn(1500, 32)
    coords = torch.randint(-50, 50, (1500, 3))
    nghb = torch.randn(6000, 3) * 8
    with torch.no_grad():
        g = gen.mlp(feats) * 0.3
        reg = (g.reshape(-1, 3) ** 2).sum(-1).mean()
        mod = (g + 0.3 * coords.float().repeat(1, 4)).reshape(-1, 3)
        d2 = torch.cdist(mod.double(), nghb.double()) ** 2
        ref = d2.min(1)[0].sum() / len(mod) + d2.min(0)[0].sum() / len(nghb) + reg * 0.01
    got = apg.npr_reconstruction_loss(gen.to(dev), feats.to(dev), coords.to(dev), nghb.to(dev), 0.3, 4)
    assert abs(float(got) - float(ref)) < 1e-4 * float(ref)


def test_grid_nn3_is_bit_identical_to_brute_force(dev):
    """apr_nn3 with a cell (two grids + fall-through passes) against cell = 0 (brute force): the same packed (d^2, index)
    for queries that resolve in the fine grid, in the coarse grid, and only in the full search; ties go to the smaller
    index on both paths."""
    from apr_amd import npr
    rng = np.random.default_rng(21)
    b = rng.uniform(-40, 40, (30000, 3)).astype(np.float32)
    b[:, 2] = 0.05 * rng.standard_normal(30000).astype(np.float32)              # a ground-like sheet: ~0.5 m spacing
    b[100] = b[7]                                                                # an exact duplicate: tie -> index 7
    near = b[rng.integers(0, len(b), 20000)] + rng.normal(0, 0.15, (20000, 3)).astype(np.float32)
    mid = b[rng.integers(0, len(b), 3000)] + np.array([0, 0, 2.5], np.float32)   # 2.5 m above the sheet: coarse grid
    far = rng.uniform(-300, 300, (500, 3)).astype(np.float32)                    # tens of metres away: full search
    a = np.concatenate([near, mid, far, b[7:8]]).astype(np.float32)
    A, B = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    i0, d0, s0 = npr.nn3(A, B, cell=0.0)
    for cell in (2.4, 0.6, 0.25, 8.0):
        i1, d1, s1 = npr.nn3(A, B, cell=cell)
        assert torch.equal(i0, i1) and torch.equal(d0, d1) and float(s0) == float(s1), cell
    assert int(i0[-1]) == 7
    # against a float64 brute force: same neighbours up to exact-distance ties
    d = torch.cdist(A[:5000].double(), B.double())
    assert float((d.argmin(1) != i0[:5000]).float().mean()) < 1e-3
    assert torch.allclose(d.min(1)[0].float() ** 2, d0[:5000], rtol=1e-4, atol=1e-6)
    # the other direction of the Chamfer term (few targets, many queries) and a single-point target cloud
    i2, d2, _ = npr.nn3(B, A[:3000])
    i3, d3, _ = npr.nn3(B, A[:3000], cell=0.0)
    assert torch.equal(i2, i3) and torch.equal(d2, d3)
    i4, _, _ = npr.nn3(A[:100], B[:1])
    assert int(i4.max()) == 0


def test_batched_chamfer_equals_cloud_by_cloud(dev):
    """npr.chamfer_distance_batch (apr_nn3_batch: the clouds of a batch in two searches) against chamfer_distance per cloud:
    the same values (same bits per cloud from the searches), the same gradients; clouds of very different sizes and one far
    away from its partner (the full-search fall-through inside a batch)."""
    from apr_amd import npr
    rng = np.random.default_rng(33)
    sizes = [(3000, 4000), (50, 7000), (6000, 300), (2500, 2500)]
    As, Bs = [], []
    for k, (n, m) in enumerate(sizes):
        b = rng.uniform(-30, 30, (m, 3)).astype(np.float32)
        b[:, 2] *= 0.05
        a = b[rng.integers(0, m, n)] + rng.normal(0, 0.3, (n, 3)).astype(np.float32)
        if k == 1:
            a += np.array([0, 0, 40], np.float32)              # every query tens of metres from its cloud
        As.append(torch.from_numpy(a).to(dev))
        Bs.append(torch.from_numpy(b).to(dev))
    ao = np.concatenate([[0], np.cumsum([len(a) for a in As])]).tolist()
    bo = np.concatenate([[0], np.cumsum([len(b) for b in Bs])]).tolist()
    A = torch.cat(As).requires_grad_(True)
    B = torch.cat(Bs).requires_grad_(True)
    vals = npr.chamfer_distance_batch(A, ao, B, bo)
    w = torch.tensor([1.0, 0.5, 2.0, -1.0], device=dev)
    (vals * w).sum().backward()
    for k in range(4):
        a = As[k].clone().requires_grad_(True)
        b = Bs[k].clone().requires_grad_(True)
        v = npr.chamfer_distance(a, b)
        (v * w[k]).backward()
        assert abs(float(v) - float(vals[k])) <= 1e-6 * abs(float(v)), k
        assert torch.allclose(A.grad[ao[k]:ao[k + 1]], a.grad, rtol=1e-5, atol=1e-7), k
        assert torch.allclose(B.grad[bo[k]:bo[k + 1]], b.grad, rtol=1e-5, atol=1e-7), k
        i1, d1, _ = npr.nn3(As[k], Bs[k])
        ib, db, _ = npr.nn3_batch(torch.cat(As), ao, torch.cat(Bs), bo)
        assert torch.equal(ib[ao[k]:ao[k + 1]] - bo[k], i1) and torch.equal(db[ao[k]:ao[k + 1]], d1), k


def test_far_queries_of_a_batch_prune_by_cell_and_keep_the_tie_rule(dev):
    """Queries tens of metres from their partner cloud (a complement cloud's far points against the generated points) take the
    cell-pruned search of k_nn3_grid_wave: same indices and distances as the brute-force search of each cloud, and of two
    targets at EXACTLY the same distance in different cells the smaller index wins."""
    from apr_amd import npr
    rng = np.random.default_rng(5)
    clouds = []
    for k in range(3):
        b = rng.uniform(-20, 20, (5000, 3)).astype(np.float32)
        b[:, 2] *= 0.1
        a = rng.uniform(-150, 150, (3000, 3)).astype(np.float32)          # most of them far outside the target's extent
        a[:, 2] = rng.uniform(-3, 60, 3000)
        if k == 1:                                                          # exact ties across cells
            b[100], b[4000] = (7.0, 0.0, 3.0), (-7.0, 0.0, 3.0)             # above every other target (|z| <= 2)
            a[0] = (0.0, 0.0, 90.0)
            a[1] = (0.0, -2.0, 55.0)
        clouds.append((torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)))
    ao = np.concatenate([[0], np.cumsum([len(a) for a, _ in clouds])]).tolist()
    bo = np.concatenate([[0], np.cumsum([len(b) for _, b in clouds])]).tolist()
    A, B = torch.cat([a for a, _ in clouds]), torch.cat([b for _, b in clouds])
    ib, db, _ = npr.nn3_batch(A, ao, B, bo)
    for k, (a, b) in enumerate(clouds):
        i0, d0, _ = npr.nn3(a, b, cell=0.0)                                 # brute force
        assert torch.equal(ib[ao[k]:ao[k + 1]] - bo[k], i0), k
        assert torch.equal(db[ao[k]:ao[k + 1]], d0), k
    assert int(ib[ao[1]] - bo[1]) == 100 and int(ib[ao[1] + 1] - bo[1]) == 100
