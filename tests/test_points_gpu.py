"""Grid subsampling / radius neighbours / kNN: HIP vs the REFERENCE's own C++ (oracle/_ref), SURVEY 8(a) P1-P2."""
import numpy as np
import pytest
import torch

from apr_amd import synth
from apr_amd.predator import point_ops
from oracle import predator_points_oracle as REF

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not REF.available(), reason="oracle/_ref not built")]


def _pair_clouds(seed, voxel=0.3):
    """Two clouds already barycentre-subsampled at 0.3 m, like the reference's dataset does before collate."""
    a, b, _ = synth.make_pair(seed, n_beams=16, n_azimuth=1250)
    pts = np.concatenate([a, b]).astype(np.float32)
    lens = np.array([len(a), len(b)], np.int32)
    return REF.subsample_batch(pts, lens, sampleDl=voxel)


@pytest.mark.parametrize("seed,dl", [(0, 0.3), (1, 0.6), (2, 1.2), (3, 2.4)])
def test_grid_subsample_bit_exact_up_to_row_order(dev, seed, dl):
    a, b, _ = synth.make_pair(seed, n_beams=16, n_azimuth=1250)
    pts = np.concatenate([a, b]).astype(np.float32)
    lens = np.array([len(a), len(b)], np.int32)
    rp, rl = REF.subsample_batch(pts, lens, sampleDl=dl)
    gp, gl = point_ops.grid_subsample(torch.from_numpy(pts).to(dev), lens, dl)
    assert np.array_equal(gl, rl)
    assert np.array_equal(REF.canonical_rows(gp.cpu().numpy(), gl).view(np.uint32),
                          REF.canonical_rows(rp, rl).view(np.uint32))


def test_grid_subsample_cells_beyond_the_wave_sort(dev):
    """An object right at the sensor puts thousands of raw points into one 0.3 m cell: cells with more than 1024 points
    go through the workgroup-wide bitonic sort (LDS up to 8192 points, global memory beyond) and must still give the
    reference's sequential fp32 sum, in bounded time (a 20 k-point cell used to take a second)."""
    import time
    rng = np.random.default_rng(17)
    a, b, _ = synth.make_pair(3, n_beams=16, n_azimuth=1250)
    blobs = [rng.uniform(0.02, 0.28, (m, 3)).astype(np.float32) + np.float32(o)
             for m, o in ((1500, 0.0), (9000, 3.0), (20000, -6.0), (1025, 9.0))]
    cloud0 = np.concatenate([a] + blobs[:3]).astype(np.float32)
    cloud0 = cloud0[rng.permutation(len(cloud0))]          # the cell's points are scattered over the index range
    cloud1 = np.concatenate([b, blobs[3], blobs[2][:5000] + np.float32(1.2)]).astype(np.float32)
    pts = np.concatenate([cloud0, cloud1])
    lens = np.array([len(cloud0), len(cloud1)], np.int32)
    rp, rl = REF.subsample_batch(pts, lens, sampleDl=0.3)
    tp = torch.from_numpy(pts).to(dev)
    tf = tp.clone()                     # features = the points: same cells, same order, mean by division
    point_ops.grid_subsample(tp, lens, 0.3, tf)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gp, gl, gf = point_ops.grid_subsample(tp, lens, 0.3, tf)
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 0.5      # the regression this guards against took 1 s; 10x headroom for a busy box
    assert np.array_equal(gl, rl)
    assert np.array_equal(REF.canonical_rows(gp.cpu().numpy(), gl).view(np.uint32),
                          REF.canonical_rows(rp, rl).view(np.uint32))
    assert torch.allclose(gf, gp, rtol=1e-6, atol=1e-6)


def test_subsample_batch_reference_api(dev):
    from apr_amd.predator.cpp_wrappers.cpp_subsampling import grid_subsampling as G
    a = synth.make_small_frame(5)
    lens = np.array([len(a) // 2, len(a) - len(a) // 2], np.int32)
    sp, sl = G.subsample_batch(a, lens, sampleDl=0.6, max_p=0, verbose=0)
    rp, rl = REF.subsample_batch(a, lens, sampleDl=0.6)
    assert sp.dtype == np.float32 and sl.dtype == np.int32 and np.array_equal(sl, rl)
    assert np.array_equal(REF.canonical_rows(sp, sl), REF.canonical_rows(rp, rl))
    feats = np.random.default_rng(0).standard_normal((len(a), 3)).astype(np.float32)
    sp2, sl2, sf2 = G.subsample_batch(a, lens, features=feats, sampleDl=0.6)
    assert sf2.shape == (len(sp2), 3) and np.array_equal(sp2, sp)
    with pytest.raises(RuntimeError):
        G.subsample_batch(a[:, :2], lens, sampleDl=0.6)


def _check_neighbors(q, s, qb, sb, r, got):
    ref = REF.batch_query(q, s, qb, sb, radius=r)
    assert got.shape == ref.shape
    ns = len(s)
    sp = np.concatenate([s, np.full((1, 3), 1e6, np.float32)])
    d_ref = ((q[:, None, :] - sp[ref]) ** 2).sum(-1)
    d_got = ((q[:, None, :] - sp[got]) ** 2).sum(-1)
    # identical neighbour sets per query, identical sorted distances; order may differ only inside ties
    assert np.array_equal(np.sort(ref, axis=1), np.sort(got, axis=1))
    assert np.array_equal((ref == ns).sum(1), (got == ns).sum(1))
    assert np.allclose(d_ref, d_got, rtol=0, atol=0) or np.array_equal(np.sort(d_ref, 1), np.sort(d_got, 1))
    valid = got != ns
    assert (np.diff(np.where(valid, d_got, np.float32(3e38)), axis=1) >= -1e-6).all()   # numpy re-rounds d2
    mism = ref != got          # positions may differ only inside runs of equal distance (ties)
    assert np.allclose(d_ref[mism], d_got[mism], rtol=1e-5, atol=1e-7)
    assert (~mism).mean() > 0.99


@pytest.mark.parametrize("seed", [0, 1])
def test_radius_neighbors_match_reference(dev, seed):
    p0, l0 = _pair_clouds(seed)
    p1, l1 = REF.subsample_batch(p0, l0, sampleDl=0.6)
    r = 0.3 * 4.25
    tq, ts = torch.from_numpy(p0).to(dev), torch.from_numpy(p1).to(dev)
    # conv (P,P,r), pool (coarse queries, fine supports, r), upsample (fine queries, coarse supports, 2r)
    _check_neighbors(p0, p0, l0, l0, r, point_ops.radius_neighbors(tq, tq, l0, l0, r).cpu().numpy())
    _check_neighbors(p1, p0, l1, l0, r, point_ops.radius_neighbors(ts, tq, l1, l0, r).cpu().numpy())
    _check_neighbors(p0, p1, l0, l1, 2 * r, point_ops.radius_neighbors(tq, ts, l0, l1, 2 * r).cpu().numpy())
    # the dataloader's truncation to the calibrated limit (dataloader.py:66-68)
    lim = 20
    ref = REF.batch_query(p0, p0, l0, l0, radius=r)[:, :lim]
    got = point_ops.radius_neighbors(tq, tq, l0, l0, r, limit=lim).cpu().numpy()
    assert got.shape == ref.shape and (got == ref).mean() > 0.999


def test_radius_neighbors_async_equals_sync(dev):
    """The synchronisation-free tables (collate with calibrated limits) equal the two-pass ones bit for bit, both
    when the limit truncates rows and when it exceeds the largest neighbour count (all-padding columns dropped)."""
    p0, l0 = _pair_clouds(3)
    p1, l1 = REF.subsample_batch(p0, l0, sampleDl=0.6)
    tq, ts = torch.from_numpy(p0).to(dev), torch.from_numpy(p1).to(dev)
    r = 0.3 * 4.25
    cases = [(tq, tq, l0, l0, r, 20), (ts, tq, l1, l0, r, 35), (tq, ts, l0, l1, 2 * r, 500), (ts, ts, l1, l1, r, 1)]
    flags = torch.empty((len(cases), 2), dtype=torch.int32, device=dev)
    tabs = [point_ops.radius_neighbors_async(q, s_, lq, ls, rad, lim, flags[i])
            for i, (q, s_, lq, ls, rad, lim) in enumerate(cases)]
    done = point_ops.finish_radius_tables(tabs, flags)
    for (q, s_, lq, ls, rad, lim), got in zip(cases, done):
        ref = point_ops.radius_neighbors(q, s_, lq, ls, rad, limit=lim)
        assert got.shape == ref.shape and torch.equal(got, ref), (lim, got.shape, ref.shape)
    assert done[2].shape[1] < 500        # the limit above the largest count was cut back


def test_batch_query_reference_api(dev):
    from apr_amd.predator.cpp_wrappers.cpp_neighbors import radius_neighbors as RN
    p0, l0 = _pair_clouds(4)
    out = RN.batch_query(p0, p0, l0, l0, radius=1.0)
    assert out.dtype == np.int32
    _check_neighbors(p0, p0, l0, l0, 1.0, out)
    with pytest.raises(RuntimeError):
        RN.batch_query(p0, p0, l0, l0[:1], radius=1.0)
    # no cross-talk between the two clouds of the batch
    n0 = l0[0]
    assert (out[:n0][out[:n0] != len(p0)] < n0).all() and (out[n0:] >= n0).all()


def test_knn_matches_dense_topk(dev):
    rng = np.random.default_rng(0)
    pts = torch.from_numpy(rng.uniform(-20, 20, (700, 3)).astype(np.float32))
    d = ((pts[:, None] - pts[None]) ** 2).sum(-1).clamp_min(1e-12)
    ref = d.topk(11, dim=-1, largest=False, sorted=True)[1][:, 1:]
    got = point_ops.knn(pts.to(dev), 10).cpu().long()
    assert got.shape == (700, 10)
    assert (got == ref).float().mean() > 0.999
    assert torch.equal(got.sort(1)[0], ref.sort(1)[0]) or (got.sort(1)[0] == ref.sort(1)[0]).float().mean() > 0.999


def test_radius_neighbors_beyond_the_rank_buffer(dev):
    """A clump with more than 1024 points inside the radius (a scan taken next to a wall): the `limit` nearest are
    still selected exactly (bisection on the distance, ties by index), like the reference's nanoflann search followed
    by the dataloader's truncation (Predator_APR/cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:211-333,
    datasets/dataloader.py:66-68); duplicates of one point make a run of exact ties."""
    rng = np.random.default_rng(9)
    clump = rng.normal(0, 0.25, (2600, 3)).astype(np.float32)
    clump[100:160] = clump[100]                                    # 60 exact duplicates: ties at one distance
    far = rng.uniform(-30, 30, (4000, 3)).astype(np.float32)
    pts = np.concatenate([clump, far]).astype(np.float32)
    lens = np.array([len(pts)], np.int32)
    t = torch.from_numpy(pts).to(dev)
    r, lim = 1.0, 48
    got = point_ops.radius_neighbors(t, t, lens, lens, r, limit=lim).cpu().numpy()
    ref = REF.batch_query(pts, pts, lens, lens, radius=r)[:, :lim]
    assert got.shape == ref.shape == (len(pts), lim)
    d_ref = ((pts[np.minimum(ref, len(pts) - 1)] - pts[:, None]) ** 2).sum(-1)
    d_got = ((pts[np.minimum(got, len(pts) - 1)] - pts[:, None]) ** 2).sum(-1)
    pad_ref, pad_got = ref == len(pts), got == len(pts)
    assert np.array_equal(pad_ref, pad_got)
    assert np.allclose(np.where(pad_ref, 0, d_ref), np.where(pad_got, 0, d_got), rtol=1e-5, atol=1e-7)
    assert (got == ref).mean() > 0.97                              # positions differ only inside runs of equal distance
    # the asynchronous table (collate path) takes the same route
    flags = torch.empty(2, dtype=torch.int32, device=dev)
    tab = point_ops.finish_radius_tables([point_ops.radius_neighbors_async(t, t, lens, lens, r, lim, flags)],
                                         flags.view(1, 2))[0]
    assert torch.equal(tab.cpu(), torch.from_numpy(got))
    # without a limit the full width cannot be ranked: loud error, not a truncated table
    with pytest.raises(Exception):
        point_ops.radius_neighbors(t, t, lens, lens, r)
