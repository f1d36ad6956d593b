"""Feature NN (bit-exact), RANSAC + Kabsch and IRLS pose vs the CPU oracle."""
import numpy as np
import pytest
import torch

from apr_amd import ops, synth
from apr_amd.fcgf import registration
from apr_amd.fcgf.lib.eval import find_nn_gpu
from apr_amd.fcgf.util.transform_estimation import est_quad_linear_robust
from oracle import match_pose_oracle as MO
from oracle import me_oracle as OME

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n0,n1,c", [(5000, 5000, 32), (777, 3001, 32), (1500, 900, 128), (1000, 1000, 16),
                                      (300, 700, 64), (257, 513, 20), (64, 1, 32)])
def test_feature_nn_bit_exact(dev, n0, n1, c):
    rng = np.random.default_rng(n0 + n1 + c)
    F0 = rng.standard_normal((n0, c)).astype(np.float32)
    F1 = rng.standard_normal((n1, c)).astype(np.float32)
    F0 /= np.linalg.norm(F0, axis=1, keepdims=True)
    F1 /= np.linalg.norm(F1, axis=1, keepdims=True)
    F1[n1 // 2] = F1[0]  # exact tie -> smallest index must win
    oi, od = MO.feature_nn(F0, F1)
    for impl in ("brute", "fast"):      # "fast" falls back to the brute-force kernel for C outside 32/64/128
        idx, d2 = ops.feature_nn(torch.from_numpy(F0).to(dev), torch.from_numpy(F1).to(dev), return_distance=True,
                                 impl=impl)
        assert np.array_equal(idx.cpu().numpy(), oi)
        assert np.array_equal(d2.cpu().numpy().view(np.uint32), od.view(np.uint32))


@pytest.mark.parametrize("case", ["clustered", "unnormalised", "duplicates", "tiny", "one_target", "scaled_small"])
def test_feature_nn_filter_refine_equals_brute_force(dev, case):
    """The bf16-MFMA filter + exact refine path must return the SAME bits as the brute-force kernel (and the oracle)
    on inputs built to stress the bound: near-ties, large and tiny magnitudes, exact duplicates, ragged sizes."""
    rng = np.random.default_rng(abs(hash(case)) % 1000)
    c = 32
    if case == "clustered":         # thousands of near-ties: targets are small perturbations of a few centres
        cent = rng.standard_normal((7, c)).astype(np.float32)
        F1 = (cent[rng.integers(0, 7, 4000)] + 1e-3 * rng.standard_normal((4000, c))).astype(np.float32)
        F0 = (cent[rng.integers(0, 7, 1777)] + 1e-3 * rng.standard_normal((1777, c))).astype(np.float32)
    elif case == "unnormalised":    # magnitudes from 1e-2 to 1e3 in one problem
        F0 = (rng.standard_normal((1500, c)) * 10.0 ** rng.uniform(-2, 3, (1500, 1))).astype(np.float32)
        F1 = (rng.standard_normal((2100, c)) * 10.0 ** rng.uniform(-2, 3, (2100, 1))).astype(np.float32)
    elif case == "duplicates":      # every target appears 3 times: smallest index must win, d2 = 0 for copies
        base = rng.standard_normal((500, c)).astype(np.float32)
        F1 = np.concatenate([base, base, base])[rng.permutation(1500)]
        F0 = np.concatenate([base[:300], rng.standard_normal((211, c)).astype(np.float32)])
    elif case == "tiny":
        F0 = rng.standard_normal((5, c)).astype(np.float32)
        F1 = rng.standard_normal((3, c)).astype(np.float32)
    elif case == "one_target":
        F0 = rng.standard_normal((1000, c)).astype(np.float32)
        F1 = rng.standard_normal((1, c)).astype(np.float32)
    else:                           # scaled_small: |a| ~ 1e-3, squared distances ~1e-6
        F0 = (1e-3 * rng.standard_normal((900, c))).astype(np.float32)
        F1 = (1e-3 * rng.standard_normal((1100, c))).astype(np.float32)
    a, b = torch.from_numpy(F0).to(dev), torch.from_numpy(F1).to(dev)
    idx, d2 = ops.feature_nn(a, b, return_distance=True, impl="fast")
    idx_b, d2_b = ops.feature_nn(a, b, return_distance=True, impl="brute")
    assert torch.equal(idx, idx_b)
    assert torch.equal(d2.view(torch.int32), d2_b.view(torch.int32))
    oi, od = MO.feature_nn(F0, F1)
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert np.array_equal(d2.cpu().numpy().view(np.uint32), od.view(np.uint32))


@pytest.mark.parametrize("c", [32, 64, 128])
def test_feature_nn_collapsed_features_take_the_dense_blocks(dev, c):
    """A collapsed encoder (every row within 1e-6 of one vector): every 64 x 16 block of the search is listed as a dense
    block by the refine pass (appended per workgroup) and evaluated by k_nn_resolve's dense role.  Same bits as brute force
    and as the oracle."""
    rng = np.random.default_rng(c)
    v = rng.standard_normal((1, c)).astype(np.float32)
    v /= np.linalg.norm(v)
    F0 = (v + 1e-6 * rng.standard_normal((1333, c))).astype(np.float32)
    F1 = (v + 1e-6 * rng.standard_normal((2501, c))).astype(np.float32)
    a, b = torch.from_numpy(F0).to(dev), torch.from_numpy(F1).to(dev)
    idx, d2 = ops.feature_nn(a, b, return_distance=True, impl="fast")
    idx_b, d2_b = ops.feature_nn(a, b, return_distance=True, impl="brute")
    assert torch.equal(idx, idx_b)
    assert torch.equal(d2.view(torch.int32), d2_b.view(torch.int32))
    oi, od = MO.feature_nn(F0, F1)
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert np.array_equal(d2.cpu().numpy().view(np.uint32), od.view(np.uint32))


@pytest.mark.parametrize("c", [32, 128])
def test_feature_nn_list_overflow_takes_the_exact_fallback(dev, c):
    """More dense blocks than the 2^20-entry list holds (20 k x 60 k identical rows: 1.17 M blocks): the overflow flag is up
    when the refine pass ends and every wave of k_nn_resolve redoes the search exactly.  Copies tie at d2 = 0: index 0."""
    rng = np.random.default_rng(c + 1)
    v = rng.standard_normal((1, c)).astype(np.float32)
    a = torch.from_numpy(np.repeat(v, 20000, axis=0)).to(dev)
    t = np.repeat(v, 60000, axis=0)
    t[7::1000] += 0.5          # a few distinct rows, never the nearest
    b = torch.from_numpy(t).to(dev)
    a[123] = b[7]              # one query whose copy sits at index 7, 1007, ...: smallest index wins
    idx, d2 = ops.feature_nn(a, b, return_distance=True, impl="fast")
    idx_b, d2_b = ops.feature_nn(a, b, return_distance=True, impl="brute")
    assert torch.equal(idx, idx_b)
    assert torch.equal(d2.view(torch.int32), d2_b.view(torch.int32))
    assert int(idx[0]) == 0 and int(idx[123]) == 7 and float(d2.max()) == 0.0


def test_find_nn_gpu_reference_api(dev):
    """find_nn_gpu(F0, F1, nn_max_n=500) as called by find_corr (scripts/test_apr.py:52)."""
    rng = np.random.default_rng(0)
    F0 = torch.from_numpy(rng.standard_normal((1200, 32)).astype(np.float32))
    F1 = torch.from_numpy(rng.standard_normal((900, 32)).astype(np.float32))
    inds = find_nn_gpu(F0.to(dev), F1.to(dev), nn_max_n=500)
    ref = MO.pdist(F0, F1, 'SquareL2').min(1)[1]
    assert inds.device.type == "cpu" and inds.dtype == torch.int64
    assert (inds == ref).float().mean() > 0.999   # torch's sum order may flip a near tie
    inds2, d = find_nn_gpu(F0.to(dev), F1.to(dev), nn_max_n=500, return_distance=True, dist_type='L2')
    assert d.shape == (1200, 1)
    assert torch.allclose(d[:, 0], MO.pdist(F0, F1, 'L2').min(1)[0], rtol=1e-5, atol=1e-6)


def _synthetic_pair(seed, n=3000, inlier=0.5, noise=0.03):
    """Source points, a known rigid transform, and features whose NN is the true match for ~inlier of them."""
    rng = np.random.default_rng(seed)
    xyz0 = rng.uniform(-30, 30, size=(n, 3)).astype(np.float32)
    a = np.deg2rad(rng.uniform(-20, 20))
    R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    t = rng.uniform(-5, 5, size=3)
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
    perm = rng.permutation(n)
    xyz1 = np.empty_like(xyz0)
    xyz1[perm] = (xyz0 @ R.T + t + rng.normal(0, noise, size=(n, 3))).astype(np.float32)
    F0 = rng.standard_normal((n, 32)).astype(np.float32)
    F1 = np.empty_like(F0)
    F1[perm] = F0 + 0.05 * rng.standard_normal((n, 32)).astype(np.float32)
    bad = rng.random(n) > inlier
    F0[bad] = rng.standard_normal((int(bad.sum()), 32)).astype(np.float32)
    return xyz0, xyz1, F0, F1, T


@pytest.mark.parametrize("seed,inlier", [(0, 0.5), (1, 0.25), (2, 0.8)])
def test_ransac_matches_oracle_and_ground_truth(dev, seed, inlier):
    xyz0, xyz1, F0, F1, T_gt = _synthetic_pair(seed, inlier=inlier)
    corr_o, _ = MO.feature_nn(F0, F1)
    T_o, info_o = MO.ransac_feature_matching(xyz0, xyz1, corr_o, 0.3, 0.9, max_iter=60000, seed=seed)
    T, info = registration.ransac_feature_matching(xyz0, xyz1, F0, F1, 0.3, ransac_n=4, edge_length=0.9,
                                                    max_iteration=60000, seed=seed, return_info=True)
    assert info["n_valid"] == info_o["n_valid"] > 0
    assert info["inliers"] == info_o["inliers"]
    assert info["best_iteration"] == info_o["best_iteration"]
    rte, rre = registration.rte_rre(T, T_o)
    assert rte < 1e-3 and rre < 1e-3          # bar: 1e-3 m / 1e-3 deg vs the CPU path
    assert abs(info["rmse"] - info_o["rmse"]) < 1e-9
    rte, rre = registration.rte_rre(T, T_gt)  # and it actually registers the pair
    assert rte < 0.3 and rre < 1.0


def test_ransac_feature_matching_honours_max_validation(dev):
    """`max_validation` (open3d <= 0.11 reading of RANSACConvergenceCriteria(max_iteration, max_validation)) is no longer
    accepted and ignored: an integer selects the stop-after-N-validations / geometric-scoring flavour (== the oracle's
    restatement of it), None keeps the open3d >= 0.12 reading, nonsense raises."""
    xyz0, xyz1, F0, F1, T_gt = _synthetic_pair(4, inlier=0.5)
    corr_o, _ = MO.feature_nn(F0, F1)
    for mv in (50, 1000):
        T_o, info_o = MO.ransac_feature_matching_geometric(xyz0, xyz1, corr_o, 0.3, 0.9, 60000, mv, seed=2)
        T, info = registration.ransac_feature_matching(xyz0, xyz1, F0, F1, 0.3, max_iteration=60000, max_validation=mv,
                                                        seed=2, return_info=True)
        assert info["best_iteration"] == info_o["best_iteration"] and info["inliers"] == info_o["inliers"]
        rte, rre = registration.rte_rre(T, T_o)
        assert rte < 1e-3 and rre < 1e-3
    T_all, info_all = registration.ransac_feature_matching(xyz0, xyz1, F0, F1, 0.3, max_iteration=60000, seed=2,
                                                            return_info=True)
    assert info_all["n_valid"] > 0 and registration.rte_rre(T_all, T_gt)[0] < 0.3
    with pytest.raises(ValueError):
        registration.ransac_feature_matching(xyz0, xyz1, F0, F1, 0.3, max_validation=0)


def test_ransac_no_valid_hypothesis_returns_identity(dev):
    rng = np.random.default_rng(0)
    xyz0 = rng.uniform(-30, 30, (500, 3)).astype(np.float32)
    xyz1 = rng.uniform(-30, 30, (500, 3)).astype(np.float32)
    corr = torch.from_numpy(rng.integers(0, 500, 500)).to(dev)
    T, info = ops.ransac_pose(torch.from_numpy(xyz0).to(dev), torch.from_numpy(xyz1).to(dev), corr, 0.05, 0.99,
                              max_iter=2000, seed=1)
    T_o, info_o = MO.ransac_feature_matching(xyz0, xyz1, corr.cpu().numpy(), 0.05, 0.99, max_iter=2000, seed=1)
    assert info["n_valid"] == info_o["n_valid"]
    if info["n_valid"] == 0:
        assert np.array_equal(T, np.eye(4))


def test_ransac_full_reference_criteria(dev):
    """The reference's criteria (4 000 000 iterations) on a KITTI-sized correspondence set; checked
    against ground truth (the oracle would take hours at this size)."""
    xyz0, xyz1, F0, F1, T_gt = _synthetic_pair(11, n=15000, inlier=0.2)
    T, info = registration.ransac_feature_matching(xyz0, xyz1, F0, F1, 0.3, max_iteration=4000000, seed=5,
                                                    return_info=True)
    rte, rre = registration.rte_rre(T, T_gt)
    assert rte < 0.2 and rre < 0.5 and info["inliers"] > 0.15 * 15000


def test_ransac_single_round_and_chunked_rounds_agree(dev, monkeypatch):
    """The fast path (all iterations in one round) and the 2^20-iteration rounds are the same search."""
    xyz0, xyz1, F0, F1, _ = _synthetic_pair(3, n=4000, inlier=0.3)
    args = dict(ransac_n=4, edge_length=0.9, max_iteration=2500000, seed=9, return_info=True)
    T_a, info_a = registration.ransac_feature_matching(xyz0, xyz1, F0, F1, 0.3, **args)
    with ops.ransac_options(force_rounds=1):       # the library reads APR_RANSAC_* once per process: switches go through the API
        T_b, info_b = registration.ransac_feature_matching(xyz0, xyz1, F0, F1, 0.3, **args)
    assert info_a == info_b and np.array_equal(T_a, T_b)


def test_ransac_hypothesis_list_overflow_replays_in_rounds(dev):
    """Perfect correspondences: every one of 1.2 M hypotheses survives both checkers, more than the 2^20-entry
    list holds, so the call must fall back to rounds; all of them are counted and scored."""
    rng = np.random.default_rng(5)
    n = 300
    xyz0 = rng.uniform(-20, 20, (n, 3)).astype(np.float32)
    R = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    xyz1 = (xyz0.astype(np.float64) @ R.T + np.array([1.0, -2.0, 0.5])).astype(np.float32)
    corr = torch.arange(n, device=dev)
    T, info = ops.ransac_pose(torch.from_numpy(xyz0).to(dev), torch.from_numpy(xyz1).to(dev), corr, 0.3, 0.9,
                              max_iter=1200000, seed=2)
    assert info["n_valid"] == 1200000 and info["inliers"] == n
    assert np.allclose(T[:3, :3], R, atol=1e-5) and np.allclose(T[:3, 3], [1.0, -2.0, 0.5], atol=1e-4)
    # the winner is the lowest-rmse hypothesis of ALL rounds: same answer as the oracle on its own iteration
    T_o, info_o = MO.ransac_feature_matching(xyz0, xyz1, np.arange(n), 0.3, 0.9, max_iter=info["best_iteration"] + 1,
                                             seed=2) if info["best_iteration"] < 20000 else (None, None)
    if info_o is not None:
        assert info_o["best_iteration"] == info["best_iteration"]


@pytest.mark.parametrize("seed,weighted", [(0, False), (1, True)])
def test_irls_matches_oracle_and_recovers_pose(dev, seed, weighted):
    rng = np.random.default_rng(seed)
    n = 5000
    p0 = rng.uniform(-20, 20, (n, 3)).astype(np.float32)
    ang = np.deg2rad(rng.uniform(-3, 3, 3))
    x = torch.tensor([[ang[0]], [ang[1]], [ang[2]], [0.4], [-0.3], [0.2]], dtype=torch.float32)
    T_gt = MO._get_trans(x)
    p1 = (torch.from_numpy(p0) @ T_gt[:3, :3].t() + T_gt[:3, 3]).numpy()
    p1 += rng.normal(0, 0.01, p1.shape).astype(np.float32)
    out = rng.random(n) < 0.1
    p1[out] += rng.uniform(-5, 5, (int(out.sum()), 3)).astype(np.float32)
    w = torch.from_numpy(rng.uniform(0.2, 1.0, (n, 1)).astype(np.float32)) if weighted else None
    T_o = MO.est_quad_linear_robust(torch.from_numpy(p0), torch.from_numpy(p1), w)
    T = est_quad_linear_robust(torch.from_numpy(p0), torch.from_numpy(p1), w)
    assert T.dtype == torch.float32 and T.shape == (4, 4) and T.device.type == "cpu"
    rte, rre = registration.rte_rre(T.numpy(), T_o.numpy())
    assert rte < 1e-3 and rre < 1e-3
    rte, rre = registration.rte_rre(T.numpy(), T_gt.numpy())
    assert rte < 0.02 and rre < 0.05


def test_end_to_end_pair_pipeline(dev):
    """voxelise -> encode both frames -> feature NN -> RANSAC on a small synthetic pair, vs the oracle chain."""
    from apr_amd import MinkowskiEngine as ME
    from tests.helpers import model_pair, rel_l2
    om, hm = model_pair("ResUNetBN2C", 32)
    om.eval(); hm.eval()
    xyz0, xyz1, T_gt = synth.make_pair(0, n_beams=16, n_azimuth=1250)
    feats_o, feats_h, pts = [], [], []
    for xyz in (xyz0, xyz1):
        c, sel = OME.sparse_quantize(xyz / np.float32(0.3), return_index=True)
        C = OME.batched_coordinates([c]); F = np.ones((len(C), 1), np.float32)
        pts.append(xyz[sel])
        with torch.no_grad():
            feats_o.append(om(OME.SparseTensor(F, coordinates=C)).F.numpy())
            # product path: GPU voxel hash end to end
            cg = ops.voxelize(torch.from_numpy(xyz).to(dev), 0.3, 0)
            m = ops.build_map(cg, want_first=True); ops.finalize_maps([m])
            assert np.array_equal(m.first.cpu().numpy(), sel)
            x = ME.SparseTensor(torch.ones((m.n, 1), device=dev), coordinates=m.coords)
            feats_h.append(hm(x).F)
    assert rel_l2(feats_h[0].cpu(), feats_o[0]) < 2e-5 and rel_l2(feats_h[1].cpu(), feats_o[1]) < 2e-5
    # same features in -> identical correspondences and pose out
    corr_o, _ = MO.feature_nn(feats_o[0], feats_o[1])
    T_o, info_o = MO.ransac_feature_matching(pts[0], pts[1], corr_o, 0.3, 0.9, max_iter=20000, seed=3)
    F0 = torch.from_numpy(feats_o[0]).to(dev); F1 = torch.from_numpy(feats_o[1]).to(dev)
    T, info = registration.ransac_feature_matching(pts[0], pts[1], F0, F1, 0.3, max_iteration=20000, seed=3,
                                                    return_info=True)
    assert info["n_valid"] == info_o["n_valid"] and info["inliers"] == info_o["inliers"]
    rte, rre = registration.rte_rre(T, T_o)
    assert rte < 1e-3 and rre < 1e-3


def test_register_batch_equals_pair_by_pair(dev):
    """B pairs through one batched encoder call give the per-pair results (features to fp32 summation-order noise;
    with the same correspondences the pose is the same hypothesis)."""
    from apr_amd.fcgf.pipeline import PairRegistration
    from tests.helpers import model_pair, rel_l2
    _, hm = model_pair("ResUNetBN2C", 32)
    hm.eval()
    pipe = PairRegistration(hm, 0.3, ransac_iters=30000)
    pairs = []
    for s in (0, 1, 2):
        a, b, _ = synth.make_pair(s, n_beams=16, n_azimuth=900 + 100 * s)
        pairs.append((torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)))
    single = [pipe(a, b, seed=7 + i) for i, (a, b) in enumerate(pairs)]
    batch = pipe.register_batch(pairs, seeds=[7, 8, 9])
    # features: batched vs separate encoder calls
    cm, counts, first, poffs, pts_all = pipe.voxelize_batch([c for p in pairs for c in p])
    F = pipe.encode_batch(cm)
    off = 0
    for k, (a, b) in enumerate(pairs):
        c1, p0, p1, n0, n1 = pipe.voxelize_pair(a, b)
        F0, F1 = pipe.encode_pair(c1, n0)
        assert counts[2 * k] == n0 and counts[2 * k + 1] == n1
        # one batched hash build == per-frame builds: same voxels in the same order, same representative points
        assert torch.equal(cm.get_coordinates(1)[off:off + n0, 1:], c1[:n0, 1:])
        assert torch.equal(a[first[off:off + n0] - poffs[2 * k]], p0) and torch.equal(pts_all[off:off + n0], p0)
        assert rel_l2(F[off:off + n0].cpu(), F0.cpu()) < 1e-6 and rel_l2(F[off + n0:off + n0 + n1].cpu(), F1.cpu()) < 1e-6
        off += n0 + n1
    for (T1, i1), (T2, i2) in zip(single, batch):
        assert i1["n0"] == i2["n0"] and i1["n1"] == i2["n1"]
        if i1["best_iteration"] == i2["best_iteration"]:      # identical correspondences -> identical search
            assert i1["inliers"] == i2["inliers"] and np.allclose(T1, T2, atol=1e-9)
        else:                                                # an NN near-tie flipped: still a valid registration
            assert abs(i1["inliers"] - i2["inliers"]) <= max(5, 0.05 * i1["inliers"])


def test_match_pose_batch_equals_per_pair_calls(dev):
    """apr_match_pose_batch (one call, one sync) == feature_nn + ransac_pose pair by pair, bit for bit; one of the
    pairs has perfect correspondences and more than 2^20 valid hypotheses, i.e. takes the overflow replay."""
    cases = []
    for s, (n, inl) in enumerate([(3000, 0.4), (2200, 0.25), (1500, 0.6)]):
        xyz0, xyz1, F0, F1, _ = _synthetic_pair(20 + s, n=n, inlier=inl)
        cases.append((xyz0, xyz1, F0, F1))
    # perfect pair: identical features in the same order -> every hypothesis survives both checkers
    rng = np.random.default_rng(77)
    q = rng.uniform(-15, 15, (260, 3)).astype(np.float32)
    R = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    t = (q.astype(np.float64) @ R.T + np.array([0.5, 1.5, -0.25])).astype(np.float32)
    Fp = rng.standard_normal((260, 32)).astype(np.float32)
    cases.append((q, t, Fp, Fp.copy()))
    g = lambda a: torch.from_numpy(a).to(dev)
    iters = 1200000
    seeds = [3, 4, 5, 6]
    batch = ops.match_pose_batch([g(c[2]) for c in cases], [g(c[3]) for c in cases], [g(c[0]) for c in cases],
                                 [g(c[1]) for c in cases], 0.3, 0.9, iters, seeds=seeds)
    for (xyz0, xyz1, F0, F1), (Tb, ib), seed in zip(cases, batch, seeds):
        corr = ops.feature_nn(g(F0), g(F1))
        T, info = ops.ransac_pose(g(xyz0), g(xyz1), corr, 0.3, 0.9, iters, seed)
        assert info == ib and np.array_equal(T, Tb)
    assert batch[3][1]["n_valid"] == iters and batch[3][1]["inliers"] == 260
    # the pairs dealt over 2 / 3 / 4 streams inside the library (fork / join around the caller's stream), 7 pairs so
    # that lanes carry different numbers of them: the same bits as on one stream, and the caller's stream still orders
    # what follows behind all of them (the features are overwritten right after the call)
    seven = (cases + cases)[:7]
    seeds7 = list(range(11, 18))
    ref7 = None
    try:
        for lanes in (1, 2, 3, 4):
            ops.set_match_lanes(lanes)
            f0s, f1s = [g(c[2]) for c in seven], [g(c[3]) for c in seven]
            pend = ops.match_pose_batch_async(f0s, f1s, [g(c[0]) for c in seven], [g(c[1]) for c in seven], 0.3, 0.9,
                                              iters, seeds=seeds7)
            for f in f0s + f1s:
                f.zero_()                      # enqueued behind the join
            pend.event.synchronize()
            got = pend.finish()
            if ref7 is None:
                ref7 = got
            for (Ta, ia), (Tb, ib) in zip(ref7, got):
                assert ia == ib and np.array_equal(Ta, Tb), lanes
        with pytest.raises(Exception):
            ops.set_match_lanes(5)
    finally:
        ops.set_match_lanes(1)


def test_batch_path_replays_overflow_at_full_reference_criteria(dev):
    """KITTI-sized pair, 80 % true correspondences, the reference's 4 000 000 iterations, through the BATCH entry point
    (apr_match_pose_batch) and without any test hook: ~1.6 M hypotheses survive both checkers, more than the 2^20-entry
    list holds, so the pair is redone in rounds inside the call.  Same result as the per-pair entry point."""
    xyz0, xyz1, F0, F1, T_gt = _synthetic_pair(21, n=14000, inlier=0.8)
    t = lambda a: torch.from_numpy(a).to(dev)
    (T, info), = ops.match_pose_batch([t(F0)], [t(F1)], [t(xyz0)], [t(xyz1)], 0.3, 0.9, 4000000, seeds=[4])
    assert info["n_valid"] > (1 << 20)
    rte, rre = registration.rte_rre(T, T_gt)
    assert rte < 0.1 and rre < 0.2 and info["inliers"] > 0.7 * 14000
    corr = ops.feature_nn(t(F0), t(F1))
    T2, info2 = ops.ransac_pose(t(xyz0), t(xyz1), corr, 0.3, 0.9, 4000000, 4)
    assert info2 == {k: info[k] for k in info2} and np.array_equal(T, T2)


@pytest.mark.parametrize("share,iters", [(0.5, 4000000), (0.3, 1000000), (0.62, 300000)])
def test_many_survivors_count_path_equals_full_fp64_scoring(dev, share, iters):
    """Trained-descriptor regime (FCGF_APR/scripts/test_apr.py:148-156): 10^4 ... 10^6 hypotheses survive
    the checkers.  Above 2048 survivors the inlier counts come from the fp32-screened count kernels (k_count +
    k_count_fix) and only the hypotheses at the maximum count get their squared error summed; the result must be THE SAME
    BITS as scoring every survivor in fp64 (APR_RANSAC_COUNT=0): transform, inliers, rmse, best iteration, n_valid.  The
    last case overflows the 2^20-entry hypothesis list at once (replay in rounds)."""
    import os
    from apr_amd.fcgf.lib import apg
    from apr_amd.fcgf.pipeline import PairRegistration
    from tests.helpers import model_pair
    a_h, b_h, T_gt = synth.make_pair(0)
    ta, tb = torch.from_numpy(a_h).to(dev), torch.from_numpy(b_h).to(dev)
    _, hm = model_pair("ResUNetBN2C")
    _, pts0, pts1, n0, n1 = PairRegistration(hm, voxel_size=0.3).voxelize_pair(ta, tb)
    pairs_gt = apg.get_matching_indices(pts0, pts1, T_gt, 0.3, K=1)
    g = torch.Generator(device="cpu").manual_seed(3)
    F1 = torch.nn.functional.normalize(torch.randn(n1, 32, generator=g), dim=1).to(dev)
    F0 = torch.nn.functional.normalize(torch.randn(n0, 32, generator=g), dim=1).to(dev)
    pick = pairs_gt[torch.randperm(len(pairs_gt), generator=g)[:int(share * n0)].to(dev)]
    F0[pick[:, 0]] = torch.nn.functional.normalize(F1[pick[:, 1]] + 0.02 * torch.randn(len(pick), 32, generator=g).to(dev),
                                                   dim=1)
    res = {}
    for mode in ("1", "0"):
        with ops.ransac_options(count=int(mode)):
            (T, info), = ops.match_pose_batch([F0], [F1], [pts0], [pts1], 0.3, 0.9, iters, seeds=[7])
        res[mode] = (T, info)
    (T1, i1), (T0, i0) = res["1"], res["0"]
    assert i1["n_valid"] == i0["n_valid"] > 2048
    assert i1["inliers"] == i0["inliers"] and i1["rmse"] == i0["rmse"] and i1["best_iteration"] == i0["best_iteration"]
    assert np.array_equal(T1, T0)
    rte, rre = registration.rte_rre(T1, T_gt)
    assert rte < 0.1 and rre < 0.2


@pytest.mark.parametrize("n,scale,noise", [(100, 1.0, 0.0), (777, 1.0, 0.05), (3001, 1000.0, 30.0), (64, 1.0, 0.02)])
def test_count_path_edge_cases_equal_full_fp64_scoring(dev, n, scale, noise):
    """The fp32-screened count path (above 2048 surviving hypotheses) on small and awkward correspondence sets: fewer
    correspondences than one 64-wide mini-chunk, an odd count (padded row), coordinates 1000x KITTI's (the guard band then
    spans most of the threshold: nearly every cell is recounted in fp64) -- always the same bits as scoring every
    survivor in fp64 (APR_RANSAC_COUNT=0)."""
    import os
    rng = np.random.default_rng(n)
    src = (rng.uniform(-40, 40, (n, 3)) * scale).astype(np.float32)
    ang = 0.3
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    t = np.array([3.0, -2.0, 0.5]) * scale
    tgt = (src.astype(np.float64) @ R.T + t + rng.normal(0, noise, (n, 3))).astype(np.float32)
    corr = np.arange(n, dtype=np.int64)
    bad = rng.random(n) < 0.2
    corr[bad] = rng.integers(0, n, int(bad.sum()))
    thr = 0.3 * scale
    res = {}
    for mode in ("1", "0"):
        with ops.ransac_options(count=int(mode)):
            res[mode] = ops.ransac_pose(torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev),
                                        torch.from_numpy(corr).to(dev), thr, 0.9, 60000, 3)
    (T1, i1), (T0, i0) = res["1"], res["0"]
    assert i1["n_valid"] == i0["n_valid"] > 2048, i1
    assert i1["inliers"] == i0["inliers"] and i1["rmse"] == i0["rmse"] and i1["best_iteration"] == i0["best_iteration"]
    assert np.array_equal(T1, T0)
    assert i1["inliers"] >= 0.7 * n


@pytest.mark.parametrize("n,second_share,spread", [(4000, 0.35, 60.0), (2500, 0.45, 8.0), (900, 0.0, 400.0)])
def test_count_path_pruning_with_far_hypotheses_equals_full_scoring(dev, n, second_share, spread, monkeypatch):
    """The count path prunes the correspondences to those within reach of the FIRST surviving hypothesis and counts every
    survivor NEAR it over the pruned list, the others over all.  Correspondence sets where many survivors are FAR from the
    first one -- two rigid motions in one set (whichever the first survivor belongs to, the other motion's hypotheses are
    far), a small cloud with a long lever arm, coordinates 400 m out -- must give the same bits as scoring everything in
    fp64 and as the unpruned count path (APR_RANSAC_PRUNE=0)."""
    rng = np.random.default_rng(n)
    src = rng.uniform(-spread, spread, (n, 3)).astype(np.float32)

    def motion(ang, t):
        R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
        return src.astype(np.float64) @ R.T + np.asarray(t)

    tgt = motion(0.3, [3.0, -2.0, 0.5])
    second = rng.random(n) < second_share
    tgt[second] = motion(-0.2, [-15.0, 9.0, 1.0])[second]
    tgt = (tgt + rng.normal(0, 0.03, (n, 3))).astype(np.float32)
    corr = np.arange(n, dtype=np.int64)
    bad = rng.random(n) < 0.15
    corr[bad] = rng.integers(0, n, int(bad.sum()))
    args = (torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev), torch.from_numpy(corr).to(dev), 0.3, 0.9, 200000, 5)
    res = {}
    for name, opts in (("pruned", {}), ("unpruned", {"prune": 0}), ("fp64", {"count": 0})):
        with ops.ransac_options(**opts):
            res[name] = ops.ransac_pose(*args)
    T0, i0 = res["fp64"]
    assert i0["n_valid"] > 2048, i0
    for name in ("pruned", "unpruned"):
        T1, i1 = res[name]
        assert i1 == i0, (name, i1, i0)
        assert np.array_equal(T1, T0), name


def _screen_ab(fn):
    """fn() with the LDS-screened sampling kernel (the default) and with the plain kernel."""
    out = {}
    for mode in ("1", "0"):
        with ops.ransac_options(screen=int(mode)):
            out[mode] = fn()
    return out["1"], out["0"]


@pytest.mark.parametrize("share,iters,seed", [(0.0, 4000000, 1), (0.3, 4000000, 2), (0.55, 600000, 3), (0.1, 262144, 4)])
def test_lds_screened_sampling_equals_the_plain_kernel(dev, share, iters, seed):
    """k_sample_screen (first RANSAC edge tested on 10-bit quantised points out of LDS, rigorous guard band, survivors
    re-checked exactly) selects THE SAME candidates as k_sample_check: identical transform bits, inliers, rmse, best
    iteration and n_valid, from no true matches (the bench's random-init features) to a majority of them; full-size pair,
    the reference's 4 M iterations (FCGF_APR/scripts/test_apr.py:148-156)."""
    from apr_amd.fcgf.lib import apg
    from apr_amd.fcgf.pipeline import PairRegistration
    from tests.helpers import model_pair
    a_h, b_h, T_gt = synth.make_pair(seed)
    ta, tb = torch.from_numpy(a_h).to(dev), torch.from_numpy(b_h).to(dev)
    _, hm = model_pair("ResUNetBN2C")
    _, pts0, pts1, n0, n1 = PairRegistration(hm, voxel_size=0.3).voxelize_pair(ta, tb)
    g = torch.Generator(device="cpu").manual_seed(seed)
    F1 = torch.nn.functional.normalize(torch.randn(n1, 32, generator=g), dim=1).to(dev)
    F0 = torch.nn.functional.normalize(torch.randn(n0, 32, generator=g), dim=1).to(dev)
    if share > 0:
        pairs_gt = apg.get_matching_indices(pts0, pts1, T_gt, 0.3, K=1)
        pick = pairs_gt[torch.randperm(len(pairs_gt), generator=g)[:int(share * n0)].to(dev)]
        F0[pick[:, 0]] = torch.nn.functional.normalize(
            F1[pick[:, 1]] + 0.02 * torch.randn(len(pick), 32, generator=g).to(dev), dim=1)
    (T1, i1), (T0, i0) = _screen_ab(
        lambda: ops.match_pose_batch([F0], [F1], [pts0], [pts1], 0.3, 0.9, iters, seeds=[seed])[0])
    assert i1 == i0 and np.array_equal(T1, T0), (i1, i0)
    if share >= 0.3:
        rte, rre = registration.rte_rre(T1, T_gt)
        assert i1["n_valid"] > 1000 and rte < 0.1 and rre < 0.2


@pytest.mark.parametrize("case", ["one_point", "tiny", "huge_coords", "collinear", "too_many_for_lds"])
def test_lds_screened_sampling_edge_cases(dev, case):
    """Degenerate tables for the quantised screen: every point the same (R = 0: the screen must pass everything), 5
    correspondences, coordinates of 1e6 m (cells of 2 km), points on a line, and more correspondences than the LDS holds
    (plain kernel taken silently): always the plain kernel's result."""
    rng = np.random.default_rng(11)
    n = {"one_point": 300, "tiny": 5, "huge_coords": 4000, "collinear": 2000, "too_many_for_lds": 19000}[case]
    src = rng.uniform(-40, 40, (n, 3))
    if case == "one_point":
        src[:] = 0.0
    if case == "huge_coords":
        src *= 25000.0
    if case == "collinear":
        src[:, 1:] = 0.0
    ang = 0.4
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    tgt = src @ R.T + np.array([1.0, 2.0, -0.5])
    corr = np.arange(n, dtype=np.int64)
    bad = rng.random(n) < 0.5
    corr[bad] = rng.integers(0, n, int(bad.sum()))
    s_d, t_d, c_d = (torch.from_numpy(x).to(dev) for x in (src.astype(np.float32), tgt.astype(np.float32), corr))
    thr = 0.3 if case != "huge_coords" else 7500.0
    (T1, i1), (T0, i0) = _screen_ab(lambda: ops.ransac_pose(s_d, t_d, c_d, thr, 0.9, 300000, 5))
    assert i1 == i0 and np.array_equal(T1, T0), (i1, i0)
    if case in ("huge_coords", "too_many_for_lds"):
        assert i1["inliers"] >= 0.4 * n
