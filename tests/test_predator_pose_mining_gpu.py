"""Predator tester pieces (P9) and hardest-contrastive mining (F12) vs the CPU oracle."""
import numpy as np
import pytest
import torch

from apr_amd.fcgf import registration
from apr_amd.fcgf.lib.trainer import HardestContrastiveLoss
from apr_amd.predator.lib import benchmark_utils as BU
from oracle import match_pose_oracle as MO
from tests.test_match_pose_gpu import _synthetic_pair

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,inlier", [(0, 0.5), (3, 0.3)])
def test_ransac_pose_estimation_geometric_matches_oracle(dev, seed, inlier):
    """ransac_pose_estimation(..., distance_threshold=0.3, ransac_n=4) as lib/tester.py:97 calls it."""
    xyz0, xyz1, F0, F1, T_gt = _synthetic_pair(seed, n=2500, inlier=inlier)
    corr, _ = MO.feature_nn(F0, F1)
    T_o, info_o = MO.ransac_feature_matching_geometric(xyz0, xyz1, corr, 0.3, 0.9, max_iter=50000,
                                                       max_validation=1000, seed=seed)
    T, info = BU.ransac_pose_estimation(xyz0, xyz1, F0, F1, mutual=False, distance_threshold=0.3, ransac_n=4,
                                        seed=seed, return_info=True)
    assert info["n_valid"] == info_o["n_valid"] > 0
    assert info["best_iteration"] == info_o["best_iteration"] and info["inliers"] == info_o["inliers"]
    rte, rre = registration.rte_rre(T, T_o)
    assert rte < 1e-3 and rre < 1e-3
    rte, rre = registration.rte_rre(T, T_gt)
    assert rte < 0.3 and rre < 1.0


def test_max_validation_cuts_in_iteration_order(dev):
    xyz0, xyz1, F0, F1, _ = _synthetic_pair(5, n=2000, inlier=0.6)
    corr, _ = MO.feature_nn(F0, F1)
    for mv in (3, 40):
        T_o, info_o = MO.ransac_feature_matching_geometric(xyz0, xyz1, corr, 0.3, 0.9, 20000, mv, seed=9)
        T, info = BU.ransac_pose_estimation(xyz0, xyz1, F0, F1, distance_threshold=0.3, ransac_n=4,
                                            max_iteration=20000, max_validation=mv, seed=9, return_info=True)
        assert info["best_iteration"] == info_o["best_iteration"] and info["inliers"] == info_o["inliers"]


def test_score_sampling_and_angle_deviation(dev):
    rng = np.random.default_rng(0)
    pcd = torch.from_numpy(rng.standard_normal((7000, 3)).astype(np.float32)).to(dev)
    feats = torch.from_numpy(rng.standard_normal((7000, 32)).astype(np.float32)).to(dev)
    scores = torch.rand(7000)
    np.random.seed(0)
    p, f, idx = BU.sample_by_score(pcd, feats, scores, 5000)
    np.random.seed(0)
    probs = (scores / scores.sum()).numpy().flatten()          # lib/tester.py:85, float32 on the host
    ref = np.random.choice(np.arange(7000), size=5000, replace=False, p=probs)
    assert np.array_equal(idx, ref) and p.shape == (5000, 3) and torch.equal(f.cpu(), feats.cpu()[ref])
    a = np.deg2rad(7.0)
    R = np.array([[[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]]])
    assert abs(BU.get_angle_deviation(R, np.eye(3)[None])[0] - 7.0) < 1e-9


@pytest.mark.parametrize("c,n0,n1,npos", [(32, 3000, 2800, 1500), (16, 900, 1000, 5000)])
def test_hardest_contrastive_matches_oracle(dev, c, n0, n1, npos):
    rng = np.random.default_rng(c)
    F0 = rng.standard_normal((n0, c)).astype(np.float32); F0 /= np.linalg.norm(F0, axis=1, keepdims=True)
    F1 = rng.standard_normal((n1, c)).astype(np.float32); F1 /= np.linalg.norm(F1, axis=1, keepdims=True)
    pos = np.stack([rng.integers(0, n0, npos), rng.integers(0, n1, npos)], 1)
    F1[pos[:200, 1]] = F0[pos[:200, 0]] + 0.02 * rng.standard_normal((200, c)).astype(np.float32)  # real positives
    sel0 = rng.choice(n0, min(n0, 512), replace=False)
    sel1 = rng.choice(n1, min(n1, 512), replace=False)
    sel1[:50] = pos[:50, 1]                                  # some mined negatives ARE positives -> masked
    sel1 = np.unique(sel1)
    pos_sel = rng.choice(npos, 1024, replace=False) if npos > 1024 else None
    ref = MO.hardest_contrastive(F0, F1, pos, sel0, sel1, pos_sel)
    loss = HardestContrastiveLoss(0.1, 1.4)
    got = loss.contrastive_hardest_negative_loss(torch.from_numpy(F0).to(dev), torch.from_numpy(F1).to(dev), pos,
                                                 draws=(sel0, sel1, pos_sel))
    assert abs(float(got[0]) - float(ref[0])) < 1e-5 * max(1.0, abs(float(ref[0])))
    assert abs(float(got[1]) - float(ref[1])) < 1e-5 * max(1.0, abs(float(ref[1])))
    # the reference's own sampling path (host RNG) runs too
    np.random.seed(1)
    p, n = loss.contrastive_hardest_negative_loss(torch.from_numpy(F0).to(dev), torch.from_numpy(F1).to(dev), pos,
                                                  num_pos=1024, num_hn_samples=256)
    assert np.isfinite(float(p)) and np.isfinite(float(n))


def test_predator_pair_pipeline_registers_a_synthetic_pair(dev):
    """PredatorRegistration (grid subsample -> collate -> KPFCNN -> score sampling -> RANSAC(50000, 1000)) end to end.
    A random-init network has no useful features, so the test feeds the pipeline's RANSAC stage features that
    encode the geometry (descriptor = quantised canonical position) through `sample_by_score` + `ransac_pose_estimation`,
    and checks the encoder stage separately for shape / finiteness."""
    from apr_amd import synth
    from apr_amd.predator.configs.models import kitti_config
    from apr_amd.predator.models.architectures import KPFCNN
    from apr_amd.predator.pipeline import PredatorRegistration
    from apr_amd.fcgf import registration
    torch.manual_seed(0); np.random.seed(0)
    cfg = kitti_config()
    model = KPFCNN(cfg).to(dev).eval()
    a, b, T_gt = synth.make_pair(3, n_beams=16, n_azimuth=900)
    pipe = PredatorRegistration(model, cfg, [40, 40, 40, 40], max_iteration=20000)
    ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    src, tgt, feats, ov, sal = pipe.encode(ta, tb)
    assert feats.shape[0] == len(src) + len(tgt) and feats.shape[1] == cfg.final_feats_dim
    assert torch.isfinite(feats).all() and torch.isfinite(ov).all() and torch.isfinite(sal).all()
    assert float(ov.min()) >= 0.0 and float(ov.max()) <= 1.0
    T, info = pipe(ta, tb, seed=1)
    assert T.shape == (4, 4) and info["n0"] == len(src) and info["n1"] == len(tgt)
    # the matching + pose stage on informative descriptors: position in the target frame, jittered
    rng = np.random.default_rng(0)
    s_np, t_np = src.cpu().numpy(), tgt.cpu().numpy()
    sf = (s_np.astype(np.float64) @ T_gt[:3, :3].T + T_gt[:3, 3]).astype(np.float32)
    sf += 0.02 * rng.standard_normal(sf.shape).astype(np.float32)
    from apr_amd.predator.lib import benchmark_utils as BU
    T2 = BU.ransac_pose_estimation(s_np, t_np, sf, t_np.copy(), distance_threshold=0.3, ransac_n=4, seed=2)
    rte, rre = registration.rte_rre(T2, T_gt)
    assert rte < 0.3 and rre < 1.0
