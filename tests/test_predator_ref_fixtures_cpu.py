"""CPU side of the round-5 reference pins (tests/golden/predator_ref.npz, generator make_predator_ref_golden.py): the
oracle's restatements and the host-side mirrors against outputs of the REFERENCE's own text.  No GPU compute here."""
import os

import numpy as np
import pytest
import torch

from apr_amd import synth
from apr_amd.fcgf.lib import metrics as MT
from apr_amd.predator.configs.models import kitti_config
from apr_amd.predator.lib import benchmark_utils as BU
from apr_amd.predator.models import mlp as MLP
from oracle import kpfcnn_oracle as KO
from oracle import predator_points_oracle as PREF

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "predator_ref.npz"))


def _sd(prefix):
    return {k[len(prefix) + 1:]: torch.from_numpy(np.asarray(G[k])) for k in G.files if k.startswith(prefix + "/")}


def test_generative_mlp_state_dict_loads_strictly_from_the_reference_module():
    """Keys / shapes of Predator_APR/models/mlp.py:103-179 (`list_modules.N.{0,2}.*`), incl. the BatchNorm1d the
    reference also gives the LAST block."""
    cfg = type("C", (), dict(generative_model="GenerativeMLP_54", final_feats_dim=32, point_generation_ratio=4,
                             batch_norm_momentum=0.02))
    m = MLP.get_GenerativeMLP(cfg, radius=None)
    res = m.load_state_dict(_sd("mlp54_sd"), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert m.list_modules[2][2].num_features == 12 and m.list_modules[0][2].momentum == 0.02
    cfg4 = type("C", (), dict(generative_model="GenerativeMLP_4", final_feats_dim=125, point_generation_ratio=6,
                              batch_norm_momentum=0.1))
    m4 = MLP.get_GenerativeMLP(cfg4, radius=2.5, in_channels=32)
    m4.load_state_dict(_sd("mlp4_sd"), strict=True)
    assert m4.radius == float(G["mlp4_radius"])
    # the wide variants only differ in CHANNELS
    assert [tuple(p.shape) for p in MLP.GenerativeMLP_98(32, 4).parameters()][:4] == [(512, 32), (512,), (512,), (512,)]
    assert MLP.GenerativeMLP_11_10_9.CHANNELS[1:4] == [2048, 1024, 512] and MLP.GenerativeMLP_99.CHANNELS[2] == 512


def test_angle_deviation_matches_reference_text():
    got = BU.get_angle_deviation(G["angle_R_pred"], G["angle_R_gt"])
    assert np.allclose(got, G["angle_degs"], rtol=0, atol=1e-12)
    assert got[0] < 1e-5 and abs(got[1] - 180.0) < 1e-5


def test_corr_dist_matches_reference_import():
    t = torch.from_numpy
    est, T, x0, w = t(G["cd_est"]), t(G["cd_T"]), t(G["cd_xyz0"]), t(G["cd_w"])
    assert abs(float(MT.corr_dist(est, T, x0, None, max_dist=0.25)) - float(G["cd_plain"])) < 1e-6
    assert abs(float(MT.corr_dist(est, T, x0, None, weight=w, max_dist=0.25)) - float(G["cd_weighted"])) < 1e-6
    assert abs(float(MT.corr_dist(est, T, x0, None)) - float(G["cd_default"])) < 1e-6


def test_score_sampling_draws_the_reference_indices():
    """lib/tester.py:80-92: np.random.choice(idx, 5000, replace=False, p=probs) from the legacy global stream -- same
    indices, and the stream is left at the same position."""
    scores = torch.from_numpy(G["tester_src_overlap"]) * torch.from_numpy(G["tester_src_saliency"])
    np.random.seed(123)
    idx = BU.draw_by_score(len(scores), scores, 5000)
    assert np.array_equal(idx, G["tester_idx_src"])
    t_scores = torch.from_numpy(G["tester_tgt_overlap"]) * torch.from_numpy(G["tester_tgt_saliency"])
    assert BU.draw_by_score(len(t_scores), t_scores, 5000) is None          # below the cap: every point kept, no draw
    assert np.random.random_sample() == float(G["tester_next_uniform"])
    pcd, feats = BU.take_drawn(G["tester_src_pcd"], G["tester_src_feats"], idx)
    assert np.array_equal(pcd, G["tester_src_pcd_out"]) and np.array_equal(feats, G["tester_src_feats_out"])


@pytest.mark.skipif(not PREF.available(), reason="oracle/_ref not built")
def test_oracle_collate_equals_reference_collate_text():
    """oracle/kpfcnn_oracle.collate (the restatement the full-size GPU tests check against) == the reference's own
    collate_fn_descriptor run over the reference's C++ (datasets/dataloader.py:72-198)."""
    cfg = kitti_config()
    caps = [int(v) for v in G["calib_caps"]]
    ref = KO.collate(G["collate_src"], G["collate_tgt"], cfg, caps + [caps[-1]])
    assert int(G["collate_levels"]) == len(ref["points"]) == 4
    for l in range(4):
        assert np.array_equal(ref["points"][l].numpy(), G[f"collate_points_{l}"])
        assert np.array_equal(ref["stack_lengths"][l].numpy(), G[f"collate_lengths_{l}"])
        for key in ("neighbors", "pools", "upsamples"):
            assert np.array_equal(ref[key][l].numpy().astype(np.int32), G[f"collate_{key}_{l}"]), (key, l)


@pytest.mark.skipif(not PREF.available(), reason="oracle/_ref not built")
def test_oracle_calibration_rule_equals_reference_text():
    """The rule the GPU test of calibrate_neighbors restates (histogram of un-capped counts, 80th percentile), driven by
    the oracle's collate, gives the caps the reference's calibrate_neighbors text gave (dataloader.py:200-232)."""
    cfg = kitti_config()
    hist_n = int(np.ceil(4 / 3 * np.pi * (cfg.deform_radius + 1) ** 3))
    hists = np.zeros((cfg.num_layers, hist_n), np.int64)
    for s in G["calib_seeds"]:
        a, b, _ = synth.make_pair(int(s), n_beams=16, n_azimuth=400)
        lens = np.array([len(a), len(b)], np.int32)
        pts, ln = PREF.subsample_batch(np.concatenate([a, b]), lens, sampleDl=cfg.first_subsampling_dl)
        ref = KO.collate(pts[:ln[0]], pts[ln[0]:], cfg, [hist_n] * 5)
        for l, m in enumerate(ref["neighbors"]):
            hists[l] += np.bincount((m < m.shape[0]).sum(1).numpy(), minlength=hist_n)[:hist_n]
        if hists.sum(1).min() > 2000:
            break
    cum = np.cumsum(hists.T, axis=0)
    assert np.array_equal((cum < 0.8 * cum[hist_n - 1]).sum(0), G["calib_caps"])
