"""CPU tests: the oracle against the golden fixtures, independent re-derivations and edge cases."""
import os

import numpy as np
import pytest
import torch

from apr_amd import synth
from oracle import match_pose_oracle as MO
from oracle import me_oracle as OME
from oracle import resunet_oracle as OR

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_pdist_pinned_by_reference_output():
    """pdist_ref.npz holds outputs of the reference's own lib/metrics.py:pdist."""
    g = np.load(os.path.join(GOLD, "pdist_ref.npz"))
    A, B = torch.from_numpy(g["A"]), torch.from_numpy(g["B"])
    assert torch.allclose(MO.pdist(A, B, 'SquareL2'), torch.from_numpy(g["d2"]), rtol=1e-6, atol=1e-7)
    assert torch.allclose(MO.pdist(A, B, 'L2'), torch.from_numpy(g["d"]), rtol=1e-6, atol=1e-7)
    # the fixed-order C restatement agrees with the reference's values and arg-min
    d2c = MO.sqdist_matrix(g["A"], g["B"])
    assert np.allclose(d2c, g["d2"], rtol=1e-5, atol=1e-6)
    idx, d2 = MO.feature_nn(g["A"], g["B"])
    assert np.array_equal(idx, g["argmin"])
    assert np.array_equal(d2, d2c[np.arange(len(idx)), idx])


def test_feature_nn_ties_and_ragged():
    F1 = np.zeros((5, 7), np.float32); F1[3] = 1
    F0 = np.zeros((2, 7), np.float32); F0[1] = 1
    idx, d2 = MO.feature_nn(F0, F1)
    assert list(idx) == [0, 3] and list(d2) == [0.0, 0.0]   # ties -> smallest index; c % 4 != 0 ok


def test_sparse_quantize_against_bruteforce():
    rng = np.random.default_rng(0)
    x = rng.uniform(-3, 3, (500, 3)).astype(np.float32)
    c, idx = OME.sparse_quantize(x, return_index=True)
    seen, exp = set(), []
    for i, v in enumerate(np.floor(x).astype(np.int32)):
        if tuple(v) not in seen:
            seen.add(tuple(v)); exp.append(i)
    assert list(idx) == exp and np.array_equal(c, np.floor(x[exp]).astype(np.int32))
    assert len(OME.sparse_quantize(np.zeros((0, 3), np.float32))) == 0


def test_kernel_map_and_conv_against_dense_conv3d():
    """The sparse conv restatement equals a dense zero-padded conv3d sampled at the output voxels."""
    rng = np.random.default_rng(1)
    G = 12
    occ = rng.random((G, G, G)) < 0.15
    xyz = np.argwhere(occ).astype(np.int32)            # (x,y,z)
    rng.shuffle(xyz)
    C = OME.batched_coordinates([xyz])
    cin, cout = 3, 5
    F = rng.standard_normal((len(C), cin)).astype(np.float32)
    W = torch.from_numpy(rng.standard_normal((27, cin, cout)).astype(np.float32))
    st = OME.SparseTensor(F, coordinates=C)
    out = OME.conv_forward(st, W, 3, 1).F
    dense = torch.zeros(1, cin, G, G, G)
    dense[0, :, xyz[:, 2], xyz[:, 1], xyz[:, 0]] = torch.from_numpy(F).t()   # [c, z, y, x]
    # offset index = (ox+1) + 3(oy+1) + 9(oz+1): weight[co, ci, kz, ky, kx] = W[idx, ci, co] (correlation)
    Wd = W.view(3, 3, 3, cin, cout).permute(4, 3, 0, 1, 2).contiguous()
    ref = torch.nn.functional.conv3d(dense, Wd, padding=1)[0][:, xyz[:, 2], xyz[:, 1], xyz[:, 0]].t()
    assert torch.allclose(out, ref, rtol=1e-4, atol=1e-5)
    # strided conv: output voxel (2i,2j,2k) gathers inputs at 2*idx + o
    out2 = OME.conv_forward(st, W, 3, 2)
    c2 = out2.C.numpy()
    assert (c2[:, 1:] % 2 == 0).all()
    dense_o = torch.nn.functional.conv3d(dense, Wd, padding=1)[0]
    ref2 = dense_o[:, c2[:, 3], c2[:, 2], c2[:, 1]].t()
    assert torch.allclose(out2.F, ref2, rtol=1e-4, atol=1e-5)
    # transposed conv back to the fine map is the adjoint of the strided conv's gather
    Wt = torch.from_numpy(rng.standard_normal((27, cout, cin)).astype(np.float32))
    up = OME.conv_forward(out2, Wt, 3, 2, transpose=True)
    assert up.coordinate_map_key == 1 and len(up.F) == len(C)
    nbr = st.coordinate_manager.get_map(1, 2, 3)
    exp = torch.zeros(len(C), cin)
    for j in range(nbr.shape[0]):
        for o in range(27):
            i = nbr[j, o]
            if i >= 0:
                exp[i] += out2.F[j] @ Wt[o]
    assert torch.allclose(up.F, exp, rtol=1e-4, atol=1e-5)


def test_fcgf_golden_config1():
    """BASELINE config 1 on CPU: 20 k-ray frame, voxel 0.3, ResUNetBN2C -> committed oracle features."""
    g = np.load(os.path.join(GOLD, "fcgf_small.npz"))
    xyz = synth.make_small_frame(0)
    assert len(xyz) == int(g["n_points"])
    c, sel = OME.sparse_quantize(xyz / np.float32(0.3), return_index=True)
    assert np.array_equal(c, g["coords"]) and np.array_equal(sel, g["sel"])
    C = OME.batched_coordinates([c])
    torch.manual_seed(0)
    om = OR.ResUNetBN2C(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3).eval()
    OR.randomize_bn_stats(om, 0)
    wsum = float(sum(p.detach().double().abs().sum() for p in om.parameters()))
    if abs(wsum - float(g["weight_abs_sum"])) > 1e-6 * wsum:
        pytest.skip("torch RNG stream differs from the one the fixture was generated with")
    st = OME.SparseTensor(np.ones((len(C), 1), np.float32), coordinates=C)
    with torch.no_grad():
        out = om(st).F.numpy()
    cm = st.coordinate_manager
    assert [len(cm.get_coords(s)) for s in (1, 2, 4, 8)] == list(g["sizes"])
    assert np.array_equal(cm.get_coords(8), g["coords_s8"])
    for (ti, to, k) in [(1, 1, 5), (1, 1, 3), (1, 2, 3), (2, 2, 3), (2, 4, 3), (4, 4, 3), (4, 8, 3), (8, 8, 3)]:
        assert int((cm.get_map(ti, to, k) >= 0).sum()) == int(g[f"p{ti}_{to}_{k}"])
    assert np.linalg.norm(out - g["features"]) / np.linalg.norm(g["features"]) < 1e-5
    assert np.allclose(np.linalg.norm(out, axis=1), 1.0, atol=1e-5)


def test_state_dict_keys_match_reference_layout():
    m = OR.ResUNetBN2C(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3)
    sd = m.state_dict()
    assert sd["conv1.kernel"].shape == (125, 1, 32)
    assert sd["block1.conv1.kernel"].shape == (27, 32, 32)
    assert sd["conv1_tr.kernel"].shape == (96, 64)           # kernel_size 1 -> [Cin, Cout]
    assert sd["final.bias"].shape == (1, 32)
    assert "norm4_tr.bn.running_var" in sd and "block2_tr.norm2.bn.num_batches_tracked" in sd
    from apr_amd.fcgf.model import load_model
    hm = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3)
    assert list(hm.state_dict().keys()) == list(sd.keys())
    assert all(hm.state_dict()[k].shape == sd[k].shape for k in sd)
    assert load_model("NoSuchNet") is None
    for name, nparam in [("ResUNetBN2C", 8753408), ("ResUNetFatBN", None)]:
        M = load_model(name)
        assert M is not None and M.__name__ == name


def test_pose_golden_and_analytic():
    g = np.load(os.path.join(GOLD, "pose_small.npz"))
    T = MO.est_quad_linear_robust(torch.from_numpy(g["p0"]), torch.from_numpy(g["p1"])).numpy()
    assert np.allclose(T, g["T_irls"], atol=1e-5)
    assert np.allclose(T, g["T_gt"], atol=5e-3)               # recovers the known (R,t)
    assert np.array_equal(MO.sample_indices(7, 0, 16, len(g["p0"])), g["sample_idx"])
    Tr, info = MO.ransac_feature_matching(g["p0"], g["q1"], g["corr"], 0.3, 0.9, max_iter=20000, seed=7)
    assert info["inliers"] == int(g["inliers"]) and info["n_valid"] == int(g["n_valid"])
    assert info["best_iteration"] == int(g["best_iteration"])
    assert np.allclose(Tr, g["T_ransac"], atol=1e-9)
    assert np.allclose(Tr, g["T_gt"], atol=2e-2)


def test_kabsch_recovers_rotation_and_handles_reflection_case():
    rng = np.random.default_rng(0)
    src = rng.standard_normal((6, 4, 3))
    a = 0.7
    R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
    tgt = src @ R.T + np.array([1.0, -2.0, 0.5])
    T = MO.kabsch(src, tgt)
    assert np.allclose(T[:, :3, :3], R, atol=1e-10) and np.allclose(T[:, :3, 3], [1.0, -2.0, 0.5], atol=1e-10)
    planar = src.copy(); planar[:, :, 2] = 0
    T = MO.kabsch(planar, planar @ R.T)
    assert np.allclose(np.linalg.det(T[:, :3, :3]), 1.0)      # proper rotation, never a reflection


def test_hardest_contrastive_restatement_runs():
    rng = np.random.default_rng(0)
    F0 = torch.from_numpy(rng.standard_normal((300, 16)).astype(np.float32))
    F1 = torch.from_numpy(rng.standard_normal((280, 16)).astype(np.float32))
    pos = np.stack([rng.integers(0, 300, 100), rng.integers(0, 280, 100)], 1)
    sel0, sel1 = rng.choice(300, 64, replace=False), rng.choice(280, 64, replace=False)
    p, n = MO.hardest_contrastive(F0, F1, pos, sel0, sel1, None)
    assert p.item() > 0 and n.item() >= 0
