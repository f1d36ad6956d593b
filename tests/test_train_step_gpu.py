"""The APR training iteration (apr_amd/fcgf/lib/complement_trainer.GenerativePairTrainStep = the loop body of
FCGF_APR/lib/complement_trainer.py:350-512) end to end on a small synthetic pair: both encoder call forms give the same
iteration, the loss goes down, nothing inside the step depends on the launch order (same bits run to run)."""
import numpy as np
import pytest
import torch

from apr_amd import ops, synth
from apr_amd.fcgf.lib import apg
from apr_amd.fcgf.lib.complement_trainer import GenerativePairTrainStep
from apr_amd.fcgf.model import load_model

pytestmark = pytest.mark.gpu


def _batch(dev, seed=3):
    xyz0, xyz1, T = synth.make_pair(seed, n_beams=16, n_azimuth=600)
    out, pts = {}, []
    rng = np.random.default_rng(seed)
    for tag, xyz in (("0", xyz0), ("1", xyz1)):
        key = torch.from_numpy(xyz).to(dev)
        m = ops.build_map(ops.voxelize(key, 0.3, 0), want_first=True)
        ops.finalize_maps([m])
        out[f"sinput{tag}_C"] = m.coords
        out[f"sinput{tag}_F"] = torch.ones((m.n, 1), device=dev)
        p = key[m.first.long()].contiguous()
        pts.append(p)
        # a stand-in APG cloud: the frame's own voxel points, jittered and doubled (denser than the key frame, as APG is)
        jit = torch.from_numpy(rng.normal(0, 0.1, (2 * m.n, 3)).astype(np.float32)).to(dev)
        out[f"pcd_nghb{tag}"] = [(p.repeat(2, 1) + jit).contiguous()]
    out["correspondences"] = apg.get_matching_indices(pts[0], pts[1], torch.from_numpy(T).float().to(dev), 0.45).cpu()
    out["len_batch"] = [[int(out["sinput0_C"].shape[0]), int(out["sinput1_C"].shape[0])]]
    return out


def _step(dev, stack):
    torch.manual_seed(0)
    enc = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3).to(dev)
    gen = apg.GenerativeMLP_54(in_channel=32, out_points=4, bn_momentum=0.05).to(dev)
    opt = torch.optim.SGD([{'params': enc.parameters()}, {'params': gen.parameters()}], lr=0.05, momentum=0.8, weight_decay=1e-4)
    st = GenerativePairTrainStep(enc, gen, opt, point_generation_ratio=4, regularization_strength=0.1, loss_ratio=2e-3,
                                 num_pos_per_batch=256, num_hn_samples_per_batch=128)
    st.stack_frames = stack
    return st, enc, gen


def test_train_step_runs_learns_and_both_encoder_call_forms_agree(dev):
    batch = _batch(dev)
    assert len(batch["correspondences"]) > 300
    hist = {}
    for stack in (True, False, True):
        st, enc, gen = _step(dev, stack)
        losses = []
        for it in range(6):
            np.random.seed(it)
            r = st(batch, timed=(it == 0))
            losses.append(float(r["loss"]))
            assert np.isfinite(losses[-1])
        hist.setdefault(stack, []).append((losses, [p.detach().clone() for p in enc.parameters()],
                                           {k: v.clone() for k, v in enc.state_dict().items() if "num_batches" in k}))
    (la, pa, na), (lc, pc, nc) = hist[True]
    lb, pb, nb = hist[False][0]
    assert la == lc and all(torch.equal(u, v) for u, v in zip(pa, pc))          # the same bits run to run
    assert la[-1] < la[0]                                                       # it learns
    assert all(int(v) == 12 for v in na.values()) and all(int(v) == 12 for v in nb.values())   # 6 iterations x 2 calls
    assert abs(la[0] - lb[0]) < 1e-4 * abs(lb[0])                               # first iteration: the same numbers
    assert max(abs(a - b) for a, b in zip(la, lb)) < 2e-2 * abs(lb[0])          # then two fp32 trajectories of the same run


def test_train_step_with_a_batch_of_two_pairs(dev):
    """batch_size 2 (scripts/train_apr_kitti.sh trains with 4): each frame tensor holds two clouds (BatchNorm statistics over
    both, as MinkowskiBatchNorm does), correspondences carry the collate's row offsets (complement_data_loader.py:1240-1256),
    the NPR loss loops over the clouds (complement_trainer.py:424-449).  Stacked and separate encoder calls agree."""
    b0, b1 = _batch(dev, seed=3), _batch(dev, seed=5)
    n00, n01 = b0["sinput0_C"].shape[0], b0["sinput1_C"].shape[0]
    shift = lambda C, k: torch.cat((C[:, :1] + k, C[:, 1:]), 1)
    batch = {}
    for tag in ("0", "1"):
        batch[f"sinput{tag}_C"] = torch.cat((b0[f"sinput{tag}_C"], shift(b1[f"sinput{tag}_C"], 1)), 0).contiguous()
        batch[f"sinput{tag}_F"] = torch.cat((b0[f"sinput{tag}_F"], b1[f"sinput{tag}_F"]), 0)
        batch[f"pcd_nghb{tag}"] = b0[f"pcd_nghb{tag}"] + b1[f"pcd_nghb{tag}"]
    off = torch.tensor([[n00, n01]])
    batch["correspondences"] = torch.cat((b0["correspondences"], b1["correspondences"] + off), 0)
    batch["len_batch"] = b0["len_batch"] + b1["len_batch"]
    res = {}
    for stack in (True, False):
        st, enc, gen = _step(dev, stack)
        np.random.seed(0)
        r = st(batch)
        res[stack] = (float(r["loss"]), float(r["pos_loss"]), float(r["neg_loss"]),
                      {k: v.clone() for k, v in enc.state_dict().items() if k.endswith("running_mean")})
    assert np.isfinite(res[True][0]) and abs(res[True][0] - res[False][0]) < 1e-4 * abs(res[False][0])
    assert abs(res[True][1] - res[False][1]) < 1e-4 and abs(res[True][2] - res[False][2]) < 1e-4
    for k, v in res[True][3].items():
        assert torch.allclose(v, res[False][3][k], rtol=1e-4, atol=1e-6), k
