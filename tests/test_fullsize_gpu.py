"""BASELINE configs 3 and 5 and the APR encoder (ResUNetFatBN, 128 features) at FULL size (2 x ~118 k-point frames):
the HIP path against the reference's own C++ index build (oracle/_ref), the CPU oracles, and size-independent
properties.  Round 1 covered these workloads with 16-beam pairs only."""
import numpy as np
import pytest
import torch

from apr_amd import MinkowskiEngine as ME
from apr_amd import ops, synth
from apr_amd.fcgf.lib import apg
from apr_amd.predator import point_ops
from apr_amd.predator.configs.models import kitti_config
from apr_amd.predator.datasets.dataloader import collate_fn_descriptor
from apr_amd.predator.models.architectures import KPFCNN
from apr_amd.predator.pipeline import PredatorRegistration
from oracle import kpfcnn_oracle as KO
from oracle import me_oracle as OME
from oracle import predator_points_oracle as PREF
from tests.helpers import model_pair, rel_l2

pytestmark = pytest.mark.gpu
LIMITS = [58, 59, 58, 57]        # 80th-percentile caps of this generator at full size (calibrate_neighbors)


@pytest.fixture(scope="module")
def full_pair():
    a, b, T = synth.make_pair(0)
    return a, b, T


@pytest.fixture(scope="module")
def predator_full(full_pair, dev):
    """Full-size pair through the reference C++ (subsample at 0.3 m, collate) and through the GPU collate."""
    if not PREF.available():
        pytest.skip("oracle/_ref not built")
    a, b, _ = full_pair
    lens = np.array([len(a), len(b)], np.int32)
    ref_pts, ref_len = PREF.subsample_batch(np.concatenate([a, b]), lens, sampleDl=0.3)
    cfg = kitti_config()
    ref_batch = KO.collate(ref_pts[:ref_len[0]], ref_pts[ref_len[0]:], cfg, LIMITS)
    pts, glen = point_ops.grid_subsample(torch.cat([torch.from_numpy(a), torch.from_numpy(b)]).to(dev), lens, 0.3)
    return cfg, ref_pts, ref_len, ref_batch, pts, glen


def test_config3_subsample_and_collate_match_reference_cpp(predator_full, dev):
    cfg, ref_pts, ref_len, ref_batch, pts, glen = predator_full
    assert np.array_equal(np.asarray(glen), ref_len) and 12000 < ref_len[0] < 18000
    # barycentres of the 0.3 m grid: identical per cell (row order differs: libstdc++ unordered_map)
    assert np.array_equal(PREF.canonical_rows(pts.cpu().numpy(), ref_len), PREF.canonical_rows(ref_pts, ref_len))
    # feed the GPU collate the reference's rows so that level 0 is in the same order on both sides
    src, tgt = ref_pts[:ref_len[0]], ref_pts[ref_len[0]:]
    got = collate_fn_descriptor([(src, tgt, np.ones((len(src), 1), np.float32), np.ones((len(tgt), 1), np.float32))],
                                cfg, LIMITS)
    assert [len(p) for p in got["points"]] == [len(p) for p in ref_batch["points"]]
    assert [n.shape[1] for n in got["neighbors"]] == [n.shape[1] for n in ref_batch["neighbors"]]
    from scipy.spatial import cKDTree
    for l in range(4):
        gp, rp = got["points"][l].cpu().numpy(), ref_batch["points"][l].numpy()
        gl, rl = got["stack_lengths"][l].numpy(), ref_batch["stack_lengths"][l].numpy()
        assert np.array_equal(gl, rl)
        if l <= 1:      # same input order on both sides -> bit-identical barycentres
            assert np.array_equal(PREF.canonical_rows(gp, gl), PREF.canonical_rows(rp, rl))
        else:           # levels >= 2 sum level-1 points whose ROW ORDER differs (unordered_map): last-bit differences,
            s0 = 0      # so match the two point sets geometrically, cloud by cloud
            for n in gl:
                d, j = cKDTree(rp[s0:s0 + n]).query(gp[s0:s0 + n])
                assert d.max() < 1e-4, (l, float(d.max()))
                assert len(np.unique(j)) == n, (l, n, len(np.unique(j)))
                s0 += n
    # level-0 tables.  `neighbors`: rows and values both index level 0 (same order on both sides) -> entry by entry.
    g0, r0 = got["neighbors"][0].cpu().long(), ref_batch["neighbors"][0]
    assert (g0 == r0).float().mean() > 0.995          # positions may differ only inside runs of equal distance
    assert (g0.sort(1)[0] == r0.sort(1)[0]).float().mean() > 0.9995
    # `pools` (rows = level-1 points) and `upsamples` (values = level-1 indices): level 1 holds the SAME points in a
    # different row order (unordered_map), so translate through the exact coordinate match before comparing sets
    g1, r1 = got["points"][1].cpu().numpy(), ref_batch["points"][1].numpy()
    where = {p.tobytes(): i for i, p in enumerate(r1)}
    perm = np.array([where[p.tobytes()] for p in g1])              # got level-1 row -> reference level-1 row
    n0, n1 = len(ref_pts), len(r1)
    gp, rp = got["pools"][0].cpu().numpy(), ref_batch["pools"][0].numpy()[perm]
    assert gp.shape == rp.shape and (np.sort(gp, 1) == np.sort(rp, 1)).mean() > 0.9995
    gu = got["upsamples"][0].cpu().numpy()
    gu = np.where(gu < n1, perm[np.minimum(gu, n1 - 1)], n1)
    ru = ref_batch["upsamples"][0].numpy()
    assert gu.shape == ru.shape and (np.sort(gu, 1) == np.sort(ru, 1)).mean() > 0.9995
    assert n0 == len(got["points"][0])


def test_config3_kpfcnn_matches_cpu_oracle_and_is_reproducible(predator_full, dev):
    cfg, ref_pts, ref_len, ref_batch, _, _ = predator_full
    np.random.seed(0)
    torch.manual_seed(0)
    model = KPFCNN(cfg).to(dev).eval()
    ref = KO.kpfcnn_forward({k: v.cpu() for k, v in model.state_dict().items()}, cfg, ref_batch)
    src, tgt = ref_pts[:ref_len[0]], ref_pts[ref_len[0]:]
    mk = lambda: collate_fn_descriptor([(src, tgt, np.ones((len(src), 1), np.float32),
                                         np.ones((len(tgt), 1), np.float32))], cfg, LIMITS)
    feats, ov, sal = model(mk())
    assert feats.shape == (len(ref_pts), 32)
    assert torch.allclose(feats.norm(dim=1).cpu(), torch.ones(len(feats)), atol=1e-5)
    assert rel_l2(feats.cpu(), ref[0]) < 1e-4                         # north_star: features 1e-4 relative
    assert (ov.cpu() - ref[1]).abs().max() < 1e-4 and (sal.cpu() - ref[2]).abs().max() < 1e-4
    f2, ov2, sal2 = model(mk())                                       # run-to-run bit identity
    assert torch.equal(feats, f2) and torch.equal(ov, ov2) and torch.equal(sal, sal2)


def test_config3_pipeline_recovers_the_pose_when_features_match(full_pair, dev):
    """The whole Predator pair path at full size.  A random-init network has no matching power, so the RANSAC stage is
    given geometry-derived descriptors here (the point of this test is the path and its shapes at size, not learning):
    the ground-truth pose must come back within the tester's success bounds (RTE < 2 m, RRE < 5 deg,
    Predator_APR/lib/tester.py:99-105)."""
    from apr_amd.predator.lib import benchmark_utils as BU
    a, b, T = full_pair
    cfg = kitti_config()
    np.random.seed(0)
    torch.manual_seed(0)
    pipe = PredatorRegistration(KPFCNN(cfg).to(dev).eval(), cfg, LIMITS)
    src, tgt, feats, ov, sal = pipe.encode(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev))
    assert feats.shape == (len(src) + len(tgt), 32) and bool(torch.isfinite(feats).all())
    assert float(ov.min()) >= 0 and float(ov.max()) <= 1 and float(sal.min()) >= 0 and float(sal.max()) <= 1
    # descriptors = world position of the point (source moved by the ground truth): matches are the true neighbours
    Tt = torch.from_numpy(T.astype(np.float32)).to(dev)
    f_src = src @ Tt[:3, :3].t() + Tt[:3, 3]
    rng = np.random.RandomState(0)
    s_p, s_f, _ = BU.sample_by_score(src, f_src, torch.ones(len(src), device=dev), 5000, rng=rng)
    t_p, t_f, _ = BU.sample_by_score(tgt, tgt, torch.ones(len(tgt), device=dev), 5000, rng=rng)
    pad = lambda f: torch.cat([f, torch.zeros((len(f), 29), device=dev)], 1).contiguous()
    Tm, info = BU.ransac_pose_estimation(s_p, t_p, pad(s_f), pad(t_f), distance_threshold=0.3, ransac_n=4, seed=1,
                                         return_info=True)
    rte = np.linalg.norm(Tm[:3, 3] - T[:3, 3])
    rre = np.degrees(np.arccos(np.clip((np.trace(Tm[:3, :3].T @ T[:3, :3]) - 1) / 2, -1, 1)))
    assert rte < 2.0 and rre < 5.0, (rte, rre, info)


def test_fatbn128_fullsize_matches_oracle(dev):
    """APR's real encoder (FCGF_APR/scripts/train_apr_kitti.sh:12-13: ResUNetFatBN, 128 output features) on one full
    118 k-point frame against the CPU oracle; rows stay in voxel order, unit norm."""
    om, hm = model_pair("ResUNetFatBN", out_channels=128)
    om.eval(); hm.eval()
    xyz = synth.make_frame(3)
    c, sel = OME.sparse_quantize(xyz / np.float32(0.3), return_index=True)
    C = OME.batched_coordinates([c])
    with torch.no_grad():
        ref = om(OME.SparseTensor(np.ones((len(C), 1), np.float32), coordinates=C)).F
        out = hm(ME.SparseTensor(torch.ones((len(C), 1), device=dev), coordinates=torch.from_numpy(C).to(dev))).F
    assert out.shape == (len(C), 128) and len(C) > 12000
    assert rel_l2(out.cpu(), ref) < 2e-5
    assert torch.allclose(out.norm(dim=1).cpu(), torch.ones(len(C)), atol=1e-5)


def test_config5_apg_fatbn_npr_at_size(dev):
    """LoNuScenes-shaped distant pair: 16-beam source (~30 k pts) vs 64-beam target (~118 k pts) 40 m apart, APG over
    10 complement frames (~300 k pts), FatBN-128 encode of the key frame, NPR reconstruction loss."""
    import scripts.bench_config5 as C5
    xyz0, xyz1, T = synth.make_pair(0, n_beams1=16, dist=40.0)
    assert 25000 < len(xyz0) < 35000 and 100000 < len(xyz1) < 125000 and abs(np.linalg.norm(T[:3, 3]) - 40.0) < 1e-6
    frames, poses = C5.complement_frames(0, 16, 0.0)
    assert len(frames) == 10
    key = torch.from_numpy(xyz0).to(dev)
    nghb, sel = apg.aggregate_frames(key, [torch.from_numpy(f).to(dev) for f in frames], poses, 0.3)
    # numpy restatement of complement_data_loader.py:65-70, 576-579, 620-628 at full size
    moved = np.concatenate([f @ M.astype(np.float32)[:3, :3].T + M.astype(np.float32)[:3, 3]
                            for f, M in zip(frames, poses)])
    ref = moved[(moved ** 2).sum(-1) < np.max((xyz0 ** 2).sum(-1))]
    assert abs(len(nghb) - len(ref)) <= 3
    if len(nghb) == len(ref):
        assert np.allclose(nghb.cpu().numpy(), ref, atol=1e-4)
    # voxel subset: one point per occupied 0.3 m voxel, all distinct voxels
    v = torch.floor(nghb[sel] / 0.3).to(torch.int64)
    assert len(torch.unique(v, dim=0)) == len(sel)
    _, ref_sel = OME.sparse_quantize(ref / np.float32(0.3), return_index=True)
    assert abs(len(sel) - len(ref_sel)) <= 3
    # encoder + NPR loss: finite, and equal to the brute-force value on a sub-sample of the generated points
    torch.manual_seed(0)
    from apr_amd.fcgf.model import load_model
    enc = load_model("ResUNetFatBN")(1, 128, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3)
    enc = enc.to(dev).eval()
    gen = apg.GenerativeMLP_98(in_channel=128, out_points=4).to(dev).eval()
    m = ops.build_map(ops.voxelize(key, 0.3, 0), want_first=True)
    ops.finalize_maps([m])
    with torch.no_grad():
        F = enc(ME.SparseTensor(torch.ones((m.n, 1), device=dev), coordinates=m.coords)).F
    assert F.shape == (m.n, 128) and bool(torch.isfinite(F).all())
    cloud = nghb[sel]
    loss = apg.npr_reconstruction_loss(gen, F, m.coords[:, 1:], cloud, 0.3, 4)
    assert np.isfinite(float(loss)) and float(loss) > 0
    g = gen(F) * 0.3
    pts = apg.npr_points(g, m.coords[:, 1:], 0.3, 4)
    sub = pts[:: max(1, len(pts) // 2000)][:2000]
    d = torch.cdist(sub.double(), cloud.double()).min(1)[0] ** 2
    assert abs(float(apg.chamfer_sum(sub, cloud)) - float(d.sum())) < 1e-6 * float(d.sum())
    assert float(apg.chamfer_distance(cloud[:5000], cloud[:5000])) < 1e-9


def test_config5_distant_pair_registration_end_to_end(dev):
    """BASELINE config 5, the registration itself (FCGF_APR/lib/complement_trainer.py:514-681, scripts/test_apr.py:130-163):
    a 16-beam source (~30 k points) against the 64-beam target 40 m away -- a 4x density ratio -- through both FatBN-128
    encodes, feature NN, RANSAC(4 M) and Kabsch in one batched call.  A random-init encoder cannot match, so the check of
    the POSE uses descriptors planted after the encode (30 % of the source voxels carry their true match, the rest
    noise); the encoder's own output is checked for shape / unit norm at both densities."""
    from apr_amd import ops, synth
    from apr_amd.fcgf import registration
    from apr_amd.fcgf.lib import apg
    from apr_amd.fcgf.pipeline import PairRegistration
    xyz0, xyz1, T = synth.make_pair(0, n_beams1=16, dist=40.0)
    assert 25000 < len(xyz0) < 35000 and len(xyz1) > 3.5 * len(xyz0)
    from tests.helpers import model_pair
    _, hm = model_pair("ResUNetFatBN", out_channels=128)
    pipe = PairRegistration(hm, voxel_size=0.3, ransac_iters=4000000)
    d0, d1 = torch.from_numpy(xyz0).to(dev), torch.from_numpy(xyz1).to(dev)
    _, q0, q1, m0, m1 = pipe.voxelize_pair(d0, d1)
    seen = {}

    def plant(F, counts, prs):
        assert counts == [m0, m1] and F.shape == (m0 + m1, 128)
        seen["norm"] = F.norm(dim=1).clone()
        gt = apg.get_matching_indices(q0, q1, T, 0.3, K=1)
        g = torch.Generator(device="cpu").manual_seed(0)
        H1 = torch.nn.functional.normalize(torch.randn(m1, 128, generator=g), dim=1).to(dev)
        H0 = torch.nn.functional.normalize(torch.randn(m0, 128, generator=g), dim=1).to(dev)
        pk = gt[torch.randperm(len(gt), generator=g)[:int(0.3 * m0)].to(dev)]
        H0[pk[:, 0]] = H1[pk[:, 1]]
        F[:m0], F[m0:] = H0, H1
        return F

    pipe.feature_hook = plant
    (T_est, info), = pipe.register_batch([(d0, d1)], seeds=[1])
    assert torch.allclose(seen["norm"].cpu(), torch.ones(m0 + m1), atol=1e-5)
    rte, rre = registration.rte_rre(T_est, T)
    assert info["n0"] == m0 and info["n1"] == m1 and info["n_valid"] > 1000
    assert rte < 0.3 and rre < 0.5, (rte, rre, info)


def test_bn2c_headline_batch_matches_oracle_and_single_frames(dev):
    """The headline's encoder call at its real shape: ResUNetBN2C / 32 on a batched sparse tensor of 6 full 118 k-point
    frames (a bench step has 12) -- the shape at which the 64-channel residual blocks run on the output-stationary kernel
    with tiles of several hundred rows, the 128- / 256-channel levels on the weight-stationary pair and the rest on the
    tile kernel.  Every frame's rows against the CPU oracle's encode of that frame alone (north_star: 1e-4 relative;
    asserted 2e-5), and the same call with every stage forced onto the tile kernel (exact-fp32 MFMA) for the default
    path's own deviation."""
    import os
    from apr_amd.fcgf.pipeline import PairRegistration
    om, hm = model_pair("ResUNetBN2C", out_channels=32)
    om.eval(); hm.eval()
    frames = [synth.make_frame(s) for s in range(6)]
    pipe = PairRegistration(hm, voxel_size=0.3)
    clouds = [torch.from_numpy(f).to(dev) for f in frames]
    cm, counts, first, offs, pts_all = pipe.voxelize_batch(clouds)
    F = pipe.encode_batch(cm)
    assert F.shape == (sum(counts), 32) and sum(counts) > 70000
    o = 0
    for b, xyz in enumerate(frames):
        c, sel = OME.sparse_quantize(xyz / np.float32(0.3), return_index=True)
        C = OME.batched_coordinates([c])
        with torch.no_grad():
            ref = om(OME.SparseTensor(np.ones((len(C), 1), np.float32), coordinates=C)).F
        assert counts[b] == len(C)
        assert rel_l2(F[o:o + counts[b]].cpu(), ref) < 2e-5, b
        o += counts[b]
    again = pipe.encode_batch(cm)
    assert torch.equal(again, F)                                   # bitwise reproducible, run to run
    # ... and the same bits whether the plan leaves through ONE library call (apr_resunet_encode, the default) or stage by
    # stage from Python
    from apr_amd.fcgf.model import resunet as R
    assert R.ENCODE_PLAN
    R.ENCODE_PLAN = False
    try:
        staged = pipe.encode_batch(pipe.voxelize_batch(clouds)[0])
    finally:
        R.ENCODE_PLAN = True
    assert torch.equal(staged, F)
    os.environ["APR_WS_STAGES"], os.environ["APR_OS_STAGES"] = "none", "none"
    try:
        tile = pipe.encode_batch(pipe.voxelize_batch(clouds)[0])
    finally:
        os.environ.pop("APR_WS_STAGES"); os.environ.pop("APR_OS_STAGES")
    assert rel_l2(F.cpu(), tile.cpu()) < 5e-6
