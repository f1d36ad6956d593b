"""FCGF encoder parity: HIP ResUNet vs the CPU oracle on the same weights and inputs.

Bar (BASELINE.json north_star): relative L2 feature error < 1e-4 (we assert 2e-5).
"""
import numpy as np
import pytest
import torch

from apr_amd import MinkowskiEngine as ME
from oracle import me_oracle as OME
from tests.helpers import batched_input, model_pair, rel_l2

pytestmark = pytest.mark.gpu
TOL = 2e-5  # relative L2; the stated bar is 1e-4


def _run_both(om, hm, C, F, dev):
    with torch.no_grad():
        ref = om(OME.SparseTensor(F, coordinates=C)).F
        x = ME.SparseTensor(torch.from_numpy(F).to(dev), coordinates=torch.from_numpy(C).to(dev))
        out = hm(x)
    return ref, out


@pytest.mark.parametrize("name,n_out", [("ResUNetBN2C", 32), ("ResUNetFatBN", 128), ("ResUNetBN2", 32),
                                         ("ResUNetBN2B", 16), ("ResUNetBN2E", 32)])
def test_eval_fused_matches_oracle(dev, name, n_out):
    """BASELINE config 1: one synthetic 20 k-point frame, voxel 0.3, eval mode."""
    om, hm = model_pair(name, n_out)
    om.eval(); hm.eval()
    C, F = batched_input([0])
    ref, out = _run_both(om, hm, C, F, dev)
    assert out.F.shape == ref.shape
    assert rel_l2(out.F.cpu(), ref) < TOL
    # row order is the input order (scripts index xyz[i] against F[i], test_apr.py:142-146)
    assert np.array_equal(out.C.cpu().numpy(), C)
    assert torch.allclose(out.F.norm(dim=1).cpu(), torch.ones(len(C)), atol=1e-5)


def test_eval_modular_matches_fused_and_oracle(dev):
    om, hm = model_pair("ResUNetBN2C", 32)
    om.eval(); hm.eval()
    C, F = batched_input([1, 2])          # batch of two clouds
    ref, out_f = _run_both(om, hm, C, F, dev)
    hm.use_fused = False
    _, out_m = _run_both(om, hm, C, F, dev)
    assert rel_l2(out_m.F.cpu(), ref) < TOL
    assert rel_l2(out_f.F.cpu(), ref) < TOL
    assert rel_l2(out_f.F.cpu(), out_m.F.cpu()) < TOL


@pytest.mark.parametrize("name,n_out,fused", [("ResUNetBN2C", 32, True), ("ResUNetBN2C", 32, False),
                                               ("ResUNetFatBN", 128, True)])
def test_unit_feature_input_runs_conv1_on_occupancy_with_the_same_bits(dev, name, n_out, fused, monkeypatch):
    """FCGF's input features are ones: flagged `unit_features`, conv1 runs from the occupancy bitmap (no 5^3 kernel
    map is built) and the encoder's output is bit-identical to the kernel-map path and within the bar of the oracle."""
    om, hm = model_pair(name, n_out)
    om.eval(); hm.eval()
    hm.use_fused = fused
    C, F = batched_input([5, 6])
    assert np.all(F == 1.0) and F.shape[1] == 1
    Cd, Fd = torch.from_numpy(C).to(dev), torch.from_numpy(F).to(dev)
    with torch.no_grad():
        ref = om(OME.SparseTensor(F, coordinates=C)).F
        plain = hm(ME.SparseTensor(Fd, coordinates=Cd))
        x = ME.SparseTensor(Fd, coordinates=Cd, unit_features=True)
        occ = hm(x)
        k1 = hm.conv1.kernel_size
        assert (1, 1, k1, False) in plain.coordinate_manager._kmaps
        assert (1, 1, k1, False) not in x.coordinate_manager._kmaps         # the table was never needed
        assert not occ.unit_features                                         # never inherited by outputs
        monkeypatch.setenv("APR_OCC_CONV", "0")
        off = hm(ME.SparseTensor(Fd, coordinates=Cd, unit_features=True))
        assert (1, 1, k1, False) in off.coordinate_manager._kmaps
    assert torch.equal(occ.F, plain.F) and torch.equal(off.F, plain.F)
    assert rel_l2(occ.F.cpu(), ref) < TOL
    # the promise can be checked: a caller that flags features which are not ones is told so
    monkeypatch.delenv("APR_OCC_CONV")
    monkeypatch.setenv("APR_CHECK_UNIT_FEATURES", "1")
    with torch.no_grad():
        hm(ME.SparseTensor(Fd, coordinates=Cd, unit_features=True))
        with pytest.raises(Exception, match="not all 1.0"):
            hm(ME.SparseTensor(Fd * 2.0, coordinates=Cd, unit_features=True))


def test_batched_equals_separate(dev):
    """Encoding two clouds in one batch == encoding them one by one (eval BN), as the eval script does."""
    om, hm = model_pair("ResUNetBN2C", 32)
    hm.eval()
    C, F = batched_input([3, 4])
    with torch.no_grad():
        both = hm(ME.SparseTensor(torch.from_numpy(F).to(dev), coordinates=torch.from_numpy(C).to(dev))).F.cpu()
        outs = []
        for b in (0, 1):
            m = C[:, 0] == b
            Cb = C[m].copy(); Cb[:, 0] = 0
            outs.append(hm(ME.SparseTensor(torch.from_numpy(F[m]).to(dev),
                                           coordinates=torch.from_numpy(Cb).to(dev))).F.cpu())
    assert rel_l2(both, torch.cat(outs)) < 1e-6


def test_train_mode_batchnorm_matches_oracle(dev):
    """Training-mode forward (batch statistics over all clouds) + running-stat update."""
    om, hm = model_pair("ResUNetBN2C", 32)
    om.train(); hm.train()
    C, F = batched_input([5, 6])
    ref, out = _run_both(om, hm, C, F, dev)
    assert rel_l2(out.F.cpu(), ref) < 1e-4
    osd, hsd = om.state_dict(), hm.state_dict()
    for k in osd:
        if "running" in k:
            assert torch.allclose(hsd[k].cpu(), osd[k], rtol=1e-4, atol=1e-5), k
        if "num_batches_tracked" in k:
            assert int(hsd[k]) == int(osd[k]) == 1


def test_instance_norm_variant(dev):
    om, hm = model_pair("ResUNetIN2C", 32)
    om.eval(); hm.eval()
    C, F = batched_input([7, 8])
    ref, out = _run_both(om, hm, C, F, dev)
    assert rel_l2(out.F.cpu(), ref) < 1e-4


def test_color_input_and_k7(dev):
    """in_channels=3 (the ctor default, resunet.py:19) and conv1_kernel_size=7 (FCGF 3DMatch setting)."""
    om, hm = model_pair("ResUNetBN2C", 32, conv1_kernel_size=7, in_channels=3)
    om.eval(); hm.eval()
    C, _ = batched_input([9])
    F = np.random.default_rng(0).standard_normal((len(C), 3)).astype(np.float32)
    ref, out = _run_both(om, hm, C, F, dev)
    assert rel_l2(out.F.cpu(), ref) < TOL


def test_decomposed_and_autograd_switch(dev):
    _, hm = model_pair("ResUNetBN2C", 32)
    hm.eval()
    C, F = batched_input([1, 2])
    x = ME.SparseTensor(torch.from_numpy(F).to(dev), coordinates=torch.from_numpy(C).to(dev))
    # grad enabled: the modules record autograd (training path, tests/test_backward_gpu.py) instead of the fused plan
    y = hm(x)
    assert y.F.requires_grad and y.F.grad_fn is not None
    with torch.no_grad():
        out = hm(x)
    assert not out.F.requires_grad
    assert rel_l2(y.F.detach().cpu(), out.F.cpu()) < 1e-5          # same numbers on both paths (eval-mode BN)
    coords, feats = out.decomposed_coordinates_and_features
    assert len(coords) == 2 and sum(len(c) for c in coords) == len(C)
    assert np.array_equal(coords[1].cpu().numpy(), C[C[:, 0] == 1][:, 1:])
    assert len(out) == len(C)


def test_full_size_properties_translation_and_batch_order(dev):
    """Size-independent properties at BASELINE's full frame size (2 x ~118 k points, where the oracle is too slow to be
    the checker): (1) shifting every coordinate by a multiple of the coarsest tensor stride (8 voxels: the strided maps snap to
    floor(c / 2ts) * 2ts) changes the hash layout but not a single kernel-map row, so the features must be BIT-identical; (2) two encodes of the same input are bit-identical although
    the weight-stationary pair lists are laid out by atomics (positions vary run to run, results must not);
    (3) swapping the two frames of the batch permutes the rows; a row's partial sums are then grouped differently (the
    tile kernel deals the ACTIVE offsets of a 64-row tile to its 4 waves), so this one holds to summation-order
    rounding, not bit for bit."""
    from apr_amd import ops, synth
    _, hm = model_pair("ResUNetBN2C", 32)
    hm.eval()
    xyz0, xyz1, _ = synth.make_pair(0)
    frames = []
    for b, xyz in enumerate((xyz0, xyz1)):
        m = ops.build_map(ops.voxelize(torch.from_numpy(xyz).to(dev), 0.3, b))
        ops.finalize_maps([m])
        frames.append(m.coords.clone())
    assert min(len(f) for f in frames) > 10000

    def encode(C):
        with torch.no_grad():
            return hm(ME.SparseTensor(torch.ones((len(C), 1), device=dev), coordinates=C)).F

    C = torch.cat(frames)
    base = encode(C)
    assert torch.equal(base, encode(C))                                   # (2)
    shift = torch.tensor([0, 40, -512, 8], dtype=torch.int32, device=dev)
    assert torch.equal(base, encode(C + shift))                           # (1)
    f0, f1 = frames[0].clone(), frames[1].clone()
    f0[:, 0], f1[:, 0] = 1, 0
    swapped = encode(torch.cat([f1, f0]))                                 # (3)
    n0, n1 = len(frames[0]), len(frames[1])
    assert rel_l2(swapped[:n1].cpu(), base[n0:].cpu()) < 1e-6 and rel_l2(swapped[n1:].cpu(), base[:n0].cpu()) < 1e-6


@pytest.mark.parametrize("name,n_out,seeds", [("ResUNetBN2C", 32, [0]), ("ResUNetBN2C", 32, [1, 2, 3]), ("ResUNetFatBN", 128, [4, 5]),
                                               ("ResUNetBN2E", 32, [6]), ("ResUNetBN2", 32, [7, 8])])
def test_one_call_encode_plan_equals_the_stage_by_stage_plan(dev, name, n_out, seeds, monkeypatch):
    """apr_resunet_encode (the whole fused plan enqueued from C over one arena) == forward_fused walking its stages from
    Python, bit for bit: the same kernels with the same arguments in the same order.  Also with other routing switches,
    and twice on one coordinate manager (the second call clears counter blocks of its own)."""
    from apr_amd.fcgf.model import resunet as R
    om, hm = model_pair(name, n_out)
    hm.eval()
    C, F = batched_input(seeds)
    Cd, Fd = torch.from_numpy(C).to(dev), torch.from_numpy(F).to(dev)

    def run(plan, x=None):
        monkeypatch.setattr(R, "ENCODE_PLAN", plan)
        if x is None:
            x = ME.SparseTensor(Fd, coordinates=Cd, unit_features=True)
        with torch.no_grad():
            return hm(x), x

    ref, xr = run(False)
    assert xr.coordinate_manager._kmaps                       # the stage walk filled the manager's caches ...
    got, xg = run(True)
    assert not xg.coordinate_manager._kmaps                   # ... the one-call plan never touched them
    assert torch.equal(got.F, ref.F)
    again, _ = run(True, xg)
    assert torch.equal(again.F, ref.F)
    for env in ({"APR_WS3": "0"}, {"APR_OS_STAGES": "none"}, {"APR_WS_STAGES": "none", "APR_OS_STAGES": "none"},
                {"APR_OS_MIN_ROWS": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        monkeypatch.setattr(R, "OS_MIN_ROWS", int(env.get("APR_OS_MIN_ROWS", R.OS_MIN_ROWS)))
        a, _ = run(False)
        b, _ = run(True)
        assert torch.equal(a.F, b.F), env
        for k in env:
            monkeypatch.delenv(k)


@pytest.mark.parametrize("npts", [1, 7, 300])
def test_one_call_front_end_and_encode_on_tiny_clouds(dev, npts, monkeypatch):
    """The stage-level calls at the small end: frames of 1 / 7 / 300 points (maps of a single row on every level, tiles and
    pair lists with one entry) through apr_voxel_pyramid + apr_resunet_encode == the tensor-by-tensor front end + the stage
    walk, bit for bit, and finite."""
    from apr_amd.fcgf import pipeline as P
    from apr_amd.fcgf.model import resunet as R
    from apr_amd.fcgf.pipeline import PairRegistration
    _, hm = model_pair("ResUNetBN2C", 32)
    hm.eval()
    rng = np.random.default_rng(npts)
    clouds = [torch.from_numpy((rng.standard_normal((npts, 3)) * 3).astype(np.float32)).to(dev) for _ in range(2)]
    pipe = PairRegistration(hm, voxel_size=0.3)
    outs = []
    for one_call in (True, False):
        monkeypatch.setattr(P, "FRONT_END_CALL", one_call)
        monkeypatch.setattr(R, "ENCODE_PLAN", one_call)
        cm, counts, first, offs, pts = pipe.voxelize_batch(clouds)
        outs.append((pipe.encode_batch(cm), counts, pts))
    (Fa, ca, pa), (Fb, cb, pb) = outs
    assert ca == cb and torch.equal(pa, pb) and torch.equal(Fa, Fb)
    assert bool(torch.isfinite(Fa).all()) and Fa.shape == (sum(ca), 32)
