"""CPU tests of the C-ABI boundary and host logic (no compute calls without a GPU)."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "apr_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(apr_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    from apr_amd import _lib
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.apr_version() >= 100


def test_struct_layouts_of_the_binding_match_the_library(lib):
    """Every struct that crosses the C ABI has the same size in apr_amd/_lib.py (ctypes) as in the compiled library
    (apr_struct_sizes): a field added on one side only would corrupt descriptors silently."""
    import ctypes as C
    from apr_amd import _lib
    mine = [_lib.PairDesc, _lib.SpconvDesc, _lib.ResunetLayer, _lib.ResunetPlan, _lib.LevelMap, _lib.Pyramid, _lib.KpResnetDesc,
            _lib.GcnLayer, _lib.GcnDesc]
    out = (C.c_int32 * 16)()
    n = lib.apr_struct_sizes(out, 16)
    assert n == len(mine)
    assert [C.sizeof(t) for t in mine] == list(out[:n])


def test_stage_level_entries_size_and_reject_without_a_gpu(lib):
    """apr_voxel_pyramid / apr_resunet_encode / apr_gcn_forward: the sizing walks run on the host (dry run of the same walk
    that launches), bad descriptors are rejected before any launch."""
    import ctypes as C
    from apr_amd import _lib
    assert lib.apr_voxel_pyramid_scratch_bytes(0, 2) == 0 and lib.apr_voxel_pyramid_scratch_bytes(1000, 0) == 0
    assert lib.apr_voxel_pyramid_scratch_bytes(1000, 65) == 0                       # more than APR_MAX_FRAMES
    small, big = lib.apr_voxel_pyramid_scratch_bytes(40000, 2), lib.apr_voxel_pyramid_scratch_bytes(1400000, 12)
    assert 40000 * 40 < small < big < (1 << 30)
    plan, lv, box = _lib.ResunetPlan(), (_lib.LevelMap * 4)(), (C.c_int32 * 8)(0, 0, 0, 10, 10, 10, 0, 0)
    assert lib.apr_resunet_encode_supported(C.byref(plan), lv, box) == 0            # empty plan / maps
    assert lib.apr_resunet_encode_scratch_bytes(C.byref(plan), lv, box) == 0
    assert lib.apr_resunet_encode(C.byref(plan), lv, box, None, 0, None, 0, None, 32, None) == -1
    assert b"not covered" in lib.apr_last_error()
    d = _lib.GcnDesc()
    assert lib.apr_gcn_scratch_bytes(C.byref(d), 100, 100) == 0                     # no layers
    d.n_layers, d.c = 1, 96
    d.layer[0].kind, d.layer[0].k = 0, 10
    assert lib.apr_gcn_scratch_bytes(C.byref(d), 100, 100) == 0                     # c % 64 != 0
    d.c = 256
    d.layer[0].w1 = d.layer[0].w2 = d.layer[0].w3 = 1                               # non-NULL stand-ins: only sized, never read
    assert lib.apr_gcn_scratch_bytes(C.byref(d), 1400, 1300) > 1400 * 10 * 512 * 4
    assert lib.apr_gcn_scratch_bytes(C.byref(d), 5, 1300) == 0                      # fewer points than neighbours asked for


def test_host_side_sizing_helpers(lib):
    assert lib.apr_hash_capacity(0) == 1024
    assert lib.apr_hash_capacity(15000) == 32768
    cap = lib.apr_hash_capacity(120000)
    assert cap >= 240000 and cap & (cap - 1) == 0
    assert lib.apr_spconv_packed_size(27, 64, 64) == 27 * 64 * 64
    assert lib.apr_map_scratch_bytes(1000) > 5000
    assert lib.apr_ransac_scratch_bytes(15000, 4000000) > 15000 * 12


def test_argument_errors_are_reported_without_a_gpu(lib):
    from apr_amd import _lib
    rc = lib.apr_kernel_map(None, 10, None, None, None, 1000, 3, 1, None, None)   # cap not a power of two
    assert rc == -1 and b"power of two" in lib.apr_last_error()
    rc = lib.apr_spconv_fwd(None, 8, None, 10, 27, 32, 32, None, None, None, None, 0, 0, None, 32, None)
    assert rc == -1 and b"identity map needs K == 1" in lib.apr_last_error()
    rc = lib.apr_spconv_fwd(None, 8, None, 10, 1, 32, 32, None, None, None, None, 0, 0, None, 32, None)
    assert rc == -1 and b"leading dimension" in lib.apr_last_error()
    with pytest.raises(_lib.AprHipError):
        _lib.check(rc)


def test_ops_refuse_cpu_tensors():
    from apr_amd import ops
    from apr_amd._lib import AprHipError
    with pytest.raises(AprHipError):
        ops.voxelize(torch.zeros(4, 3), 0.3)
    with pytest.raises(AprHipError):
        ops.feature_nn(torch.zeros(4, 32), torch.zeros(4, 32))
    with pytest.raises(AprHipError):
        ops.build_map(torch.zeros((4, 4), dtype=torch.int32))
    from apr_amd import MinkowskiEngine as ME
    with pytest.raises(AprHipError):
        ME.SparseTensor(torch.ones(3, 1), coordinates=torch.zeros((3, 4), dtype=torch.int32))


def test_shim_install_and_collate():
    import sys
    import apr_amd
    ME = apr_amd.install_shims()
    import MinkowskiEngine
    import MinkowskiEngine.MinkowskiFunctional as MEF
    assert MinkowskiEngine is ME and hasattr(MEF, "relu")
    c0 = np.array([[1, 2, 3], [4, 5, 6]], np.int32)
    c1 = torch.tensor([[7.0, 8.0, 9.0]])
    bc, bf = ME.utils.sparse_collate([c0, c1], [np.ones((2, 1), np.float32), torch.ones(1, 1)])
    assert bc.tolist() == [[0, 1, 2, 3], [0, 4, 5, 6], [1, 7, 8, 9]] and bc.dtype == torch.int32
    assert bf.shape == (3, 1)
    sys.modules.pop("MinkowskiEngine", None)
    sys.modules.pop("MinkowskiEngine.MinkowskiFunctional", None)
    sys.modules.pop("MinkowskiEngine.utils", None)


def test_synthetic_generator_is_deterministic_and_kitti_shaped():
    from apr_amd import synth
    a, b = synth.make_small_frame(0), synth.make_small_frame(0)
    assert np.array_equal(a, b) and a.dtype == np.float32 and 15000 < len(a) <= 20000
    x0, x1, T = synth.make_pair(1, n_beams=16, n_azimuth=625)
    assert np.allclose(T[:3, :3] @ T[:3, :3].T, np.eye(3), atol=1e-12)
    assert 5.0 <= np.linalg.norm(T[:3, 3]) <= 20.0


def test_rte_rre():
    from apr_amd.fcgf.registration import rte_rre
    a = np.deg2rad(3.0)
    T = np.eye(4); T[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]; T[:3, 3] = [1, 2, 2]
    rte, rre = rte_rre(T, np.eye(4))
    assert abs(rte - 3.0) < 1e-12 and abs(rre - 3.0) < 1e-9


def test_nn_scratch_size_is_monotone(lib):
    """apr_match_pose_batch sizes ONE feature-NN scratch for a batch from (max n0, max n1): the size function must
    not shrink when either count grows (it once did: a batch of unequal pairs failed with 'scratch too small')."""
    for c in (32, 64, 128):
        for n0 in (300, 5000, 14000, 41000):
            col = [lib.apr_feature_nn_fast_scratch_bytes(n0, n1, c) for n1 in range(200, 60000, 1499)]
            assert col == sorted(col)
        for n1 in (300, 5000, 14000, 41000):
            row = [lib.apr_feature_nn_fast_scratch_bytes(n0, n1, c) for n0 in range(200, 60000, 1499)]
            assert row == sorted(row)
    big = lib.apr_match_pose_batch_scratch_bytes(3, 9000, 9500, 32, 200000)
    assert big > lib.apr_feature_nn_fast_scratch_bytes(6000, 9500, 32)
