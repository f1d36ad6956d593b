"""Voxel hashing / coordinate maps / kernel maps: HIP vs oracle, bit-exact."""
import numpy as np
import pytest
import torch

from apr_amd import ops, synth
from apr_amd._lib import AprHipError
from oracle import me_oracle as OME

pytestmark = pytest.mark.gpu


def _quantize_gpu(xyz, vs, dev):
    c = ops.voxelize(torch.from_numpy(xyz).to(dev), vs, 0)
    m = ops.build_map(c, want_first=True)
    ops.finalize_maps([m])
    return m


@pytest.mark.parametrize("seed,small", [(0, True), (1, True), (0, False)])
def test_sparse_quantize_matches_oracle(dev, seed, small):
    xyz = synth.make_small_frame(seed) if small else synth.make_frame(seed)
    oc, osel = OME.sparse_quantize(xyz / np.float32(0.3), return_index=True)
    m = _quantize_gpu(xyz, 0.3, dev)
    assert m.n == len(oc)
    assert np.array_equal(m.first.cpu().numpy(), osel)
    assert np.array_equal(m.coords[:, 1:].cpu().numpy(), oc)


def test_sparse_quantize_shim_api(dev):
    from apr_amd import MinkowskiEngine as ME
    xyz = synth.make_small_frame(3)
    oc, osel = OME.sparse_quantize(xyz / np.float32(0.3), return_index=True)
    c, sel = ME.utils.sparse_quantize(xyz / np.float32(0.3), return_index=True)
    assert isinstance(c, np.ndarray) and np.array_equal(c, oc) and np.array_equal(sel, osel)
    ct, selt = ME.utils.sparse_quantize(torch.from_numpy(xyz) / 0.3, return_index=True)
    assert torch.is_tensor(selt) and np.array_equal(selt.numpy(), osel)
    bc = ME.utils.batched_coordinates([oc, oc[:10]])
    assert np.array_equal(bc.numpy(), OME.batched_coordinates([oc, oc[:10]]))


def test_empty_and_duplicates(dev):
    from apr_amd import MinkowskiEngine as ME
    from apr_amd._lib import AprHipError
    c = torch.zeros((0, 4), dtype=torch.int32, device=dev)
    m = ops.build_map(c)
    ops.finalize_maps([m])
    assert m.n == 0
    # all points in one voxel
    xyz = np.full((1000, 3), 0.1, np.float32)
    uc, sel = ME.utils.sparse_quantize(xyz / np.float32(0.3), return_index=True)
    assert len(uc) == 1 and sel[0] == 0
    # duplicate coordinates handed to SparseTensor are an error, as in ME
    C = torch.tensor([[0, 1, 2, 3], [0, 1, 2, 3]], dtype=torch.int32)
    st = ME.SparseTensor(torch.ones(2, 1).to(dev), coordinates=C.to(dev))
    with pytest.raises(AprHipError):
        st.coordinate_manager.get_map(1)
    # out-of-range coordinate
    C = torch.tensor([[0, 1 << 20, 0, 0]], dtype=torch.int32)
    st = ME.SparseTensor(torch.ones(1, 1).to(dev), coordinates=C.to(dev))
    with pytest.raises(AprHipError):
        st.coordinate_manager.get_map(1)


def test_pyramid_and_kernel_maps(dev):
    from apr_amd.MinkowskiEngine.core import CoordinateManager
    C0 = OME.batched_coordinates([OME.sparse_quantize(synth.make_small_frame(s) / np.float32(0.3)) for s in (0, 5)])
    ocm = OME.CoordinateManager(C0)
    cm = CoordinateManager(torch.from_numpy(C0).to(dev))
    cm.build_pyramid([2, 4, 8])
    for ts in (1, 2, 4, 8):
        assert np.array_equal(cm.get_map(ts).coords.cpu().numpy(), ocm.get_coords(ts)), ts
    # regular + strided maps
    for (ti, to, k) in [(1, 1, 3), (1, 1, 5), (1, 2, 3), (2, 2, 3), (2, 4, 3), (4, 4, 3), (4, 8, 3), (8, 8, 3)]:
        nbr = cm.kernel_map(ti, to, k).cpu().numpy()
        assert np.array_equal(nbr, ocm.get_map(ti, to, k)), (ti, to, k)
    # same-level maps take the symmetric half-probe kernel; a copy of the map object forces the generic probe kernel
    import copy
    for ts, k in [(1, 3), (1, 5), (2, 3), (8, 3)]:
        m = cm.get_map(ts)
        sym = torch.empty((m.n, k ** 3), dtype=torch.int32, device=dev)
        ops.check(ops._lib_().apr_kernel_map_same(ops.ptr(m.coords), m.n, ops.ptr(m.keys), ops.ptr(m.vals), m.cap, k, ts,
                                                  ops.ptr(sym), ops.stream()))
        gen = ops.kernel_map(m, copy.copy(m), k, ts)
        assert torch.equal(sym, gen), (ts, k)
    # transposed maps = forward strided map with in/out swapped
    for (tc, tf) in [(8, 4), (4, 2), (2, 1)]:
        nbr = cm.kernel_map(tc, tf, 3, True).cpu().numpy()
        ref = OME.transpose_map(ocm.get_map(tf, tc, 3), len(ocm.get_coords(tf)))
        assert np.array_equal(nbr, ref), (tc, tf)
        # ... and the hash-probe build of the same table (the path taken when the strides are not 2:1)
        probe = ops.kernel_map(cm.get_map(tf), cm.get_map(tc), 3, -tf).cpu().numpy()
        assert np.array_equal(probe, ref), (tc, tf)
    # transpose of an empty / single-row table
    e = ops.kernel_map_transpose(torch.empty((0, 27), dtype=torch.int32, device=dev), 5)
    assert e.shape == (5, 27) and bool((e == -1).all())
    with pytest.raises(AprHipError):
        ops.kernel_map_transpose(torch.empty((3, 27), dtype=torch.int64, device=dev), 5)


def test_negative_coordinates_floor(dev):
    # floor semantics for negative coords (floor-to-multiple, not truncation)
    rng = np.random.default_rng(0)
    C0 = np.unique(np.concatenate([np.zeros((500, 1), np.int32),
                                   rng.integers(-40, 40, size=(500, 3)).astype(np.int32)], 1), axis=0)
    rng.shuffle(C0)
    from apr_amd.MinkowskiEngine.core import CoordinateManager
    cm = CoordinateManager(torch.from_numpy(C0).to(dev))
    ocm = OME.CoordinateManager(C0)
    for ts in (2, 4):
        assert np.array_equal(cm.get_map(ts).coords.cpu().numpy(), ocm.get_coords(ts))


def test_batched_dedup_map_compaction_and_fallback(dev, monkeypatch):
    """PairRegistration.voxelize_batch builds ONE hash over all raw points and re-inserts the unique voxels into a
    table sized by a guess (rows / 4).  Dense scans take that compact table; a sparse cloud (every point its own voxel)
    exceeds the guess and must fall back to the big table — either way coordinates, sizes and kernel maps equal the
    plain per-tensor path."""
    import torch
    from apr_amd import MinkowskiEngine as ME
    from apr_amd.fcgf.pipeline import PairRegistration
    rng = np.random.default_rng(0)
    sparse = [rng.uniform(-200, 200, (160000, 3)).astype(np.float32) for _ in range(2)]      # ~1 point per voxel
    dense = [np.repeat(rng.uniform(-40, 40, (20000, 3)), 9, axis=0).astype(np.float32) +      # 9 points per voxel
             rng.uniform(0, 0.01, (180000, 3)).astype(np.float32) for _ in range(2)]
    from apr_amd.fcgf import pipeline as P
    for clouds, expect_fallback, one_call in ((sparse, True, False), (dense, False, False), (sparse, True, True), (dense, False, True)):
        pipe = PairRegistration(torch.nn.Identity(), 0.3)
        tc = [torch.from_numpy(c).to(dev) for c in clouds]
        monkeypatch.setattr(P, "FRONT_END_CALL", one_call)     # the front end through ONE library call / tensor by tensor
        cm, counts, first, poffs, _ = pipe.voxelize_batch(tc)
        if one_call:
            assert (ops.VoxelPyramid(tc, 0.3).finish() is None) == expect_fallback
        else:
            assert (cm._dedup is None) == expect_fallback
        # reference: per-frame quantisation, then a plain coordinate manager
        rows = []
        for b, c in enumerate(tc):
            m = ops.build_map(ops.voxelize(c, 0.3, b), want_first=True)
            ops.finalize_maps([m])
            rows.append(m.coords)
        ref = ME.CoordinateManager(torch.cat(rows))
        assert counts == [len(r) for r in rows]
        for ts in (1, 2, 4, 8):
            assert torch.equal(cm.get_coordinates(ts), ref.get_coordinates(ts)), ts
        assert torch.equal(cm.kernel_map(1, 1, 3), ref.kernel_map(1, 1, 3))
        assert torch.equal(cm.kernel_map(2, 4, 3), ref.kernel_map(2, 4, 3))
        assert torch.equal(cm.kernel_map(4, 2, 3, True), ref.kernel_map(4, 2, 3, True))


def test_frame_table_front_end_equals_the_concatenated_path(dev):
    """apr_voxelize_frames / apr_gather_frame_points / apr_pack_i32 (the step's front end without torch glue): the same
    coordinates, offsets, representative points and fetched words as torch.cat + apr_voxelize_segments + indexing; ragged
    frames, an empty frame in the middle, one frame, MAX_FRAMES frames."""
    rng = np.random.default_rng(5)
    for sizes in ([5000, 1, 0, 7321, 256], [4097], [37] * ops.MAX_FRAMES):
        clouds = [torch.from_numpy((rng.standard_normal((n, 3)) * 20).astype(np.float32)).to(dev) for n in sizes]
        coords, offs_dev, offs = ops.voxelize_frames(clouds, 0.3)
        assert offs == [0] + list(np.cumsum(sizes)) and offs_dev.tolist() == offs
        xyz_all = torch.cat(clouds)
        ref = ops.voxelize_segments(xyz_all, 0.3, torch.tensor(offs, dtype=torch.int64, device=dev))
        assert torch.equal(coords, ref)
        m = ops.build_map(coords, want_first=True)
        pts = ops.gather_frame_points(clouds, m)            # before the sizes are known on the host
        counts = ops.segment_counts(m, offs_dev)
        bbox = ops.coords_bbox(coords)
        zero = torch.full((777,), 5, dtype=torch.int32, device=dev)
        packed = ops.pack_i32([m.n_dev, m.status, counts, bbox], zero=zero)
        assert torch.equal(packed, torch.cat([m.n_dev, m.status, counts, bbox])) and int(zero.abs().sum()) == 0
        ops.finalize_maps([m])
        assert torch.equal(pts[:m.n], xyz_all[m.first])
        assert int(counts.sum()) == m.n
    many = [torch.arange(i, i + 3, dtype=torch.int32, device=dev) for i in range(250)]       # more sources than one launch takes
    assert torch.equal(ops.pack_i32(many), torch.cat(many))
    with pytest.raises(AprHipError):
        ops.voxelize_frames([clouds[0]] * (ops.MAX_FRAMES + 1), 0.3)


def test_transposed_tables_from_one_prefilled_pool(dev):
    """The three coarse -> fine tables of an encoder come out of ONE -1-filled allocation
    (apr_kernel_map_transpose_prefilled): equal to the stand-alone transpose and to hash probing."""
    from apr_amd.MinkowskiEngine.core import CoordinateManager
    xyz = synth.make_small_frame(2)
    c = ops.voxelize(torch.from_numpy(xyz).to(dev), 0.3, 0)
    m = ops.build_map(c)
    ops.finalize_maps([m])
    cm = CoordinateManager(m.coords.contiguous())
    cm.build_pyramid([2, 4, 8])
    for ts in (4, 2, 1):
        got = cm.kernel_map(2 * ts, ts, 3, True)
        fwd = cm.kernel_map(ts, 2 * ts, 3, False)
        assert torch.equal(got, ops.kernel_map_transpose(fwd, cm.size(ts)))
        probe = ops.kernel_map(cm.get_map(ts), cm.get_map(2 * ts), 3, -ts)
        assert torch.equal(got, probe)
    assert cm._tpool == {}                      # every table handed out exactly once


def test_one_call_front_end_equals_the_tensor_by_tensor_front_end(dev, monkeypatch):
    """apr_voxel_pyramid (frames -> de-duplicated map -> compact table -> coarser maps, one call over one arena) against
    PairRegistration's tensor-by-tensor front end: the same rows in the same order on every level, the same kernel maps,
    rows per frame, bounding box, representative points; small frames (no compact table), full frames (compact table),
    ragged / one frame, and a batch with more distinct voxels than a quarter of its points (the compact table is too small:
    the call reports it and the pipeline falls back)."""
    from apr_amd.fcgf import pipeline as P
    from apr_amd.fcgf.pipeline import PairRegistration
    pipe = PairRegistration(torch.nn.Identity(), voxel_size=0.3)
    rng = np.random.default_rng(11)
    cases = [[torch.from_numpy(synth.make_small_frame(s)).to(dev) for s in (0, 1)],
             [torch.from_numpy(synth.make_frame(s)).to(dev) for s in (0, 1, 2)],
             [torch.from_numpy((rng.standard_normal((n, 3)) * 20).astype(np.float32)).to(dev) for n in (5000, 1, 7321, 256)],
             [torch.from_numpy((rng.standard_normal((n, 3)) * 20).astype(np.float32)).to(dev) for n in (300, 0, 4100)],      # an empty frame in the middle
             [torch.from_numpy(synth.make_frame(3)).to(dev)]]
    for clouds in cases:
        monkeypatch.setattr(P, "FRONT_END_CALL", False)
        cm0, counts0, first0, offs0, pts0 = pipe.voxelize_batch(clouds)
        monkeypatch.setattr(P, "FRONT_END_CALL", True)
        vp = ops.VoxelPyramid(clouds, 0.3)
        assert vp.finish() is not None                                  # the one-call path was taken
        cm1, counts1, first1, offs1, pts1 = pipe.voxelize_batch(clouds)
        assert counts0 == counts1 and offs0 == offs1 and cm0.get_bbox() == cm1.get_bbox()
        assert torch.equal(first0, first1) and torch.equal(pts0, pts1)
        for ts in (1, 2, 4, 8):
            assert cm0.size(ts) == cm1.size(ts) and torch.equal(cm0.get_map(ts).coords, cm1.get_map(ts).coords)
        for key in ((1, 1, 3, False), (1, 2, 3, False), (2, 2, 3, False), (4, 8, 3, False), (8, 4, 3, True)):
            assert torch.equal(cm0.kernel_map(*key), cm1.kernel_map(*key)), key
        assert int(cm1._plist_counters.abs().sum()) == 0
    # scattered points: (almost) every point its own voxel -> the quarter-size compact table cannot hold them
    sparse = [torch.from_numpy((rng.uniform(-400, 400, (150000, 3))).astype(np.float32)).to(dev) for _ in range(2)]
    assert ops.VoxelPyramid(sparse, 0.3).finish() is None
    cm1, counts1, first1, offs1, pts1 = pipe.voxelize_batch(sparse)
    monkeypatch.setattr(P, "FRONT_END_CALL", False)
    cm0, counts0, first0, offs0, pts0 = pipe.voxelize_batch(sparse)
    assert counts0 == counts1 and sum(counts1) > 290000 and torch.equal(pts0, pts1)
    for ts in (1, 2, 4, 8):
        assert torch.equal(cm0.get_map(ts).coords, cm1.get_map(ts).coords)
