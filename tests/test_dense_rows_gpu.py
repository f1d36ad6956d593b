"""apr_dense_rows_bf3 (dense_rows.hip): the K = 1 layers at the end of the encoders as a row stream past LDS-resident weights
(`FCGF_APR/model/resunet.py:126-142`).  Checked against fp64, bit for bit against apr_dense_gemm_bf3 where both kernels take
the shape, and against the separate row normalisation."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(m, cin, cout, seed, ld_pad=0):
    from apr_amd import ops
    g = torch.Generator(device="cpu").manual_seed(seed)
    dev = torch.device("cuda:0")
    xw = torch.randn(m, cin + ld_pad, generator=g).to(dev)
    x = xw[:, ld_pad:] if ld_pad else xw
    W = (torch.randn(1, cin, cout, generator=g) * 0.1).to(dev)
    scale = (torch.rand(cout, generator=g) + 0.5).to(dev)
    shift = torch.randn(cout, generator=g).to(dev)
    res = torch.randn(m, cout, generator=g).to(dev)
    return ops, x, W, scale, shift, res


@pytest.mark.parametrize("cin,cout", [(64, 32), (96, 64), (128, 128), (160, 128), (192, 128), (64, 64), (128, 64), (160, 32)])
@pytest.mark.parametrize("m", [1, 255, 256, 257, 5000, 70001])
def test_row_stream_gemm_matches_fp64(cin, cout, m):
    ops, x, W, scale, shift, res = _case(m, cin, cout, 7 * m + cin + cout)
    w3 = ops.pack_weights_bf3(W)
    assert w3 is not None
    for kw in (dict(), dict(scale=scale, shift=shift, relu=True), dict(scale=scale, shift=shift, residual=res)):
        out = ops.dense_rows_bf3(x, w3, cin, cout, **kw)
        ref = x.double() @ W[0].double()
        if "scale" in kw:
            ref = ref * scale.double() + shift.double()
        if "residual" in kw:
            ref = ref + res.double()
        if kw.get("relu"):
            ref = ref.clamp_min(0)
        err = float((out.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
        assert err < 2e-6, (cin, cout, m, sorted(kw), err)
        assert torch.equal(out, ops.dense_rows_bf3(x, w3, cin, cout, **kw))      # same bits run to run


@pytest.mark.parametrize("cin,cout", [(64, 64), (128, 128), (128, 64), (192, 128)])
def test_row_stream_gemm_has_the_tiled_kernels_bits(cin, cout):
    m = 33333
    ops, x, W, scale, shift, res = _case(m, cin, cout, 3)
    w3 = ops.pack_weights_bf3(W)
    for kw in (dict(), dict(scale=scale, shift=shift, relu=True), dict(scale=scale, shift=shift, residual=res, relu=True)):
        a = ops.dense_rows_bf3(x, w3, cin, cout, **kw)
        b = ops.dense_gemm_bf3(x, w3, cin, cout, **kw)
        assert torch.equal(a, b), (cin, cout, sorted(kw), float((a - b).abs().max()))


@pytest.mark.parametrize("cin,cout", [(64, 32), (96, 64), (128, 128), (160, 128)])
def test_fused_row_normalisation_has_the_separate_kernels_bits(cin, cout):
    m = 40961
    ops, x, W, scale, shift, _ = _case(m, cin, cout, 11)
    w3 = ops.pack_weights_bf3(W)
    plain = ops.dense_rows_bf3(x, w3, cin, cout, scale=scale, shift=shift)
    fused = ops.dense_rows_bf3(x, w3, cin, cout, scale=scale, shift=shift, l2norm=True)
    assert torch.equal(fused, ops.l2_normalize(plain))
    n = fused.double().norm(dim=1)
    assert float((n - 1).abs().max()) < 1e-6


def test_strided_rows_and_column_slices():
    """conv1_tr reads a concatenation buffer and writes a column slice of a wider one (ld > channels)."""
    m, cin, cout = 50000, 96, 64
    ops, x, W, scale, shift, _ = _case(m, cin, cout, 5, ld_pad=32)
    w3 = ops.pack_weights_bf3(W)
    wide = torch.full((m, cout + 64), -7.0, device=x.device)
    out = ops.dense_rows_bf3(x, w3, cin, cout, scale=scale, shift=shift, relu=True, out=wide[:, 64:])
    ref = ops.dense_rows_bf3(x.contiguous(), w3, cin, cout, scale=scale, shift=shift, relu=True)
    assert torch.equal(out, ref)
    assert bool((wide[:, :64] == -7.0).all())


def test_the_library_routes_by_shape_and_rows():
    from apr_amd import _lib
    lib = _lib.load()
    assert lib.apr_dense_rows_bf3_ok(160, 128) == 1 and lib.apr_dense_rows_bf3_ok(256, 128) == 0
    assert lib.apr_dense_rows_bf3_ok(64, 48) == 0 and lib.apr_dense_rows_bf3_ok(32, 32) == 0
    assert lib.apr_dense_rows_bf3_route(189191, 96, 64) == 1
    assert lib.apr_dense_rows_bf3_route(1000, 96, 64) == 0
    # K = 1 through the batched descriptor: same bits as the direct call, row normalisation included
    from apr_amd import ops
    m, cin, cout = 60000, 160, 128
    _, x, W, scale, shift, _ = _case(m, cin, cout, 13)
    w3, wp = ops.pack_weights_bf3(W), ops.pack_weights(W)
    b = ops.SpconvBatch()
    out = b.add(x, None, 1, cin, cout, wp, scale=scale, shift=shift, w_bf3=w3, l2norm=True)
    b.launch()
    assert torch.equal(out, ops.dense_rows_bf3(x, w3, cin, cout, scale=scale, shift=shift, l2norm=True))
    assert torch.equal(ops.spconv(x, None, 1, cin, cout, wp, scale=scale, shift=shift, w_bf3=w3),
                       ops.dense_rows_bf3(x, w3, cin, cout, scale=scale, shift=shift))
