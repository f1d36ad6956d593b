"""Sparse-conv / norm kernels vs the torch-CPU oracle on the same seeded inputs."""
import numpy as np
import pytest
import torch

from apr_amd import ops
from oracle import me_oracle as OME
from tests.helpers import rel_l2

pytestmark = pytest.mark.gpu


def _oracle_conv(x, nbr, W, scale=None, shift=None, residual=None, relu=False):
    n_out = nbr.shape[0]
    out = torch.zeros(n_out, W.shape[2], dtype=torch.float64)
    xd, Wd = x.double(), W.double()
    for o in range(nbr.shape[1]):
        j = np.nonzero(nbr[:, o] >= 0)[0]
        if len(j):
            out.index_add_(0, torch.from_numpy(j), xd[torch.from_numpy(nbr[j, o].astype(np.int64))] @ Wd[o])
    if scale is not None:
        out = out * scale.double()
    if shift is not None:
        out = out + shift.double()
    if residual is not None:
        out = out + residual.double()
    return torch.relu(out) if relu else out


def _random_map(rng, n_in, n_out, K, density):
    nbr = rng.integers(0, n_in, size=(n_out, K)).astype(np.int32)
    nbr[rng.random((n_out, K)) > density] = -1
    return nbr


@pytest.mark.parametrize("cin,cout,K,n_in,n_out", [
    (32, 32, 27, 3000, 3000), (32, 64, 27, 3000, 1100), (64, 64, 27, 5000, 5000), (64, 128, 27, 2000, 700),
    (128, 128, 27, 900, 900), (256, 128, 27, 300, 900), (192, 64, 27, 900, 2500), (96, 64, 1, 4000, 4000),
    (64, 32, 1, 4001, 4001), (384, 128, 27, 500, 1300), (1, 32, 125, 3000, 3000), (3, 32, 27, 1000, 1000),
    (64, 64, 27, 70000, 70000), (32, 32, 27, 40000, 40000), (5, 7, 27, 333, 333), (64, 64, 27, 31, 31),
])
def test_spconv_matches_oracle(dev, cin, cout, K, n_in, n_out):
    rng = np.random.default_rng(cin * 1000 + cout + K)
    x = torch.from_numpy(rng.standard_normal((n_in, cin)).astype(np.float32))
    W = torch.from_numpy((rng.standard_normal((K, cin, cout)) / np.sqrt(cin * 8)).astype(np.float32))
    if K == 1:
        nbr = None
        nbr_o = np.arange(n_out, dtype=np.int32)[:, None]
    else:
        nbr = nbr_o = _random_map(rng, n_in, n_out, K, 0.27)
    wp = ops.pack_weights(W.to(dev))
    out = ops.spconv(x.to(dev), None if nbr is None else torch.from_numpy(nbr).to(dev), K, cin, cout, wp,
                     n_out=n_out)
    ref = _oracle_conv(x, nbr_o, W)
    assert rel_l2(out.cpu(), ref) < 2e-6


@pytest.mark.parametrize("cin,cout,K,n_in,n_out,density", [
    (64, 64, 27, 5000, 5000, 0.27), (256, 128, 27, 300, 900, 0.1), (128, 128, 27, 900, 900, 0.3),
    (384, 128, 27, 500, 1300, 0.08), (192, 64, 8, 700, 257, 0.5),
    (64, 64, 27, 31, 31, 0.3), (256, 256, 27, 1246, 1246, 0.29), (64, 64, 27, 40000, 40000, 0.27),
    (64, 64, 27, 500, 500, 0.0), (128, 64, 27, 100, 1000, 1.0), (512, 64, 27, 400, 700, 0.3),
])
@pytest.mark.parametrize("gemm", ["fp32", "bf3", "ws3"])
def test_weight_stationary_path_matches_oracle(dev, cin, cout, K, n_in, n_out, density, gemm):
    """apr_pairlist_build + apr_spconv_ws_fwd[_bf3] (strided / transposed / deep layers) against the same oracle,
    with the fused epilogue and strided in / out / residual rows; also bit-stable run to run.  `gemm`: the fp32-MFMA
    kernel (k_ws_gemm) or the default of the encoder, the bf16 3-way split (k_ws_gemm_bf3<1|2|4>), which must keep
    fp32 accuracy ROW BY ROW on inputs whose row magnitudes span 8 decades."""
    rng = np.random.default_rng(cin * 977 + cout + K + n_out)
    xw = torch.from_numpy(rng.standard_normal((n_in, cin + 32)).astype(np.float32))
    if gemm == "bf3":
        if cin not in (64, 128, 192, 256, 384):
            pytest.skip("k_ws_gemm_bf3 covers cin 64 / 128 / 192 / 256 / 384")
        xw = xw * torch.from_numpy(np.exp(rng.uniform(-9.2, 9.2, (n_in, 1))).astype(np.float32))
    if gemm == "ws3":      # the x-triple entry lists (k_ws3_gemm_bf3: one product row per entry, three offsets per accumulator)
        if cin not in (64, 128) or K != 27:
            pytest.skip("k_ws3_gemm_bf3 covers 27-offset maps with cin 64 / 128")
        xw = xw * torch.from_numpy(np.exp(rng.uniform(-9.2, 9.2, (n_in, 1))).astype(np.float32))
    x = xw[:, 32:]
    W = torch.from_numpy((rng.standard_normal((K, cin, cout)) / np.sqrt(cin * 8)).astype(np.float32))
    nbr = _random_map(rng, n_in, n_out, K, density)
    scale = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32))
    shift = torch.from_numpy(rng.standard_normal(cout).astype(np.float32))
    resw = torch.from_numpy(rng.standard_normal((n_out, cout + 64)).astype(np.float32))
    res = resw[:, :cout]
    wp = ops.pack_weights(W.to(dev))
    w3 = ops.pack_weights_bf3(W.to(dev)) if gemm in ("bf3", "ws3") else None
    assert (w3 is not None) == (gemm in ("bf3", "ws3"))
    nbr_d = torch.from_numpy(nbr).to(dev)
    mk = ops.build_pairlist3 if gemm == "ws3" else ops.build_pairlist
    pl = mk(nbr_d)
    outw = torch.zeros(n_out, cout + 32, device=dev)
    xd, resd = xw.to(dev)[:, 32:], resw.to(dev)[:, :cout]
    out = ops.spconv(xd, nbr_d, K, cin, cout, wp, scale=scale.to(dev), shift=shift.to(dev), residual=resd, relu=True,
                     out=outw[:, 32:], plist=pl, w_bf3=w3)
    ref = _oracle_conv(x, nbr, W, scale, shift, res, relu=True)
    assert rel_l2(out.cpu(), ref) < 2e-6
    # the bare contraction, row by row: every output row keeps fp32 accuracy at the magnitude of ITS OWN inputs (the
    # bound is relative to sum_k |x_row| |W_k|, the scale a row's fp32 rounding errors live on: a row that cancels to
    # near zero is not asked for more digits than fp32 has)
    bare = ops.spconv(xd, nbr_d, K, cin, cout, wp, plist=pl, w_bf3=w3).cpu().double()
    ref0 = _oracle_conv(x, nbr, W)
    mag = _oracle_conv(x.abs(), nbr, W.abs()).norm(dim=1).clamp_min(1e-300)
    assert float(((bare - ref0).norm(dim=1) / mag).max()) < 5e-6
    assert rel_l2(bare, ref0) < 2e-6
    assert float(outw[:, :32].abs().max()) == 0.0          # neighbouring columns untouched
    # pair lists: one region per offset (per x-triple: rows with any of its three offsets), counts consistent with the map
    if gemm == "ws3":
        assert np.array_equal(pl.counts(), (nbr.reshape(n_out, 9, 3) >= 0).any(axis=2).sum(axis=0))
    else:
        assert np.array_equal(pl.counts(), (nbr >= 0).sum(axis=0))
    again = ops.spconv(xd, nbr_d, K, cin, cout, wp, scale=scale.to(dev), shift=shift.to(dev), residual=resd,
                       relu=True, plist=mk(nbr_d), w_bf3=w3)
    assert torch.equal(again, out)
    # and agrees with the tile kernel to fp32 summation-order noise
    tile = ops.spconv(xd, nbr_d, K, cin, cout, wp, scale=scale.to(dev), shift=shift.to(dev), residual=resd, relu=True)
    assert rel_l2(tile.cpu(), out.cpu()) < 2e-6


def test_spconv_fused_epilogue_and_slices(dev):
    rng = np.random.default_rng(7)
    n, cin, cout, K = 2111, 64, 64, 27
    wide_in = torch.from_numpy(rng.standard_normal((n, cin + 32)).astype(np.float32)).to(dev)
    x = wide_in[:, 32:]                      # column slice as input
    W = torch.from_numpy((rng.standard_normal((K, cin, cout)) / 20).astype(np.float32))
    nbr = _random_map(rng, n, n, K, 0.3)
    scale = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32))
    shift = torch.from_numpy(rng.standard_normal(cout).astype(np.float32))
    res = torch.from_numpy(rng.standard_normal((n, cout)).astype(np.float32))
    wide_out = torch.zeros((n, cout + 64), device=dev)
    ops.spconv(x, torch.from_numpy(nbr).to(dev), K, cin, cout, ops.pack_weights(W.to(dev)), scale=scale.to(dev),
               shift=shift.to(dev), residual=res.to(dev), relu=True, out=wide_out[:, 64:])
    ref = _oracle_conv(x.cpu(), nbr, W, scale, shift, res, True)
    assert rel_l2(wide_out[:, 64:].cpu(), ref) < 2e-6
    assert float(wide_out[:, :64].abs().max()) == 0.0   # nothing written outside the slice


@pytest.mark.parametrize("m,cin,cout", [
    (4001, 64, 64), (27674, 960, 64), (129, 128, 256), (1246, 1920, 128), (64, 512, 512), (1, 64, 64),
    (10187, 192, 128), (33000, 64, 128),
])
def test_dense_gemm_identity_map(dev, m, cin, cout):
    """K = 1 with an identity map (Linear layers, KPConv step 2, K = 1 convolutions) runs on k_dense_gemm: column
    slices as input / output / residual, fused scale + shift + residual + ReLU, ragged last tile."""
    rng = np.random.default_rng(m + cin + cout)
    wide_in = torch.from_numpy(rng.standard_normal((m, cin + 32)).astype(np.float32)).to(dev)
    x = wide_in[:, 32:]
    W = torch.from_numpy((rng.standard_normal((cin, cout)) / np.sqrt(cin)).astype(np.float32))
    scale = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32))
    shift = torch.from_numpy(rng.standard_normal(cout).astype(np.float32))
    res_wide = torch.from_numpy(rng.standard_normal((m, cout + 4)).astype(np.float32)).to(dev)
    res = res_wide[:, 4:]
    wp = ops.pack_weights(W.to(dev))
    plain = ops.spconv(x, None, 1, cin, cout, wp, n_out=m)
    assert rel_l2(plain.cpu(), x.cpu().double() @ W.double()) < 2e-6
    wide_out = torch.zeros((m, cout + 64), device=dev)
    ops.spconv(x, None, 1, cin, cout, wp, scale=scale.to(dev), shift=shift.to(dev), residual=res, relu=True,
               out=wide_out[:, 64:], n_out=m)
    ref = torch.relu((x.cpu().double() @ W.double()) * scale.double() + shift.double() + res.cpu().double())
    assert rel_l2(wide_out[:, 64:].cpu(), ref) < 2e-6
    assert float(wide_out[:, :64].abs().max()) == 0.0


@pytest.mark.parametrize("m,cin,cout", [
    (4001, 64, 64), (27674, 960, 64), (129, 128, 256), (1382, 7680, 512), (64, 512, 512), (1, 64, 64),
    (40000, 256, 256), (9918, 1920, 128),
])
def test_dense_gemm_bf3_matches_fp64(dev, m, cin, cout):
    """apr_dense_gemm_bf3 (bf16 MFMA, 3-way split of both operands, LDS-DMA weight staging; both tile heights): fp32
    accuracy against the float64 product, with column slices as input / output / residual, the fused epilogue and a
    ragged last tile; operands over several orders of magnitude."""
    rng = np.random.default_rng(m + cin + cout)
    mag = np.exp(rng.uniform(-4, 4, (m, 1))).astype(np.float32)
    wide_in = torch.from_numpy((rng.standard_normal((m, cin + 32)) * mag).astype(np.float32)).to(dev)
    x = wide_in[:, 32:]
    W = torch.from_numpy((rng.standard_normal((cin, cout)) / np.sqrt(cin)).astype(np.float32))
    scale = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32))
    shift = torch.from_numpy(rng.standard_normal(cout).astype(np.float32))
    res_wide = torch.from_numpy(rng.standard_normal((m, cout + 4)).astype(np.float32)).to(dev)
    res = res_wide[:, 4:]
    w3 = ops.pack_weights_bf3(W[None].contiguous().to(dev))
    assert w3 is not None
    plain = ops.dense_gemm_bf3(x, w3, cin, cout)
    ref0 = x.cpu().double() @ W.double()
    assert rel_l2(plain.cpu(), ref0) < 2e-6
    # per row: every row keeps fp32 accuracy at its own magnitude (a plain bf16 product would be off by 4e-3)
    err_row = (plain.cpu().double() - ref0).norm(dim=1) / ref0.norm(dim=1).clamp_min(1e-30)
    assert float(err_row.max()) < 5e-6
    wide_out = torch.zeros((m, cout + 64), device=dev)
    ops.dense_gemm_bf3(x, w3, cin, cout, scale=scale.to(dev), shift=shift.to(dev), residual=res, relu=True,
                       out=wide_out[:, 64:])
    ref = torch.relu(ref0 * scale.double() + shift.double() + res.cpu().double())
    assert rel_l2(wide_out[:, 64:].cpu(), ref) < 2e-6
    assert float(wide_out[:, :64].abs().max()) == 0.0
    with pytest.raises(Exception):
        ops.dense_gemm_bf3(wide_in[:, 1:1 + cin], w3, cin, cout)        # rows not 16-byte aligned: refused, loudly


@pytest.mark.parametrize("cin,cout,bf3", [(128, 64, 0), (64, 64, 1), (128, 64, 1), (256, 128, 1)])
def test_weight_stationary_many_rounds(dev, cin, cout, bf3):
    """More units than resident workgroups (APR_WS_TARGET=4 is read once per process, hence the child process): every
    workgroup walks several units, re-staging its weight slice, and the result must not change -- for the fp32 gemm
    and for k_ws_gemm_bf3<1|2|4> (cin = 256 x K = 27 at many rounds included)."""
    import os, subprocess, sys
    code = (
        "import numpy as np, torch, sys\n"
        "from apr_amd import ops\n"
        "rng = np.random.default_rng(5)\n"
        f"n, cin, cout, K, bf3 = 6000, {cin}, {cout}, 27, {bf3}\n"
        "x = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32)).cuda()\n"
        "W = torch.from_numpy((rng.standard_normal((K, cin, cout)) / 30).astype(np.float32)).cuda()\n"
        "nbr = rng.integers(0, n, size=(n, K)).astype(np.int32); nbr[rng.random((n, K)) > 0.3] = -1\n"
        "nbr = torch.from_numpy(nbr).cuda()\n"
        "wp = ops.pack_weights(W)\n"
        "tile = ops.spconv(x, nbr, K, cin, cout, wp)\n"
        "w3 = ops.pack_weights_bf3(W) if bf3 else None\n"
        "assert (w3 is not None) == bool(bf3)\n"
        "ws = ops.spconv(x, nbr, K, cin, cout, wp, plist=ops.build_pairlist(nbr), w_bf3=w3)\n"
        "err = float((ws - tile).norm() / tile.norm())\n"
        "print('REL', err); sys.exit(0 if err < 1e-6 else 1)\n")
    env = dict(os.environ, APR_WS_TARGET="4", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


def test_weight_stationary_bf3_empty_offsets_and_map(dev):
    """k_ws_gemm_bf3 on a map whose offsets are mostly EMPTY (only 3 of 27 carry pairs) and on an all-empty map: the
    unit table must skip the empty offsets, and the reduce must still write the epilogue of every row."""
    rng = np.random.default_rng(11)
    n, cin, cout, K = 3000, 64, 128, 27
    x = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32))
    W = torch.from_numpy((rng.standard_normal((K, cin, cout)) / 20).astype(np.float32))
    shift = torch.from_numpy(rng.standard_normal(cout).astype(np.float32))
    nbr = np.full((n, K), -1, np.int32)
    for k in (0, 13, 26):
        rows = rng.random(n) < 0.4
        nbr[rows, k] = rng.integers(0, n, int(rows.sum()))
    wp, w3 = ops.pack_weights(W.to(dev)), ops.pack_weights_bf3(W.to(dev))
    for m in (nbr, np.full((n, K), -1, np.int32)):
        md = torch.from_numpy(m).to(dev)
        out = ops.spconv(x.to(dev), md, K, cin, cout, wp, shift=shift.to(dev), plist=ops.build_pairlist(md), w_bf3=w3)
        ref = _oracle_conv(x, m, W, shift=shift)
        assert rel_l2(out.cpu(), ref) < 2e-6


def test_all_empty_offsets(dev):
    n, c = 100, 32
    nbr = torch.full((n, 27), -1, dtype=torch.int32, device=dev)
    x = torch.ones((n, c), device=dev)
    W = torch.ones((27, c, c), device=dev)
    shift = torch.full((c,), 2.0, device=dev)
    out = ops.spconv(x, nbr, 27, c, c, ops.pack_weights(W), shift=shift)
    assert torch.equal(out.cpu(), torch.full((n, c), 2.0))


def test_bn_stats_affine_l2(dev):
    rng = np.random.default_rng(3)
    x = torch.from_numpy((rng.standard_normal((7777, 96)) * 3 + 5).astype(np.float32))
    mean, var = ops.bn_stats(x.to(dev))
    assert torch.allclose(mean.cpu(), x.double().mean(0).float(), rtol=1e-6, atol=1e-6)
    assert torch.allclose(var.cpu(), x.double().var(0, unbiased=False).float(), rtol=1e-5, atol=1e-6)
    y = ops.affine_act(x.to(dev), scale=mean, shift=var, relu=True)
    assert torch.allclose(y.cpu(), torch.relu(x * mean.cpu() + var.cpu()), rtol=1e-6, atol=1e-5)
    z = ops.l2_normalize(x.to(dev))
    assert rel_l2(z.cpu(), x / x.norm(dim=1, keepdim=True)) < 1e-6


def test_cpu_tensors_fail_loudly():
    from apr_amd._lib import AprHipError
    with pytest.raises(AprHipError):
        ops.affine_act(torch.ones(4, 4))


@pytest.mark.parametrize("cin,cout,K,n_in,n_out,density", [
    (64, 64, 27, 5000, 5000, 0.27), (64, 64, 27, 70001, 70001, 0.27), (64, 128, 27, 3000, 1100, 0.3),
    (64, 256, 27, 1000, 9000, 0.12), (64, 64, 8, 700, 257, 0.5), (64, 64, 27, 31, 31, 0.3), (64, 64, 27, 500, 500, 0.0),
    (64, 64, 27, 100, 1000, 1.0), (64, 64, 27, 40, 2000, 0.02), (64, 64, 27, 3000, 3000, 0.9),
    (128, 128, 27, 2500, 2500, 0.3), (128, 64, 27, 1000, 9000, 0.12), (128, 128, 27, 20000, 20000, 0.27),
    (128, 64, 27, 100, 700, 1.0),
    # 128 input channels (k_os_conv<2>): every group count per offset from 0 to 13 (density 1.0), an odd number of items, one
    # item, no item, K = 8 and K = 1, 256 output columns, a strided map, a 60 k-row map
    (128, 128, 27, 100, 700, 1.0), (128, 128, 27, 31, 31, 0.3), (128, 128, 27, 500, 500, 0.0), (128, 128, 8, 700, 257, 0.5),
    (128, 256, 27, 3000, 1100, 0.3), (128, 128, 27, 40, 2000, 0.02), (128, 128, 27, 60000, 60000, 0.27),
    (128, 128, 1, 1000, 1000, 1.0),
])
def test_output_stationary_path_matches_oracle(dev, cin, cout, K, n_in, n_out, density):
    """apr_spconv_os_pairs_build + apr_spconv_os_fwd (accumulators in LDS, no product rows) against the fp64 oracle:
    fused epilogue, strided in / out / residual rows, row magnitudes over 8 decades (fp32 accuracy row by row), ragged
    last tile, empty offsets / empty map, more than 8 groups per (tile, offset) (density 1.0), and bit-stable run to run."""
    rng = np.random.default_rng(cin * 977 + cout + K + n_out)
    xw = torch.from_numpy(rng.standard_normal((n_in, cin + 32)).astype(np.float32))
    xw = xw * torch.from_numpy(np.exp(rng.uniform(-9.2, 9.2, (n_in, 1))).astype(np.float32))
    x = xw[:, 32:]
    W = torch.from_numpy((rng.standard_normal((K, cin, cout)) / np.sqrt(cin * 8)).astype(np.float32))
    nbr = _random_map(rng, n_in, n_out, K, density)
    scale = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32))
    shift = torch.from_numpy(rng.standard_normal(cout).astype(np.float32))
    resw = torch.from_numpy(rng.standard_normal((n_out, cout + 64)).astype(np.float32))
    res = resw[:, :cout]
    w3 = ops.pack_weights_bf3(W.to(dev))
    nbr_d = torch.from_numpy(nbr).to(dev)
    R = ops.os_tile_rows(n_out, cin, cout)
    assert R > 0 and R % 16 == 0 and ops.os_tile_rows(n_out, 256, cout) == 0      # 64 or 128 input channels only
    for rows in sorted({R, 64, 400 if cin == 64 else 208}):
        pairs = ops.build_os_pairs(nbr_d, n_in, rows)
        outw = torch.zeros(n_out, cout + 32, device=dev)
        xd, resd = xw.to(dev)[:, 32:], resw.to(dev)[:, :cout]
        out = ops.spconv_os(xd, pairs, cin, cout, w3, scale=scale.to(dev), shift=shift.to(dev), residual=resd, relu=True,
                            out=outw[:, 32:])
        ref = _oracle_conv(x, nbr, W, scale, shift, res, relu=True)
        assert rel_l2(out.cpu(), ref) < 2e-6
        assert float(outw[:, :32].abs().max()) == 0.0          # neighbouring columns untouched
        bare = ops.spconv_os(xd, pairs, cin, cout, w3).cpu().double()
        ref0 = _oracle_conv(x, nbr, W)
        mag = _oracle_conv(x.abs(), nbr, W.abs()).norm(dim=1).clamp_min(1e-300)
        assert float(((bare - ref0).norm(dim=1) / mag).max()) < 5e-6
        again = ops.spconv_os(xd, ops.build_os_pairs(nbr_d, n_in, rows), cin, cout, w3, scale=scale.to(dev),
                              shift=shift.to(dev), residual=resd, relu=True)
        assert torch.equal(again, out)


@pytest.mark.parametrize("cin,cout,n,density", [(64, 64, 30011, 0.27), (128, 128, 9000, 0.3), (64, 64, 3000, 0.05),
                                                 (64, 128, 5000, 1.0)])
def test_os_conv_debug_pipeline_matches_production(dev, cin, cout, n, density):
    """Run-time guard of k_os_conv's hand-counted row pipeline (inline-asm loads + s_waitcnt vmcnt(N); the build only checks
    the disassembly): the production kernel, the diagnostics instantiation (other registers and schedule, same counted
    waits) and the diagnostics instantiation with a FULL DRAIN in front of every counted wait give the same bits.  A
    toolchain that moves a row-register read above its wait breaks this equality.  Odd item counts per tile included
    (the loop is unrolled twice: the surplus item must be empty)."""
    from apr_amd import _lib
    rng = np.random.default_rng(cin + cout + n)
    x = torch.from_numpy((rng.standard_normal((n, cin)) * np.exp(rng.uniform(-3, 3, (n, 1)))).astype(np.float32)).to(dev)
    W = torch.from_numpy((rng.standard_normal((27, cin, cout)) / np.sqrt(cin * 8)).astype(np.float32)).to(dev)
    nbr = torch.from_numpy(_random_map(rng, n, n, 27, density)).to(dev)
    w3 = ops.pack_weights_bf3(W)
    lib = _lib.load()
    outs = []
    try:
        for rows in (ops.os_tile_rows(n, cin, cout), 64):
            pairs = ops.build_os_pairs(nbr, n, rows)
            items = pairs.blob[: 4 * 32 * ((n + rows - 1) // rows)].view(torch.int32).view(-1, 32)[:, 31]
            assert bool((items % 2 == 1).any()) or density == 1.0      # tiles with an odd item count exist
            for mode in (0, 1, 2):
                _lib.check(lib.apr_spconv_os_set_debug(mode))
                outs.append(ops.spconv_os(x, pairs, cin, cout, w3, relu=True))
            assert torch.equal(outs[-3], outs[-2]) and torch.equal(outs[-3], outs[-1])
    finally:
        _lib.check(lib.apr_spconv_os_set_debug(0))
    assert torch.equal(outs[0], outs[3])                 # tile height does not change the bits either (fixed offset order)


def _clustered_coords(rng, n, nbatch, lo, hi, spread):
    """Unique int32 (batch, x, y, z) rows: blobs with plenty of occupied neighbour cells, some isolated voxels, the box
    corners occupied (the bitmap's padding is exercised), negative coordinates."""
    pts = []
    for b in range(nbatch):
        centres = rng.integers(lo + spread, hi - spread, size=(6, 3))
        c = centres[rng.integers(0, 6, n)] + np.rint(rng.standard_normal((n, 3)) * spread / 3).astype(np.int64)
        c = np.clip(c, lo, hi)
        far = rng.integers(lo, hi + 1, size=(n // 10, 3))
        corners = np.array([[lo, lo, lo], [hi, hi, hi], [lo, hi, lo], [hi, lo, hi]])
        xyz = np.concatenate([c, far, corners])
        pts.append(np.concatenate([np.full((len(xyz), 1), b), xyz], 1))
    coords = np.unique(np.concatenate(pts), axis=0)
    return coords[rng.permutation(len(coords))].astype(np.int32)


@pytest.mark.parametrize("bks,n,nbatch,lo,hi,spread", [
    (5, 4000, 3, -40, 55, 6), (3, 3000, 2, -21, 20, 4), (7, 1500, 2, -25, 30, 5), (5, 60000, 2, -160, 161, 30), (5, 1, 1, 7, 7, 0),
])
def test_kernel_map_with_occupancy_prefilter(dev, bks, n, nbatch, lo, hi, spread):
    """apr_kernel_map_occ (probe conv1's occupancy bitmap first, the hash table only for occupied cells) == apr_kernel_map,
    on the same-level 3^3 map and on the strided 1 -> 2 map (output voxels floor below the box's minimum: cells outside the
    bitmap go to the table), with the exact box, a superset box and a box that MISSES voxels."""
    rng = np.random.default_rng(bks * 1000 + n)
    coords = np.array([[0, 7, 7, 7]], dtype=np.int32) if n == 1 else _clustered_coords(rng, n, nbatch, lo, hi, spread)
    cd = torch.from_numpy(coords).to(dev)
    N = len(coords)
    m = ops.build_map(cd)
    m2 = ops.build_map(cd, floor_to=2)
    ops.finalize_maps([m, m2])
    bbox = ops.coords_bbox(cd).tolist()
    w = torch.zeros((bks ** 3, 8), dtype=torch.float32, device=dev)
    ref_same, ref_down = ops.kernel_map(m, m, 3, 1), ops.kernel_map(m2, m, 3, 1)
    assert int((ref_same >= 0).sum()) >= N and int((ref_down >= 0).sum()) >= N
    boxes = [bbox, [bbox[0] - 3, bbox[1] - 40, bbox[2], bbox[3] + 37, bbox[4], bbox[5] + 2, bbox[6] + 1, 0]]
    if n > 1:
        boxes.append([bbox[0] + 2, bbox[1], bbox[2] + 1, bbox[3] - 3, bbox[4], bbox[5], max(bbox[6] - 1, 0), 0])
    for box in boxes:
        keep = []
        ops.occ_conv(m.coords, N, box, bks, w, keep=keep)
        assert torch.equal(ops.kernel_map_occ(m, m, 3, 1, keep[0]), ref_same)
        assert torch.equal(ops.kernel_map_occ(m2, m, 3, 1, keep[0]), ref_down)
    if N > 8:
        # a bitmap built from a SUBSET of the map's rows (round-4 advice): a clear bit inside its box would read as "no voxel"
        # -- the wrapper notices that the bitmap does not cover the map it is asked about and takes the hash-table path
        part = []
        ops.occ_conv(m.coords, N // 2, bbox, bks, w, keep=part)
        assert part[0][3] == N // 2
        assert torch.equal(ops.kernel_map_occ(m, m, 3, 1, part[0]), ref_same)
        assert torch.equal(ops.kernel_map_occ(m2, m, 3, 1, part[0]), ref_down)


@pytest.mark.parametrize("ks,cout,n,nbatch,lo,hi,spread", [
    (5, 32, 4000, 3, -40, 55, 6), (3, 32, 3000, 2, -20, 20, 4), (7, 32, 1500, 2, -25, 30, 5), (5, 8, 500, 1, 0, 31, 3),
    (5, 40, 2500, 4, -100, -36, 5), (5, 32, 60000, 2, -160, 160, 30), (5, 64, 1, 1, 7, 7, 0),
])
def test_occupancy_conv_equals_kernel_map_path(dev, ks, cout, n, nbatch, lo, hi, spread):
    """conv1 on constant-1 features from the occupancy bitmap (apr_occ_conv) == the kernel-map path (apr_kernel_map_same
    + k_spconv_smallcin) bit for bit, and == the fp64 oracle; epilogue (scale, shift, ReLU) and a strided output slice
    included; the bounding box comes from apr_coords_bbox."""
    rng = np.random.default_rng(ks * 100 + cout + n)
    if n == 1:
        coords = np.array([[0, 7, 7, 7]], dtype=np.int32)
    else:
        coords = _clustered_coords(rng, n, nbatch, lo, hi, spread)
    N, K = len(coords), ks ** 3
    cd = torch.from_numpy(coords).to(dev)
    bbox = ops.coords_bbox(cd).tolist()
    assert bbox[:7] == [int(coords[:, 1].min()), int(coords[:, 2].min()), int(coords[:, 3].min()),
                        int(coords[:, 1].max()), int(coords[:, 2].max()), int(coords[:, 3].max()), int(coords[:, 0].max())]
    assert ops.occ_conv_supported(bbox, ks, cout)
    W = torch.from_numpy((rng.standard_normal((K, 1, cout)) / np.sqrt(K / 4)).astype(np.float32))
    scale = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32))
    shift = torch.from_numpy(rng.standard_normal(cout).astype(np.float32))
    m = ops.build_map(cd)
    ops.finalize_maps([m])
    assert m.n == N
    nbr = ops.kernel_map(m, m, ks, 1)
    ones = torch.ones((N, 1), dtype=torch.float32, device=dev)
    wp = ops.pack_weights(W.to(dev))
    for sc, sh, relu in ((None, None, False), (scale, shift, True)):
        scd, shd = (None if t is None else t.to(dev) for t in (sc, sh))
        ref = ops.spconv(ones, nbr, K, 1, cout, wp, scale=scd, shift=shd, relu=relu, n_out=N)
        buf = torch.full((N, cout + 8), -7.0, dtype=torch.float32, device=dev)
        got = ops.occ_conv(m.coords, N, bbox, ks, W.view(K, cout).to(dev), scale=scd, shift=shd, relu=relu, out=buf[:, 4:4 + cout])
        assert torch.equal(got, ref)
        assert bool((buf[:, :4] == -7.0).all()) and bool((buf[:, 4 + cout:] == -7.0).all())
        orc = _oracle_conv(torch.ones(N, 1), nbr.cpu().numpy(), W, sc, sh, relu=relu)
        assert rel_l2(got.cpu(), orc) < 2e-6
    # a superset box (the raw points' box in the pipeline) gives the same result
    big = [bbox[0] - 3, bbox[1] - 40, bbox[2], bbox[3] + 37, bbox[4], bbox[5] + 2, bbox[6] + 1, 0]
    assert torch.equal(ops.occ_conv(m.coords, N, big, ks, W.view(K, cout).to(dev)), ops.spconv(ones, nbr, K, 1, cout, wp, n_out=N))
    # a box that would need more than 2 GB of bitmap is declined (the caller keeps the kernel-map path)
    assert not ops.occ_conv_supported([0, 0, 0, 1 << 20, 1 << 20, 64, 0, 0], ks, cout)
    assert not ops.occ_conv_supported(bbox, ks, cout + 1)
    # a box stretched by one far outlier voxel: still under 2 GB, but far more bitmap than voxels -> declined when the
    # caller states the voxel count (the kernel-map path is cheaper there)
    far, ext = list(bbox), 64
    while ops.occ_conv_supported(far, ks, cout, n=N) and ext < (1 << 22):
        ext *= 2
        far[3], far[4] = bbox[3] + ext, bbox[4] + ext // 8
    assert ops.occ_conv_supported(far, ks, cout) and not ops.occ_conv_supported(far, ks, cout, n=N)
    assert ops.occ_conv_supported(bbox, ks, cout, n=N)
    # a box that MISSES voxels (host-supplied, so it can be wrong): no access leaves the bitmap; the missed voxels come out
    # as NaN rows, voxels whose whole neighbourhood avoids the missed ones are unchanged
    if n > 1:
        small = list(bbox)
        small[3] = bbox[3] - 3                       # cut the top 3 x-layers
        small[6] = bbox[6]
        out_small = ops.occ_conv(m.coords, N, small, ks, W.view(K, cout).to(dev))
        ref_full = ops.spconv(ones, nbr, K, 1, cout, wp, n_out=N)
        cx = m.coords[:N, 1]
        missed = cx > small[3]
        assert bool(missed.any()) and bool(torch.isnan(out_small[missed]).all())
        safe = cx <= small[3] - ks // 2
        assert bool(torch.isfinite(out_small[~missed]).all()) and torch.equal(out_small[safe], ref_full[safe])
        if nbatch > 1:                               # a batch index above the box's: NaN rows as well
            nb_small = list(bbox)
            nb_small[6] = bbox[6] - 1
            out_b = ops.occ_conv(m.coords, N, nb_small, ks, W.view(K, cout).to(dev))
            gone = m.coords[:N, 0] > nb_small[6]
            assert bool(torch.isnan(out_b[gone]).all()) and torch.equal(out_b[~gone], ref_full[~gone])


@pytest.mark.parametrize("cin,cout,n,K", [(64, 32, 5000, 1), (64, 32, 70001, 1), (96, 64, 3000, 1), (32, 32, 2500, 27),
                                          (64, 128, 4000, 1), (32, 64, 900, 27), (64, 64, 40000, 1)])
def test_row_normalisation_fused_behind_the_last_layer(dev, cin, cout, n, K):
    """`l2norm` on a batched launch (the encoder's normalize_feature riding in the final layer's epilogue) == the conv
    followed by ops.l2_normalize, bit for bit, whichever kernel family runs the layer (tile kernel: fused; dense GEMM:
    a second launch inside the library); scale / shift / ReLU in front of it; strided output rows."""
    rng = np.random.default_rng(cin + cout + n)
    x = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32)).to(dev)
    W = torch.from_numpy((rng.standard_normal((K, cin, cout)) / np.sqrt(cin * 4)).astype(np.float32)).to(dev)
    nbr = None if K == 1 else torch.from_numpy(_random_map(rng, n, n, K, 0.3)).to(dev)
    scale = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32)).to(dev)
    shift = torch.from_numpy(rng.standard_normal(cout).astype(np.float32)).to(dev)
    wp = ops.pack_weights(W)
    for sc, sh, relu in ((None, None, False), (scale, shift, True)):
        plain = ops.spconv(x, nbr, K, cin, cout, wp, scale=sc, shift=sh, relu=relu, n_out=n)
        ref = ops.l2_normalize(plain.clone())
        buf = torch.full((n, cout + 8), 3.0, dtype=torch.float32, device=dev)
        b = ops.SpconvBatch()
        got = b.add(x, nbr, K, cin, cout, wp, scale=sc, shift=sh, relu=relu, out=buf[:, 4:4 + cout], n_out=n, l2norm=True)
        b.launch()
        assert torch.equal(got, ref)
        assert bool((buf[:, :4] == 3.0).all()) and bool((buf[:, 4 + cout:] == 3.0).all())
        nrm = got.norm(dim=1)
        ok = torch.isfinite(nrm)                 # an all-zero row (possible behind the ReLU) is 0 / 0 in both forms
        assert torch.allclose(nrm[ok], torch.ones_like(nrm[ok]), atol=1e-5)
