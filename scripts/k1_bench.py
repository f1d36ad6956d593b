"""K = 1 (identity-map) layers of the two encoders at 12 frames: the tile kernel against the dense bf16-split GEMM, and
the per-layer conv times of a ResUNetBN2C encode (the FatBN list is scripts/fatbn_layers.py)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from apr_amd import synth, ops
from apr_amd.fcgf.pipeline import PairRegistration
dev = torch.device("cuda:0")


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


M = int(os.environ.get("ROWS", "189191"))
for cin, cout in [(128, 128), (192, 128), (160, 128), (64, 64), (96, 64), (64, 32), (128, 64)]:
    x = torch.randn(M, cin, device=dev)
    W = torch.randn(1, cin, cout, device=dev) * 0.05
    out = torch.empty(M, cout, device=dev)
    wp = ops.pack_weights(W)
    t_tile = timeit(lambda: ops.spconv(x, None, 1, cin, cout, wp, out=out))
    line = f"K=1 M={M} {cin:4d}->{cout:4d}: tile {t_tile:7.1f} us"
    if cin % 64 == 0 and cout % 64 == 0:
        w3 = ops.pack_weights_bf3(W)
        t_d = timeit(lambda: ops.dense_gemm_bf3(x, w3, cin, cout, out=out))
        line += f"  dense-bf3 {t_d:7.1f} us"
    if ops._lib_().apr_dense_rows_bf3_ok(cin, cout):
        w3 = ops.pack_weights_bf3(W)
        t_r = timeit(lambda: ops.dense_rows_bf3(x, w3, cin, cout, out=out))
        t_n = timeit(lambda: ops.dense_rows_bf3(x, w3, cin, cout, l2norm=True, out=out))
        line += f"  rows-bf3 {t_r:7.1f} us (+ row norm {t_n:7.1f}; separate norm {timeit(lambda: ops.l2_normalize(out, out=out)):6.1f})"
    line += f"   roof {4.0 * M * (cin + cout) / 8e6:6.1f} us"
    print(line, flush=True)

if os.environ.get("LAYERS", "1") == "1":
    m = bench.build_model("ResUNetBN2C", 32, dev)
    pipe = PairRegistration(m, voxel_size=0.3, ransac_iters=4000000)
    pool6 = [tuple(torch.from_numpy(x).to(dev) for x in synth.make_pair(s)[:2]) for s in range(6)]
    prof = ops.SpconvProfile(); ops.PROFILE = prof
    for i in range(3): pipe.encode_batch(pipe.voxelize_batch([c for p in pool6 for c in p])[0])
    ops.PROFILE = None
    fs = prof.summary()
    print("bn2c conv us/encode", round(1e3 * fs["ms"] / 3, 1), {k: round(1e3 * d["ms"] / 3, 1) for k, d in fs["by_path"].items()})
    n = len(prof.records) // 3
    for (P, cin, cout, mfma, ms, _e1, path) in prof.records[-n:]:
        print(f"  {path:5s} cin {cin:4d} cout {cout:4d} P {P:8d}  {1e3 * ms:7.1f} us  {(4.0 * P * (cin + cout) + 8 * P) / (ms * 1e-3) / 1e9:7.0f} GB/s")
