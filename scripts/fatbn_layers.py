"""ResUNetFatBN (APR's encoder, 128 features): pairs/s with three steps in flight (one scheduler thread) and the per-layer
conv times of one 12-frame encode (HIP events on the launch stream)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from apr_amd import synth, ops
from apr_amd.fcgf.pipeline import PairRegistration, run_pipelined
dev = torch.device("cuda:0")
fat = bench.build_model("ResUNetFatBN", 128, dev)
pipe = PairRegistration(fat, voxel_size=0.3, ransac_iters=4000000)
pool6 = [tuple(torch.from_numpy(x).to(dev) for x in synth.make_pair(s)[:2]) for s in range(6)]
st = [torch.cuda.Stream(device=dev) for _ in range(3)]
mk = lambda i: pipe.register_batch_phases([pool6[(i * 6 + j) % 6] for j in range(6)], seeds=[i * 6 + j for j in range(6)])
r = []
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run_pipelined(mk, range(24), st)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    if rep: r.append(24 * 6 / (t1 - t0))
prof = ops.SpconvProfile(); ops.PROFILE = prof
batch = [pool6[j % 6] for j in range(6)]
for i in range(3): pipe.encode_batch(pipe.voxelize_batch([c for p in batch for c in p])[0])
ops.PROFILE = None
fs = prof.summary()
print("fatbn pairs/s", [round(x, 1) for x in r], "conv us/encode", round(1e3 * fs["ms"] / 3, 1), {k: round(1e3 * d["ms"] / 3, 1) for k, d in fs["by_path"].items()})
rec = prof.records[-23:]
for (P, cin, cout, mfma, ms, _e1, path) in rec:
    print(f"  {path:5s} cin {cin:4d} cout {cout:4d} P {P:8d}  {1e3 * ms:7.1f} us  {2.0 * P * cin * cout / (ms * 1e-3) / 1e12:6.1f} TF  {(4.0 * P * (cin + cout) + 8 * P) / (ms * 1e-3) / 1e9:7.0f} GB/s")
