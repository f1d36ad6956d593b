"""RANSAC stage alone on the bench pair's correspondences (timing + survivor statistics)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apr_amd import ops, synth
from apr_amd.fcgf.pipeline import PairRegistration
from bench import build_model
dev = torch.device("cuda:0")
model = build_model("ResUNetBN2C", 32, dev)
xyz0, xyz1, _ = synth.make_pair(0)
a, b = torch.from_numpy(xyz0).to(dev), torch.from_numpy(xyz1).to(dev)
pipe = PairRegistration(model, 0.3)
coords, p0, p1, n0, n1 = pipe.voxelize_pair(a, b)
F0, F1 = pipe.encode_pair(coords, n0)
corr = ops.feature_nn(F0.contiguous(), F1.contiguous())
p0, p1 = p0.contiguous(), p1.contiguous()
for it in (4000000, 1000000):
    for _ in range(3): T, info = ops.ransac_pose(p0, p1, corr, 0.3, 0.9, it, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for r in range(20): T, info = ops.ransac_pose(p0, p1, corr, 0.3, 0.9, it, r)
    torch.cuda.synchronize()
    print(f"max_iter={it}: {(time.perf_counter()-t0)*50:.3f} ms per call   {info}", flush=True)
