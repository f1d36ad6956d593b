"""Per-stage wall time of one pair on the GPU (debug / DESIGN.md numbers)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from apr_amd import ops, synth
from apr_amd.fcgf.pipeline import PairRegistration
from bench import build_model

def T(msg, t0):
    torch.cuda.synchronize(); print(f"{msg}: {1000*(time.perf_counter()-t0):.2f} ms", flush=True)

dev = torch.device("cuda:0")
t0 = time.perf_counter(); model = build_model("ResUNetBN2C", 32, dev); T("model", t0)
t0 = time.perf_counter(); xyz0, xyz1, _ = synth.make_pair(0); T("synth", t0)
a, b = torch.from_numpy(xyz0).to(dev), torch.from_numpy(xyz1).to(dev)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
pipe = PairRegistration(model, 0.3, ransac_iters=iters)
for rep in range(3):
    print("--- rep", rep, flush=True)
    t0 = time.perf_counter(); coords, p0, p1, n0, n1 = pipe.voxelize_pair(a, b); T(f"voxelize {n0}+{n1}", t0)
    t0 = time.perf_counter(); F0, F1 = pipe.encode_pair(coords, n0); T("encode", t0)
    t0 = time.perf_counter(); corr = ops.feature_nn(F0.contiguous(), F1.contiguous()); T("feature_nn", t0)
    t0 = time.perf_counter(); Tm, info = ops.ransac_pose(p0.contiguous(), p1.contiguous(), corr, 0.3, 0.9, iters, rep); T("ransac", t0)
    print(info, flush=True)
