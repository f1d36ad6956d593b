"""GPU time of the fused encoder alone on one synthetic pair (A/B aid for the conv paths; APR_WS_STAGES=...)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apr_amd import synth
from apr_amd.fcgf.pipeline import PairRegistration
from bench import build_model

dev = torch.device("cuda:0")
model = build_model("ResUNetBN2C", 32, dev)
xyz0, xyz1, _ = synth.make_pair(0)
a, b = torch.from_numpy(xyz0).to(dev), torch.from_numpy(xyz1).to(dev)
pipe = PairRegistration(model, 0.3, ransac_iters=1000)
ref = None
for rep in range(3):
    ts = []
    for _ in range(30):
        coords, p0, p1, n0, n1 = pipe.voxelize_pair(a, b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); F0, F1 = pipe.encode_pair(coords, n0); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1000)
    ts.sort()
    print(f"APR_WS_STAGES={os.environ.get('APR_WS_STAGES', '<default>')}: encode median {ts[len(ts)//2]:.0f} us  min {ts[0]:.0f} us"
          f"  |F0|={float(F0.norm()):.6f}", flush=True)
