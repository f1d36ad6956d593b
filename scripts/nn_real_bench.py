"""Feature NN on the features the bench actually produces (random-init encoder: collapsed, ~60 near-ties per query):
time per call for ResUNetBN2C-32 and ResUNetFatBN-128 features of one synthetic pair."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apr_amd import ops, synth
from apr_amd.fcgf.pipeline import PairRegistration
from bench import build_model
dev = torch.device("cuda:0")
xyz0, xyz1, _ = synth.make_pair(0)
a, b = torch.from_numpy(xyz0).to(dev), torch.from_numpy(xyz1).to(dev)
for name, c in (("ResUNetBN2C", 32), ("ResUNetFatBN", 128)):
    if os.environ.get("ONLY_C") and int(os.environ["ONLY_C"]) != c:
        continue
    model = build_model(name, c, dev)
    pipe = PairRegistration(model, 0.3)
    coords, p0, p1, n0, n1 = pipe.voxelize_pair(a, b)
    F0, F1 = pipe.encode_pair(coords, n0)
    F0, F1 = F0.contiguous(), F1.contiguous()
    ref = ops.feature_nn(F0, F1, impl="brute")
    got = ops.feature_nn(F0, F1, impl="fast")
    same = all(torch.equal(x, y) for x, y in zip(ref, got)) if isinstance(ref, (tuple, list)) else torch.equal(ref, got)
    for _ in range(3): ops.feature_nn(F0, F1, impl="fast")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.feature_nn(F0, F1, impl="fast")
    e1.record(); torch.cuda.synchronize()
    print(f"{name} {n0}x{n1}x{c}: {e0.elapsed_time(e1) * 50:8.1f} us per search, equal to brute force: {same}", flush=True)
