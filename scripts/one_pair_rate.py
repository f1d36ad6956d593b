"""BASELINE config 2 literally, on its own: ONE pair per call, one stream, the pose on the host before the next pair starts
(bench.py's workloads.fcgf_one_pair).  REPS runs of N pairs; prints ms per pair per run.  A/B aid for routing switches:
  APR_OS_MIN_ROWS=0 python scripts/one_pair_rate.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apr_amd import synth
from apr_amd.fcgf.pipeline import PairRegistration
from bench import build_model
dev = torch.device("cuda:0")
N, REPS = int(os.environ.get("N", "60")), int(os.environ.get("REPS", "3"))
pipe = PairRegistration(build_model("ResUNetBN2C", 32, dev), voxel_size=0.3, ransac_iters=4000000)
pipe.fetch_wait = os.environ.get("FETCH_WAIT", "sync")
pool = [tuple(torch.from_numpy(x).to(dev) for x in synth.make_pair(s)[:2]) for s in range(6)]
for i in range(12):
    pipe.register_batch([pool[i % 6]], seeds=[i])
out = []
for r in range(REPS):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(N):
        pipe.register_batch([pool[i % 6]], seeds=[i])
    torch.cuda.synchronize(); out.append(1e3 * (time.perf_counter() - t0) / N)
print("ms/pair " + " ".join(f"{v:.3f}" for v in out) + f"  best {min(out):.3f} = {1e3 / min(out):.0f} pairs/s", flush=True)
