"""Is the batched pair pipeline host- or GPU-bound?  One thread / one stream: host time to ENQUEUE a step (the loop
returns before the GPU is done unless the step syncs) vs wall time per step, and the GPU time between events."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apr_amd import synth
from apr_amd.fcgf.pipeline import PairRegistration
from bench import build_model
dev = torch.device("cuda:0")
B = int(os.environ.get("B", "6"))
model = build_model("ResUNetBN2C", 32, dev)
pipe = PairRegistration(model, 0.3, ransac_iters=int(os.environ.get("ITERS", "4000000")))
pairs = [tuple(torch.from_numpy(x).to(dev) for x in synth.make_pair(s)[:2]) for s in range(12)]
def step(i):
    batch = [pairs[(i * B + j) % len(pairs)] for j in range(B)]
    return pipe.register_batch(batch, seeds=[i * B + j for j in range(B)])
for i in range(10): step(i)
torch.cuda.synchronize()
N = 40
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for i in range(N): step(i)
t1 = time.perf_counter(); e1.record()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B}: host loop {1e3*(t1-t0)/N:.2f} ms/step, wall {1e3*(t2-t0)/N:.2f} ms/step, gpu span {e0.elapsed_time(e1)/N:.2f} ms/step", flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(N): step(i)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
