"""Host-side (Python) cost of the pair pipeline: cProfile, sorted by self time."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apr_amd import synth
from apr_amd.fcgf.pipeline import PairRegistration
from bench import build_model
dev = torch.device("cuda:0")
model = build_model("ResUNetBN2C", 32, dev)
pipe = PairRegistration(model, 0.3)
pairs = [tuple(torch.from_numpy(x).to(dev) for x in synth.make_pair(s)[:2]) for s in range(4)]
for i in range(8): pipe(*pairs[i % 4], seed=i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(100): pipe(*pairs[i % 4], seed=i)
torch.cuda.synchronize()
print("ms/pair (1 thread, 1 stream)", (time.perf_counter() - t0) * 10)
pr = cProfile.Profile(); pr.enable()
for i in range(100): pipe(*pairs[i % 4], seed=i)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
