"""Where one pair's 1.4 ms goes (one pair per call, one stream): host time between the fetches (Python + launch calls) against
time blocked waiting for the GPU, per phase, plus cProfile of the host part."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apr_amd import ops, synth
from apr_amd.fcgf.pipeline import PairRegistration
from bench import build_model
dev = torch.device("cuda:0")
pipe = PairRegistration(build_model("ResUNetBN2C", 32, dev), voxel_size=0.3, ransac_iters=4000000)
pool = [tuple(torch.from_numpy(x).to(dev) for x in synth.make_pair(s)[:2]) for s in range(6)]
for i in range(12):
    pipe.register_batch([pool[i % 6]], seeds=[i])
N = 60
waits, hosts = [], []


def run_one(i):
    gen = pipe.register_batch_phases([pool[i % 6]], seeds=[i])
    t = time.perf_counter()
    h, w = [], []
    try:
        pend = next(gen)
        while True:
            t1 = time.perf_counter(); h.append(t1 - t)
            pend.wait("sync")
            t = time.perf_counter(); w.append(t - t1)
            pend = next(gen)
    except StopIteration:
        h.append(time.perf_counter() - t)
    return h, w


torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(N):
    h, w = run_one(i); hosts.append(h); waits.append(w)
torch.cuda.synchronize(); total = time.perf_counter() - t0
print(f"ms/pair {1e3 * total / N:.3f}")
nh = len(hosts[0])
for p in range(nh):
    print(f"  host phase {p}: {1e6 * sum(h[p] for h in hosts) / N:7.1f} us" + (f"   then blocked {1e6 * sum(w[p] for w in waits) / N:7.1f} us" if p < len(waits[0]) else ""))
pr = cProfile.Profile(); pr.enable()
for i in range(N):
    run_one(i)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
