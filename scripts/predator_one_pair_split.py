"""Where one Predator pair's ~6 ms goes (one pair per call, one stream): host time between the fetches against time blocked on
the GPU, per phase."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from apr_amd import synth
from apr_amd.predator.configs.models import kitti_config
from apr_amd.predator.models.architectures import KPFCNN
from apr_amd.predator.pipeline import PredatorRegistration
dev = torch.device("cuda:0")
np.random.seed(0); torch.manual_seed(0)
cfg = kitti_config()
pred = PredatorRegistration(KPFCNN(cfg).to(dev).eval(), cfg, [58, 59, 58, 57])
pool = [tuple(torch.from_numpy(x).to(dev) for x in synth.make_pair(s)[:2]) for s in range(4)]
for i in range(6):
    pred(*pool[i % 4], seed=i)
N = 30
hosts, waits = [], []
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(N):
    gen = pred.register_batch_phases([pool[i % 4]], seeds=[i])
    t = time.perf_counter(); h, w = [], []
    try:
        pend = next(gen)
        while True:
            t1 = time.perf_counter(); h.append(t1 - t)
            pend.wait("sync")
            t = time.perf_counter(); w.append(t - t1)
            pend = next(gen)
    except StopIteration:
        h.append(time.perf_counter() - t)
    hosts.append(h); waits.append(w)
torch.cuda.synchronize(); total = time.perf_counter() - t0
print(f"ms/pair {1e3 * total / N:.3f}")
for p in range(len(hosts[0])):
    print(f"  host phase {p}: {1e6 * sum(h[p] for h in hosts) / N:8.1f} us" + (f"   then blocked {1e6 * sum(w[p] for w in waits) / N:8.1f} us" if p < len(waits[0]) else ""))
