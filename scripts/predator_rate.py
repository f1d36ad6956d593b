"""Predator pairs/s in the bench's harness (4 pairs stacked per forward, one host thread, S batches in flight) -- for A/B
runs under environment switches.  S=4 B=4 NB=12 by default."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from apr_amd import synth
from apr_amd.fcgf.pipeline import run_pipelined
from apr_amd.predator.configs.models import kitti_config
from apr_amd.predator.models.architectures import KPFCNN
from apr_amd.predator.pipeline import PredatorRegistration
dev = torch.device("cuda:0")
np.random.seed(0); torch.manual_seed(0)
cfg = kitti_config()
pred = PredatorRegistration(KPFCNN(cfg).to(dev).eval(), cfg, [58, 59, 58, 57])
pool = [tuple(torch.from_numpy(x).to(dev) for x in synth.make_pair(s)[:2]) for s in range(8)]
B, S, NB = int(os.environ.get("B", "4")), int(os.environ.get("S", "4")), int(os.environ.get("NB", "12"))
batches = [[pool[(i * B + j) % len(pool)] for j in range(B)] for i in range(NB)]
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
mk = lambda i: pred.register_batch_phases(batches[i], seeds=range(i * B, i * B + B))
rates, host = [], []
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run_pipelined(mk, range(NB), streams)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    if rep:
        rates.append(NB * B / (t1 - t0)); host.append(1e3 * run_pipelined.last_host_busy_s / (NB * B))
print(f"B={B} S={S}: {sorted(rates)[1]:.1f} pairs/s  (runs {[round(r, 1) for r in rates]}), host busy {sorted(host)[1]:.2f} ms/pair")
