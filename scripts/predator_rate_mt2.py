"""Predator pairs/s with T scheduler threads, each keeping S/T batches in flight on its own streams (run_pipelined per thread):
the library calls (~1.1 ms of the 2.6 ms of host time per pair) release the GIL, so two schedulers can overlap one's C time
with the other's Python.  T=1,2,3 S=8 B=4 NB=24."""
import os, sys, threading, time
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
B, S, NB = int(os.environ.get("B", "4")), int(os.environ.get("S", "8")), int(os.environ.get("NB", "24"))
batches = [[pool[(i * B + j) % len(pool)] for j in range(B)] for i in range(NB)]
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
mk = lambda i: pred.register_batch_phases(batches[i], seeds=range(i * B, i * B + B))
if os.environ.get("SWITCH"):
    sys.setswitchinterval(float(os.environ["SWITCH"]))
for T in [int(x) for x in os.environ.get("T", "1,2,3").split(",")]:
    rates = []
    for rep in range(4):
        def work(t):
            torch.cuda.set_device(dev)
            run_pipelined(mk, range(t, NB, T), streams[t::T])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if T == 1:
            work(0)
        else:
            th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
            [x.start() for x in th]; [x.join() for x in th]
        torch.cuda.synchronize(); t1 = time.perf_counter()
        if rep:
            rates.append(NB * B / (t1 - t0))
    print(f"T={T} S={S} B={B}: {sorted(rates)[1]:.1f} pairs/s (runs {[round(r, 1) for r in rates]})", flush=True)
