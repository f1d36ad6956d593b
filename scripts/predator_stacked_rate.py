"""Predator config 3 the way bench.py's `workloads.predator_config3` measures it (8 pairs stacked per forward, one host thread
keeping 8 batches in flight on 8 streams; median of 3 runs of 192 pairs; NBATCH, STREAMS override) + one pair at a time.  For A/B runs of the kernels
behind it (APR_KP_FUSED_NORM=0/1 ...)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from apr_amd import synth
from apr_amd.fcgf.pipeline import run_pipelined
from apr_amd.predator.configs.models import kitti_config
from apr_amd.predator.models.architectures import KPFCNN
from apr_amd.predator.pipeline import PredatorRegistration

dev = torch.device("cuda:0")
np.random.seed(0)
torch.manual_seed(0)
cfg = kitti_config()
pred = PredatorRegistration(KPFCNN(cfg).to(dev).eval(), cfg, [58, 59, 58, 57])
pool = []
for sd in range(8):
    pa, pb, _ = synth.make_pair(sd)
    pool.append((torch.from_numpy(pa).to(dev), torch.from_numpy(pb).to(dev)))


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


ta, tb = pool[0]
for i in range(3):
    pred(ta, tb, seed=i)
t0 = sync()
for i in range(10):
    pred(ta, tb, seed=i)
one = (sync() - t0) / 10
B, S, nbatch = int(os.environ.get("STACK", "8")), int(os.environ.get("STREAMS", "8")), int(os.environ.get("NBATCH", "24"))
batches = [[pool[(i * B + j) % len(pool)] for j in range(B)] for i in range(nbatch)]
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
mk = lambda i: pred.register_batch_phases(batches[i], seeds=range(i * B, i * B + B))
rates = []
for rep in range(4):
    t0 = sync()
    run_pipelined(mk, range(nbatch), streams)
    t1 = sync()
    if rep:
        rates.append(nbatch * B / (t1 - t0))
rates.sort()
print(json.dumps({"stacked_pairs_per_s": rates[1], "runs": rates, "one_pair_ms": 1e3 * one,
                  "fused_norm": os.environ.get("APR_KP_FUSED_NORM", "1")}))
