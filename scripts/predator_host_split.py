"""Where the Predator scheduler thread's time goes (verdict r3 #5: host- or GPU-bound?): wall time and thread CPU time inside
the generators (run_pipelined.last_host_busy_s), split into time inside libapr_hip calls (ctypes: argument marshalling + the
kernel launches, which block when a stream's queue is full) and the Python around them.  Also the same loop with every kernel
launch of the library skipped is not possible from here; instead the call count and the mean time per call tell whether calls
block (a launch that does not block costs ~5-8 us through ctypes).   B=4 S=8 NB=16 by default."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from apr_amd import _lib, synth
from apr_amd.fcgf.pipeline import run_pipelined
from apr_amd.predator.configs.models import kitti_config
from apr_amd.predator.models.architectures import KPFCNN
from apr_amd.predator.pipeline import PredatorRegistration

dev = torch.device("cuda:0")
np.random.seed(0); torch.manual_seed(0)
cfg = kitti_config()
pred = PredatorRegistration(KPFCNN(cfg).to(dev).eval(), cfg, [58, 59, 58, 57])
pool = [tuple(torch.from_numpy(x).to(dev) for x in synth.make_pair(s)[:2]) for s in range(8)]
B, S, NB = int(os.environ.get("B", "4")), int(os.environ.get("S", "8")), int(os.environ.get("NB", "16"))
batches = [[pool[(i * B + j) % len(pool)] for j in range(B)] for i in range(NB)]
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
mk = lambda i: pred.register_batch_phases(batches[i], seeds=range(i * B, i * B + B))
for rep in range(2):
    run_pipelined(mk, range(NB), streams)
torch.cuda.synchronize()

# timing proxies around every exported function
lib = _lib.load()
stat = {}


class Proxy:
    def __init__(self, name, fn):
        self.name, self.fn = name, fn
        stat[name] = [0, 0.0, 0.0]

    def __call__(self, *a):
        t0 = time.perf_counter()
        r = self.fn(*a)
        dt = time.perf_counter() - t0
        s = stat[self.name]
        s[0] += 1
        s[1] += dt
        if dt > s[2]:
            s[2] = dt
        return r


class LibProxy:
    def __init__(self, lib):
        self._lib = lib
        self._cache = {}

    def __getattr__(self, name):
        p = self._cache.get(name)
        if p is None:
            p = self._cache[name] = Proxy(name, getattr(self._lib, name))
        return p


_lib._lib = LibProxy(lib)
c0, t0 = time.thread_time(), time.perf_counter()
run_pipelined(mk, range(NB), streams)
torch.cuda.synchronize()
c1, t1 = time.thread_time(), time.perf_counter()
_lib._lib = lib
npairs = NB * B
busy = run_pipelined.last_host_busy_s
incall = sum(s[1] for s in stat.values())
ncall = sum(s[0] for s in stat.values())
print(f"B={B} S={S} NB={NB}: {npairs / (t1 - t0):.1f} pairs/s (with proxies); wall {1e3 * (t1 - t0) / npairs:.3f} ms/pair, scheduler CPU "
      f"{1e3 * (c1 - c0) / npairs:.3f} ms/pair, inside generators {1e3 * busy / npairs:.3f} ms/pair")
print(f"library calls: {ncall / npairs:.0f} per pair, {1e3 * incall / npairs:.3f} ms/pair inside them (mean {1e6 * incall / ncall:.1f} us), "
      f"Python around them {1e3 * (busy - incall) / npairs:.3f} ms/pair")
for name, s in sorted(stat.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"  {name:38s} {s[0] / npairs:7.1f} calls/pair  mean {1e6 * s[1] / s[0]:7.1f} us  max {1e6 * s[2]:8.1f} us  total {1e3 * s[1] / npairs:.3f} ms/pair")
