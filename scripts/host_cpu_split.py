#!/usr/bin/env python3
"""Where a rank's host CPU time goes in the 3-steps-in-flight FCGF loop: per worker thread, CPU seconds spent waiting for a
fetch event vs enqueueing; process CPU time vs the sum over the worker threads (the remainder = threads of the HIP runtime).
APR_BLOCKING_EVENTS=0/1, HIP/ROCr wait knobs via the environment.   python scripts/host_cpu_split.py [steps]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from apr_amd import ops, synth
from apr_amd.fcgf.pipeline import PairRegistration

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
S, B = 3, 6
dev = torch.device("cuda:0")
model = bench.build_model("ResUNetBN2C", 32, dev)
pipe = PairRegistration(model, voxel_size=0.3, ransac_iters=4000000)
pairs = [tuple(torch.from_numpy(x).to(dev) for x in synth.make_pair(s)[:2]) for s in range(12)]
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
stat = [dict(cpu=0.0, wait_cpu=0.0, wait_wall=0.0, fin_cpu=0.0) for _ in range(S)]


def run(w, first, last, rec):
    torch.cuda.set_device(dev)
    st = stat[w]
    c0 = time.thread_time()
    with torch.cuda.stream(streams[w]):
        for i in range(first + w, last, S):
            gen = pipe.register_batch_phases([pairs[(i * B + j) % 12] for j in range(B)], seeds=[i * B + j for j in range(B)])
            try:
                pending = next(gen)
                while True:
                    a, ta = time.thread_time(), time.perf_counter()
                    pending.wait()
                    if rec:
                        st["wait_cpu"] += time.thread_time() - a
                        st["wait_wall"] += time.perf_counter() - ta
                    pending = gen.send(None)
            except StopIteration:
                pass
        streams[w].synchronize()
    if rec:
        st["cpu"] += time.thread_time() - c0


def loop(first, last, rec):
    ts = [threading.Thread(target=run, args=(w, first, last, rec)) for w in range(S)]
    [t.start() for t in ts]
    [t.join() for t in ts]


loop(0, 24, False)
torch.cuda.synchronize()
p0, t0 = time.process_time(), time.perf_counter()
loop(24, 24 + steps, True)
torch.cuda.synchronize()
p1, t1 = time.process_time(), time.perf_counter()
wall = t1 - t0
print(f"fetch_wait={ops.FETCH_WAIT} poll_us={1e6 * ops.FETCH_POLL_S:.0f} blocking_events={ops.BLOCKING_EVENTS} steps={steps} wall={wall:.3f}s pairs/s={steps * B / wall:.0f}")
print(f"process CPU {p1 - p0:.3f}s = {(p1 - p0) / wall:.2f} CPUs busy; per step {1e3 * (p1 - p0) / steps:.2f} ms")
tot = 0.0
for w, st in enumerate(stat):
    tot += st["cpu"]
    print(f"  worker {w}: cpu {st['cpu']:.3f}s (waiting on events: cpu {st['wait_cpu']:.3f}s over wall {st['wait_wall']:.3f}s; "
          f"enqueue + finish {st['cpu'] - st['wait_cpu']:.3f}s)")
print(f"  sum of workers {tot:.3f}s; other threads of the process (HIP runtime) {p1 - p0 - tot:.3f}s")
try:
    print("cpu.stat:", open("/sys/fs/cgroup/cpu.stat").read().replace("\n", " "))
except OSError:
    pass
