"""Host time of the phases of one batched step (no profiler): perf_counter around the enqueue code between the syncs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apr_amd import synth, ops
from apr_amd import MinkowskiEngine as ME
from apr_amd.fcgf.pipeline import PairRegistration
from bench import build_model
dev = torch.device("cuda:0")
B = int(os.environ.get("B", "6"))
model = build_model("ResUNetBN2C", 32, dev)
pipe = PairRegistration(model, 0.3)
pairs = [tuple(torch.from_numpy(x).to(dev) for x in synth.make_pair(s)[:2]) for s in range(12)]
acc = {}
def T(name, t0):
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
def step(i):
    batch = [pairs[(i * B + j) % len(pairs)] for j in range(B)]
    clouds = [c for p in batch for c in p]
    t = time.perf_counter()
    offs = [0]
    for c in clouds: offs.append(offs[-1] + int(c.shape[0]))
    xyz_all = torch.cat(clouds)
    offs_dev = torch.tensor(offs, dtype=torch.int64).to(dev, non_blocking=True)
    T("A1 cat+offsets", t); t = time.perf_counter()
    coords_all = ops.voxelize_segments(xyz_all, 0.3, offs_dev)
    m = ops.build_map(coords_all, want_first=True)
    counts_dev = ops.segment_counts(m, offs_dev)
    T("A2 voxelize+map+counts", t); t = time.perf_counter()
    cm = ME.CoordinateManager(base_map=m)
    pending = cm.build_pyramid_async([2, 4, 8], extras=[counts_dev])
    T("A3 pyramid enqueue", t); t = time.perf_counter()
    pending.event.synchronize()
    T("wait sizes", t); t = time.perf_counter()
    (counts,) = pending.finish()
    counts = [int(c) for c in counts]
    pts_all = xyz_all[m.first]
    T("B finish+gather", t); t = time.perf_counter()
    F = pipe.encode_batch(cm)
    T("C encode enqueue", t); t = time.perf_counter()
    o = [0]
    for n in counts: o.append(o[-1] + n)
    pts = [pts_all[o[b]:o[b + 1]] for b in range(len(clouds))]
    pend = ops.match_pose_batch_async([F[o[2 * k]:o[2 * k + 1]] for k in range(B)], [F[o[2 * k + 1]:o[2 * k + 2]] for k in range(B)],
                                      pts[0::2], pts[1::2], 0.3, 0.9, 4000000, seeds=list(range(B)))
    T("D match enqueue", t); t = time.perf_counter()
    pend.event.synchronize()
    T("wait poses", t); t = time.perf_counter()
    r = pend.finish()
    T("E decode", t)
    return r
for i in range(10): step(i)
acc.clear()
N = 40
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(N): step(i)
torch.cuda.synchronize(); tot = time.perf_counter() - t0
print(f"wall {1e3 * tot / N:.3f} ms/step")
for k, v in acc.items(): print(f"  {k:28s} {1e3 * v / N:7.3f} ms")
print(f"  host busy (no waits)         {1e3 * sum(v for k, v in acc.items() if not k.startswith('wait')) / N:7.3f} ms")
