"""Soak of the stacked Predator path: N distinct synthetic pairs in batches of 4; every batch timed (a pathological
kernel shows as an outlier: the one-second barycentre sort was found this way), every 5th batch compared pair by pair
with the one-pair-at-a-time result."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from apr_amd import synth
from apr_amd.predator.configs.models import kitti_config
from apr_amd.predator.models.architectures import KPFCNN
from apr_amd.predator.pipeline import PredatorRegistration
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
np.random.seed(0); torch.manual_seed(0)
cfg = kitti_config()
pipe = PredatorRegistration(KPFCNN(cfg).to(dev).eval(), cfg, [58, 59, 58, 57])
times, worst = [], 0.0
for b0 in range(0, N, 4):
    pairs = []
    for s in range(b0, b0 + 4):
        rng = np.random.default_rng(s)
        a, b, _ = synth.make_pair(1000 + s, n_beams=int(rng.choice([32, 64])), n_azimuth=int(rng.integers(900, 2000)))
        pairs.append((torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)))
    pipe.register_batch(pairs, seeds=range(b0, b0 + 4))          # sizes seen once (allocator)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    got = pipe.register_batch(pairs, seeds=range(b0, b0 + 4))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    times.append(dt)
    if (b0 // 4) % 5 == 0:
        for i, (p, (T, info)) in enumerate(zip(pairs, got)):
            T1, info1 = pipe(*p, seed=b0 + i)
            d = float(np.abs(T - T1).max())
            worst = max(worst, d)
            if d > 5e-2 or info["n0"] != info1["n0"]:
                print("MISMATCH batch", b0, "pair", i, d, info, info1, flush=True)
    print(f"batch {b0 // 4:3d}: {dt:7.2f} ms  ({dt / 4:.2f} ms/pair)  n0 {[i['n0'] for _, i in got]}", flush=True)
t = np.array(times)
print(f"batches {len(t)}: median {np.median(t):.2f} ms, max {t.max():.2f} ms (batch {int(t.argmax())}), worst pose difference stacked vs single {worst:.2e}")
