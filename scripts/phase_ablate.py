"""Where the three-steps-in-flight FCGF loop spends the card: the full step against the step without matching + RANSAC
(encoder and index building only) and without the encoder (matching on stored features).  One scheduler thread, 3 streams,
6 pairs per step (the harness of workloads.fcgf_planted_30pct)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from apr_amd import ops, synth
from apr_amd.fcgf.model import load_model
from apr_amd.fcgf.pipeline import PairRegistration, run_pipelined

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3).to(dev).eval()
pipe = PairRegistration(model, voxel_size=0.3)
pool = [tuple(torch.from_numpy(x).to(dev) for x in synth.make_pair(s)[:2]) for s in range(12)]
B, S, NS = 6, int(os.environ.get("S", "3")), 60
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
batches = [[pool[(i * B + j) % len(pool)] for j in range(B)] for i in range(NS)]


def rate(make):
    out = []
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run_pipelined(make, range(NS), streams)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        if rep:
            out.append(NS * B / (t1 - t0))
    return sorted(out)[1]


full = lambda i: pipe.register_batch_phases(batches[i], seeds=range(i * B, i * B + B))

# encoder only: the generator is abandoned after the encode (its match phase never enqueued)
orig_match = ops.match_pose_batch_async


class _Done:
    def __init__(self):
        self.event = torch.cuda.Event(); self.event.record()
    def finish(self):
        return []


def no_match(f0, f1, p0, p1, *a, **k):
    return _Done()


def enc_only(i):
    return pipe.register_batch_phases(batches[i], seeds=range(i * B, i * B + B))


print(f"full step            : {rate(full):8.1f} pairs/s", flush=True)
ops.match_pose_batch_async = no_match
import apr_amd.fcgf.pipeline as P
print(f"without NN + RANSAC  : {rate(enc_only):8.1f} pairs/s", flush=True)
ops.match_pose_batch_async = orig_match

# matching only: features and points of every batch stored once, the step is the batched matching call alone
stored = []
for i in range(NS if NS < 12 else 12):
    cm, counts, first, offs, pts_all = pipe.voxelize_batch([c for p in batches[i] for c in p])
    F = pipe.encode_batch(cm)
    o = np.concatenate([[0], np.cumsum(counts)])
    stored.append(([F[o[2 * j]:o[2 * j + 1]].contiguous() for j in range(B)], [F[o[2 * j + 1]:o[2 * j + 2]].contiguous() for j in range(B)],
                   [pts_all[o[2 * j]:o[2 * j + 1]].contiguous() for j in range(B)], [pts_all[o[2 * j + 1]:o[2 * j + 2]].contiguous() for j in range(B)]))
torch.cuda.synchronize()


def match_only(i):
    f0, f1, p0, p1 = stored[i % len(stored)]
    pend = ops.match_pose_batch_async(f0, f1, p0, p1, pipe.distance_threshold, pipe.edge_length, pipe.ransac_iters,
                                      seeds=list(range(i * B, i * B + B)))
    yield pend
    return pend.finish()


print(f"NN + RANSAC only     : {rate(match_only):8.1f} pairs/s", flush=True)
