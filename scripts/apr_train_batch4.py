"""The APR training iteration at the reference's batch size (scripts/train_apr_kitti.sh: BATCH_SIZE 4): four full-size pairs
per iteration (two distinct synthetic pairs, each twice), frames stacked per side.  ms per iteration and pairs/s."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from apr_amd.fcgf.lib import complement_trainer as CT

dev = torch.device("cuda:0")
_r = torch.empty(8 << 30, dtype=torch.uint8, device=dev); del _r
parts = [CT.synthetic_batch(dev, seed=s) for s in (0, 1)]
parts = [parts[0], parts[1], parts[0], parts[1]]
shift = lambda C, k: torch.cat((C[:, :1] + k, C[:, 1:]), 1)
batch = {"len_batch": [], "pcd_nghb0": [], "pcd_nghb1": []}
corr, o0, o1 = [], 0, 0
for i, p in enumerate(parts):
    batch["len_batch"] += p["len_batch"]
    for t in ("0", "1"):
        batch[f"pcd_nghb{t}"] += p[f"pcd_nghb{t}"]
    corr.append(p["correspondences"] + torch.tensor([[o0, o1]]))
    o0 += p["sinput0_C"].shape[0]; o1 += p["sinput1_C"].shape[0]
for t in ("0", "1"):
    batch[f"sinput{t}_C"] = torch.cat([shift(p[f"sinput{t}_C"], i) for i, p in enumerate(parts)], 0).contiguous()
    batch[f"sinput{t}_F"] = torch.cat([p[f"sinput{t}_F"] for p in parts], 0)
batch["correspondences"] = torch.cat(corr, 0)
step = CT.build_step(dev)
step.num_pos, step.num_hn = 1024 * 4, 256 * 4
rows, walls = [], []
for it in range(8):
    np.random.seed(it)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = step(batch, timed=True)
    torch.cuda.synchronize(); walls.append((time.perf_counter() - t0) * 1e3); rows.append(r["ms"])
    print(it, float(r["loss"]), walls[-1], r["ms"], file=sys.stderr)
torch.cuda.synchronize(); t0 = time.perf_counter()
for it in range(10):
    np.random.seed(50 + it); step(batch)
torch.cuda.synchronize(); free = (time.perf_counter() - t0) * 100
ms = {k: float(np.median([x[k] for x in rows[2:]])) for k in rows[0]}
print(json.dumps({"batch_size": 4, "ms_per_iteration": float(np.median(walls[2:])), "ms_per_iteration_unsynchronised": free,
                  "pairs_per_s": 4e3 / free, "stages_ms": ms, "rows_per_side": [int(o0), int(o1)]}))
