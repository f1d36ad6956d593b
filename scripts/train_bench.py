"""FCGF training iteration on the HIP path: encode both frames of a KITTI-shaped pair in train
mode, hardest-contrastive loss, backward, SGD step — the body of FCGF_APR/lib/trainer.py:454-527.  Stage times."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from apr_amd import MinkowskiEngine as ME, ops, synth
from apr_amd.fcgf.lib.trainer import HardestContrastiveLoss
from apr_amd.fcgf.lib.apg import get_matching_indices
from bench import build_model
dev = torch.device("cuda:0")
model = build_model("ResUNetBN2C", 32, dev).train()
opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.8)
crit = HardestContrastiveLoss()
xyz0, xyz1, T = synth.make_pair(0)
a, b = torch.from_numpy(xyz0).to(dev), torch.from_numpy(xyz1).to(dev)

def frame(xyz, bidx):
    m = ops.build_map(ops.voxelize(xyz, 0.3, bidx), want_first=True)
    ops.finalize_maps([m])
    return m.coords, xyz[m.first].contiguous()

def sync():
    torch.cuda.synchronize(); return time.perf_counter()

rows = []
for it in range(6):
    t0 = sync()
    c0, p0 = frame(a, 0); c1, p1 = frame(b, 0)
    pairs = get_matching_indices(p0, p1, torch.from_numpy(T).float().to(dev), 0.3)
    t1 = sync()
    opt.zero_grad()
    F0 = model(ME.SparseTensor(torch.ones(len(c0), 1, device=dev), coordinates=c0)).F
    F1 = model(ME.SparseTensor(torch.ones(len(c1), 1, device=dev), coordinates=c1)).F
    t2 = sync()
    np.random.seed(it)
    pl, nl = crit.contrastive_hardest_negative_loss(F0, F1, pairs.cpu().numpy() if torch.is_tensor(pairs) else pairs,
                                                    num_pos=1024, num_hn_samples=256)
    loss = pl + nl
    t3 = sync()
    loss.backward()
    t4 = sync()
    opt.step()
    t5 = sync()
    if it:
        rows.append([t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4])
    print(f"iter {it}: loss {loss.item():.4f} (pos {pl.item():.4f} neg {nl.item():.4f}), voxels {len(c0)}+{len(c1)}, pairs {len(pairs)}", flush=True)
r = np.array(rows).mean(0) * 1e3
print(f"ms: voxelise+GT pairs {r[0]:.2f} | forward x2 (train mode) {r[1]:.2f} | loss (mining + value) {r[2]:.2f} | backward {r[3]:.2f} | SGD {r[4]:.2f} | total {r.sum():.2f}  -> {1e3 / r.sum():.1f} iterations/s")
