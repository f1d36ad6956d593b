"""How discriminative are the bench's features?  Near-tie counts around the NN distance (decides the NN strategy)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apr_amd import ops, synth
from apr_amd.fcgf.pipeline import PairRegistration
from bench import build_model
dev = torch.device("cuda:0")
model = build_model("ResUNetBN2C", 32, dev)
xyz0, xyz1, _ = synth.make_pair(0)
a, b = torch.from_numpy(xyz0).to(dev), torch.from_numpy(xyz1).to(dev)
pipe = PairRegistration(model, 0.3)
coords, p0, p1, n0, n1 = pipe.voxelize_pair(a, b)
F0, F1 = pipe.encode_pair(coords, n0)
q = F0[torch.randperm(n0, device=dev)[:1024]].double()
t = F1.double()
d = ((q[:, None, :] - t[None, :, :]) ** 2).sum(-1)          # [1024, n1]
mn = d.min(1).values
print("features: |F| mean", float(F0.norm(dim=1).mean()), " per-channel std over voxels", float(F0.std(0).mean()))
print("NN d2: median %.3e  p10 %.3e  p90 %.3e" % (float(mn.median()), float(mn.quantile(0.1)), float(mn.quantile(0.9))))
print("all-pairs d2 median %.3e" % float(d.flatten()[::97].median()))
for w in (0.0, 1e-7, 3e-6, 2e-5, 2e-4, 2e-3, 1.6e-2):
    c = (d <= (mn[:, None] + w)).sum(1).double()
    print(f"targets within +{w:.1e} of the min: mean {float(c.mean()):9.1f}  median {float(c.median()):7.0f}  max {int(c.max())}")
uq = torch.unique(F1, dim=0).shape[0]
print("distinct target rows:", uq, "of", n1)

# candidate count of the filter + refine path (overflow flag, then per-wave counts, sit after the U array)
from apr_amd import _lib
from apr_amd._lib import ptr, stream, check
lib = _lib.load()
f0, f1 = F0.contiguous(), F1.contiguous()
c = f0.shape[1]
al = lambda x: (x + 255) & ~255
sb = int(lib.apr_feature_nn_fast_scratch_bytes(n0, n1, c))
scratch = torch.zeros(sb + 256, dtype=torch.uint8, device=dev)
best = torch.empty(n0, dtype=torch.int64, device=dev)
check(lib.apr_feature_nn_fast(ptr(f0), n0, ptr(f1), n1, c, ptr(best), ptr(scratch), sb, stream()))
torch.cuda.synchronize()
qblocks = (n0 + 255) // 256
want = (768 + qblocks - 1) // qblocks
chunk = max(256, ((((n1 + want - 1) // want) + 63) // 64) * 64)
nwaves = qblocks * ((n1 + chunk - 1) // chunk) * 4
base = (-scratch.data_ptr()) % 256
off = base + 2 * al(n0 * c * 2) + 2 * al(n1 * c * 2) + al(n0 * 16) + al(n1 * 16) + al(n0 * 4)
ovf = int(scratch[off:off + 4].view(torch.int32).cpu()[0])
cnt = scratch[off + 256:off + 256 + 4 * nwaves].view(torch.int32).cpu().numpy()
shared = int(scratch[off + 4:off + 8].view(torch.int32).cpu()[0])
print("fast path: candidates", int(cnt.sum()) + shared, "=", (cnt.sum() + shared) / n0, "per query; per-wave regions", int(cnt.sum()), "(max", int(cnt.max()), "of 2048), shared list", shared, "; fallback flag", ovf)

dl = int(scratch[off + 8:off + 12].view(torch.int32).cpu()[0])
print("dense blocks listed by the refine pass:", dl)
