"""apr_occ_conv on the stride-1 voxels of a 12-frame batch (conv1 of the headline step)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apr_amd import ops, synth
dev = torch.device("cuda:0")
frames = []
for s in range(6):
    a, b, _ = synth.make_pair(s); frames += [a, b]
maps = []
for i, xyz in enumerate(frames):
    maps.append(ops.build_map(ops.voxelize(torch.from_numpy(xyz).to(dev), 0.3, i)))
ops.finalize_maps(maps)
coords = torch.cat([m.coords for m in maps]).contiguous()
bbox = ops.coords_bbox(coords).tolist()
W = torch.randn(125, 32, device=dev)
out = torch.empty(coords.shape[0], 32, device=dev)
for _ in range(3): ops.occ_conv(coords, coords.shape[0], bbox, 5, W, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.occ_conv(coords, coords.shape[0], bbox, 5, W, out=out)
e1.record(); torch.cuda.synchronize()
print(f"occ_conv {coords.shape[0]} voxels: {e0.elapsed_time(e1) * 1000 / 20:.1f} us (memset + set + conv)")
