import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apr_amd import ops, synth
from apr_amd.MinkowskiEngine.core import CoordinateManager
dev = torch.device("cuda:0")
xyz0, xyz1, _ = synth.make_pair(0)
maps = []
for b, xyz in enumerate((xyz0, xyz1)):
    c = ops.voxelize(torch.from_numpy(xyz).to(dev), 0.3, b); maps.append(ops.build_map(c))
ops.finalize_maps(maps)
cm = CoordinateManager(torch.cat([m.coords for m in maps])); cm.build_pyramid([2, 4, 8])
cin = cout = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ts = int(sys.argv[2]) if len(sys.argv) > 2 else 1
nbr = cm.kernel_map(ts, ts, 3)
x = torch.randn(cm.size(ts), cin, device=dev); wp = ops.pack_weights(torch.randn(27, cin, cout, device=dev) * 0.05)
out = torch.empty(cm.size(ts), cout, device=dev)
for _ in range(5): ops.spconv(x, nbr, 27, cin, cout, wp, out=out)
torch.cuda.synchronize()
