"""Run one weight-stationary conv layer on the real kernel maps of a synthetic pair (PMC / trace target)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apr_amd import ops, synth
from apr_amd.MinkowskiEngine.core import CoordinateManager
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "b2tr"
cfg = {"b2tr": (1, 1, 64, 64, False), "b4": (8, 8, 256, 256, False), "c4tr": (8, 4, 256, 128, True),
       "b3": (4, 4, 128, 128, False), "c3tr": (4, 2, 256, 64, True), "b2": (2, 2, 64, 64, False)}[which]
NF = int(os.environ.get("FRAMES", "2"))          # frames per encoder call (bench default: 12)
frames = []
for s in range((NF + 1) // 2):
    xyz0, xyz1, _ = synth.make_pair(s); frames += [xyz0, xyz1]
maps = []
for b, xyz in enumerate(frames[:NF]):
    c = ops.voxelize(torch.from_numpy(xyz).to(dev), 0.3, b); maps.append(ops.build_map(c))
ops.finalize_maps(maps)
cm = CoordinateManager(torch.cat([m.coords for m in maps]))
cm.build_pyramid([2, 4, 8])
ti, to, cin, cout, tr = cfg
nbr = cm.kernel_map(ti, to, 3, tr)
x = torch.randn(cm.size(ti), cin, device=dev)
wraw = torch.randn(27, cin, cout, device=dev) * 0.05
wp = ops.pack_weights(wraw)
w3 = ops.pack_weights_bf3(wraw) if os.environ.get("APR_WS_BF3", "1") != "0" else None
out = torch.empty(cm.size(to), cout, device=dev)
pl = cm.pair_list(ti, to, 3, tr).build()
for _ in range(10):
    ops.spconv(x, nbr, 27, cin, cout, wp, out=out, plist=pl, w_bf3=w3)
torch.cuda.synchronize()
print("done", which, int((nbr >= 0).sum()))
