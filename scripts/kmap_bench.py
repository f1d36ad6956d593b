"""Time every kernel-map build of one 6-pair step (which maps dominate k_kernel_map)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apr_amd import ops, synth
from apr_amd.fcgf.pipeline import PairRegistration
dev = torch.device("cuda:0")
pipe = PairRegistration(torch.nn.Identity(), 0.3)
clouds = []
for s in range(6):
    a, b, _ = synth.make_pair(s)
    clouds += [torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)]
cm = pipe.voxelize_batch(clouds)[0]
print("rows per level:", [cm.size(ts) for ts in (1, 2, 4, 8)])
specs = [(1, 1, 5, False), (1, 1, 3, False), (1, 2, 3, False), (2, 2, 3, False), (2, 4, 3, False), (4, 4, 3, False),
         (4, 8, 3, False), (8, 8, 3, False), (8, 4, 3, True), (4, 2, 3, True), (2, 1, 3, True)]
tot = 0.0
for ti, to, k, tr in specs:
    out_map, in_map = cm.get_map(to), cm.get_map(ti)
    scale = -to if tr else ti
    for _ in range(2): ops.kernel_map(out_map, in_map, k, scale)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): nbr = ops.kernel_map(out_map, in_map, k, scale)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    tot += us
    print(f"map ({ti}->{to}, k={k}, T={tr}): {nbr.shape[0]:7d} x {nbr.shape[1]:3d}  {us:7.1f} us  valid {float((nbr >= 0).float().mean()):.3f}")
print("total", tot, "us per step =", tot / 6, "us per pair")
