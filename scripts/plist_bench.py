"""Time apr_pairlist_build alone on the real kernel maps of a synthetic pair."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apr_amd import ops, synth, _lib
from apr_amd._lib import ptr, stream, check
from apr_amd.MinkowskiEngine.core import CoordinateManager
dev = torch.device("cuda:0")
xyz0, xyz1, _ = synth.make_pair(0)
maps = []
for b, xyz in enumerate((xyz0, xyz1)):
    c = ops.voxelize(torch.from_numpy(xyz).to(dev), 0.3, b); maps.append(ops.build_map(c))
ops.finalize_maps(maps)
cm = CoordinateManager(torch.cat([m.coords for m in maps]))
cm.build_pyramid([2, 4, 8])
lib = _lib.load()
for name, ti, to, tr in [("1>1", 1, 1, False), ("2>1T", 2, 1, True), ("2>2", 2, 2, False), ("4>4", 4, 4, False), ("8>8", 8, 8, False), ("8>4T", 8, 4, True)]:
    nbr = cm.kernel_map(ti, to, 3, tr)
    n_out, K = nbr.shape
    nb = int(lib.apr_pairlist_bytes(n_out, K))
    blob = torch.empty(nb, dtype=torch.uint8, device=dev)
    cnt = torch.zeros(32, dtype=torch.int32, device=dev)
    st = stream()
    for _ in range(3): cnt.zero_(); check(lib.apr_pairlist_build(ptr(nbr), n_out, K, ptr(cnt), ptr(blob), nb, st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): cnt.zero_(); check(lib.apr_pairlist_build(ptr(nbr), n_out, K, ptr(cnt), ptr(blob), nb, st))
    e1.record(); torch.cuda.synchronize()
    print(f"map {name}: rows={n_out} build {e0.elapsed_time(e1)*20:.1f} us (counter fill + kernel)", flush=True)
