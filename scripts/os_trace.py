"""Per-wave timeline of one workgroup of the output-stationary conv (APR_OS_TRACE=1): prints, per wave, the cycles
between consecutive stamps.  Tags: 0 start, 1 after prologue, 4 item top, 5 operands ready (rows waited for and split,
accumulators read), 6 loads issued (slice pieces, pair words, rows of item + 2), 7 MFMAs + write-back done, 8 after the
offset's barrier, 9 loop end, 10 kernel end."""
import os, sys, ctypes
os.environ["APR_OS_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from apr_amd import ops, synth, _lib
from apr_amd.MinkowskiEngine.core import CoordinateManager
dev = torch.device("cuda:0")
NF = int(os.environ.get("FRAMES", "12"))
frames = []
for s in range((NF + 1) // 2):
    a, b, _ = synth.make_pair(s); frames += [a, b]
maps = []
for b, xyz in enumerate(frames[:NF]):
    c = ops.voxelize(torch.from_numpy(xyz).to(dev), 0.3, b); maps.append(ops.build_map(c))
ops.finalize_maps(maps)
cm = CoordinateManager(torch.cat([m.coords for m in maps]))
cm.build_pyramid([2, 4, 8])
ts = int(os.environ.get("TS", "1"))
nbr = cm.kernel_map(ts, ts, 3, False)
n = cm.size(ts)
x = torch.randn(n, 64, device=dev)
w = torch.randn(27, 64, 64, device=dev) * 0.05
w3 = ops.pack_weights_bf3(w)
R = ops.os_tile_rows(n, 64, 64)
osp = ops.build_os_pairs(nbr, n, R)
for _ in range(5):
    ops.spconv_os(x, osp, 64, 64, w3)
torch.cuda.synchronize()
buf = (ctypes.c_uint64 * (8 * 512))()
_lib.check(_lib.load().apr_spconv_os_trace(buf, 8 * 512))
a = np.frombuffer(buf, dtype=np.uint64).reshape(8, 512)
print("rows", n, "R", R)
t00 = min(int(a[w_][0] & 0x00FFFFFFFFFFFFFF) for w_ in range(8))
for w_ in (0, 3, 7):
    cnt = int(a[w_][511])
    tags = [(int(v) >> 56) for v in a[w_][:cnt]]
    t = [int(v) & 0x00FFFFFFFFFFFFFF for v in a[w_][:cnt]]
    print(f"wave {w_}: {cnt} stamps, total {t[-1] - t[0]} cycles (s_memtime ticks)")
    line = []
    for i in range(1, cnt):
        line.append(f"{tags[i]}:{t[i] - t[i - 1]}")
    print("  " + " ".join(line[:140]))
# summary over waves: time per tag transition
import collections
acc = collections.defaultdict(list)
for w_ in range(8):
    cnt = int(a[w_][511])
    tags = [(int(v) >> 56) for v in a[w_][:cnt]]
    t = [int(v) & 0x00FFFFFFFFFFFFFF for v in a[w_][:cnt]]
    for i in range(1, cnt):
        acc[(tags[i - 1], tags[i])].append(t[i] - t[i - 1])
for k in sorted(acc):
    v = np.array(acc[k])
    print(f"  {k[0]}->{k[1]}: n={len(v):4d} mean {v.mean():8.0f} median {np.median(v):8.0f} max {v.max():8d} sum {v.sum():9d}")
