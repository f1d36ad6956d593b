"""Time single sparse-conv layers on the real kernel maps of one synthetic pair (A/B tuning aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from apr_amd import ops, synth
from apr_amd.MinkowskiEngine.core import CoordinateManager
dev = torch.device("cuda:0")
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
layers = [("b1 32>32 N1", 1, 1, 32, 32, False), ("b2tr 64>64 N1", 1, 1, 64, 64, False), ("c2tr 128>64 2>1", 2, 1, 128, 64, True),
          ("c2 32>64 1>2", 1, 2, 32, 64, False), ("b2 64>64 N2", 2, 2, 64, 64, False), ("c3 64>128 2>4", 2, 4, 64, 128, False),
          ("b3 128>128 N3", 4, 4, 128, 128, False), ("c4 128>256 4>8", 4, 8, 128, 256, False),
          ("b4 256>256 N4", 8, 8, 256, 256, False), ("c4tr 256>128 8>4", 8, 4, 256, 128, True),
          ("c3tr 256>64 4>2", 4, 2, 256, 64, True),
          # ResUNetFatBN's decoder blocks (128 channels on the two finest levels)
          ("f2tr 128>128 N1", 1, 1, 128, 128, False), ("f3tr 128>128 N2", 2, 2, 128, 128, False)]


def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / 20


ONLY = os.environ.get("ONLY")
for name, ti, to, cin, cout, tr in layers:
    if ONLY and name.split()[0] not in ONLY.split(","):
        continue
    nbr = cm.kernel_map(ti, to, 3, tr)
    P = int((nbr >= 0).sum())
    x = torch.randn(cm.size(ti), cin, device=dev)
    wraw = torch.randn(27, cin, cout, device=dev) * 0.05
    wp = ops.pack_weights(wraw)
    out = torch.empty(cm.size(to), cout, device=dev)
    def batched(plist=None, o=out, w3=None):      # 20 launches through ONE library call: GPU time, not host time
        b = ops.SpconvBatch()
        for _ in range(20):
            b.add(x, nbr, 27, cin, cout, wp, out=o, plist=plist, w_bf3=w3)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); b.launch(); e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1000 / 20
    ops.spconv(x, nbr, 27, cin, cout, wp, out=out); batched()
    us = batched()
    line = f"{name:18s} rows={cm.size(to):6d} P={P:7d} tile {us:7.1f} us {2.0*P*cin*cout/us/1e6:6.1f} TF {(4.0*P*(cin+cout)+8*P)/us/1e3:6.0f} GB/s"
    if ops.ws_supported(27, cin, cout):
        t_build = timeit(lambda: ops.build_pairlist(nbr))
        pl = cm.pair_list(ti, to, 3, tr).build()
        out2 = torch.empty_like(out)
        ops.spconv(x, nbr, 27, cin, cout, wp, out=out2, plist=pl); batched(pl, out2)
        us2 = batched(pl, out2)
        err = float((out2 - out).norm() / out.norm())
        line += f" | ws {us2:7.1f} us {2.0*P*cin*cout/us2/1e6:6.1f} TF {(4.0*P*(cin+cout)+8*P)/us2/1e3:6.0f} GB/s  (pairlist build {t_build:5.1f} us, rel diff {err:.1e})"
        w3 = ops.pack_weights_bf3(wraw)
        if w3 is not None:
            out3 = torch.empty_like(out)
            ops.spconv(x, nbr, 27, cin, cout, wp, out=out3, plist=pl, w_bf3=w3); batched(pl, out3, w3)
            us3 = batched(pl, out3, w3)
            err3 = float((out3 - out).norm() / out.norm())
            line += f" | ws-bf3 {us3:7.1f} us {2.0*P*cin*cout/us3/1e6:6.1f} TF  (rel diff {err3:.1e})"
        if w3 is not None and ops.ws3_supported(27, cin, cout):      # x-triple entry lists: one product row per entry
            t_b3 = timeit(lambda: ops.build_pairlist3(nbr))
            pl3 = ops.build_pairlist3(nbr)
            out5 = torch.empty_like(out)
            ops.spconv(x, nbr, 27, cin, cout, wp, out=out5, plist=pl3, w_bf3=w3); batched(pl3, out5, w3)
            us5 = batched(pl3, out5, w3)
            err5 = float((out5 - out).norm() / out.norm())
            ent = int(pl3.counts().sum())
            line += f" | ws3 {us5:7.1f} us (entries/pairs {ent / P:.2f}, build {t_b3:5.1f} us, rel diff {err5:.1e})"
        R = ops.os_tile_rows(cm.size(to), cin, cout) if w3 is not None else 0
        if R:
            t_ob = timeit(lambda: ops.build_os_pairs(nbr, cm.size(ti), R))
            osp = ops.build_os_pairs(nbr, cm.size(ti), R)
            out4 = torch.empty_like(out)
            def os_batched():
                b = ops.SpconvBatch()
                for _ in range(20):
                    b.add(x, nbr, 27, cin, cout, wp, out=out4, w_bf3=w3, os_pairs=osp)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); b.launch(); e1.record(); torch.cuda.synchronize()
                return e0.elapsed_time(e1) * 1000 / 20
            os_batched()
            us4 = os_batched()
            err4 = float((out4 - out).norm() / out.norm())
            line += f" | os R={R} {us4:7.1f} us {2.0*P*cin*cout/us4/1e6:6.1f} TF {(4.0*P*(cin+cout)+8*P)/us4/1e3:6.0f} GB/s (build {t_ob:5.1f} us, rel diff {err4:.1e})"
    print(line, flush=True)
