"""KPConv step 1 (kernel-point correlation, `k_kpconv_weighted_mfma`) per level of the KPFCNN pyramid on a synthetic
KITTI pair: time, algorithmic bytes 4*N*H*(3+Cin) + 8*N*H + 4*N*15*Cin and FLOPs
2*N*H*15*(3+Cin) against the HBM / fp32-MFMA roofs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from apr_amd import synth
from apr_amd.predator import kp_ops, point_ops
from apr_amd.predator.configs.models import kitti_config
from apr_amd.predator.datasets.dataloader import collate_fn_descriptor
from apr_amd.predator.kernels.kernel_points import load_kernels
from bench_predator import LIMITS
dev = torch.device("cuda:0")
cfg = kitti_config()
a, b, _ = synth.make_pair(0)
pts, lens = point_ops.grid_subsample(torch.cat([torch.from_numpy(a), torch.from_numpy(b)]).to(dev),
                                     np.array([len(a), len(b)], np.int32), 0.3)
src, tgt = pts[:lens[0]], pts[lens[0]:]
ones = lambda p: torch.ones((len(p), 1), device=dev)
batch = collate_fn_descriptor([(src, tgt, ones(src), ones(tgt))], cfg, LIMITS)
kp = torch.from_numpy(load_kernels(1.0, 15, dimension=3, fixed='center').astype(np.float32)).to(dev)
# KPConv input widths per level (1 -> 64 first, then 64 / 128 / 256 / 512 inside the bottlenecks)
for level, cins in enumerate([(1, 64), (128,), (256,), (512,)]):
    P, nbr = batch['points'][level], batch['neighbors'][level]
    N, H = nbr.shape
    extent = 0.6 * 2 ** level
    for cin in cins:
        x = torch.randn(N, cin, device=dev)
        for _ in range(3): kp_ops.kpconv_weighted(P, P, nbr, x, kp * extent, extent)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): kp_ops.kpconv_weighted(P, P, nbr, x, kp * extent, extent)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        byts = 4.0 * N * H * (3 + cin) + 8.0 * N * H + 4.0 * N * 15 * cin
        flops = 2.0 * N * H * 15 * (3 + cin)
        print(f"level {level}: N={N:6d} H={H:3d} Cin={cin:4d}: {us:7.1f} us  {byts / us / 1e3:7.0f} GB/s algorithmic "
              f"({byts / us / 1e3 / 80:5.1f} % of 8 TB/s)  {flops / us / 1e6:6.1f} TFLOP/s ({flops / us / 1e6 / 1.573:4.1f} % of fp32 MFMA)",
              flush=True)
