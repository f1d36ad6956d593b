"""Dense (identity-map) layers of KPFCNN / the K = 1 convolutions: k_dense_gemm vs the pair-compacted tile kernel.
Run twice: APR_DENSE_GEMM=1 (default) and APR_DENSE_GEMM=0."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apr_amd import ops
dev = torch.device("cuda:0")
# (rows, cin, cout): KPFCNN on a 0.3 m KITTI pair (27.7 k / 10.2 k / 3.7 k / 1.2 k points per level), App. C widths
shapes = [(27674, 64, 64), (27674, 960, 64), (27674, 128, 128), (10187, 960, 64), (10187, 128, 256), (10187, 1920, 128),
          (3673, 256, 512), (3673, 1920, 128), (3673, 3840, 256), (1246, 512, 1024), (1246, 3840, 256), (1246, 7680, 512),
          (1246, 1024, 256), (189191, 64, 64)]
for m, cin, cout in shapes:
    x = torch.randn(m, cin, device=dev)
    wp = ops.pack_weights(torch.randn(cin, cout, device=dev) * 0.05)
    out = torch.empty(m, cout, device=dev)
    for _ in range(3): ops.spconv(x, None, 1, cin, cout, wp, out=out, n_out=m)
    b = ops.SpconvBatch()
    for _ in range(20): b.add(x, None, 1, cin, cout, wp, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); b.launch(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 20
    print(f"M={m:6d} {cin:5d}->{cout:4d}: {us:8.1f} us  {2.0 * m * cin * cout / us / 1e6:6.1f} TFLOP/s ({2.0 * m * cin * cout / us / 1e6 / 157.3 * 100:4.1f} % of fp32 MFMA peak)", flush=True)
