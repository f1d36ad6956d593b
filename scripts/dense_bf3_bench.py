"""k_dense_gemm_bf3 on KPFCNN's Linear / KPConv-step-2 shapes (4 pairs stacked per forward: rows x 4)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apr_amd import ops
dev = torch.device("cuda:0")
S = int(os.environ.get("STACK", "4"))
shapes = [(27674, 64, 64), (27674, 960, 64), (27674, 128, 128), (27674, 256, 64), (10187, 960, 64), (10187, 128, 256),
          (10187, 1920, 128), (10187, 512, 128), (3673, 256, 512), (3673, 1920, 128), (3673, 3840, 256), (3673, 1024, 256),
          (1246, 512, 1024), (1246, 3840, 256), (1246, 7680, 512), (1246, 2048, 512)]
tot = 0.0
for m, cin, cout in shapes:
    m *= S
    x = torch.randn(m, cin, device=dev)
    W = torch.randn(1, cin, cout, device=dev) * 0.05
    w3 = ops.pack_weights_bf3(W)
    out = torch.empty(m, cout, device=dev)
    for _ in range(3): ops.dense_gemm_bf3(x, w3, cin, cout, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.dense_gemm_bf3(x, w3, cin, cout, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 20
    ref = x[:256].double() @ W[0].double()
    err = float((out[:256].double() - ref).norm() / ref.norm())
    tot += us
    print(f"M={m:6d} {cin:5d}->{cout:4d}: {us:8.1f} us  {2.0 * m * cin * cout / us / 1e6:6.1f} TFLOP/s  rel err {err:.1e}", flush=True)
print(f"sum {tot:.1f} us")
