"""k_dense_gemm_bf3<1>: time against the number of workgroups (64-row tiles x 64-column blocks) -- where are the steps?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apr_amd import ops
dev = torch.device("cuda:0")
cin, cout = 1920, 128
W = torch.randn(1, cin, cout, device=dev) * 0.05
w3 = ops.pack_weights_bf3(W)
for wgs in (128, 256, 257, 384, 512, 513, 640, 768, 769, 896, 1024, 1025, 1280, 1536, 1537, 2048, 3072):
    m = wgs // 2 * 64
    x = torch.randn(m, cin, device=dev)
    out = torch.empty(m, cout, device=dev)
    for _ in range(3): ops.dense_gemm_bf3(x, w3, cin, cout, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.dense_gemm_bf3(x, w3, cin, cout, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 20
    print(f"workgroups {wgs // 2 * 2:5d} (M = {m:6d}): {us:7.1f} us  {2.0 * m * cin * cout / us / 1e6:6.1f} TFLOP/s", flush=True)
