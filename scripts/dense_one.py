"""One compute-bound k_dense_gemm_bf3 shape, 30 launches (for PMC passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apr_amd import ops
dev = torch.device("cuda:0")
m, cin, cout = [int(v) for v in os.environ.get("SHAPE", "40748,1920,128").split(",")]
x = torch.randn(m, cin, device=dev)
W = torch.randn(1, cin, cout, device=dev) * 0.05
w3 = ops.pack_weights_bf3(W)
out = torch.empty(m, cout, device=dev)
for _ in range(30):
    ops.dense_gemm_bf3(x, w3, cin, cout, out=out)
torch.cuda.synchronize()
