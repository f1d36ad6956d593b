"""A few launches of apr_dense_gemm_bf3 on KPFCNN shapes, for rocprofv3 --pmc."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apr_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for M, cin, cout in [(5528, 7680, 512), (31520, 512, 512), (100000, 256, 256)]:
    x = torch.randn(M, cin, device=dev)
    w3 = ops.pack_weights_bf3((torch.randn(cin, cout, device=dev) / cin ** 0.5)[None].contiguous())
    for _ in range(3):
        ops.dense_gemm_bf3(x, w3, cin, cout)
torch.cuda.synchronize()
