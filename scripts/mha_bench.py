"""k_mha (scalar, interleaved channels) vs k_mha_mfma (head-major) on the cross-attention shapes of a KITTI pair."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apr_amd.predator import kp_ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for n, m in ((300, 300), (600, 640), (1000, 1000), (2400, 2400)):
    q, k, v = (torch.randn(r, 256, device=dev) for r in (n, m, m))
    for name, fn in (("scalar", kp_ops.mha), ("mfma", kp_ops.mha_headmajor)):
        if name == "scalar" and m > 1856:          # its score rows live in LDS
            continue
        for _ in range(3):
            fn(q, k, v, 4)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn(q, k, v, 4)
        e1.record(); torch.cuda.synchronize()
        print(f"n={n} m={m} {name}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
