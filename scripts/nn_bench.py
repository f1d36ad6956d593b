import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apr_amd import ops
impl = sys.argv[1] if len(sys.argv) > 1 else 'brute'
dev = torch.device("cuda:0")
for n0, n1, c in [(14152, 13522, 32), (20835, 20800, 32), (5000, 5000, 32), (14000, 14000, 128)]:
    f0 = torch.nn.functional.normalize(torch.randn(n0, c, device=dev), dim=1)
    f1 = torch.nn.functional.normalize(torch.randn(n1, c, device=dev), dim=1)
    for _ in range(3): ops.feature_nn(f0, f1, impl=impl)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.feature_nn(f0, f1, impl=impl)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"{n0}x{n1}x{c}: {us:8.1f} us  {2.0*n0*n1*c/us/1e6:6.1f} TFLOP-equivalent (sub+fma lane-ops: {2.0*n0*n1*c/us/1e6/78.6*100:4.1f}% of 78.6 T lane-op/s)")
