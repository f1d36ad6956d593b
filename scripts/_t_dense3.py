import torch, numpy as np, time
from apr_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
def bench(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M, cin, cout in [(1382, 7680, 512), (3770, 3840, 256), (9918, 1920, 128), (25900, 960, 64), (7880, 512, 512), (1382, 512, 2048),
                     (3770, 1024, 256), (9918, 128, 512), (25900, 128, 256), (25900, 64, 256), (594, 256, 256), (1382, 2048, 512), (100000, 256, 256), (77, 64, 64)]:
    x = torch.randn(M, cin, device=dev)
    w = torch.randn(cin, cout, device=dev) / cin ** 0.5
    sh = torch.randn(cout, device=dev); sc = torch.rand(cout, device=dev) + 0.5
    res = torch.randn(M, cout, device=dev)
    wp = ops.pack_weights(w[None].contiguous()); w3 = ops.pack_weights_bf3(w[None].contiguous())
    ref = torch.relu((x.double() @ w.double()) * sc.double() + sh.double() + res.double())
    y32 = ops.spconv(x, None, 1, cin, cout, wp, scale=sc, shift=sh, residual=res, relu=True, n_out=M)
    y3 = ops.dense_gemm_bf3(x, w3, cin, cout, scale=sc, shift=sh, residual=res, relu=True)
    e32 = float((y32.double() - ref).norm() / ref.norm()); e3 = float((y3.double() - ref).norm() / ref.norm())
    t32 = bench(lambda: ops.spconv(x, None, 1, cin, cout, wp, n_out=M)); t3 = bench(lambda: ops.dense_gemm_bf3(x, w3, cin, cout))
    fl = 2 * M * cin * cout
    print(f"M {M:6d} cin {cin:5d} cout {cout:5d}: fp32 {t32:7.1f} us ({fl/t32/1e6:6.1f} TF)  bf3 {t3:7.1f} us ({fl/t3/1e6:6.1f} TF)  relerr fp32 {e32:.2e} bf3 {e3:.2e}", flush=True)
