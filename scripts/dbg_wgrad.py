import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from apr_amd import ops
from tests.helpers import rel_l2
dev = torch.device("cuda:0")
rng = np.random.default_rng(11)
for (n, cin, cout) in ((6667, 128, 128), (6667, 64, 128), (6667, 64, 64), (20000, 64, 64), (14000, 128, 128)):
    K = 27
    nbr = rng.integers(0, n, size=(n, K)).astype(np.int32)
    nbr[rng.random((n, K)) > 0.25] = -1
    nbr[:, 13] = np.arange(n)
    x = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32)).to(dev)
    g = torch.from_numpy(rng.standard_normal((n, cout)).astype(np.float32)).to(dev)
    nb = torch.from_numpy(nbr).to(dev)
    ref = torch.zeros(K, cin, cout, dtype=torch.float64, device=dev)
    for k in range(K):
        j = torch.nonzero(nb[:, k] >= 0).squeeze(1)
        ref[k] = x[nb[j, k].long()].double().t() @ g[j].double()
    for same in (False, True):
        dw = ops.spconv_wgrad(x, g, nb, K, cin, cout, same_level=same)
        per_k = [rel_l2(dw[k].cpu(), ref[k].cpu()) for k in range(K)]
        print(n, cin, cout, "same" if same else "plain", "total %.2e" % rel_l2(dw.cpu(), ref.cpu()), "worst k", int(np.argmax(per_k)), "%.2e" % max(per_k),
              "bad ks", [k for k in range(K) if per_k[k] > 1e-5])
