import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from apr_amd import MinkowskiEngine as ME, synth
from apr_amd.fcgf.model import resunet as RU
from oracle import me_oracle as OME
from tests.helpers import model_pair, rel_l2
dev = torch.device("cuda:0")
om, hm = model_pair("ResUNetBN2C", 32)
om.train(); hm.train()
state0 = {k: v.clone() for k, v in hm.state_dict().items()}
xyz, _, _ = synth.make_pair(5, n_beams=16, n_azimuth=700)
c, _ = OME.sparse_quantize(xyz / np.float32(0.3), return_index=True)
C = OME.batched_coordinates([c])
F = np.ones((len(C), 1), np.float32)
oy = om(OME.SparseTensor(torch.from_numpy(F), coordinates=C)).F
proj = torch.from_numpy(np.random.default_rng(0).standard_normal(tuple(oy.shape)).astype(np.float32))
(oy * proj).sum().backward()
og = dict(om.named_parameters())
res = {}
for fused in (True,):
    RU.TRAIN_FUSED = fused
    hm.load_state_dict(state0); hm.zero_grad()
    hy = hm(ME.SparseTensor(torch.from_numpy(F).to(dev), coordinates=torch.from_numpy(C).to(dev))).F
    print("fused", fused, "feat err", rel_l2(hy.detach().cpu(), oy.detach()))
    (hy * proj.to(dev)).sum().backward()
    errs = sorted(((rel_l2(p.grad.cpu(), og[n].grad), n) for n, p in hm.named_parameters()), reverse=True)
    res[fused] = {n: p.grad.clone() for n, p in hm.named_parameters()}
    for n, p in hm.named_parameters():
        print(f"   {rel_l2(p.grad.cpu(), og[n].grad):.2e} {n}")

