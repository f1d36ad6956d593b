import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from apr_amd import MinkowskiEngine as ME, synth
from apr_amd.fcgf.model import resunet as RU
from oracle import me_oracle as OME
from tests.helpers import model_pair, rel_l2
dev = torch.device("cuda:0")
_, hm = model_pair("ResUNetFatBN", 128, seed=2)
hm.train()
state0 = {k: v.clone() for k, v in hm.state_dict().items()}
xyz, _, _ = synth.make_pair(6, n_beams=32, n_azimuth=900)
c, _ = OME.sparse_quantize(xyz / np.float32(0.3), return_index=True)
C = torch.from_numpy(OME.batched_coordinates([c])).to(dev)
proj = None
res = {}
for fused in (True, False):
    RU.TRAIN_FUSED = fused
    hm.load_state_dict(state0); hm.zero_grad()
    y = hm(ME.SparseTensor(torch.ones(len(C), 1, device=dev), coordinates=C)).F
    if proj is None:
        proj = torch.from_numpy(np.random.default_rng(1).standard_normal(tuple(y.shape)).astype(np.float32)).to(dev)
    (y * proj).sum().backward()
    res[fused] = {n: p.grad.clone() for n, p in hm.named_parameters()}
print("rows", len(C))
for n in res[True]:
    print(f"   {rel_l2(res[True][n].cpu(), res[False][n].cpu()):.2e} {n} {tuple(res[True][n].shape)}")
