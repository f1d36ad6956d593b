import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from apr_amd import npr
from apr_amd.fcgf.lib.complement_trainer import synthetic_batch
dev = torch.device("cuda:0")
b = synthetic_batch(dev)
for k in ("0", "1"):
    vox = b[f"sinput{k}_C"][:, 1:].float() * 0.3
    gen = (vox.repeat(1, 4) + 0.3 * torch.rand(len(vox), 12, device=dev)).reshape(-1, 3)
    apg_c = b[f"pcd_nghb{k}"][0]
    for name, a, t in (("gen->apg", gen, apg_c), ("apg->gen", apg_c, gen)):
        _, d2, _ = npr.nn3(a, t, cell=0.0)
        d = d2.sqrt()
        print(k, name, len(a), len(t), " ".join(f">{r}: {float((d > r).float().mean()):.3f}" for r in (0.29, 0.59, 0.88, 1.79, 3.0, 4.7, 9.0, 14.0)))
