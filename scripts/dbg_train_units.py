import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from apr_amd import MinkowskiEngine as ME, synth, ops
from apr_amd.fcgf.model import resunet as RU
from oracle import me_oracle as OME
from tests.helpers import model_pair, rel_l2
dev = torch.device("cuda:0")
om, hm = model_pair("ResUNetBN2C", 32)
hm.train()
state0 = {k: v.clone() for k, v in hm.state_dict().items()}
xyz, _, _ = synth.make_pair(5, n_beams=16, n_azimuth=700)
c, _ = OME.sparse_quantize(xyz / np.float32(0.3), return_index=True)
C = OME.batched_coordinates([c])
F = np.ones((len(C), 1), np.float32)
proj = None
rec = {}
orig_fwd, orig_bwd = ops.bn_train_fwd, ops.bn_train_bwd
for tag, env in (("ws", None), ("tile", "none")):
    if env is None: os.environ.pop("APR_WS_STAGES", None)
    else: os.environ["APR_WS_STAGES"] = env
    log = rec[tag] = []
    def f(z, bn, residual=None, relu=False, _log=log):
        out = orig_fwd(z, bn, residual=residual, relu=relu)
        _log.append(("fwd", z.clone(), out[0].clone(), out[1].clone(), out[2].clone()))
        return out
    def b(z, y, dy, mean, rstd, gamma, relu, want_dres, _log=log):
        out = orig_bwd(z, y, dy, mean, rstd, gamma, relu, want_dres)
        _log.append(("bwd", z.clone(), y.clone(), dy.clone(), out[0].clone(), out[2].clone(), out[3].clone()))
        return out
    ops.bn_train_fwd, ops.bn_train_bwd = f, b
    hm.load_state_dict(state0); hm.zero_grad()
    hy = hm(ME.SparseTensor(torch.from_numpy(F).to(dev), coordinates=torch.from_numpy(C).to(dev))).F
    if proj is None:
        proj = torch.from_numpy(np.random.default_rng(0).standard_normal(tuple(hy.shape)).astype(np.float32)).to(dev)
    (hy * proj).sum().backward()
a, b = rec["ws"], rec["tile"]
print(len(a), len(b))
for i, (ra, rb) in enumerate(zip(a, b)):
    errs = [rel_l2(x.cpu(), y.cpu()) for x, y in zip(ra[1:], rb[1:])]
    print(i, ra[0], tuple(ra[1].shape), " ".join(f"{e:.1e}" for e in errs))
print("mask flips per backward unit (y_ws > 0) != (y_tile > 0), and dbeta recomputed with the OTHER side's mask:")
for i, (ra, rb) in enumerate(zip(a, b)):
    if ra[0] != "bwd":
        continue
    ya, yb, dya, dyb = ra[2], rb[2], ra[3], rb[3]
    flips = int(((ya > 0) != (yb > 0)).sum())
    db_a = (dya * (ya > 0)).sum(0); db_b = (dyb * (yb > 0)).sum(0); db_a_mask_b = (dya * (yb > 0)).sum(0)
    print(i, "flips", flips, "dbeta ws vs tile %.1e" % rel_l2(db_a.cpu(), db_b.cpu()), "with tile's mask %.1e" % rel_l2(db_a_mask_b.cpu(), db_b.cpu()))
