"""Host (cProfile) + device (caller runs under rocprofv3) view of one Predator pair."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from apr_amd import synth
from apr_amd.predator import point_ops
from apr_amd.predator.configs.models import kitti_config
from apr_amd.predator.datasets.dataloader import collate_fn_descriptor
from apr_amd.predator.lib import benchmark_utils as BU
from apr_amd.predator.models.architectures import KPFCNN
LIMITS = [58, 59, 58, 57]
dev = torch.device("cuda:0")
np.random.seed(0); torch.manual_seed(0)
cfg = kitti_config()
model = KPFCNN(cfg).to(dev).eval()
a, b, T = synth.make_pair(0)
ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)

@torch.no_grad()
def one(rep):
    pts, lens = point_ops.grid_subsample(torch.cat([ta, tb]), np.array([len(a), len(b)], np.int32), 0.3)
    src, tgt = pts[:lens[0]], pts[lens[0]:]
    ones = lambda p: torch.ones((len(p), 1), device=dev)
    batch = collate_fn_descriptor([(src, tgt, ones(src), ones(tgt))], cfg, LIMITS)
    feats, ov, sal = model(batch)
    n0 = int(lens[0])
    np.random.seed(rep)
    s_p, s_f, _ = BU.sample_by_score(src, feats[:n0], ov[:n0] * sal[:n0], 5000)
    t_p, t_f, _ = BU.sample_by_score(tgt, feats[n0:], ov[n0:] * sal[n0:], 5000)
    return BU.ransac_pose_estimation(s_p, t_p, s_f, t_f, distance_threshold=0.3, ransac_n=4, seed=rep, return_info=True)

for r in range(3): one(r)
torch.cuda.synchronize(); t0 = time.perf_counter()
for r in range(10): one(r)
torch.cuda.synchronize(); print("ms/pair", (time.perf_counter() - t0) * 100)
pr = cProfile.Profile(); pr.enable()
for r in range(10): one(r)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
