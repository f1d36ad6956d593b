"""Hardest-contrastive mining of the FCGF/APR trainers on the GPU (forward value of the loss).

`contrastive_hardest_negative_loss(F0, F1, positive_pairs, num_pos=5192, num_hn_samples=2048)` keeps the
signature and sampling of HardestContrastiveLossTrainer (FCGF_APR/lib/trainer.py:400-452; identical copy at
lib/complement_trainer.py:296-348): the three `np.random.choice` draws stay on the host RNG in the same
order, everything after them (gathers, two nearest-negative searches, the positive-pair filter, both hinge
terms and their means) runs in HIP kernels with no host round trip until the two scalars come back.
Backward is SURVEY 8(f) next-3.
"""
import numpy as np
import torch

from ... import _lib, ops
from ..._lib import check, ptr, stream
from ...predator import kp_ops


def _hash(arr, M):
    """Pair key i + j*M (FCGF_APR/util/misc.py:6-18)."""
    if isinstance(arr, np.ndarray):
        N, D = arr.shape
    else:
        N, D = len(arr[0]), len(arr)
    h = np.zeros(N, dtype=np.int64)
    for d in range(D):
        h += (arr[:, d] if isinstance(arr, np.ndarray) else arr[d]) * M ** d
    return h


class HardestContrastiveLoss:
    def __init__(self, pos_thresh=0.1, neg_thresh=1.4):   # config.py:34-35
        self.pos_thresh, self.neg_thresh = pos_thresh, neg_thresh

    @torch.no_grad()
    def contrastive_hardest_negative_loss(self, F0, F1, positive_pairs, num_pos=5192, num_hn_samples=2048, thresh=None,
                                          draws=None):
        """-> (pos_loss, neg_loss) 0-d CPU tensors.  `draws=(sel0, sel1, pos_sel)` overrides the RNG (tests)."""
        N0, N1 = len(F0), len(F1)
        if not isinstance(positive_pairs, np.ndarray):
            positive_pairs = np.array(positive_pairs.cpu() if torch.is_tensor(positive_pairs) else positive_pairs,
                                      dtype=np.int64)
        positive_pairs = positive_pairs.astype(np.int64)
        N_pos_pairs = len(positive_pairs)
        hash_seed = max(N0, N1)
        if draws is None:
            sel0 = np.random.choice(N0, min(N0, num_hn_samples), replace=False)
            sel1 = np.random.choice(N1, min(N1, num_hn_samples), replace=False)
            pos_sel = np.random.choice(N_pos_pairs, num_pos, replace=False) if N_pos_pairs > num_pos else None
        else:
            sel0, sel1, pos_sel = draws
        sample = positive_pairs if pos_sel is None else positive_pairs[pos_sel]
        dev = F0.device
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64)).to(dev)
        sel0_d, sel1_d = t(sel0), t(sel1)
        pos0_d, pos1_d = t(sample[:, 0]), t(sample[:, 1])
        keys_d = t(np.sort(_hash(positive_pairs, hash_seed)))
        F0, F1 = F0.detach().contiguous(), F1.detach().contiguous()
        gather = lambda F, idx: kp_ops.gather_pool(F, idx.view(-1, 1), "closest")
        posF0, posF1 = gather(F0, pos0_d), gather(F1, pos1_d)
        subF0, subF1 = gather(F0, sel0_d), gather(F1, sel1_d)
        lib = _lib.load()
        p, c = posF0.shape
        nn01 = torch.empty(p, dtype=torch.int64, device=dev)
        nn10 = torch.empty(p, dtype=torch.int64, device=dev)
        check(lib.apr_feature_nn(ptr(posF0), p, ptr(subF1), subF1.shape[0], c, ptr(nn01), stream()))
        check(lib.apr_feature_nn(ptr(posF1), p, ptr(subF0), subF0.shape[0], c, ptr(nn10), stream()))
        out = torch.empty(6, dtype=torch.float64, device=dev)
        check(lib.apr_contrastive_reduce(ptr(posF0), ptr(posF1), p, c, ptr(nn01), ptr(nn10), ptr(sel0_d), ptr(sel1_d),
                                         ptr(pos0_d), ptr(pos1_d), ptr(keys_d), keys_d.shape[0], int(hash_seed),
                                         float(self.pos_thresh), float(self.neg_thresh), ptr(out), stream()))
        o = out.cpu()
        pos_loss = o[0] / o[1]
        neg_loss = (o[2] / o[3] + o[4] / o[5]) / 2     # NaN when every mined negative was a positive, as torch.mean()
        return pos_loss.float(), neg_loss.float()
