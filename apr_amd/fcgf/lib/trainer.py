"""Hardest-contrastive mining of the FCGF/APR trainers on the GPU (forward value of the loss).

`contrastive_hardest_negative_loss(F0, F1, positive_pairs, num_pos=5192, num_hn_samples=2048)` keeps the
signature and sampling of HardestContrastiveLossTrainer (FCGF_APR/lib/trainer.py:400-452; identical copy at
lib/complement_trainer.py:296-348): the three `np.random.choice` draws stay on the host RNG in the same
order, everything after them (gathers, two nearest-negative searches, the positive-pair filter, both hinge
terms and their means) runs in HIP kernels with no host round trip until the two scalars come back.

Training: when the features take part in autograd the mining (gathers + the two
nearest-negative searches) still runs on the HIP kernels under no_grad, and the loss itself is rebuilt from the
mined indices with differentiable torch ops — the same expression as the reference, whose `min` passes gradients
only to the arg-min element, i.e. to exactly the rows gathered here.
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


class PreparedDraws:
    """The host half of one loss evaluation done ahead of the encoder: the three np.random.choice draws (the reference's
    order, trainer.py:413-420), the sampled pairs, the sorted positive keys -- uploaded from pinned memory without
    blocking the host, so that nothing synchronises between the encode and the backward."""

    __slots__ = ("sel0_d", "sel1_d", "pos0_d", "pos1_d", "keys_d", "hash_seed", "N0", "N1")


class _Staging:
    """Two pinned host buffers used in turn for the per-iteration index upload: ONE asynchronous copy per loss evaluation
    and no pinned allocation in the loop (hipHostMalloc / hipHostFree synchronise the device: with the host an iteration
    ahead of the GPU, a fresh `pin_memory()` per array cost a 60-80 ms stall every few iterations)."""

    def __init__(self):
        self.buf, self.ev, self.turn = [None, None], [None, None], 0

    def upload(self, arrays, dev):
        sizes = [int(a.size) for a in arrays]
        total = sum(sizes)
        t = self.turn
        self.turn ^= 1
        if self.buf[t] is None or self.buf[t].numel() < total:
            self.buf[t] = torch.empty(max(total, 1 << 16), dtype=torch.int64).pin_memory()
            self.ev[t] = None
        if self.ev[t] is not None:
            self.ev[t].synchronize()                     # the copy that last read this buffer (two evaluations ago) is done
        host = self.buf[t].numpy()
        pos = 0
        for a, n in zip(arrays, sizes):
            host[pos:pos + n] = a
            pos += n
        d = self.buf[t][:total].to(dev, non_blocking=True)
        self.ev[t] = torch.cuda.Event()
        self.ev[t].record()
        out, pos = [], 0
        for n in sizes:
            out.append(d[pos:pos + n])
            pos += n
        return out


class HardestContrastiveLoss:
    def __init__(self, pos_thresh=0.1, neg_thresh=1.4):   # config.py:34-35
        self.pos_thresh, self.neg_thresh = pos_thresh, neg_thresh
        self._staging = _Staging()

    def prepare(self, N0, N1, positive_pairs, num_pos=5192, num_hn_samples=2048, draws=None, device=None):
        """Everything of contrastive_hardest_negative_loss that does not need the features (trainer.py:408-423): the row
        counts N0 / N1 are known from the coordinates, so a trainer can call this BEFORE the encoder runs."""
        dev = device if device is not None else torch.device('cuda', torch.cuda.current_device())
        if not isinstance(positive_pairs, np.ndarray):
            positive_pairs = (positive_pairs.detach().cpu().numpy() if torch.is_tensor(positive_pairs)
                              else np.asarray(positive_pairs)).astype(np.int64)
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
        pd = PreparedDraws()
        pd.sel0_d, pd.sel1_d, pd.pos0_d, pd.pos1_d, pd.keys_d = self._staging.upload(
            [np.asarray(sel0, dtype=np.int64), np.asarray(sel1, dtype=np.int64), sample[:, 0], sample[:, 1],
             np.sort(_hash(positive_pairs, hash_seed))], dev)
        pd.hash_seed, pd.N0, pd.N1 = int(hash_seed), int(N0), int(N1)
        return pd

    def contrastive_hardest_negative_loss(self, F0, F1, positive_pairs, num_pos=5192, num_hn_samples=2048, thresh=None,
                                          draws=None):
        """-> (pos_loss, neg_loss).  Under no_grad: 0-d CPU tensors (forward value only).  With features that
        require grad: differentiable 0-d GPU tensors.  `draws`: (sel0, sel1, pos_sel) overrides the RNG (tests), or a
        PreparedDraws from `prepare` (then `positive_pairs` is not looked at again)."""
        track = torch.is_grad_enabled() and (F0.requires_grad or F1.requires_grad)
        pd = draws if isinstance(draws, PreparedDraws) else \
            self.prepare(len(F0), len(F1), positive_pairs, num_pos, num_hn_samples, draws, F0.device)
        if pd.N0 != len(F0) or pd.N1 != len(F1):
            raise ValueError("contrastive_hardest_negative_loss: the prepared draws belong to other clouds")
        with torch.no_grad():
            res = self._mine_and_reduce(F0, F1, pd, mine_only=track)
        if not track:
            return res
        pos0_d, pos1_d, d01ind, d10ind, keys_d, hash_seed = res
        posF0, posF1 = F0[pos0_d], F1[pos1_d]
        d01 = torch.sqrt((posF0 - F1[d01ind]).pow(2).sum(1) + 1e-7)      # lib/metrics.py:pdist 'L2'
        d10 = torch.sqrt((posF1 - F0[d10ind]).pow(2).sum(1) + 1e-7)
        mask0 = ~torch.isin(pos0_d + d01ind * hash_seed, keys_d, assume_unique=False)
        mask1 = ~torch.isin(d10ind + pos1_d * hash_seed, keys_d, assume_unique=False)
        pos_loss = torch.relu((posF0 - posF1).pow(2).sum(1) - self.pos_thresh)
        neg0 = torch.relu(self.neg_thresh - d01).pow(2)
        neg1 = torch.relu(self.neg_thresh - d10).pow(2)
        # masked means without boolean indexing (which would synchronise to learn its output size): NaN for an empty
        # mask, as torch.mean() of an empty tensor
        m0, m1 = mask0.to(neg0.dtype), mask1.to(neg1.dtype)
        return pos_loss.mean(), ((neg0 * m0).sum() / m0.sum() + (neg1 * m1).sum() / m1.sum()) / 2

    def _mine_and_reduce(self, F0, F1, pd, mine_only):
        dev = F0.device
        sel0_d, sel1_d, pos0_d, pos1_d, keys_d, hash_seed = pd.sel0_d, pd.sel1_d, pd.pos0_d, pd.pos1_d, pd.keys_d, pd.hash_seed
        F0, F1 = F0.detach().contiguous(), F1.detach().contiguous()
        gather = lambda F, idx: kp_ops.gather_pool(F, idx.view(-1, 1), "closest")
        posF0, posF1 = gather(F0, pos0_d), gather(F1, pos1_d)
        subF0, subF1 = gather(F0, sel0_d), gather(F1, sel1_d)
        lib = _lib.load()
        p, c = posF0.shape
        if mine_only:       # indices into the sub-samples -> row indices of the full clouds
            nn01 = ops.feature_nn(posF0, subF1)         # exact arg-min (bf16-MFMA filter + fp32 refine for c in 32 / 64 / 128)
            nn10 = ops.feature_nn(posF1, subF0)
            return (pos0_d, pos1_d, sel1_d[nn01], sel0_d[nn10], keys_d, int(hash_seed))
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
