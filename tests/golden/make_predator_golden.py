"""Generates tests/golden/predator_small.npz with the REFERENCE itself (build container only).

  * imports /root/reference/Predator_APR/models/architectures.py:KPFCNN (cwd = a temp dir holding a copy
    of the kernel-disposition .ply, because kernel_points.py:391 uses a relative path and would mkdir),
  * builds it with np.random.seed(0) / torch.manual_seed(0) and the configs/test/kitti.yaml model section,
  * builds the batch dict with the reference's own C++ core (oracle/_ref) following
    datasets/dataloader.py:72-198,
  * stores inputs (the two 0.3 m clouds, neighbour limits) and the reference's outputs.
It also asserts that apr_amd's KPFCNN and oracle/kpfcnn_oracle.py reproduce the reference here.
"""
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/Predator_APR"
sys.path.insert(0, ROOT)

from apr_amd import synth  # noqa: E402
from apr_amd.predator.configs.models import kitti_config  # noqa: E402
from oracle import kpfcnn_oracle as KO  # noqa: E402
from oracle import predator_points_oracle as PREF  # noqa: E402

LIMITS = [40, 36, 36, 38]


def reference_model():
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "kernels", "dispositions"))
    shutil.copy(os.path.join(REF, "kernels/dispositions/k_015_center_3D.ply"),
                os.path.join(tmp, "kernels", "dispositions"))
    cwd = os.getcwd()
    os.chdir(tmp)
    sys.path.insert(0, REF)
    try:
        from models.architectures import KPFCNN
        np.random.seed(0)
        torch.manual_seed(0)
        m = KPFCNN(kitti_config()).eval()
    finally:
        os.chdir(cwd)
        sys.path.remove(REF)
    return m


def small_pair():
    a, b, T = synth.make_pair(3, n_beams=16, n_azimuth=625)
    pts, lens = PREF.subsample_batch(np.concatenate([a, b]), np.array([len(a), len(b)], np.int32), sampleDl=0.3)
    return pts[:lens[0]], pts[lens[0]:], T


def main():
    ref = reference_model()
    cfg = kitti_config()
    src, tgt, T = small_pair()
    batch = KO.collate(src, tgt, cfg, LIMITS)
    with torch.no_grad():
        f, ov, sal = ref(batch)
    # restatement vs reference
    fo, ovo, salo = KO.kpfcnn_forward(ref.state_dict(), cfg, batch)
    print("oracle vs reference: feats", float((fo - f).abs().max()), "ov", float((ovo - ov).abs().max()),
          "sal", float((salo - sal).abs().max()))
    assert (fo - f).abs().max() < 1e-4 and (ovo - ov).abs().max() < 1e-4 and (salo - sal).abs().max() < 1e-4
    # apr_amd model built under the same seeds has the same weights
    from apr_amd.predator.models.architectures import KPFCNN as MyKPFCNN
    np.random.seed(0)
    torch.manual_seed(0)
    mine = MyKPFCNN(cfg)
    sd_r, sd_m = ref.state_dict(), mine.state_dict()
    assert list(sd_r.keys()) == list(sd_m.keys())
    assert all(torch.equal(sd_r[k], sd_m[k]) for k in sd_r), "same seeds must give the same parameters"
    wsum = float(sum(v.double().abs().sum() for v in sd_r.values()))
    np.savez_compressed(os.path.join(HERE, "predator_small.npz"), src=src, tgt=tgt, T=T, limits=np.array(LIMITS),
                        feats=f.numpy(), overlap=ov.numpy(), saliency=sal.numpy(), weight_abs_sum=wsum,
                        level_sizes=np.array([len(p) for p in batch['points']]),
                        nbr_widths=np.array([n.shape[1] for n in batch['neighbors']]))
    print("levels", [len(p) for p in batch['points']], "widths", [n.shape[1] for n in batch['neighbors']],
          "wrote predator_small.npz")


if __name__ == "__main__":
    main()
