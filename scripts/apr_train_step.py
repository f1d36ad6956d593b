"""The APR training iteration at full size (one 2 x 118 k-pt pair + its two APG clouds): FatBN-128 in train mode,
hardest-contrastive (1024 / 256), GenerativeMLP_98 + Chamfer vs the APG cloud, backward, SGD (lr 0.1, momentum 0.8,
weight decay 1e-4: scripts/train_apr_kitti.sh).  Prints the stage split (HIP events) and wall time per iteration as JSON.
`make_batch` / `build` are what bench.py's `workloads.apr_train_step` uses."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from apr_amd import ops, synth
from apr_amd.fcgf.lib import apg


from apr_amd.fcgf.lib.complement_trainer import synthetic_batch as make_batch, build_step as build, measure


def main():
    dev = torch.device("cuda:0")
    iters = int(os.environ.get("ITERS", "8"))
    _reserve = torch.empty(6 << 30, dtype=torch.uint8, device=dev)      # the caching allocator keeps it: no hipMalloc in the loop
    del _reserve
    print(json.dumps(measure(dev, iters=iters, lr=float(os.environ.get("LR", "0.1")), log=lambda m: print(m, file=sys.stderr, flush=True))))


if __name__ == "__main__":
    main()
