"""BASELINE config 5: LoNuScenes-shaped distant pair (16-beam ~30 k pts vs 64-beam ~120 k pts, 40 m apart) with
APG 11-frame aggregation (key + 5 each side at 6 m spacing) + FCGF encode + NPR decoder (GenerativeMLP_98, ratio 4)
reconstruction loss, one GPU.  Prints stage times (ms) as JSON."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from apr_amd import MinkowskiEngine as ME
from apr_amd import ops, synth
from apr_amd.fcgf.lib import apg
from apr_amd.fcgf.model import load_model


def complement_frames(seed, n_beams, origin_x, k=5, spacing=6.0):
    """2k scans of the same scene along +x around the key pose, each with its pose into the key frame."""
    rng = np.random.default_rng(seed + 1000)
    scene = synth.make_scene(seed)
    frames, poses = [], []
    for j in list(range(-k, 0)) + list(range(1, k + 1)):
        dx = j * spacing
        frames.append(synth.raycast(scene, (origin_x + dx, 0.0, 0.0), 0.0, rng, n_beams, 1875))
        M = np.eye(4); M[0, 3] = dx
        poses.append(M)
    return frames, poses


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    enc = load_model("ResUNetFatBN")(1, 128, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3).to(dev).eval()
    gen = apg.GenerativeMLP_98(in_channel=128, out_points=4).to(dev).eval()
    xyz0, xyz1, T = synth.make_pair(0, n_beams1=16, dist=40.0)
    f0, p0 = complement_frames(0, 16, 0.0)
    up = lambda a: torch.from_numpy(a).to(dev)
    key0, cm0 = up(xyz0), [up(f) for f in f0]

    def sync():
        torch.cuda.synchronize(); return time.perf_counter()

    rows = []
    for rep in range(4):
        t0 = sync()
        nghb, sel = apg.aggregate_frames(key0, cm0, p0, 0.3)
        pcd_nghb = nghb[sel]
        t1 = sync()
        c = ops.voxelize(key0, 0.3, 0)
        m = ops.build_map(c, want_first=True); ops.finalize_maps([m])
        x = ME.SparseTensor(torch.ones((m.n, 1), device=dev), coordinates=m.coords)
        with torch.no_grad():
            out = enc(x)
        t2 = sync()
        loss = apg.npr_reconstruction_loss(gen, out.F, m.coords[:, 1:], pcd_nghb, 0.3, 4)
        t3 = sync()
        if rep:
            rows.append([t1 - t0, t2 - t1, t3 - t2])
    r = np.array(rows).mean(0) * 1e3
    print(json.dumps({"workload": "config 5 (frame 0 side): APG 10 complement frames + FatBN-128 encode + NPR loss",
                      "key_points": len(xyz0), "complement_points": int(sum(len(f) for f in f0)),
                      "apg_points_after_crop_and_voxel": int(len(pcd_nghb)), "key_voxels": int(m.n),
                      "ms": {"apg_aggregate": r[0], "voxelise+encode(FatBN,128)": r[1], "npr_mlp+chamfer+reg": r[2]},
                      "loss": float(loss)}, indent=1))


if __name__ == "__main__":
    main()
