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
from apr_amd.fcgf.lib.complement_trainer import GenerativePairTrainStep
from apr_amd.fcgf.model import load_model


def complement_frames(seed, origin_x, yaw, k=5, spacing=6.0, n_beams=64):
    """2k scans of the pair's scene along +x around the key pose (same heading), each with its pose into the key frame."""
    rng = np.random.default_rng(seed + 1000 + int(origin_x * 7))
    scene = synth.make_scene(seed)
    c, s = np.cos(yaw), np.sin(yaw)
    R = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    frames, poses = [], []
    for j in list(range(-k, 0)) + list(range(1, k + 1)):
        dx = j * spacing
        frames.append(synth.raycast(scene, (origin_x + dx, 0.0, 0.0), yaw, rng, n_beams, 1875))
        M = np.eye(4)
        M[:3, 3] = R.T @ np.array([dx, 0.0, 0.0])
        poses.append(M)
    return frames, poses


def make_batch(dev, seed=0, k=5):
    """input_dict of one pair (the collate's keys, complement_data_loader.py:1266-1278) built on the GPU: voxelised key
    frames, their APG clouds (2k complement frames moved into the key frame, cropped, one point per voxel) and the GT
    correspondences within 1.5 voxels (train_apr_kitti.sh: positive_pair_search_voxel_size_multiplier 1.5)."""
    xyz0, xyz1, T = synth.make_pair(seed)
    R1 = T[:3, :3].T
    d = float((-R1 @ T[:3, 3])[0])
    yaw = float(np.arctan2(R1[1, 0], R1[0, 0]))
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    out = {}
    pts = []
    for tag, xyz, ox, yw in (("0", xyz0, 0.0, 0.0), ("1", xyz1, d, yaw)):
        key = up(xyz)
        frames, poses = complement_frames(seed, ox, yw, k=k)
        nghb, sel = apg.aggregate_frames(key, [up(f) for f in frames], poses, 0.3)
        out[f"pcd_nghb{tag}"] = [nghb[sel.long()].contiguous()]
        m = ops.build_map(ops.voxelize(key, 0.3, 0), want_first=True)
        ops.finalize_maps([m])
        out[f"sinput{tag}_C"] = m.coords
        out[f"sinput{tag}_F"] = torch.ones((m.n, 1), device=dev)
        pts.append(key[m.first.long()].contiguous())
    out["correspondences"] = apg.get_matching_indices(pts[0], pts[1], torch.from_numpy(T).float().to(dev), 0.3 * 1.5).cpu()
    return out


def build(dev, n_out=128, lr=0.1):
    torch.manual_seed(0)
    enc = load_model("ResUNetFatBN")(1, n_out, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3).to(dev)
    gen = apg.GenerativeMLP_98(in_channel=n_out, out_points=4, bn_momentum=0.05).to(dev)
    opt = torch.optim.SGD([{'params': enc.parameters()}, {'params': gen.parameters()}], lr=lr, momentum=0.8,
                          weight_decay=1e-4)
    return GenerativePairTrainStep(enc, gen, opt, voxel_size=0.3, point_generation_ratio=4, regularization_strength=0.1,
                                   loss_ratio=2e-3)


def main():
    dev = torch.device("cuda:0")
    iters = int(os.environ.get("ITERS", "8"))
    batch = make_batch(dev)
    step = build(dev, lr=float(os.environ.get("LR", "0.1")))
    rows, walls, losses = [], [], []
    for it in range(iters + 2):
        np.random.seed(it)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = step(batch, timed=True)
        torch.cuda.synchronize()
        walls.append((time.perf_counter() - t0) * 1e3)
        rows.append(r["ms"])
        losses.append([float(r["loss"]), float(r["pos_loss"]), float(r["neg_loss"])])
        print(f"iter {it}: loss {losses[-1][0]:.4f} pos {losses[-1][1]:.4f} neg {losses[-1][2]:.4f}  wall {walls[-1]:.2f} ms  "
              + " ".join(f"{k} {v:.2f}" for k, v in r["ms"].items()), file=sys.stderr, flush=True)
    ms = {k: float(np.median([r[k] for r in rows[2:]])) for k in rows[0]}
    print(json.dumps({"workload": "APR training iteration: one 2 x 118 k-pt pair, ResUNetFatBN-128 (train), hardest-contrastive "
                                  "1024/256, GenerativeMLP_98 ratio 4 + Chamfer vs APG cloud, backward, SGD",
                      "voxels": [int(batch["sinput0_C"].shape[0]), int(batch["sinput1_C"].shape[0])],
                      "apg_cloud_points": [int(batch["pcd_nghb0"][0].shape[0]), int(batch["pcd_nghb1"][0].shape[0])],
                      "positive_pairs": int(len(batch["correspondences"])),
                      "ms_per_iteration": float(np.median(walls[2:])), "stages_ms": ms,
                      "backward_over_forward": ms["backward"] / ms["forward"],
                      "loss_first_last": [losses[0][0], losses[-1][0]]}))


if __name__ == "__main__":
    main()
