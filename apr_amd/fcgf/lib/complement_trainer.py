"""One training iteration of APR's FCGF trainer on the HIP kernels.

Mirrors the loop body of `GenerativePairTrainer._train_epoch`, /root/reference/FCGF_APR/lib/complement_trainer.py:350-512
(the non-symmetric branch the shipped script trains, scripts/train_apr_kitti.sh): both frames through the encoder in
train mode, the hardest-contrastive loss (:398-409), per cloud of the batch the NPR branch -- generator on the cloud's
features x voxel size, regulariser (:432-440), generated points = offsets + voxel corner (:441-442), Chamfer distance to
the APG cloud, `(chamfer + reg * strength) * loss_ratio` (:446-448) -- one backward, one optimizer step.  The epoch loop,
data loader, logging and checkpoints around it are the reference's own host code and stay out of scope.

Every stage can be bracketed by HIP events on the launch stream (`timed=True`) for the bench's forward / loss / backward
/ optimizer split.
"""
import numpy as np
import torch

from ... import MinkowskiEngine as ME
from ... import npr, ops, synth
from . import apg
from .trainer import HardestContrastiveLoss


class GenerativePairTrainStep:
    def __init__(self, encoder_model, generator_model, optimizer, voxel_size=0.3, point_generation_ratio=6,
                 regularization_strength=0.1, regularization_type='L2', alpha=0.1, loss_ratio=1e-5, neg_weight=1,
                 num_pos_per_batch=1024, num_hn_samples_per_batch=256, batch_size=1, pos_thresh=0.1, neg_thresh=1.4):
        # defaults: FCGF_APR/config.py:30-36,77-79,86
        self.encoder_model, self.generator_model, self.optimizer = encoder_model, generator_model, optimizer
        self.voxel_size = voxel_size
        self.point_generation_ratio = point_generation_ratio
        self.regularization_strength, self.regularization_type, self.alpha = regularization_strength, regularization_type, alpha
        self.loss_ratio, self.neg_weight = loss_ratio, neg_weight
        self.num_pos = num_pos_per_batch * batch_size
        self.num_hn = num_hn_samples_per_batch * batch_size
        self.crit = HardestContrastiveLoss(pos_thresh, neg_thresh)
        import os
        self.stack_frames = os.environ.get("APR_TRAIN_STACK_FRAMES", "1") != "0"      # A/B switch: 0 = one encoder call per frame

    def _recon(self, encoded, clouds, rows, C=None):
        """The per-cloud loop :424-449 for one frame's batched output; `rows`: the clouds' row counts when the caller knows
        them (the collate's `len_batch`) and `C` the frame's coordinates (rows keep the input order), else both are read off
        the tensor (`decomposed_coordinates_and_features` synchronises, as the reference's does)."""
        loss = 0
        if rows is not None and sum(rows) == encoded.F.shape[0]:
            C, F = (encoded.C if C is None else C), encoded.F
            offs = [0]
            for r in rows:
                offs.append(offs[-1] + int(r))
            coords = [C[a:b, 1:] for a, b in zip(offs[:-1], offs[1:])]
            feats = [F[a:b] for a, b in zip(offs[:-1], offs[1:])]
        else:
            coords, feats = encoded.decomposed_coordinates_and_features
        for i in range(len(coords)):
            loss = loss + apg.npr_reconstruction_loss(self.generator_model, feats[i], coords[i], clouds[i], self.voxel_size,
                                                      self.point_generation_ratio, self.regularization_strength,
                                                      self.regularization_type, self.alpha) * self.loss_ratio
        return loss

    def _recon_stacked(self, encs, Cs, clouds, rows):
        """Both frames' per-cloud loops (:424-449, :455-481) in ONE pass: the generator runs once over all clouds (per-cloud
        BatchNorm statistics through the segments, in the reference's call order: frame 0's clouds, then frame 1's), the
        Chamfer term of every cloud in two batched searches (apr_nn3_batch).  Same numbers as `_recon` per frame."""
        offs = [0]
        for fr in rows:
            for r in fr:
                offs.append(offs[-1] + int(r))
        F = torch.cat([e.F for e in encs], 0)
        Cc = torch.cat([c[:, 1:] for c in Cs], 0)
        generated = self.generator_model(F, segments=offs) * self.voxel_size
        ratio = self.point_generation_ratio
        mod = apg.npr_points(generated, Cc, self.voxel_size, ratio)
        cl = [c for fr in clouds for c in fr]
        cl = [c.to(F.device, dtype=torch.float32) if torch.is_tensor(c) else torch.as_tensor(c, dtype=torch.float32, device=F.device)
              for c in cl]
        b_offs = [0]
        for c in cl:
            b_offs.append(b_offs[-1] + int(c.shape[0]))
        cham = npr.chamfer_distance_batch(mod, [o * ratio for o in offs], torch.cat(cl, 0), b_offs)
        loss = 0
        for i, (a, b) in enumerate(zip(offs[:-1], offs[1:])):
            reg = apg.npr_regulariser(generated[a:b], self.regularization_type, self.alpha)
            loss = loss + (cham[i] + reg * self.regularization_strength) * self.loss_ratio
        return loss

    def __call__(self, input_dict, draws=None, timed=False):
        dev = torch.device('cuda', torch.cuda.current_device())
        marks = []

        def mark():
            if timed:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                marks.append(e)

        self.encoder_model.train()
        self.generator_model.train()
        mark()
        self.optimizer.zero_grad()
        # the host half of the contrastive loss (three NumPy draws + uploads) needs only the row counts: do it before the
        # encoder is enqueued so that no host->device copy sits between the forward and the backward
        n0, n1 = int(input_dict['sinput0_C'].shape[0]), int(input_dict['sinput1_C'].shape[0])
        prepared = self.crit.prepare(n0, n1, input_dict['correspondences'], self.num_pos, self.num_hn, draws, dev)
        lens = input_dict.get('len_batch')
        rows = ([int(l[0]) for l in lens], [int(l[1]) for l in lens]) if lens else (None, None)
        sinputs = [ME.SparseTensor(input_dict[f'sinput{k}_F'].to(dev), coordinates=input_dict[f'sinput{k}_C'].to(dev))
                   for k in ("0", "1")]
        if self.stack_frames and hasattr(self.encoder_model, "forward_frames"):
            enc = self.encoder_model.forward_frames(sinputs)        # both calls of :386-394 in one walk, per-call BN statistics
        else:
            enc = [self.encoder_model(t) for t in sinputs]
        mark()
        pos_loss, neg_loss = self.crit.contrastive_hardest_negative_loss(
            enc[0].F, enc[1].F, None, num_pos=self.num_pos, num_hn_samples=self.num_hn, draws=prepared)
        loss = pos_loss + self.neg_weight * neg_loss
        mark()
        Cs = [input_dict['sinput0_C'].to(dev), input_dict['sinput1_C'].to(dev)]
        if self.stack_frames and rows[0] is not None and all(sum(r) == e.F.shape[0] for r, e in zip(rows, enc)) \
                and sum(len(r) for r in rows) <= 32 and all(v >= 2 for r in rows for v in r):
            loss = loss + self._recon_stacked(enc, Cs, [input_dict['pcd_nghb0'], input_dict['pcd_nghb1']], rows)
        else:
            loss = loss + self._recon(enc[0], input_dict['pcd_nghb0'], rows[0], Cs[0]) \
                + self._recon(enc[1], input_dict['pcd_nghb1'], rows[1], Cs[1])
        mark()
        loss.backward()
        mark()
        self.optimizer.step()
        mark()
        out = {"loss": loss.detach(), "pos_loss": pos_loss.detach(), "neg_loss": neg_loss.detach()}
        if timed:
            torch.cuda.synchronize()
            names = ("forward", "contrastive", "npr", "backward", "optimizer")
            out["ms"] = {n: marks[i].elapsed_time(marks[i + 1]) for i, n in enumerate(names)}
        return out


# ---------------------------------------------------------------------------------------------------------------------
# Measurement helpers (bench.py `workloads.apr_train_step`, scripts/apr_train_step.py): a synthetic input_dict with the
# collate's keys, the model / optimizer of scripts/train_apr_kitti.sh, and the timing loop.
# ---------------------------------------------------------------------------------------------------------------------
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


def synthetic_batch(dev, seed=0, k=5):
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
    out["len_batch"] = [[int(out["sinput0_C"].shape[0]), int(out["sinput1_C"].shape[0])]]
    return out


def build_step(dev, n_out=128, lr=0.1):
    torch.manual_seed(0)
    from ..model import load_model
    enc = load_model("ResUNetFatBN")(1, n_out, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3).to(dev)
    gen = apg.GenerativeMLP_98(in_channel=n_out, out_points=4, bn_momentum=0.05).to(dev)
    opt = torch.optim.SGD([{'params': enc.parameters()}, {'params': gen.parameters()}], lr=lr, momentum=0.8,
                          weight_decay=1e-4)
    return GenerativePairTrainStep(enc, gen, opt, voxel_size=0.3, point_generation_ratio=4, regularization_strength=0.1,
                                   loss_ratio=2e-3)



def measure(dev, iters=8, lr=0.1, log=None):
    """Stage split (HIP events, one synchronised iteration at a time) and the un-synchronised iteration time of the APR
    training step on one full-size pair -> dict for the bench line."""
    import time
    batch = synthetic_batch(dev)
    step = build_step(dev, lr=lr)
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
        if log:
            log(f"train iter {it}: loss {losses[-1][0]:.4f} pos {losses[-1][1]:.4f} neg {losses[-1][2]:.4f}  wall {walls[-1]:.2f} ms  "
                + " ".join(f"{k} {v:.2f}" for k, v in r["ms"].items()))
    n_free = max(iters, 10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    per = []
    for it in range(n_free):
        np.random.seed(100 + it)
        t1 = time.perf_counter()
        step(batch)
        per.append((time.perf_counter() - t1) * 1e3)
    torch.cuda.synchronize()
    free_ms = (time.perf_counter() - t0) * 1e3 / n_free
    if log:
        log("host ms per un-synchronised iteration: " + " ".join(f"{v:.1f}" for v in per))
    ms = {k: float(np.median([r[k] for r in rows[2:]])) for k in rows[0]}
    return {"workload": "APR training iteration (FCGF_APR/lib/complement_trainer.py:350-512): one 2 x 118 k-pt pair, "
                        "ResUNetFatBN-128 in train mode, both encodes, hardest-contrastive 1024 / 256, GenerativeMLP_98 ratio 4 "
                        "+ Chamfer against the APG cloud (10 complement frames) per frame, backward, SGD",
            "value": float(np.median(walls[2:])), "unit": "ms per iteration", "higher_is_better": False,
            "ms_per_iteration_unsynchronised_loop": free_ms,
            "stages_ms": ms, "backward_over_forward": ms["backward"] / ms["forward"],
            "voxels": [int(batch["sinput0_C"].shape[0]), int(batch["sinput1_C"].shape[0])],
            "apg_cloud_points": [int(batch["pcd_nghb0"][0].shape[0]), int(batch["pcd_nghb1"][0].shape[0])],
            "positive_pairs": int(len(batch["correspondences"])),
            "round4_module_path_ms": 29.9, "loss_first_last": [losses[0][0], losses[-1][0]]}
