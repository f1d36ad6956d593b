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
import torch

from ... import MinkowskiEngine as ME
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

    def _recon(self, encoded, clouds):
        """The per-cloud loop :424-449 for one frame's batched output."""
        loss = 0
        coords, feats = encoded.decomposed_coordinates_and_features
        for i in range(len(coords)):
            loss = loss + apg.npr_reconstruction_loss(self.generator_model, feats[i], coords[i], clouds[i], self.voxel_size,
                                                      self.point_generation_ratio, self.regularization_strength,
                                                      self.regularization_type, self.alpha) * self.loss_ratio
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
        enc = []
        for k in ("0", "1"):
            sinput = ME.SparseTensor(input_dict[f'sinput{k}_F'].to(dev), coordinates=input_dict[f'sinput{k}_C'].to(dev))
            enc.append(self.encoder_model(sinput))
        mark()
        pos_loss, neg_loss = self.crit.contrastive_hardest_negative_loss(
            enc[0].F, enc[1].F, input_dict['correspondences'], num_pos=self.num_pos, num_hn_samples=self.num_hn, draws=draws)
        loss = pos_loss + self.neg_weight * neg_loss
        mark()
        loss = loss + self._recon(enc[0], input_dict['pcd_nghb0']) + self._recon(enc[1], input_dict['pcd_nghb1'])
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
