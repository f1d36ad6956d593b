"""The NPR reconstruction loss of Predator_APR's trainer on the HIP kernels.

Mirrors /root/reference/Predator_APR/lib/trainer.py: `chamfer_distance` (:131-140) and, per frame, the statements
:175-183 / :199-207 of `Trainer.inference_one_batch` -- offsets from the generative model, mean-of-squares
regulariser, `generated + pcd.repeat(1, ratio)` reshaped to points, Chamfer distance to the aggregated neighbour
cloud, `(chamfer + regulariser * strength) * loss_ratio`.  Differentiable end to end (apr_amd/npr.py); the circle /
overlap / saliency losses of the descriptor branch (lib/loss.py) are host-side training glue outside SURVEY 8.
"""
import torch

from ... import npr


def chamfer_distance(array1, array2):
    """forward_cd / n1 + backward_cd / n2 (lib/trainer.py:131-140)."""
    return npr.chamfer_distance(array1, array2)


def npr_frame_loss(generative_model, feats, pcd, nghb, point_generation_ratio, regularization_strength, loss_ratio):
    """One frame's share of `generative_loss` (lib/trainer.py:175-183): returns (loss, chamfer_loss_raw, regularize_loss,
    mod_generated)."""
    generated = generative_model(feats)
    if isinstance(generated, tuple):          # a model built with a radius returns (x, radius): models/mlp.py:140-143
        generated = generated[0]
    regularize_loss = torch.mean(torch.sum((generated.reshape(-1, 3)) ** 2, axis=-1))
    mod_generated = (generated + pcd.to(generated.device).repeat(1, point_generation_ratio)).reshape(-1, 3)
    chamfer_loss_raw = chamfer_distance(mod_generated, nghb)
    loss = (chamfer_loss_raw + regularize_loss * regularization_strength) * loss_ratio
    return loss, chamfer_loss_raw, regularize_loss, mod_generated


def npr_loss(generative_model, src_feats, tgt_feats, src_pcd, tgt_pcd, src_nghb, tgt_nghb, point_generation_ratio,
             regularization_strength, loss_ratio):
    """Both frames (lib/trainer.py:166-211): -> dict(generative_loss, chamfer_loss, regularization_loss, invalid)."""
    l0, c0, r0, _ = npr_frame_loss(generative_model, src_feats, src_pcd, src_nghb, point_generation_ratio,
                                   regularization_strength, loss_ratio)
    l1, c1, r1, _ = npr_frame_loss(generative_model, tgt_feats, tgt_pcd, tgt_nghb, point_generation_ratio,
                                   regularization_strength, loss_ratio)
    return {"generative_loss": l0 + l1, "chamfer_loss": c0 + c1, "regularization_loss": r0 + r1,
            "invalid": bool(torch.isnan(c0) or torch.isnan(c1))}
