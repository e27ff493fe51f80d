"""EgoNCE / NormSoftmaxLoss -- drop-in for the reference's model/loss.py:7-53.

Two ways in, same kernels underneath (egovlp_amd/csrc/egonce.hip):
  * the reference's own call shape  `loss(sim_matrix(text, video), sim_v, sim_n)`  (model/loss.py:34,
    trainer/trainer_egoclip.py:130-137): `forward` below, an autograd node over egv_egonce_from_sim;
  * the fused hot path  `loss.fused(text, video, noun, verb)`: ONE call computes the three similarity
    matrices, the mask, the loss and the gradients w.r.t. both embeddings (egv_egonce_fwd_bwd).
MaxMarginRankingLoss (:55-90) and AdaptiveMaxMarginRankingLoss (:92-133), the EPIC-MIR / Charades fine-tuning heads over the
same similarity matrix, run on egv_maxmargin_fwd_bwd.  CrossEntropy (:135-141), the loss of the OSCC / PNR
classification fine-tunes on the [B, classes] scores of FrozenInTime(video_only=True) (trainer/trainer_oscc.py:335-338), runs on
egv_cross_entropy_fwd_bwd.
"""
import torch
from torch import nn

from .. import loss_ops, ops


class _LossFromSimFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask_v, mask_n, temperature, use_noun, use_verb):
        loss, dx = loss_ops.egonce_from_sim(x, mask_v, mask_n, temperature, use_noun, use_verb, want_grad=True)
        ctx.save_for_backward(dx)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        return dx * g, None, None, None, None, None


class _FusedHeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, text, video, noun, verb, temperature, use_noun, use_verb):
        loss, _, dt, dv = ops.egonce_fwd_bwd(text.contiguous(), video.contiguous(),
                                             None if noun is None else noun.contiguous().float(),
                                             None if verb is None else verb.contiguous().float(),
                                             temperature, use_noun=use_noun, use_verb=use_verb)
        ctx.save_for_backward(dt, dv)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        dt, dv = ctx.saved_tensors
        return dt * g, dv * g, None, None, None, None, None


class NormSoftmaxLoss(nn.Module):
    def __init__(self, temperature=0.05):
        super().__init__()
        self.temperature = temperature

    def forward(self, x):
        "x: similarity matrix N x N in [-1, 1] (model/loss.py:13-25)"
        return _LossFromSimFn.apply(x, None, None, self.temperature, True, True)

    def fused(self, text_embeds, video_embeds):
        return _FusedHeadFn.apply(text_embeds, video_embeds, None, None, self.temperature, True, True)


class EgoNCE(nn.Module):
    def __init__(self, temperature=0.05, noun=True, verb=True):
        super().__init__()
        self.noun = noun
        self.verb = verb
        self.temperature = temperature

    def forward(self, x, mask_v, mask_n):
        # noun=False, verb=False falls into the reference's else-branch (mask_v), model/loss.py:40-41
        return _LossFromSimFn.apply(x, mask_v, mask_n, self.temperature, self.noun, self.verb or not self.noun)

    def fused(self, text_embeds, video_embeds, noun_vec, verb_vec):
        return _FusedHeadFn.apply(text_embeds, video_embeds, noun_vec, verb_vec, self.temperature, self.noun,
                                  self.verb or not self.noun)


class _MaxMarginFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, margin, fix_norm):
        loss, dx = loss_ops.maxmargin(x, weight, margin, fix_norm, want_grad=True)
        ctx.save_for_backward(dx)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        return dx * g, None, None, None


class MaxMarginRankingLoss(nn.Module):
    """model/loss.py:55-90 (same constructor; `weight` is accepted and ignored, as in the reference)."""

    def __init__(self, margin=0.2, fix_norm=True):
        super().__init__()
        self.fix_norm = fix_norm
        self.margin = margin

    def forward(self, x, weight=None):
        return _MaxMarginFn.apply(x, None, self.margin, self.fix_norm)


class AdaptiveMaxMarginRankingLoss(nn.Module):
    """model/loss.py:92-133: the margin of row i is weight[i] * margin."""

    def __init__(self, margin=0.4, fix_norm=True):
        super().__init__()
        self.fix_norm = fix_norm
        self.margin = margin

    def forward(self, x, weight=None):
        if weight is None:
            raise TypeError("AdaptiveMaxMarginRankingLoss needs the per-row weight (model/loss.py:109)")
        return _MaxMarginFn.apply(x, weight, self.margin, self.fix_norm)


class _CrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, output, target):
        loss, dx = loss_ops.cross_entropy(output, target, want_grad=True)
        ctx.save_for_backward(dx)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        return dx * g, None


class CrossEntropy(nn.Module):
    """model/loss.py:135-141: `nn.CrossEntropyLoss()(output, target)` (mean reduction, ignore_index -100), the loss of
    configs/ft/oscc.json and pnr.json."""

    def __init__(self):
        super().__init__()

    def forward(self, output, target):
        return _CrossEntropyFn.apply(output, target)
