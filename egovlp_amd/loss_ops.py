"""Tensor-level wrappers of the contrastive-head kernels (include/egovlp_hip.h: egv_sim_matrix_*,
egv_egonce_from_sim, egv_egonce_fwd_bwd)."""
import torch

from . import _lib, ops
from ._lib import check
from .ops import _p


def sim_fwd(a, b, eps):
    ops._need_cuda(a, b)
    a = a.contiguous().float()
    b = b.contiguous().float()
    n, D = a.shape
    m = b.shape[0]
    dev = a.device
    an = torch.empty_like(a)
    bn = torch.empty_like(b)
    norms = torch.empty(n + m, dtype=torch.float32, device=dev)
    out = torch.empty((n, m), dtype=torch.float32, device=dev)
    check(_lib.lib().egv_sim_matrix_fwd(_p(a), _p(b), n, m, D, float(eps), _p(an), _p(bn), _p(norms), _p(out),
                                        ops._stream()), "egv_sim_matrix_fwd")
    return out, (an, bn, norms, n, m, D, float(eps))


def sim_bwd(data, g):
    an, bn, norms, n, m, D, eps = data
    g = g.contiguous()
    da = torch.empty_like(an)
    db = torch.empty_like(bn)
    check(_lib.lib().egv_sim_matrix_bwd(_p(g), _p(an), _p(bn), _p(norms), n, m, D, eps, _p(da), _p(db), ops._stream()),
          "egv_sim_matrix_bwd")
    return da, db


def egonce_from_sim(x, sim_v, sim_n, temperature, use_noun, use_verb, want_grad=True):
    ops._need_cuda(x)
    x = x.contiguous()
    n = x.shape[0]
    if x.shape[1] != n:
        raise ValueError("EgoNCE / NormSoftmaxLoss need a square similarity matrix")
    dev = x.device
    work = torch.empty(n * n + 6 * n, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    dx = torch.empty_like(x) if want_grad else None
    sv = sim_v.contiguous() if sim_v is not None else None
    sn = sim_n.contiguous() if sim_n is not None else None
    check(_lib.lib().egv_egonce_from_sim(_p(x), _p(sv), _p(sn), n, float(temperature), int(use_noun), int(use_verb),
                                         _p(loss), _p(dx), _p(work), ops._stream()), "egv_egonce_from_sim")
    return loss, dx


def maxmargin(x, weight, margin, fix_norm, want_grad=True):
    """MaxMarginRankingLoss (weight None) / AdaptiveMaxMarginRankingLoss on a square similarity matrix -> (loss[1], dx)."""
    ops._need_cuda(x, weight)
    x = x.contiguous().float()
    n = x.shape[0]
    if x.dim() != 2 or x.shape[1] != n:
        raise ValueError("the ranking losses need a square similarity matrix")
    w = None if weight is None else weight.contiguous().float()
    if w is not None and w.numel() != n:
        raise ValueError("weight must have one entry per row")
    loss = torch.empty(1, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x) if want_grad else None
    check(_lib.lib().egv_maxmargin_fwd_bwd(_p(x), _p(w), n, float(margin), int(bool(fix_norm)), _p(loss), _p(dx), ops._stream()),
          "egv_maxmargin_fwd_bwd")
    return loss, dx


def dual_softmax(x, temp=500.0):
    """softmax(softmax(x / temp, dim=1) * x, dim=0) of a [texts, videos] similarity matrix (run/test_epic.py:137-143)."""
    ops._need_cuda(x)
    x = x.contiguous().float()
    if x.dim() != 2:
        raise ValueError("dual_softmax needs a [texts, videos] matrix")
    n, m = x.shape
    work = torch.empty_like(x)
    out = torch.empty_like(x)
    check(_lib.lib().egv_dual_softmax(_p(x), n, m, float(temp), _p(work), _p(out), ops._stream()), "egv_dual_softmax")
    return out


def cross_entropy(logits, target, ignore_index=-100, want_grad=True):
    """nn.CrossEntropyLoss (mean over targets != ignore_index) on [rows, classes] fp32 scores and int64 labels -> (loss[1], dlogits)."""
    ops._need_cuda(logits, target)
    if logits.dim() != 2 or target.dim() != 1 or target.shape[0] != logits.shape[0]:
        raise ValueError("cross_entropy: logits [rows, classes], target [rows]")
    x = logits.float()
    if x.stride(1) != 1:
        x = x.contiguous()
    t = target.to(torch.int64).contiguous()
    loss = torch.empty(1, dtype=torch.float32, device=x.device)
    dx = torch.empty((x.shape[0], x.shape[1]), dtype=torch.float32, device=x.device) if want_grad else None
    check(_lib.lib().egv_cross_entropy_fwd_bwd(_p(x), x.stride(0), _p(t), x.shape[0], x.shape[1], int(ignore_index), _p(loss),
                                               _p(dx), x.shape[1], ops._stream(x)), "egv_cross_entropy_fwd_bwd")
    return loss, dx
