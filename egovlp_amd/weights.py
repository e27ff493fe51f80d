"""Split-bf16 operand planes of the 2-D weights, refreshed lazily when a parameter changes.

Parameters stay ordinary fp32 `nn.Parameter`s (checkpoints, optimizers and DDP keep working, SURVEY
8b "Ownership"); the GEMMs read derived bf16 planes W[N,K] (forward) and W^T[K,N] (dgrad).  A plane
set is rebuilt by one egv_split_f32 launch when (param._version, EPOCH) differs from the cached one;
optimizers that update weights through raw pointers (egovlp_amd.optim.AdamW) bump EPOCH.
"""
from __future__ import annotations

import torch

from . import ops

EPOCH = 0


def bump_epoch():
    global EPOCH
    EPOCH += 1


class WeightCache:
    def __init__(self):
        self._c = {}

    def get(self, param: torch.Tensor, need_t: bool):
        """-> (Planes [N,K], Planes [K,N] | None).  `param` is [N, ...] (conv weights are flattened to [N, K])."""
        key = id(param)
        ver = (param._version, EPOCH, param.data_ptr())
        ent = self._c.get(key)
        if ent is None or ent[0] != ver or (need_t and ent[2] is None):
            w2 = param.detach().reshape(param.shape[0], -1)
            passes = 3  # planes always carry lo; single-pass GEMMs simply ignore it
            pl, tp, _ = ops.split_f32(w2, passes, want_rowmajor=True, want_transposed=need_t or (ent is not None and ent[2] is not None))
            ent = (ver, pl, tp)
            self._c[key] = ent
        return ent[1], ent[2]

    def get_cat(self, params, need_t: bool):
        """Planes of the row-wise concatenation of several [N_i, K] weights (DistilBERT's q/k/v projections run as ONE
        [3*768, 768] GEMM) -> (Planes [sum N_i, K], Planes [K, sum N_i] | None)."""
        key = tuple(id(p) for p in params)
        ver = tuple((p._version, EPOCH, p.data_ptr()) for p in params)
        ent = self._c.get(key)
        if ent is None or ent[0] != ver or (need_t and ent[2] is None):
            w2 = torch.cat([p.detach().reshape(p.shape[0], -1) for p in params], dim=0)
            pl, tp, _ = ops.split_f32(w2, 3, want_rowmajor=True, want_transposed=need_t or (ent is not None and ent[2] is not None))
            ent = (ver, pl, tp)
            self._c[key] = ent
        return ent[1], ent[2]

    def clear(self):
        self._c.clear()
