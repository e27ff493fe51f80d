"""Split-bf16 operand planes of the 2-D weights, refreshed when a parameter changes.

Parameters stay ordinary fp32 `nn.Parameter`s (checkpoints, optimizers and DDP keep working, SURVEY
8b "Ownership"); the GEMMs read derived bf16 planes W[N,K] (forward) and W^T[K,N] (dgrad).  The planes of a
weight are allocated once and live as long as the cache entry; they are stale when (param._version, EPOCH,
data_ptr) differs from what they were built from -- optimizers that update weights through raw pointers
(egovlp_amd.optim.AdamW) bump EPOCH.  The first stale weight touched after an optimizer step refreshes EVERY stale
entry of the cache in place with ONE egv_split_f32_multi launch (~100 tensors; one ~8 us launch each otherwise).
"""
from __future__ import annotations

import torch

from . import ops

EPOCH = 0


def bump_epoch():
    global EPOCH
    EPOCH += 1


class _Entry:
    __slots__ = ("params", "ver", "pl", "p2", "tp", "t16", "ok_token", "ok_epoch", "ok_vers")

    def __init__(self, params):
        # pl: split-bf16 planes W[N,K]; p2: the same weight in the f16x2 operand format, second-operand role (forward of the f16x2 mode); tp: split-bf16
        # W^T[K,N] (dgrad); t16: W^T[K,N] as ONE plane of plain fp16 (dgrad of the fp16 backward).  An entry owns whichever of the four its
        # callers have asked for so far.
        self.params, self.ver, self.pl, self.p2, self.tp, self.t16 = params, None, None, None, None, None
        self.ok_token, self.ok_epoch, self.ok_vers = -1, -1, None

    def mark_valid(self, token):
        self.ok_token, self.ok_epoch = token, EPOCH
        self.ok_vers = [p._version for p in self.params]

    def still_valid(self, token):
        """The cheap test of the per-call fast path: validated in THIS step (token), no optimizer step since (EPOCH), no in-place
        write since (tensor version counters).  Everything else (a replaced storage: `p.data = ...`, `.to()`) is caught by the
        full check -- data_ptr() included -- that `begin_step` runs over the whole cache once per forward."""
        if self.ok_token != token or self.ok_epoch != EPOCH:
            return False
        for p, v in zip(self.params, self.ok_vers):
            if p._version != v:
                return False
        return True

    def version(self):
        return tuple((p._version, EPOCH, p.data_ptr()) for p in self.params)

    def shapes_ok(self):
        n = sum(p.shape[0] for p in self.params)
        k = self.params[0][0].numel() if self.params[0].dim() > 1 else 1
        own = [x for x in (self.pl, self.p2) if x is not None]
        for t in (self.tp, self.t16):
            if t is not None and (t.rows, t.cols) != (k, n):
                return False
        return (bool(own) or self.tp is not None or self.t16 is not None) and all((x.rows, x.cols) == (n, k) for x in own)

    def has(self, need_t, fmt, t_fmt="bf16"):
        return (self.p2 if fmt == "f16x2" else self.pl) is not None and ((self.t16 if t_fmt == "f16" else self.tp) is not None or not need_t)


class WeightCache:
    def __init__(self):
        self._c = {}
        self._c_bias = {}
        self._token = 0
        self.param_structs = {}  # (id of a layer's first weight, direction) -> (plane objects, small-parameter addresses, C struct): the block / layer calls
        self._prep = None        # (signature of the stale set, replay of the split launch, replay of the f16x2 launch, a parameter)

    # -- one launch for every stale entry that already owns its planes
    def _stale(self, mark=False):
        """[(entry, current version)] of the entries whose planes are out of date (and can be refreshed in place); `mark`: stamp the
        up-to-date ones valid for this step while walking the cache (one version() -- a tuple of data_ptr() calls -- per entry)."""
        stale = []
        for ent in self._c.values():
            ver = ent.version()
            if ent.ver == ver:
                if mark:
                    ent.mark_valid(self._token)
            elif ent.shapes_ok():
                stale.append((ent, ver))
        return stale

    def _refresh_all(self, stale=None):
        stale = self._stale() if stale is None else stale
        if not stale:
            return
        # after an optimizer step the SAME entries are stale every time, with the same parameter and plane addresses: the argument
        # tables of the two multi-tensor launches are built once and replayed (what changes is the stream)
        def at(pl):
            return 0 if pl is None else (pl.hi.data_ptr(), pl.lo.data_ptr() if pl.lo is not None else 0)
        sig = tuple((at(ent.pl), at(ent.tp), at(ent.p2), at(ent.t16), tuple(v[2] for v in ver)) for ent, ver in stale)   # every address in the tables
        if self._prep is None or self._prep[0] != sig:
            jobs, jobs2 = [], []
            for ent, _ in stale:
                off = 0
                for i, p in enumerate(ent.params):
                    w2 = p.detach().reshape(p.shape[0], -1)
                    if not w2.is_contiguous():
                        # a non-contiguous parameter is converted through a temporary: nothing to replay
                        w2, sig = w2.contiguous(), None
                    n_i, last = w2.shape[0], i == len(ent.params) - 1
                    if ent.pl is not None or ent.tp is not None or ent.t16 is not None:
                        hi = lo = None
                        ldo = 0
                        if ent.pl is not None:
                            hi = ent.pl.hi.data_ptr() + off * ent.pl.ld * 2
                            lo = ent.pl.lo.data_ptr() + off * ent.pl.ld * 2
                            ldo = ent.pl.ld
                        if ent.tp is not None:
                            thi, tlo = ent.tp.hi.data_ptr() + off * 2, ent.tp.lo.data_ptr() + off * 2
                            ldt, tcols = ent.tp.ld, (ent.tp.ld - off if last else n_i)
                        else:
                            thi = tlo = None
                            ldt, tcols = 0, n_i
                        t16 = None
                        if ent.t16 is not None:       # same geometry as tp (both are [K, pad32(N)]): one job serves both
                            t16 = ent.t16.hi.data_ptr() + off * 2
                            ldt, tcols = ent.t16.ld, (ent.t16.ld - off if last else n_i)
                        jobs.append((w2, hi, lo, ldo, thi, tlo, ldt, tcols, t16))
                    if ent.p2 is not None:
                        jobs2.append((w2, ent.p2.hi.data_ptr() + off * ent.p2.ld * 2, ent.p2.lo.data_ptr() + off * ent.p2.ld * 2, ent.p2.ld))
                    off += n_i
            self._prep = (sig, ops.split_f32_multi(jobs, prepare=True), ops.f16x2_encode_multi(jobs2, prepare=True),
                          stale[0][0].params[0])
        _, run_split, run_x2, any_param = self._prep
        st = ops._stream(any_param)
        for run in (run_split, run_x2):
            if run is not None:
                run(st)
        for ent, ver in stale:
            ent.ver = ver

    def refresh(self):
        """Bring every cached plane set up to date NOW, on the current stream (callers that are about to fork work onto a
        second stream do this first, so that no stream finds a stale entry and refreshes the cache under the other one)."""
        self._refresh_all()

    def begin_step(self):
        """Start of a forward pass (ExecContext.begin_step): ONE full validation of the cache (version counters, plane epoch and
        storage addresses of ~100 weights), one multi-tensor refresh if anything is stale, and a new token -- within the step every
        `get` of a validated entry is three integer compares instead of a tuple of data_ptr() calls (15 us x 300 GEMMs)."""
        self._token += 1
        stale = self._stale(mark=True)
        self._refresh_all(stale)
        for ent, _ in stale:
            ent.mark_valid(self._token)

    def _get(self, params, need_t: bool, fmt: str = "bf16", t_fmt: str = "bf16"):
        key = id(params[0]) if len(params) == 1 else tuple(id(p) for p in params)
        ent = self._c.get(key)
        out = lambda: (ent.p2 if fmt == "f16x2" else ent.pl, ent.t16 if t_fmt == "f16" else ent.tp)
        if ent is None:
            ent = self._c[key] = _Entry(list(params))
        elif ent.has(need_t, fmt, t_fmt) and ent.still_valid(self._token):
            return out()
        if ent.ver == ent.version() and ent.has(need_t, fmt, t_fmt):
            ent.mark_valid(self._token)
            return out()
        if ent.has(need_t, fmt, t_fmt) and ent.shapes_ok():
            self._refresh_all()                 # stale after an optimizer step: refresh the whole cache in one launch
            return out()
        # first use (or another form of this weight is wanted for the first time): allocate what is missing, fill this entry alone
        w2 = torch.cat([p.detach().reshape(p.shape[0], -1) for p in params], dim=0) if len(params) > 1 \
            else params[0].detach().reshape(params[0].shape[0], -1)
        if not ent.shapes_ok():
            ent.pl = ent.p2 = ent.tp = ent.t16 = None
        want_pl = fmt == "bf16" or ent.pl is not None
        want_t = (need_t and t_fmt == "bf16") or ent.tp is not None
        want_t16 = (need_t and t_fmt == "f16") or ent.t16 is not None
        if want_pl or want_t:
            # planes always carry lo; single-pass GEMMs simply ignore it
            pl, tp, _ = ops.split_f32(w2, 3, want_rowmajor=want_pl, want_transposed=want_t)
            ent.pl = pl if want_pl else None
            ent.tp = tp
        if want_t16:
            n, k = w2.shape
            w2c = w2.contiguous()
            ld = ops.pad32(n)
            t = torch.empty((k, ld), dtype=torch.float16, device=w2.device)
            ent.t16 = ops.Planes(t, None, k, n, "f16")
            ops.split_f32_multi([(w2c, None, None, 0, None, None, ld, ld, t.data_ptr())])
        if fmt == "f16x2" or ent.p2 is not None:
            ent.p2 = ops.f16x2_encode(w2.contiguous(), role=1)
        ent.ver = ent.version()
        return out()

    def get(self, param: torch.Tensor, need_t: bool, fmt: str = "bf16", t_fmt: str = "bf16"):
        """-> (Planes [N,K], Planes [K,N] | None).  `param` is [N, ...] (conv weights are flattened to [N, K]).
        fmt 'f16x2': the [N,K] planes in the f16x2 operand format (second-operand role); t_fmt 'f16': the transposed weight as ONE plane of
        plain fp16 (the dgrad operand of the fp16 backward) instead of split-bf16 planes."""
        return self._get((param,), need_t, fmt, t_fmt)

    def get_cat(self, params, need_t: bool, fmt: str = "bf16"):
        """Planes of the row-wise concatenation of several [N_i, K] weights (DistilBERT's q/k/v projections run as ONE
        [3*768, 768] GEMM) -> (Planes [sum N_i, K], Planes [K, sum N_i] | None)."""
        return self._get(tuple(params), need_t, fmt)

    def get_bias_cat(self, biases):
        """The concatenation of several bias vectors (DistilBERT's fused q/k/v projection), rebuilt only when one of them
        changed -- not one torch.cat per layer per step."""
        key = ("bias",) + tuple(id(b) for b in biases)
        ent = self._c_bias.get(key)
        ver = tuple((b._version, EPOCH, b.data_ptr()) for b in biases)
        if ent is None or ent[0] != ver:
            ent = (ver, torch.cat([b.detach() for b in biases]))
            self._c_bias[key] = ent
        return ent[1]

    def clear(self):
        self._c.clear()
        self._c_bias.clear()
        self.param_structs.clear()
        self._prep = None
