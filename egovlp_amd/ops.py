"""Thin tensor-level wrappers over the C ABI (include/egovlp_hip.h).

PyTorch is used here for what the task statement calls plumbing: device memory (caching
allocator), streams and autograd bookkeeping.  Every arithmetic op below is one enqueue of a
hand-written gfx950 kernel on the current HIP stream; nothing falls back to ATen math.

Data format: an fp32-grade activation that feeds a GEMM lives as `Planes` = (hi, lo) bf16
tensors with hi = bf16(x), lo = bf16(x - hi).  `passes` = 3 uses both (fp32-grade product on bf16
MFMA), `passes` = 1 uses hi only (plain bf16).
"""
from __future__ import annotations

import ctypes as C
import os
import time
import sys
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from ._lib import GemmDesc, check

ACT_NONE, ACT_GELU, ACT_GELU_BWD, ACT_RELU_BWD = 0, 1, 2, 3


# ---------------------------------------------------------------------------------------------- execution context
# Everything that used to be process-global on the hot path (precision policy, side streams and their bookkeeping, the
# persistent-grid cap of the big GEMM, the backward poll of the data-parallel exchange, the kernel timer) lives in an
# `ExecContext`.  Every model owns one (FrozenInTime.exec_ctx, shared with its two towers) and hands it to its autograd
# functions, so two models -- or one model per thread / per device -- share nothing: the C ABI underneath is stateless and
# re-entrant (include/egovlp_hip.h; the reference's DDP reducer thread is the caller this has to survive,
# base/base_trainer.py:258).  SETTINGS a context does not set itself are inherited from its parent; `DEFAULT` is the root, and
# the module-level names of earlier rounds (`Precision.set`, `ops.WGRAD_SIDE_STREAM = ...`, `ops.BACKWARD_POLL = ...`,
# `ops.KERNEL_TIMER`, `ops.set_gemm_grid`) are views of DEFAULT's settings, kept for scripts and tests.  STATE (stream objects,
# the dirty / queued flags of the wgrad stream, the weight-plane cache) is private to a context and never inherited.
# "f16x2" (forward only; passes code 2 = its two MFMA products): the video blocks' qkv / fc1 / fc2 Linears run as TWO fp16 products on
# operands in the f16x2 format (include/egovlp_hip.h, csrc/f16x2.h) -- the accuracy of the three-product split-bf16 scheme at two
# thirds of its MFMA work; attention, the proj Linears, the patch embedding, the text tower and the heads stay split-bf16
# three-product.
# Inside the f16x2 mode a Linear may run ONE fp16 product (passes code 4: plain fp16(activation) x fp16(weight), 2^-11 per operand) where
# its share of the 1e-3 parity budget allows it: an error injected late in the tower reaches the embedding almost unamplified, one
# injected in the first blocks is amplified by everything behind it, so the policy is "op X runs single-product from block k_X on"
# (setting `f16_single`, see single_product_policy and profiles/r05_precision_table.txt).
# the backward 'f16x2' / 'f16mix' pair with when none is named: the fp16 backward (round 6); EGV_X2_BWD=bf16: round 5's pairing (A/B runs)
_DEFAULT_X2_BWD = os.environ.get("EGV_X2_BWD", "f16")
_AUX_BWD = int(os.environ.get("EGV_AUX_BWD", "3"))              # A/B: 1 = the text tower / patch embedding / heads back on ONE bf16 product next to an fp16 backward
_ENV_F16_SINGLE = os.environ.get("EGV_F16_SINGLE", "auto")     # read once: later Precision.set calls of the process agree
# "f16" (backward only; passes code 4 = ONE fp16 product): the backward of the video blocks on fp16 operands -- gradients carry a dynamic
# loss scale (egovlp_amd.optim.LossScaler; S lives in device memory, overflow -> skipped step + halved scale, no host sync), dY planes are
# un-clamped fp16 written by their producers, X is the forward's own fp16 operand plane (so the f16x2 / f16mix forward writes NO bf16
# copies), W^T is an fp16 plane.  2^-11 per operand instead of bf16's 2^-8: the weight gradients of the benchmarked mode were 2e-2 from
# fp32 with the bf16 backward (profiles/r05_backward_fp16_table.txt priced this).  Everything outside the video blocks (text tower, patch
# embedding, heads: 2 % of the FLOPs) then runs its backward on three bf16 products (`bwd_passes_split`).
_PASSES = {"bf16x3": 3, "bf16": 1, "f16x2": 2, "f16": 4}
F16_SINGLE_BITS = {"fc1": 1, "fc2": 2, "qkv": 4, "proj": 8}
_PASSES_INV = {3: "bf16x3", 1: "bf16", 2: "f16x2", 4: "f16"}
A1_INV = 1.0 / (1.0 - 2.0 ** -6)      # csrc/f16x2.h: plane 1 of a first-operand f16x2 encoding is fp16((1 - 2^-6) x)
_HARD_DEFAULTS = {
    "fwd_passes": 3, "bwd_passes": 3,
    # weight-gradient GEMMs on their own HIP stream.  OFF unless the owner of the gradient hooks turns it on (bench.py and
    # Multi_BaseTrainer_dist do): code that reads a weight gradient from an autograd hook DURING backward without calling
    # `join_streams_for_gradient_hook()` first (torch's DistributedDataParallel does) would race with the side stream.
    "wgrad_side_stream": os.environ.get("EGV_WGRAD_SIDE", "0") == "1",
    # the DistilBERT tower on a second HIP stream under the video tower (model/model.py FrozenInTime.forward)
    "text_side_stream": os.environ.get("EGV_TEXT_SIDE", "1") == "1",
    # persistent workgroups of the big GEMM: 256 = one per CU; data-parallel runs use 248 so that the RCCL kernels of the
    # overlapped gradient exchange find free CUs.  Travels in egv_gemm_desc.grid_cap; the wgrad split-K policy follows it.
    "gemm_grid": 256,
    # called (if set) at points of backward where earlier gradients are final (entry of every SpaceTimeBlock backward, of the
    # patch-embed / CLS-norm / projection nodes on the main stream): the hook-free gradient exchange hangs off it
    # (egovlp_amd.dist.Bf16GradSync(use_hooks=False).poll)
    "backward_poll": None,
    "kernel_timer": None,
    # one C-ABI call per SpaceTimeBlock forward / backward (egv_block_fwd / egv_block_bwd, one workspace arena per direction)
    # instead of the per-kernel calls: the same launches, ~10x less host work.  Off: the per-kernel reference path.
    "block_calls": os.environ.get("EGV_BLOCK_CALLS", "1") == "1",
    # how many steps the host may have enqueued beyond the one the GPU is executing.  Workspaces that side streams touch go back to
    # the caching allocator only when the GPU has passed them, so the memory a training loop holds grows with the host's lead: at
    # B = 16, T = 16 (66 GB of such workspaces per step) a lead of four steps reserved 273 of the 288 GB and one slower box went into
    # allocator retries (a device synchronisation each).  Two steps of lead keep the GPU fed (the host needs ~10 ms per step) and
    # bound the memory.  0: no limit.
    "max_steps_in_flight": int(os.environ.get("EGV_MAX_STEPS_IN_FLIGHT", "2")),
    # fwd_passes == 2: which Linears of which video blocks run ONE fp16 product.  "none": every qkv / fc1 / fc2 keeps its two products
    # (precision 'f16x2', round 4's benchmarked mode); "auto": single_product_policy(depth) (precision 'f16mix': what the per-op /
    # per-block table allows inside 5e-4); or "fc2:3,fc1:3,qkv:3" / a dict {"fc2": first single-product block, ...} (absent op: never).
    "f16_single": "none",
}


def single_product_policy(depth):
    """-> {"fc2": k, "fc1": k, "qkv": k, "proj": k2}: the op runs ONE fp16 product in blocks [k, depth); before that qkv / fc1 / fc2
    run two fp16 products (f16x2) and proj three bf16 products.  k = depth / 4, k2 = depth / 2.  profiles/r05_precision_table.txt
    (CPU oracle, operands rounded as the hardware rounds them; video-embedding error of the whole model against fp32): a
    single-product Linear costs (0.65 .. 0.9)e-4 in the later blocks but 2.4e-4 (fc1 / fc2) .. 9.4e-4 (qkv) in block 0, and the
    contributions add in quadrature -- with this policy: ViT-B/16 T = 4 4.7e-4, T = 16 4.5e-4, ViT-L/14 4.0e-4 (north_star's bar:
    1e-3; text tower untouched, 2.7e-5)."""
    return {"fc2": -(-depth // 4), "fc1": -(-depth // 4), "qkv": -(-depth // 4), "proj": -(-depth // 2)}


def parse_f16_single(spec, depth):
    """The `f16_single` setting -> {"op": first single-product block} (empty: none)."""
    if spec is None or spec == "none" or spec == "":
        return {}
    if spec == "auto":
        return single_product_policy(depth)
    if isinstance(spec, str):        # "fc2:0,fc1:4,qkv:6"
        out = {}
        for part in spec.split(","):
            op, _, k = part.partition(":")
            out[op.strip()] = int(k)
        spec = out
    unknown = set(spec) - set(F16_SINGLE_BITS)
    if unknown:
        raise ValueError(f"f16_single: unknown ops {sorted(unknown)} (known: {sorted(F16_SINGLE_BITS)})")
    return dict(spec)


class ExecContext:
    def __init__(self, parent=None, **settings):
        unknown = set(settings) - set(_HARD_DEFAULTS)
        if unknown:
            raise TypeError(f"ExecContext: unknown settings {sorted(unknown)}")
        self.parent = parent
        self._s = dict(settings)
        self._side = {"stream": None, "main": None, "dirty": False, "queued": False, "extra": [], "rr": 0, "held": []}
        self._text = {"stream": None, "main": None}     # "main": the stream forward() forked the text tower from
        self._inflight = []                             # events of the steps the host has enqueued (see _throttle)
        self.flow_wait_s = 0.0                          # seconds the host has waited in _throttle so far
        self._wc = None
        self._scaler = None
        self._pol_cache = {}

    # ---- settings (inherited) ------------------------------------------------------------------------------------------
    def get(self, key):
        c = self
        while c is not None:
            if key in c._s:
                return c._s[key]
            c = c.parent
        return _HARD_DEFAULTS[key]

    def set(self, **settings):
        unknown = set(settings) - set(_HARD_DEFAULTS)
        if unknown:
            raise TypeError(f"ExecContext: unknown settings {sorted(unknown)}")
        if "gemm_grid" in settings:
            g = int(settings["gemm_grid"])
            if not (8 <= g <= 256 and g % 8 == 0):
                raise ValueError("gemm_grid: a multiple of 8 in [8, 256]")
        self._s.update(settings)
        return self

    def unset(self, *keys):
        for k in keys:
            self._s.pop(k, None)
        return self

    def set_precision(self, fwd: str = "bf16x3", bwd: Optional[str] = None, f16_single=None):
        """'bf16x3' = split-bf16, three MFMA products, fp32-grade (meets the 1e-3 parity bar); 'bf16' = single pass;
        'f16x2' (forward only; backward 'f16' -- fp16 operands under a dynamic loss scale, the default -- or single-pass 'bf16') = two fp16
        products, fp32-grade like 'bf16x3' (3e-5 on the embeddings);
        'f16mix' = 'f16x2' in the first quarter of the video blocks, ONE fp16 product in their qkv / fc1 / fc2 Linears behind it and in
        the proj Linears from the middle of the tower on (4e-4).  `f16_single` ('f16mix' only): an explicit single-product policy
        ("fc2:3,fc1:3", a dict, "auto"); default: the policy this context already carries if it is a custom one (so that
        `set_precision(*precision_name())` restores a mode instead of resetting it), else EGV_F16_SINGLE (read ONCE, at import), else "auto"."""
        single = "none"
        if fwd == "f16mix":
            # 'f16mix' (the benchmarked mode of round 5) = 'f16x2' with ONE fp16 product where the parity budget allows it
            # (single_product_policy); EGV_F16_SINGLE overrides the policy (A/B runs: "none", "fc2:0", "fc2:3,fc1:3,qkv:3")
            cur = self._s.get("f16_single")
            fwd = "f16x2"
            single = f16_single if f16_single is not None else (cur if cur not in (None, "none", "", "auto") else _ENV_F16_SINGLE)
        elif f16_single not in (None, "none", ""):
            raise ValueError("f16_single is the per-block policy of the 'f16mix' forward")
        bwd = bwd if bwd is not None else (_DEFAULT_X2_BWD if fwd == "f16x2" else fwd)
        if bwd in ("f16x2", "f16mix") or fwd == "f16" or (fwd == "f16x2" and bwd not in ("bf16", "f16")) or (bwd == "f16" and fwd != "f16x2"):
            raise ValueError("'f16x2' / 'f16mix' are forward formats; they pair with the single-product backwards 'bf16' and 'f16' "
                             "('f16': fp16 operands under a dynamic loss scale, only behind these fp16 forwards)")
        return self.set(fwd_passes=_PASSES[fwd], bwd_passes=_PASSES[bwd], f16_single=single)

    def precision_name(self):
        fwd = _PASSES_INV[self.fwd_passes]
        if fwd == "f16x2" and self.get("f16_single") not in (None, "none", ""):
            fwd = "f16mix"
        return fwd, _PASSES_INV[self.bwd_passes]

    fwd_passes = property(lambda self: self.get("fwd_passes"))
    # what every forward product OUTSIDE the video blocks' qkv / fc1 / fc2 Linears runs with (patch embedding, text tower, heads,
    # attention, proj): the f16x2 mode keeps them split-bf16 three-product
    fwd_passes_split = property(lambda self: 3 if self.get("fwd_passes") == 2 else self.get("fwd_passes"))
    bwd_passes = property(lambda self: self.get("bwd_passes"))
    # the backward of everything OUTSIDE the video blocks (text tower, patch embedding, projection heads; 2 % of the step's FLOPs): next
    # to an fp16 backward it runs three bf16 products, so that every weight gradient of the step is fp32-grade
    bwd_passes_split = property(lambda self: _AUX_BWD if self.get("bwd_passes") == 4 else self.get("bwd_passes"))
    wgrad_side_stream = property(lambda self: self.get("wgrad_side_stream"))
    text_side_stream = property(lambda self: self.get("text_side_stream"))
    gemm_grid = property(lambda self: self.get("gemm_grid"))
    backward_poll = property(lambda self: self.get("backward_poll"))
    kernel_timer = property(lambda self: self.get("kernel_timer"))
    block_calls = property(lambda self: self.get("block_calls"))
    f16_single = property(lambda self: self.get("f16_single"))
    max_steps_in_flight = property(lambda self: self.get("max_steps_in_flight"))

    def f16_single_mask(self, layer, depth):
        """Bit mask (F16_SINGLE_BITS) of the Linears of video block `layer` (of `depth`) that run ONE fp16 product in the f16x2 mode."""
        if layer is None or depth is None:
            return 0
        spec = self.get("f16_single")
        # keyed on the CONTENT of the policy (a dict that is mutated or whose id is recycled must not hit a stale entry)
        key = (tuple(sorted(spec.items())) if isinstance(spec, dict) else spec, depth)
        pol = self._pol_cache.get(key)
        if pol is None:
            pol = self._pol_cache[key] = parse_f16_single(spec, depth)
        m = 0
        for op, k in pol.items():
            if layer >= k:
                m |= F16_SINGLE_BITS[op]
        return m

    def f16_single_policy(self, depth):
        return parse_f16_single(self.get("f16_single"), depth)

    def poll_backward(self):
        fn = self.get("backward_poll")
        if fn is not None:
            fn()

    # ---- state (private) -----------------------------------------------------------------------------------------------
    @property
    def wc(self):
        """The split-bf16 operand planes of this context's weights (egovlp_amd.weights.WeightCache)."""
        if self._wc is None:
            from .weights import WeightCache
            self._wc = WeightCache()
        return self._wc

    def loss_scaler(self, **kwargs):
        """The dynamic loss scale of this model's fp16 backward (egovlp_amd.optim.LossScaler), created on first use: what
        `egoclip_step` multiplies the loss with and hands to `AdamW.step(scaler=...)` when the backward precision is 'f16'."""
        if self._scaler is None:
            from .optim import LossScaler
            self._scaler = LossScaler(**kwargs)
        return self._scaler

    def text_stream(self):
        if self._text["stream"] is None:
            self._text["stream"] = torch.cuda.Stream()
        return self._text["stream"]

    def on_text_stream(self):
        return self._text["stream"] is not None and torch.cuda.current_stream() == self._text["stream"]

    def on_side_stream(self):
        if self._side["stream"] is None:
            return False
        cur = torch.cuda.current_stream()
        return cur == self._side["stream"] or any(cur == s_ for s_ in self._side["extra"])

    # The wgrad side stream.  dW = dY^T X is needed only by the optimizer, while the dgrad chain of backward waits for nothing
    # but dX.  With `wgrad_side_stream` on, every wgrad GEMM (and its split-K reduce) is enqueued on a second HIP stream behind
    # an event of the main stream: its persistent workgroups fill the CUs that the main stream's kernels leave idle (last
    # partial round of a tile grid, drain tails, the launch gaps of the small kernels).  The main stream re-joins at the end of
    # backward (an autograd engine callback queued by the first wgrad of the pass), before any gradient hook reads a wgrad
    # (`join_streams_for_gradient_hook`) and, unconditionally, in egoclip_step before the optimizer.
    def side_stream(self, *inputs, cost=1.0):
        """`with ec.side_stream(*inputs):` -- enqueue the body on the side stream, ordered after everything already enqueued on
        the current stream; `inputs` are the tensors the body reads (kept alive for the side stream by the allocator).  With
        several side streams the body goes to the one with the least work dealt to it so far in this step (`cost`: any additive
        measure, the callers pass MACs)."""
        return _SideStream(self, inputs, cost)

    def assign_side_streams(self, costs):
        """The dealing of `side_stream()` for a batch of weight-gradient GEMMs that a C block call will enqueue itself: -> for each
        cost a (torch stream, torch event) pair -- the event is recorded on the current stream and waited for by the side stream
        INSIDE the C call -- and the same bookkeeping (`dirty`, the end-of-backward join callback, load accounting)."""
        sd = self._side
        main = torch.cuda.current_stream()
        if sd["stream"] is None:
            sd["stream"] = torch.cuda.Stream()
            sd["extra"] = [torch.cuda.Stream() for _ in range(_wgrad_stream_count() - 1)]
        pool = [sd["stream"]] + sd["extra"]
        load = sd.get("load")
        if load is None or len(load) != len(pool):
            load = sd["load"] = [0.0] * len(pool)
        evs = sd.get("events")
        if evs is None or len(evs) < len(costs):
            evs = sd["events"] = [torch.cuda.Event() for _ in range(max(6, len(costs)))]
            for e in evs:
                e.record(main)                 # materialises the underlying hipEvent_t
        out = []
        for i, c in enumerate(costs):
            if _WGRAD_DEAL == "rr":
                k = sd["rr"] % len(pool)
                sd["rr"] += 1
            else:
                k = min(range(len(pool)), key=load.__getitem__)
            load[k] += float(c)
            out.append((pool[k], evs[i]))
        sd["main"], sd["dirty"] = main, True
        if not sd["queued"]:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(self._join_callback)
                sd["queued"] = True
            except RuntimeError:
                pass
        return out

    def begin_step(self):
        """Start of a forward / backward pass: forget a join callback that never ran (a backward that raised leaves
        'queued' set and later passes would not queue theirs), and validate / refresh the weight-plane cache once for the step."""
        self._side["queued"] = False
        self._side["load"] = None
        # a backward that raised before its join leaves the side streams dirty and their workspaces held: wait for them and let go
        self.join_side_stream()
        if torch.is_grad_enabled():
            self._throttle()                  # flow control of the TRAINING loop; evaluation forwards are not throttled
        if self._wc is not None:
            self._wc.begin_step()

    def _throttle(self):
        """Flow control of the training loop (setting max_steps_in_flight): an event per begin_step on the current stream -- the side
        streams of the previous step were joined into it -- and a host wait for the event of `limit` steps ago."""
        limit = self.get("max_steps_in_flight")
        if limit <= 0 or not torch.cuda.is_available() or torch.cuda.is_current_stream_capturing():
            return
        q = self._inflight
        ev = torch.cuda.Event()
        ev.record()
        q.append(ev)
        while len(q) > limit:
            t0 = time.perf_counter()
            q.pop(0).synchronize()
            self.flow_wait_s += time.perf_counter() - t0       # host time spent waiting here (bench.py reports it)

    def join_side_stream(self):
        """Make the main stream (the one the side work was forked from) and the current stream wait for everything enqueued on
        the side stream; workspaces held for the side streams (hold_until_join) go back to the allocator behind that wait."""
        sd = self._side
        if sd["dirty"]:
            cur = torch.cuda.current_stream()
            for st in [sd["stream"]] + sd["extra"]:
                sd["main"].wait_stream(st)
                if cur != sd["main"]:
                    cur.wait_stream(st)
            sd["dirty"] = False
        sd["held"].clear()

    def reset_side_streams(self):
        """Forget the wgrad side streams (after joining them): the next weight gradient re-creates them under the CURRENT policy
        (ops._wgrad_stream_count)."""
        self.join_side_stream()
        sd = self._side
        sd["stream"], sd["extra"], sd["load"], sd["events"], sd["rr"] = None, [], None, None, 0

    def hold_until_join(self, *tensors):
        """Keep main-stream allocations that side-stream kernels read or write alive until the next join_side_stream().  The
        alternative, Tensor.record_stream, parks a freed block behind an event of the side stream: it comes back to the pool only
        when the GPU has got there, so with multi-GB workspaces and a host that runs ahead the pool keeps growing by hipMalloc
        (BASELINE config 4: 55 ms of host time per step in hipMalloc and 171 GB reserved, against 73 GB and none; profiles/r04z_*).  Behind the join the main stream is ordered after the
        side streams, and a block freed then is reusable at once."""
        self._side["held"].extend(t for t in tensors if t is not None)

    def _join_callback(self):
        self._side["queued"] = False
        if self._wc is not None:
            self._wc.begin_step()
        self.join_side_stream()

    def order_behind_gradient_streams(self, stream):
        """Make `stream` (a private stream of the gradient exchange) wait for everything enqueued so far on the streams gradients
        are produced on -- the current one, the wgrad side stream, the text tower's -- WITHOUT making any of those wait: the
        exchange reads finished gradients, the backward pass that produces the next ones keeps running."""
        stream.wait_stream(torch.cuda.current_stream())
        if self._side["stream"] is not None:
            stream.wait_stream(self._side["stream"])
            for st in self._side["extra"]:
                stream.wait_stream(st)
        if self._text["stream"] is not None:
            stream.wait_stream(self._text["stream"])

    def join_streams_for_gradient_hook(self):
        """Gradient hooks run when a gradient has been ENQUEUED, on the stream of the node that produced it; a hook that reads
        gradients of several parameters (a bucket of the data-parallel exchange) must first order its stream behind the other
        streams gradients are produced on: the wgrad side stream and the text tower's stream."""
        self.join_side_stream()
        tx = self._text
        if tx["stream"] is not None:
            cur = torch.cuda.current_stream()
            if cur != tx["stream"]:
                cur.wait_stream(tx["stream"])
            elif tx.get("main") is not None:
                cur.wait_stream(tx["main"])       # a hook on the text stream that also reads gradients of the video tower


class _SideStream:
    def __init__(self, ec: ExecContext, inputs, cost=1.0):
        self.ec = ec
        self.cost = float(cost)
        self.inputs = [t for t in inputs if t is not None]

    def __enter__(self):
        sd = self.ec._side
        main = torch.cuda.current_stream()
        if sd["stream"] is None:
            sd["stream"] = torch.cuda.Stream()
            sd["extra"] = [torch.cuda.Stream() for _ in range(_wgrad_stream_count() - 1)]
        pool = [sd["stream"]] + sd["extra"]
        load = sd.get("load")
        if load is None or len(load) != len(pool):
            load = sd["load"] = [0.0] * len(pool)
        if _WGRAD_DEAL == "rr":
            k = sd["rr"] % len(pool)
            sd["rr"] += 1
        else:
            k = min(range(len(pool)), key=load.__getitem__)      # least-loaded stream (ties: the first)
        load[k] += self.cost
        side = pool[k]
        side.wait_stream(main)
        for t in self.inputs:
            t.record_stream(side)
        sd["main"], sd["dirty"] = main, True
        if not sd["queued"]:
            try:   # inside a backward pass: join when the pass ends, whoever called backward()
                torch.autograd.Variable._execution_engine.queue_callback(self.ec._join_callback)
                sd["queued"] = True
            except RuntimeError:
                pass
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        return side

    def __exit__(self, *exc):
        return self.ctx.__exit__(*exc)


DEFAULT = ExecContext()


def new_context(**settings) -> ExecContext:
    """A context with private state whose unset settings follow DEFAULT (what every model creates for itself)."""
    return ExecContext(DEFAULT, **settings)


class _PrecisionMeta(type):
    fwd_passes = property(lambda cls: DEFAULT.fwd_passes)
    bwd_passes = property(lambda cls: DEFAULT.bwd_passes)


class Precision(metaclass=_PrecisionMeta):
    """The DEFAULT context's precision policy (what every model follows unless its own context overrides it)."""

    @classmethod
    def set(cls, fwd: str = "bf16x3", bwd: Optional[str] = None):
        DEFAULT.set_precision(fwd, bwd)

    @classmethod
    def name(cls):
        return DEFAULT.precision_name()


def set_gemm_grid(workgroups: int) -> int:
    """DEFAULT's persistent-workgroup cap of the big GEMM -> the previous cap.  Multiples of 8 in [8, 256]."""
    prev = DEFAULT.gemm_grid
    if 8 <= workgroups <= 256 and workgroups % 8 == 0:
        DEFAULT.set(gemm_grid=int(workgroups))
    return prev


def _stream(t=None):
    """The HIP stream the next kernel is enqueued on: the current stream of the device `t` lives on.  Launching for a tensor
    of another device than the current one would run the kernel in the wrong HIP context: refuse (wrap the call in
    `torch.cuda.device(t.device)`)."""
    if t is not None and t.is_cuda and t.device.index != torch.cuda.current_device():
        raise _lib.EgovlpHipError(f"tensor on {t.device} but the current HIP device is cuda:{torch.cuda.current_device()}: "
                                  "egovlp_amd ops launch on the current device's current stream "
                                  "(use `with torch.cuda.device(t.device):`)")
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """Opt-in HIP-event timing of individual C-ABI calls (bench.py's roofline leg).  Events are recorded on
    the stream the kernel is launched on, immediately before and after the enqueue; `key` groups the launches of one
    problem shape."""

    def __init__(self):
        self.records = {}

    def time(self, name, flops, fn, passes=1, key=None):
        st = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        fn()
        e1.record(st)
        self.records.setdefault(name, []).append((e0, e1, flops, flops * passes, key))

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, recs in self.records.items():
            shapes = {}
            for e0, e1, fl, ifl, key in recs:
                d = shapes.setdefault(key, {"launches": 0, "seconds": 0.0, "flops": 0.0})
                d["launches"] += 1
                d["seconds"] += e0.elapsed_time(e1) * 1e-3
                d["flops"] += fl
            out[name] = {"launches": len(recs), "seconds": sum(d["seconds"] for d in shapes.values()),
                         "flops": float(sum(r[2] for r in recs)), "issue_flops": float(sum(r[3] for r in recs)),
                         "shapes": shapes}
        return out


def _p(t):
    return None if t is None else t.data_ptr()


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.EgovlpHipError("egovlp_amd ops need tensors resident in HBM (device 'cuda'); "
                                      "there is no CPU path in the product")


@dataclass
class Planes:
    hi: torch.Tensor                 # bf16 [rows, ld]                                  | fmt 'f16x2': fp16 plane 1 [rows, ld]
    lo: Optional[torch.Tensor]       # bf16 [rows, ld] or None (passes == 1)            | fmt 'f16x2': fp16 plane 2 [rows, ld]
    rows: int
    cols: int                        # logical columns (<= ld)
    fmt: str = "bf16"                # 'bf16' (split planes), 'f16x2' (include/egovlp_hip.h: egv_f16x2_encode; role: first / second operand)
                                     # or 'f16s' (an fp16 SPLIT: hi = fp16(x), lo = fp16(x - hi): the qkv planes of the fp16 attention)
                                     # or 'f16' (ONE plane of plain fp16 in `hi`, lo = None: the first operand of a single-fp16-product GEMM)
                                     # or 'bf16+f16' (attention output ahead of a single-product proj: hi = bf16(value) for the backward,
                                     # lo = fp16(value), the GEMM's operand)
    bf: Optional[torch.Tensor] = None  # fmt 'f16x2' / 'f16' only: bf16(value) [rows, ld], what the single-pass backward GEMMs read

    @property
    def ld(self):
        return self.hi.stride(0)

    def float(self):
        if self.fmt == "bf16+f16":
            return self.lo[:, : self.cols].float()
        v = self.hi[:, : self.cols].float()
        if self.lo is not None and self.fmt == "bf16":
            v = v + self.lo[:, : self.cols].float()
        return v

    def f16_plane(self):
        """The plain-fp16 plane a single-product GEMM (passes == 4) reads."""
        return self.hi if self.fmt == "f16" else self.lo

    def bwd16(self):
        """-> (Planes fmt 'f16' = plane 1 of this fp16 forward operand, alpha): the X operand of a weight gradient of the fp16 backward.
        alpha = 1 for plain fp16(x) ('f16'), 1 / (1 - 2^-6) for a1 = fp16((1 - 2^-6) x) of an f16x2 first-operand encoding."""
        if self.fmt == "f16":
            return self, 1.0
        if self.fmt == "f16x2":
            return Planes(self.hi, None, self.rows, self.cols, "f16"), A1_INV
        raise ValueError(f"the fp16 backward needs an fp16 forward operand, not '{self.fmt}' planes")

    def bwd(self):
        """The operand view the backward GEMMs take: the planes themselves, or the bf16 copy of an f16x2 operand."""
        if self.fmt == "bf16":
            return self
        if self.fmt == "bf16+f16":
            return Planes(self.hi, None, self.rows, self.cols)
        if self.bf is None:
            raise ValueError("this fp16 operand was produced without its bf16 plane (forward outside a training step)")
        return Planes(self.bf, None, self.rows, self.cols)


def empty_planes(rows, cols, passes, device, ld=None, zero=False):
    ld = cols if ld is None else ld
    mk = (lambda shp, dtype, device: zeros(shp, dtype, device)) if zero else torch.empty
    hi = mk((rows, ld), dtype=torch.bfloat16, device=device)
    lo = mk((rows, ld), dtype=torch.bfloat16, device=device) if passes == 3 else None
    return Planes(hi, lo, rows, cols)


def empty_planes_f16x2(rows, cols, device, want_bf=False, single=False, split=False):
    """Uninitialised f16x2 operand planes [rows, cols] (cols % 8 == 0); `single`: ONE plain fp16 plane (fmt 'f16'); `split`: an fp16
    split (fmt 'f16s')."""
    if cols % 8:
        raise ValueError("f16x2 operands come in 16-byte pieces (cols % 8 == 0)")
    hi = torch.empty((rows, cols), dtype=torch.float16, device=device)
    lo = None if single else torch.empty((rows, cols), dtype=torch.float16, device=device)
    bf = torch.empty((rows, cols), dtype=torch.bfloat16, device=device) if want_bf else None
    return Planes(hi, lo, rows, cols, "f16" if single else ("f16s" if split else "f16x2"), bf)


def f16_cast(x2d: torch.Tensor) -> Planes:
    """fp32 [rows, cols] -> ONE plane of un-clamped fp16 (fmt 'f16'; egv_f16x2_encode role 2): a scaled gradient entering the fp16 backward."""
    _need_cuda(x2d)
    rows, cols = x2d.shape
    pl = empty_planes_f16x2(rows, cols, x2d.device, single=True)
    check(_lib.lib().egv_f16x2_encode(_p(x2d), x2d.stride(0), rows, cols, _p(pl.hi), None, None, pl.ld, 2, _stream(x2d)), "egv_f16x2_encode")
    return pl


def f16x2_encode(x2d: torch.Tensor, role: int, want_bf=False) -> Planes:
    """fp32 [rows, cols] -> f16x2 operand planes (egv_f16x2_encode); role 0 = first operand (activations), 1 = second (weights)."""
    _need_cuda(x2d)
    rows, cols = x2d.shape
    pl = empty_planes_f16x2(rows, cols, x2d.device, want_bf)
    check(_lib.lib().egv_f16x2_encode(_p(x2d), x2d.stride(0), rows, cols, _p(pl.hi), _p(pl.lo), _p(pl.bf), pl.ld, int(role),
                                      _stream(x2d)), "egv_f16x2_encode")
    return pl


def f16x2_encode_multi(jobs, prepare=False):
    """One launch for many WEIGHTS (second-operand role).  jobs: (x2d [rows, cols] fp32, plane-1 address, plane-2 address, ldo).
    prepare: as split_f32_multi."""
    n = len(jobs)
    if n == 0:
        return None
    vp, i64, i32 = C.c_void_p * n, C.c_int64 * n, C.c_int32 * n
    for j in jobs:
        _need_cuda(j[0])
    args = (n, vp(*[j[0].data_ptr() for j in jobs]), i64(*[j[0].stride(0) for j in jobs]),
            i32(*[j[0].shape[0] for j in jobs]), i32(*[j[0].shape[1] for j in jobs]),
            vp(*[j[1] for j in jobs]), vp(*[j[2] for j in jobs]), i64(*[j[3] for j in jobs]), 1)
    fn = _lib.lib().egv_f16x2_encode_multi

    def run(stream):
        check(fn(*args, stream), "egv_f16x2_encode_multi")
    if prepare:
        return run
    run(_stream(jobs[0][0]))


def zeros(shape, dtype=torch.float32, device="cuda"):
    """torch.zeros without the ATen fill kernel: caching-allocator memory + one memset node on the current stream."""
    t = torch.empty(shape, dtype=dtype, device=device)
    check(_lib.lib().egv_zero(_p(t), t.numel() * t.element_size(), _stream(t)), "egv_zero")
    return t


_WGRAD_STREAMS = int(os.environ.get("EGV_WGRAD_STREAMS", "0"))       # A/B override (0 = policy below)
_WGRAD_DEAL = os.environ.get("EGV_WGRAD_DEAL", "load")                # "rr": round robin (A/B)
_WGRAD_KSPLIT_DIV = int(os.environ.get("EGV_WGRAD_KSPLIT_DIV", "0"))   # A/B override of the wgrad k-slice divisor (0 = policy)


def _wgrad_stream_count():
    """How many side streams the weight-gradient GEMMs are dealt to.  ONE, with half the k-slices of a main-stream launch (wgrad_ksplit).
    Rounds 3 - 4 used TWO streams with a third of the slices each in single-process runs (+0.7 % then); with the round-5 forward (a
    shorter main stream) and the XCD-contiguous wgrad mapping that order has turned: one stream / half the slices 997, two streams /
    half 995, two streams / a third (the old policy) 990 pairs/s, one stream / all slices 980 (same box, interleaved,
    profiles/r05w_wgrad_stream_policy_ab.txt).  One stream is also what a process group needs (RCCL's and the exchange's streams are
    in play then: a second wgrad stream next to them collapsed the step from 39 to 48.5 ms, profiles/r03_stream_ab.txt) -- so the
    N = 1 and the N > 1 step now run the same stream policy.  EGV_WGRAD_STREAMS overrides (A/B runs)."""
    if _WGRAD_STREAMS > 0:
        return _WGRAD_STREAMS
    return 1


_SIZE_CACHE = {}


def _cached_size(kind, *geom):
    """Workspace sizes are pure functions of the geometry: ask the library once per geometry (the step asks ~100 times)."""
    key = (kind,) + geom
    v = _SIZE_CACHE.get(key)
    if v is None:
        lib = _lib.lib()
        fn = {"ln_parts": lib.egv_layernorm_bwd_parts, "attn_fwd": lib.egv_divided_attn_fwd_work_floats,
              "attn_bwd": lib.egv_divided_attn_bwd_work_floats}[kind]
        v = _SIZE_CACHE[key] = int(fn(*geom))
    return v


def pad32(n):
    return (n + 31) // 32 * 32


# ------------------------------------------------------------------------------------------------ GEMM
def gemm_nt(a: Planes, b: Planes, *, passes, bias=None, residual=None, act=ACT_NONE, aux_in=None, aux_out=None,
            out_f32=None, out_planes: Optional[Planes] = None, alpha=1.0, ksplit=None, K=None, aux_is_grad=False,
            ec: Optional[ExecContext] = None, grad_out=False):
    """C[M,N] = A[M,K] . B[N,K]^T with the fused epilogue of egv_gemm_nt.  A.rows = M, B.rows = N.
    `aux_is_grad` (bf16 aux only): the GELU epilogue saves gelu'(z) instead of z and the GELU' epilogue multiplies by it.
    `ec`: the caller's execution context (grid cap of the big kernel, kernel timer); DEFAULT when omitted."""
    ec = DEFAULT if ec is None else ec
    M, N = a.rows, b.rows
    K = a.cols if K is None else K
    # passes 4: ONE fp16 product -- A a plain fp16 plane ('f16'), B the weight's f16x2 encoding (its plane 1 IS fp16(W))
    # (b 'f16' with passes 4: the fp16 W^T plane of a dgrad of the fp16 backward)
    if (a.fmt == "f16x2") != (passes == 2) or (a.fmt in ("f16", "bf16+f16")) != (passes == 4) or \
            b.fmt not in (("f16x2", "f16") if passes == 4 else (("f16x2",) if passes == 2 else ("bf16",))):
        raise ValueError(f"gemm_nt: operand formats {a.fmt} / {b.fmt} do not go with passes = {passes}")
    a_hi = a.f16_plane() if passes == 4 else a.hi
    out_fmt = 0
    if out_planes is not None and out_planes.fmt != "bf16":
        # 'f16x2' -> 1; 'f16' -> 2 (an activation: saturating) or 4 (a scaled gradient: un-clamped -- the GELU' epilogue, or grad_out);
        # 'f16s' -> 3 (an fp16 split: the qkv planes of the fp16 attention)
        out_fmt = {"f16x2": 1, "f16": 4 if (act == ACT_GELU_BWD or grad_out) else 2, "f16s": 3}[out_planes.fmt]
    if ksplit is None:
        ksplit = 1 if passes in (2, 4) else auto_ksplit_nt(M, N, K)
    aux = aux_in if aux_in is not None else aux_out
    aux_bf16 = int(aux is not None and aux.dtype in (torch.bfloat16, torch.float16))
    if aux_is_grad:
        if not aux_bf16:
            raise ValueError("aux_is_grad needs a 16-bit aux buffer")
        aux_bf16 = 3 if aux.dtype == torch.float16 else 2       # 3: gelu' saved as fp16 (the fp16 backward)
    elif aux is not None and aux.dtype == torch.float16:
        raise ValueError("an fp16 aux buffer holds gelu' (aux_is_grad)")
    if aux_bf16 and not uses_big_gemm(M, N, K, passes):
        raise ValueError("bf16 aux buffers are only supported by the big-tile GEMM kernel (see uses_big_gemm)")
    partial = torch.empty((ksplit, M, N), dtype=torch.float32, device=a.hi.device) if ksplit > 1 else None
    # one positional construction (the field order of egv_gemm_desc): 30 attribute stores cost ~6 us of host time per GEMM,
    # and the step issues 296 of them
    d = GemmDesc(_p(a_hi), (None if passes == 4 else _p(a.lo)), a.ld, _p(b.hi), _p(b.lo), b.ld, M, N, K, passes, alpha, act, _p(bias),
                 _p(residual), (residual.stride(0) if residual is not None else 0),
                 _p(aux_in), _p(aux_out), (aux.stride(0) if aux is not None else 0),
                 _p(out_f32), (out_f32.stride(0) if out_f32 is not None else 0),
                 _p(out_planes.hi) if out_planes is not None else None, _p(out_planes.lo) if out_planes is not None else None,
                 out_planes.ld if out_planes is not None else 0,
                 ksplit, 0, _p(partial), 0, aux_bf16, None, ec.gemm_grid, out_fmt,
                 _p(out_planes.bf) if out_fmt else None)
    timer = ec.kernel_timer
    if timer is not None:
        timer.time("egv_gemm_nt", 2.0 * M * N * K,
                   lambda: check(_lib.lib().egv_gemm_nt(C.byref(d), _stream(a.hi)), "egv_gemm_nt"), 1 if passes == 4 else passes,
                   key=("gemm_big " if ksplit <= 1 and uses_big_gemm(M, N, K, passes) else "gemm_nt(128x128) ")
                   + f"NT M={M} N={N} K={K} " + ("x1 fp16" if passes == 4 else f"x{passes}"))
    else:
        check(_lib.lib().egv_gemm_nt(C.byref(d), _stream(a.hi)), "egv_gemm_nt")


def auto_ksplit_nt(M, N, K):
    """Split-K factor for the small-M NT problems that run on the 128x128 kernel (DistilBERT's M = B*L = 1024 rows make
    48..192 tiles for 256 CUs): enough k-slices to put about one workgroup on every CU, at least 6 k-steps of 32 each; the
    slabs are summed by a reduce kernel that applies the full fused epilogue.  1 for everything else."""
    if SMALL_SPLITK == 0 or uses_big_gemm(M, N, K) or M >= 4096 or K % 32:
        return 1
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    ks = max(1, min(4, round(256 / max(tiles, 1)), (K // 32) // 6))
    return ks


SMALL_SPLITK = int(os.environ.get("EGV_SMALL_SPLITK", "1"))   # 0: off (A/B diagnostics)


def uses_big_gemm(M, N, K, passes=None):
    """Mirror of the kernel choice in csrc/gemm_nt.hip (gemm_variant) for NT problems without split-K (fp16 operands, passes = 2 / 4,
    always take the big-tile kernel when it can run the shape at all)."""
    if passes in (2, 4):
        return f16x2_gemm_ok(M, N, K)
    return M >= 256 and N >= 256 and K % 64 == 0 and ((M + 255) // 256) * ((N + 255) // 256) >= 128


def f16x2_gemm_ok(M, N, K):
    """Can egv_gemm_nt(passes = 2) run this NT shape?  (one 256 x 256 tile at least, 64-deep k-tiles)"""
    return M >= 256 and N >= 256 and K % 64 == 0


def gemm_tn(a: Planes, b: Planes, *, passes, out_f32, want_colsum=False, ksplit=None, ec: Optional[ExecContext] = None, alpha=1.0):
    """C[M,N] = A^T . B with both operands stored k-major: A is [K, M] (a.rows = K, a.cols = M), B is [K, N].
    This is the weight gradient dW = dY^T X with K = #tokens; neither operand is ever transposed in HBM
    (egv_gemm_nt, trans = 1).  -> colsum[M] = sum_k A[k, :] (the bias gradient) when want_colsum.  passes 4: both operands ONE plane of plain
    fp16 (fmt 'f16', the fp16 backward); alpha rescales the product (not the column sums)."""
    ec = DEFAULT if ec is None else ec
    Kd, M, N = a.rows, a.cols, b.cols
    if (passes == 4) != (a.fmt == "f16") or (passes == 4) != (b.fmt == "f16"):
        raise ValueError(f"gemm_tn: operand formats {a.fmt} / {b.fmt} do not go with passes = {passes}")
    if passes == 4 and (M < 256 or N < 256 or M % 8 or N % 8):
        raise ValueError("gemm_tn: fp16 weight gradients need the big-tile kernel (M, N >= 256)")
    if b.rows != Kd:
        raise ValueError("gemm_tn: operands disagree on the contraction length")
    dev = a.hi.device
    if M < 256 or N < 256 or M % 8 or N % 8:
        # below one 256x256 output tile (toy widths only; every EgoClip linear is >= 256 wide): transpose both
        # operands explicitly and run the small-tile NT kernel
        a_t, cs = transpose_planes(a, passes, want_colsum=want_colsum)
        b_t, _ = transpose_planes(b, passes)
        Kc = pad32(Kd)
        gemm_nt(a_t, b_t, passes=passes, out_f32=out_f32, ksplit=pick_ksplit(M, N, Kc), K=Kc, ec=ec)
        return cs
    if ksplit is None:
        ksplit = wgrad_ksplit(M, N, Kd, ec, ec.on_side_stream())
    cs = torch.empty(M, dtype=torch.float32, device=dev) if want_colsum else None
    partial = torch.empty(ksplit * (M * N + M), dtype=torch.float32, device=dev) if ksplit > 1 else None
    d = GemmDesc(_p(a.hi), _p(a.lo), a.ld, _p(b.hi), _p(b.lo), b.ld, M, N, Kd, passes, float(alpha), ACT_NONE, None,
                 None, 0, None, None, 0, _p(out_f32), out_f32.stride(0), None, None, 0,
                 ksplit, 0, _p(partial), 1, 0, _p(cs), ec.gemm_grid, 0, None)
    timer = ec.kernel_timer
    if timer is not None:
        timer.time("egv_gemm_nt", 2.0 * M * N * Kd,
                   lambda: check(_lib.lib().egv_gemm_nt(C.byref(d), _stream(a.hi)), "egv_gemm_nt(trans)"), 1 if passes == 4 else passes,
                   key=f"gemm_big TN M={M} N={N} K={Kd} " + ("x1 fp16" if passes == 4 else f"x{passes}"))
    else:
        check(_lib.lib().egv_gemm_nt(C.byref(d), _stream(a.hi)), "egv_gemm_nt(trans)")
    return cs


def wgrad_ksplit(M, N, Kd, ec, on_side):
    """k-slices of a TN weight-gradient GEMM (output M x N, contraction over Kd token rows): as many as fit ONE round of the
    persistent grid (256 workgroups, or the data-parallel cap: a slice count sized for 256 on a 248-workgroup grid would spill 4
    work units into a second round and double the wgrad's time) ... and HALF of that when the wgrad runs on the side stream next
    to the dgrad chain: it no longer has to fill the chip by itself, half the workgroups leave CUs to the main stream's kernels,
    and the fp32 slabs (and the reduce that reads them) are half as big.  Same box: 796.6 -> 819.8 pairs/s (+2.9 %); a third:
    788, a quarter: 635 -- the wgrads then become the critical path (profiles/r02_ab_wgrad_ksplit.txt).  (A third with two wgrad
    streams: two of them then share the free CUs.)"""
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    nkt = (Kd + 63) // 64
    div = _WGRAD_KSPLIT_DIV or ((3 if ec._side["extra"] else 2) if on_side else 1)
    return max(1, min(ec.gemm_grid // max(tiles, 1) // div, nkt // 2))


def pick_ksplit(M, N, K):
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    ks = K // 32
    want = max(1, 1024 // max(tiles, 1))
    return max(1, min(want, ks // 8 if ks >= 16 else 1))


# ---------------------------------------------------------------------------------------------- formats
def split_f32(x2d: torch.Tensor, passes, *, want_rowmajor=True, want_transposed=False, want_colsum=False):
    """fp32 [rows, cols] -> (Planes row-major | None, Planes transposed [cols, pad32(rows)] | None, colsum | None)."""
    _need_cuda(x2d)
    rows, cols = x2d.shape
    dev = x2d.device
    pl = empty_planes(rows, cols, passes, dev) if want_rowmajor else None
    tp = empty_planes(cols, rows, passes, dev, ld=pad32(rows)) if want_transposed else None
    cs = torch.empty(cols, dtype=torch.float32, device=dev) if want_colsum else None
    check(_lib.lib().egv_split_f32(
        _p(x2d), x2d.stride(0), rows, cols,
        _p(pl.hi) if pl else None, _p(pl.lo) if pl else None, pl.ld if pl else 0,
        _p(tp.hi) if tp else None, _p(tp.lo) if tp else None, tp.ld if tp else 0,
        _p(cs), _stream(x2d)), "egv_split_f32")
    return pl, tp, cs


def split_f32_multi(jobs, prepare=False):
    """One launch for many fp32 -> split-plane conversions.  jobs: (x2d [rows, cols] fp32, hi, lo, ldo, t_hi, t_lo, ldt, t_cols[, t16])
    with hi / lo / t_hi / t_lo / t16 raw device addresses (or None); see egv_split_f32_multi[_t16] (t16: the transposed matrix as one
    plane of plain fp16).
    prepare=True: build the argument tables and return `run(stream_handle)` instead of launching (None for no jobs) -- the weight
    cache replays the same table after every optimizer step."""
    n = len(jobs)
    if n == 0:
        return None
    vp, i64, i32 = C.c_void_p * n, C.c_int64 * n, C.c_int32 * n
    X = vp(*[j[0].data_ptr() for j in jobs])
    LDX = i64(*[j[0].stride(0) for j in jobs])
    R = i32(*[j[0].shape[0] for j in jobs])
    Cc = i32(*[j[0].shape[1] for j in jobs])
    HI, LO = vp(*[j[1] for j in jobs]), vp(*[j[2] for j in jobs])
    LDO = i64(*[j[3] for j in jobs])
    THI, TLO = vp(*[j[4] for j in jobs]), vp(*[j[5] for j in jobs])
    LDT = i64(*[j[6] for j in jobs])
    TC = i32(*[j[7] for j in jobs])
    T16 = vp(*[(j[8] if len(j) > 8 else None) for j in jobs])
    for j in jobs:
        _need_cuda(j[0])
    fn = _lib.lib().egv_split_f32_multi_t16

    def run(stream):
        check(fn(n, X, LDX, R, Cc, HI, LO, LDO, THI, TLO, LDT, TC, T16, stream), "egv_split_f32_multi_t16")
    if prepare:
        return run
    run(_stream(jobs[0][0]))


def transpose_planes(x: Planes, passes, want_colsum=False):
    """Planes [rows, cols] -> Planes [cols, pad32(rows)] (zero pad) (+ column sums of hi+lo)."""
    dev = x.hi.device
    tp = empty_planes(x.cols, x.rows, passes, dev, ld=pad32(x.rows))
    cs = torch.empty(x.cols, dtype=torch.float32, device=dev) if want_colsum else None
    check(_lib.lib().egv_transpose_planes(_p(x.hi), _p(x.lo) if passes == 3 else None, x.ld, x.rows, x.cols,
                                          _p(tp.hi), _p(tp.lo), tp.ld, _p(cs), _stream(x.hi)), "egv_transpose_planes")
    return tp, cs


def relu_split(x2d: torch.Tensor, passes) -> Planes:
    rows, cols = x2d.shape
    pl = empty_planes(rows, cols, passes, x2d.device)
    check(_lib.lib().egv_relu_split(_p(x2d), x2d.stride(0), rows, cols, _p(pl.hi), _p(pl.lo), pl.ld, _stream(x2d)),
          "egv_relu_split")
    return pl


# --------------------------------------------------------------------------------------------- LayerNorm
def layernorm_fwd(x2d, gamma, beta, eps, passes, *, x_add=None, want_sum=False, want_f32=False, want_planes=True,
                  rows=None, ldx=None, want_bf=False, single=False):
    """rows of x2d (optionally x2d + x_add) -> (Planes | None, y_f32 | None, mean, rstd, sum | None).
    `rows`/`ldx` allow strided row selection (e.g. only the CLS row of every clip).
    passes == 2: the output is written in the f16x2 operand format, first-operand role (`want_bf`: with the bf16 plane the backward
    reads); `single`: as ONE plane of plain fp16 instead (the consumer runs a single fp16 product)."""
    _need_cuda(x2d, gamma, beta)
    cols = x2d.shape[-1]
    rows = x2d.shape[0] if rows is None else rows
    ldx = x2d.stride(0) if ldx is None else ldx
    dev = x2d.device
    if passes == 2:
        if x_add is not None or want_sum or want_f32 or not want_planes:
            raise ValueError("layernorm_fwd: the f16x2 form writes operand planes only")
        pl = empty_planes_f16x2(rows, cols, dev, want_bf, single=single)
        mean = torch.empty(rows, dtype=torch.float32, device=dev)
        rstd = torch.empty(rows, dtype=torch.float32, device=dev)
        check(_lib.lib().egv_layernorm_fwd_f16x2(_p(x2d), ldx, _p(gamma), _p(beta), float(eps), rows, cols, _p(pl.hi), _p(pl.lo),
                                                 _p(pl.bf), pl.ld, _p(mean), _p(rstd), _stream(x2d)), "egv_layernorm_fwd_f16x2")
        return pl, None, mean, rstd, None
    pl = empty_planes(rows, cols, passes, dev) if want_planes else None
    yf = torch.empty((rows, cols), dtype=torch.float32, device=dev) if want_f32 else None
    mean = torch.empty(rows, dtype=torch.float32, device=dev)
    rstd = torch.empty(rows, dtype=torch.float32, device=dev)
    s = torch.empty_like(x2d) if want_sum else None
    check(_lib.lib().egv_layernorm_fwd(_p(x2d), _p(x_add), ldx, _p(gamma), _p(beta), float(eps), rows, cols, _p(s),
                                       _p(pl.hi) if pl else None, _p(pl.lo) if pl else None, _p(yf), cols,
                                       _p(mean), _p(rstd), _stream(x2d)), "egv_layernorm_fwd")
    return pl, yf, mean, rstd, s


def layernorm_bwd(dy2d, x2d, gamma, mean, rstd, *, add1=None, add2=None, rows=None, ldx=None, dx=None, lddx=None,
                  planes_passes=0):
    """-> (dx [rows, cols] (= add1 + add2 + LN-backward), dgamma, dbeta[, Planes of dx when planes_passes in (1, 3)]).
    dy2d: fp32 [rows, cols] or Planes (hi[, lo])."""
    dy_pl = dy2d if isinstance(dy2d, Planes) else None
    cols = dy_pl.cols if dy_pl is not None else dy2d.shape[-1]
    rows = (dy_pl.rows if dy_pl is not None else dy2d.shape[0]) if rows is None else rows
    ldx = x2d.stride(0) if ldx is None else ldx
    dev = x2d.device
    if dx is None:
        dx = torch.empty((rows, cols), dtype=torch.float32, device=dev)
        lddx = cols
    dg = torch.empty(cols, dtype=torch.float32, device=dev)
    db = torch.empty(cols, dtype=torch.float32, device=dev)
    parts = _cached_size("ln_parts", rows)
    work = torch.empty(2 * cols * parts, dtype=torch.float32, device=dev)
    # planes_passes 4: dx also as ONE plane of un-clamped fp16 (fmt 'f16'): the dY operand of the fp16 backward's next GEMMs
    if planes_passes == 4:
        pl = empty_planes_f16x2(rows, cols, dev, single=True)
    else:
        pl = empty_planes(rows, cols, planes_passes, dev) if planes_passes else None
    dy_f16 = dy_pl is not None and dy_pl.fmt == "f16"      # ONE plane of un-clamped fp16 (the dgrad GEMM of the fp16 backward wrote it)
    if dy_pl is not None:
        if dy_pl.fmt not in ("bf16", "f16"):
            raise ValueError(f"layernorm_bwd: dy planes must be split-bf16 or one un-clamped fp16 plane, got {dy_pl.fmt!r}")
        dy_args = (None, _p(dy_pl.hi), None if dy_f16 else _p(dy_pl.lo), dy_pl.ld)
    else:
        dy_args = (_p(dy2d), None, None, dy2d.stride(0))
    check(_lib.lib().egv_layernorm_bwd_fmt(*dy_args, _p(x2d), ldx, _p(gamma), _p(mean), _p(rstd), rows,
                                           cols, _p(add1), _p(add2), _p(dx), lddx, _p(pl.hi) if pl else None,
                                           _p(pl.lo) if pl else None, (1 if planes_passes == 4 else 0) | (2 if dy_f16 else 0), _p(dg), _p(db),
                                           _p(work), _stream(x2d)), "egv_layernorm_bwd_fmt")
    if planes_passes:
        return dx, dg, db, pl
    return dx, dg, db


# ------------------------------------------------------------------------------------------ video tokens
IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)    # data_loader/transforms.py:34-35 defaults


def patch_gather(video5d, P, passes, norm_mean=IMAGENET_MEAN, norm_std=IMAGENET_STD, aug=None) -> Planes:
    """im2col planes [B*T*patches, K = C*P*P]; K is padded with zero columns to a multiple of 64 (the GEMM k-tile;
    588 -> 640 for ViT-L/14), `cols` of the returned planes is the PADDED width.  A uint8 `video5d` (decoded frames) is
    scaled and normalised in the kernel (x / 255, then (x - mean) / std per channel) -- the loader's host transform.
    aug = (boxes int32 [B, 5] on the device: top, left, h, w, flip; out_res): the train transform (RandomResizedCrop +
    RandomHorizontalFlip, data_loader/transforms.py:14-19) runs inside the gather on the decoded uint8 clip."""
    B, T, Cc, H, W = video5d.shape
    if aug is not None:
        boxes, R = aug
        if video5d.dtype != torch.uint8:
            raise ValueError("the fused train transform takes decoded uint8 frames")
        if boxes.dtype != torch.int32 or tuple(boxes.shape) != (B, 5) or not boxes.is_cuda or not boxes.is_contiguous():
            raise ValueError("aug boxes: contiguous int32 [B, 5] on the device (top, left, h, w, flip)")
        H = W = int(R)
    rows = B * T * (H // P) * (W // P)
    K = Cc * P * P
    Kp = (K + 63) // 64 * 64
    pl = empty_planes(rows, Kp, passes, video5d.device, zero=(Kp != K))
    if video5d.dtype == torch.uint8:
        if len(norm_mean) != Cc or len(norm_std) != Cc:
            raise ValueError("patch_gather: one mean / std per channel")
        mean, std = (C.c_float * Cc)(*norm_mean), (C.c_float * Cc)(*norm_std)
        if aug is not None:
            check(_lib.lib().egv_patch_gather_u8_aug(_p(video5d), B * T, T, Cc, video5d.shape[3], video5d.shape[4], H, P,
                                                     _p(aug[0]), mean, std, _p(pl.hi), _p(pl.lo), pl.ld, _stream(video5d)),
                  "egv_patch_gather_u8_aug")
            return pl
        check(_lib.lib().egv_patch_gather_u8(_p(video5d), B * T, Cc, H, W, P, mean, std, _p(pl.hi), _p(pl.lo), pl.ld,
                                             _stream(video5d)), "egv_patch_gather_u8")
        return pl
    check(_lib.lib().egv_patch_gather(_p(video5d), B * T, Cc, H, W, P, _p(pl.hi), _p(pl.lo), pl.ld, _stream(video5d)),
          "egv_patch_gather")
    return pl


def assemble_tokens(pe, cls, pos, temporal, B, T, n, D):
    x = torch.empty((B, 1 + T * n, D), dtype=torch.float32, device=pe.device)
    check(_lib.lib().egv_assemble_tokens(_p(pe), _p(cls), _p(pos), _p(temporal), B, T, n, D, _p(x), _stream(pe)),
          "egv_assemble_tokens")
    return x


def assemble_tokens_bwd(dx, B, T, n, D, T_model):
    dev = dx.device
    d_pe = torch.empty((B * T * n, D), dtype=torch.float32, device=dev)
    d_cls = torch.empty((1, 1, D), dtype=torch.float32, device=dev)
    d_pos = torch.empty((1, n + 1, D), dtype=torch.float32, device=dev)
    d_tmp = zeros((1, T_model, D), device=dev)
    check(_lib.lib().egv_assemble_tokens_bwd(_p(dx), B, T, n, D, T_model, _p(d_pe), _p(d_cls), _p(d_pos), _p(d_tmp),
                                             _stream(dx)), "egv_assemble_tokens_bwd")
    return d_pe, d_cls, d_pos, d_tmp


# ---------------------------------------------------------------------------------------------- attention
ATT_OUT_FMTS = {"bf16": 0, "bf16+f16": 1, "f16x2": 2, "f16": 3}      # csrc/attn_common.h ATT_OUT_*


def divided_attn_fwd(qkv: Planes, B, T, n, H, mode, passes, out_f16=False, out_fmt=None):
    """qkv planes [B*S, 3*H*64] (the qkv GEMM's out_planes) -> (Planes [B*S, H*64], lse [B,H,S]).
    out_fmt (passes == 3; out_f16=True = 'bf16+f16'): the format of the output planes = the fmt of the returned Planes --
    'bf16' split planes; 'bf16+f16' hi = bf16(value) for a bf16 backward, lo = fp16(value), the operand of a one-product proj;
    'f16x2' the f16x2 first-operand planes of a TWO-product proj (fp16 backward); 'f16' ONE plane of fp16(value) (one-product proj, fp16
    backward)."""
    S = 1 + T * n
    dev = qkv.hi.device
    if qkv.fmt == "f16s":
        mode = mode | 8                      # the qkv planes are an fp16 split: fp16 MFMA products
    out_fmt = ("bf16+f16" if out_f16 else "bf16") if out_fmt is None else out_fmt
    if out_fmt != "bf16":
        if passes != 3:
            raise ValueError("divided_attn_fwd: fp16 output planes are written by the three-pass forward")
        if out_fmt in ("f16x2", "f16"):
            out = empty_planes_f16x2(B * S, H * 64, dev, single=(out_fmt == "f16"))
        else:
            out = empty_planes(B * S, H * 64, passes, dev)
            out = Planes(out.hi, out.lo.view(torch.float16), out.rows, out.cols, "bf16+f16")
        mode = mode | (ATT_OUT_FMTS[out_fmt] << 1)
    else:
        out = empty_planes(B * S, H * 64, passes, dev)
    lse = torch.empty((B, H, S), dtype=torch.float32, device=dev)
    work = torch.empty(_cached_size("attn_fwd", B, T, n, H, mode & 1), dtype=torch.float32, device=dev)
    check(_lib.lib().egv_divided_attn_fwd(_p(qkv.hi), _p(qkv.lo), B, T, n, H, mode, passes, _p(out.hi), _p(out.lo),
                                          _p(lse), _p(work), _stream(qkv.hi)), "egv_divided_attn_fwd")
    return out, lse


def divided_attn_bwd(qkv: Planes, out: Planes, d_out: Planes, lse, B, T, n, H, mode, passes, grad_f16=False) -> Planes:
    """-> dqkv planes [B*S, 3*H*64], ready to be the dY operand of the qkv dgrad / wgrad GEMMs.  grad_f16 (passes == 1): dqkv as ONE plane of
    un-clamped fp16 (the fp16 backward; q / k / v / dO are read as bf16 planes all the same)."""
    S = 1 + T * n
    dev = qkv.hi.device
    if grad_f16:
        dqkv = empty_planes_f16x2(B * S, 3 * H * 64, dev, single=True)
        mode = mode | 8
    else:
        dqkv = empty_planes(B * S, 3 * H * 64, passes, dev)
    mode = mode | (ATT_OUT_FMTS[out.fmt] << 1)          # how the forward wrote `out` (delta = rowsum(dO o O) decodes it)
    if qkv.fmt == "f16s":
        if not grad_f16 or d_out.fmt != "f16":
            raise ValueError("divided_attn_bwd: fp16 qkv planes go with an fp16 dO plane and an fp16 dqkv plane (the fp16 backward)")
        mode = mode | 16                                # q / k / v / dO are fp16: fp16 products throughout
    work = torch.empty(_cached_size("attn_bwd", B, T, n, H), dtype=torch.float32, device=dev)
    out_lo = out.lo if out.fmt == "bf16" else None      # an fp16(value) plane is not a residual: delta comes from the first plane alone
    check(_lib.lib().egv_divided_attn_bwd(_p(qkv.hi), _p(qkv.lo), _p(out.hi), _p(out_lo), _p(d_out.hi), _p(d_out.lo),
                                          _p(lse), B, T, n, H, mode, passes, _p(dqkv.hi), _p(dqkv.lo), _p(work),
                                          _stream(qkv.hi)), "egv_divided_attn_bwd")
    return dqkv


def text_attn_fwd(q, k, v, mask, B, L, H, passes, dropout_p=0.0, seed=0, seed_dev=None):
    """q, k, v: fp32 [B*L, H*64] tensors, or column-block views of one fused [B*L, 3*H*64] projection output.
    dropout_p > 0: attention-probability dropout with the counter-based mask of (dropout_p, seed ^ *seed_dev); seed_dev: an
    optional device int64[1] holding the per-step part of the seed (HIP-graph replay, egovlp_amd/graph.py)."""
    out = empty_planes(B * L, H * 64, passes, q.device)
    lse = torch.empty((B, H, L), dtype=torch.float32, device=q.device)
    assert q.stride(0) == k.stride(0) == v.stride(0) and q.stride(1) == 1
    check(_lib.lib().egv_text_attn_fwd(_p(q), _p(k), _p(v), q.stride(0), _p(mask), B, L, H, passes, float(dropout_p),
                                       int(seed), _p(seed_dev), _p(out.hi), _p(out.lo), _p(lse), _stream(q)), "egv_text_attn_fwd")
    return out, lse


def text_attn_bwd(q, k, v, mask, d_out, lse, B, L, H, passes, fused_out=False, dropout_p=0.0, seed=0, seed_dev=None):
    """-> (dq, dk, dv); with fused_out they are the column blocks of ONE [B*L, 3*H*64] tensor (returned 4th)."""
    HD = H * 64
    if fused_out:
        dqkv = torch.empty((B * L, 3 * HD), dtype=torch.float32, device=q.device)
        dq, dk, dv = dqkv[:, :HD], dqkv[:, HD:2 * HD], dqkv[:, 2 * HD:]
    else:
        dqkv = None
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    work = torch.empty((B, H, L), dtype=torch.float32, device=q.device)
    check(_lib.lib().egv_text_attn_bwd(_p(q), _p(k), _p(v), q.stride(0), _p(mask), _p(d_out), _p(lse), B, L, H, passes,
                                       float(dropout_p), int(seed), _p(seed_dev), _p(dq), _p(dk), _p(dv), dq.stride(0), _p(work),
                                       _stream(q)), "egv_text_attn_bwd")
    return (dq, dk, dv, dqkv) if fused_out else (dq, dk, dv)


def dropout(x, p, seed, add=None, seed_dev=None):
    """out = x * M'(p, seed ^ *seed_dev) + add (elementwise, contiguous fp32); with x = dy the same call is the backward."""
    _need_cuda(x, add)
    x = x.contiguous()
    out = torch.empty_like(x)
    check(_lib.lib().egv_dropout(_p(x), _p(add), _p(out), x.numel(), float(p), int(seed), _p(seed_dev), _stream(x)), "egv_dropout")
    return out


def embed_fwd(ids, word, pos, D):
    B, L = ids.shape
    e = torch.empty((B * L, D), dtype=torch.float32, device=word.device)
    check(_lib.lib().egv_embed_fwd(_p(ids), _p(word), _p(pos), B, L, D, _p(e), _stream(word)), "egv_embed_fwd")
    return e


def embed_bwd(ids, d_e, word_shape, pos_shape, pad_id=-1):
    B, L = ids.shape
    D = word_shape[1]
    d_word = zeros(tuple(word_shape), device=d_e.device)
    d_pos = zeros(tuple(pos_shape), device=d_e.device)
    check(_lib.lib().egv_embed_bwd(_p(ids), _p(d_e), B, L, D, int(pad_id), _p(d_word), _p(d_pos), _stream(d_e)),
          "egv_embed_bwd")
    return d_word, d_pos


# ------------------------------------------------------------------------------------------- loss / optim
def egonce_fwd_bwd(text, video, noun, verb, temperature, eps=1e-8, use_noun=True, use_verb=True, want_grads=True,
                   want_sim=False):
    _need_cuda(text, video, noun, verb)
    n, D = text.shape
    dev = text.device
    dn = noun.shape[1] if noun is not None else 0
    dv = verb.shape[1] if verb is not None else 0
    wf = _lib.lib().egv_egonce_work_floats(n, D)
    work = torch.empty(wf, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    sim = torch.empty((n, n), dtype=torch.float32, device=dev) if want_sim else None
    dt = torch.empty_like(text) if want_grads else None
    dvv = torch.empty_like(video) if want_grads else None
    check(_lib.lib().egv_egonce_fwd_bwd(_p(text), _p(video), _p(noun), _p(verb), n, D, dn, dv, float(temperature),
                                        float(eps), int(use_noun), int(use_verb), _p(loss), _p(sim), _p(dt), _p(dvv),
                                        _p(work), _stream(text)), "egv_egonce_fwd_bwd")
    return loss, sim, dt, dvv


def grad_nonfinite_multi(grads, state):
    """state[2] |= 1 if any gradient holds an inf / NaN (egv_grad_nonfinite_multi); `state`: the int32[8] device block of a LossScaler."""
    n = len(grads)
    if n == 0:
        return
    G = (C.c_void_p * n)(*[g.data_ptr() for g in grads])
    N = (C.c_int64 * n)(*[g.numel() for g in grads])
    check(_lib.lib().egv_grad_nonfinite_multi(n, G, N, _p(state), _stream(grads[0])), "egv_grad_nonfinite_multi")


def loss_scale_update(state, hyper_out, lr, beta1, beta2, step, correct_bias, growth, backoff, interval, max_scale, advance):
    check(_lib.lib().egv_loss_scale_update(_p(state), _p(hyper_out), float(lr), float(beta1), float(beta2), int(step), int(correct_bias),
                                           float(growth), float(backoff), int(interval), float(max_scale), int(advance), _stream(state)),
          "egv_loss_scale_update")


def adamw_tables(params, ms, vs):
    """The argument tables of egv_adamw_multi that do not change from step to step (parameter / moment addresses, sizes)."""
    n = len(params)
    arr = C.c_void_p * n
    return (arr(*[p.data_ptr() for p in params]), arr(*[m.data_ptr() for m in ms]), arr(*[v.data_ptr() for v in vs]),
            (C.c_int64 * n)(*[p.numel() for p in params]))


def adamw_multi(params, grads, ms, vs, lr, beta1, beta2, eps, weight_decay, step, correct_bias=True, grad_scale=1.0, hyper_dev=None,
                tables=None):
    n = len(params)
    P, M_, V, N = tables if tables is not None else adamw_tables(params, ms, vs)
    G = (C.c_void_p * n)(*[g.data_ptr() for g in grads])
    check(_lib.lib().egv_adamw_multi(n, P, G, M_, V, None, None, N, float(lr), float(beta1), float(beta2),
                                     float(eps), float(weight_decay), int(step), int(correct_bias),
                                     float(grad_scale), _p(hyper_dev), _stream(params[0])), "egv_adamw_multi")


# ---- module-level views of DEFAULT's settings (earlier rounds' names; scripts, tools and tests assign to them) -----------------
class _OpsModule(type(sys.modules[__name__])):
    WGRAD_SIDE_STREAM = property(lambda m: DEFAULT.wgrad_side_stream, lambda m, v: DEFAULT.set(wgrad_side_stream=bool(v)))
    TEXT_SIDE_STREAM = property(lambda m: DEFAULT.text_side_stream, lambda m, v: DEFAULT.set(text_side_stream=bool(v)))
    BACKWARD_POLL = property(lambda m: DEFAULT.backward_poll, lambda m, v: DEFAULT.set(backward_poll=v))
    KERNEL_TIMER = property(lambda m: DEFAULT.kernel_timer, lambda m, v: DEFAULT.set(kernel_timer=v))
    GEMM_GRID = property(lambda m: DEFAULT.gemm_grid)


sys.modules[__name__].__class__ = _OpsModule
