"""Data-parallel gradient averaging for MI355X nodes: `Bf16GradSync` replaces the reference's
`DistributedDataParallel(model, find_unused_parameters=True)` (base/base_trainer.py:258).

What the reference does per step: DDP copies every fp32 gradient into 25 MB fp32 buckets, all-reduces them with NCCL
(724 MB on the wire per rank for the 180.93 M parameters), divides by the world size, and walks the autograd graph looking
for unused parameters (there are none).  What this does instead, sized for xGMI (point-to-point links, ring collectives
are per-link bound):

  * the exchange format is **bf16**: a bucket's fp32 gradients are scaled by 1/W, rounded and packed back to back into one
    flat bf16 buffer by ONE multi-tensor kernel (`egv_grad_pack_bf16`) -- half the bytes on the links;
  * buckets are **large and few** (default 64 MB of bf16 = 32 M parameters, ~6 collectives per step instead of DDP's ~29),
    built from the order in which gradients actually became ready in the first backward;
  * each bucket's `all_reduce` (RCCL, SUM of pre-scaled values = the mean) is issued **from the grad-ready hook of its last
    parameter**, asynchronously, while backward keeps running; `finish()` (called between backward and optimizer.step)
    waits for the collectives and unpacks the means into the fp32 `.grad` tensors (`egv_grad_unpack_bf16`);
  * no graph walk, no module wrapper: `model.state_dict()` keeps the reference's key names on every world size.

Two ways to move a bucket (`exchange`):
  * "direct" (default): the all-reduce is spelled out as what a fully connected xGMI node is good at (every
    GPU has a private link to every other, a ring keeps five of the seven idle) -- ONE all-to-all in which rank r receives slice
    r of every peer's bucket over all links at once, a LOCAL sum of those W slices in fp32 (`egv_slice_sum_bf16`: one rounding
    to bf16 instead of the W - 1 roundings of a bf16 all-reduce), and ONE all-gather of the reduced slices.  The same bytes per
    rank as a ring all-reduce (2 (W-1)/W of the bucket), W-1 links busy instead of 2.  The whole chain -- pack, all-to-all, sum,
    all-gather -- is enqueued on a private exchange stream behind an event of the compute streams, so backward never waits
    for it; `finish()` makes the compute stream wait for the chain's last event;
  * "allreduce": RCCL's stock `all_reduce(SUM)` of the bf16 bucket (bf16 accumulation inside the collective).

Semantics equal DDP's: after `finish()` every rank holds (1/W) * sum_r grad_r in p.grad (up to bf16 rounding of the
exchanged values, 2^-9 relative -- the single-pass backward that produces them rounds its GEMM operands the same way).
Initial parameters are broadcast from rank 0 at construction, as DDP does.

The pack / unpack kernels are the HIP ones on the device; `pack_fn` / `unpack_fn` exist so that the HOST logic (bucket
layout, hook accounting, collective calls) can be exercised on CPU tensors with gloo in tests -- there is no CPU default.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, List, Optional

import torch
import torch.distributed as dist


# Opt-in timing of the exchange steps (bench.py --gpus N): a list that receives (name, start_event, end_event) recorded on the
# compute stream around the embedding all-gather and around Bf16GradSync.finish() (= the part of the gradient all-reduce
# that backward did NOT hide, plus the unpack).
COMM_EVENTS = None


def timed(name, fn):
    if COMM_EVENTS is None or not torch.cuda.is_available():
        return fn()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    out = fn()
    e1.record(st)
    COMM_EVENTS.append((name, e0, e1))
    return out


def _hip_pack(grads, flat, offsets, scale):
    from . import _lib, ops
    n = len(grads)
    P = (C.c_void_p * n)(*[g.data_ptr() for g in grads])
    N = (C.c_int64 * n)(*[g.numel() for g in grads])
    O = (C.c_int64 * n)(*offsets)
    _lib.check(_lib.lib().egv_grad_pack_bf16(n, P, N, flat.data_ptr(), O, float(scale), ops._stream()), "egv_grad_pack_bf16")


def _hip_slice_sum(recv, world, slice_elems, out):
    from . import _lib, ops
    _lib.check(_lib.lib().egv_slice_sum_bf16(recv.data_ptr(), int(world), int(slice_elems), out.data_ptr(), ops._stream(recv)),
               "egv_slice_sum_bf16")


def _hip_unpack(grads, flat, offsets):
    from . import _lib, ops
    n = len(grads)
    P = (C.c_void_p * n)(*[g.data_ptr() for g in grads])
    N = (C.c_int64 * n)(*[g.numel() for g in grads])
    O = (C.c_int64 * n)(*offsets)
    _lib.check(_lib.lib().egv_grad_unpack_bf16(n, P, N, flat.data_ptr(), O, ops._stream()), "egv_grad_unpack_bf16")


class _Bucket:
    __slots__ = ("params", "offsets", "flat", "pending", "work", "numel", "recv", "red", "slice")

    def __init__(self, params, device, world=1, direct=False):
        self.params = params
        self.offsets, off = [], 0
        for p in params:
            self.offsets.append(off)
            off += (p.numel() + 7) // 8 * 8          # every tensor starts on a 16-byte boundary of the bf16 buffer
        if direct:                                   # W equal slices, each a whole number of 16-byte pieces
            off = (off + 8 * world - 1) // (8 * world) * (8 * world)
        self.numel = off
        self.flat = torch.zeros(off, dtype=torch.bfloat16, device=device)     # the padding stays zero forever
        self.slice = off // world if direct else 0
        self.recv = torch.empty(off, dtype=torch.bfloat16, device=device) if direct else None
        self.red = torch.empty(self.slice, dtype=torch.bfloat16, device=device) if direct else None
        self.pending = len(params)
        self.work = None


class Bf16GradSync:
    def __init__(self, params, process_group=None, bucket_mb: float = 64.0,
                 pack_fn: Optional[Callable] = None, unpack_fn: Optional[Callable] = None, broadcast: bool = True,
                 stream_of: Optional[Callable] = None, use_hooks: bool = True, order_hint: Optional[List] = None,
                 exec_ctx=None, exchange: str = "direct", slice_sum_fn: Optional[Callable] = None):
        """`stream_of(param)` -> the HIP stream that parameter's gradient is produced on (None: the current one).  A grad-ready
        hook keeps the parameter's AccumulateGrad node alive across iterations, and autograd runs that node on the stream
        that was current WHEN THE HOOK WAS REGISTERED: for the text tower (its backward runs on its own stream,
        ops.TEXT_SIDE_STREAM) a hook registered on the default stream makes the default stream wait for the text stream at
        every text gradient -- the two towers serialise (measured: -1.2 ms of the overlap, profiles/r02_d_*).

        `use_hooks=False`: no autograd hooks at all.  Merely HAVING 327 post-accumulate-grad hooks costs 1.7 ms of GPU time per
        step on this model (measured with empty hook bodies, profiles/r02_d_dp_overhead.txt); instead the owner calls `poll()`
        at points of backward where earlier gradients are known to be final (egovlp_amd.ops.BACKWARD_POLL, invoked at the entry
        of every SpaceTimeBlock backward): every bucket whose parameters all have a gradient is packed and all-reduced there.
        Needs `zero_grad(set_to_none=True)` (a gradient is "ready" when it is not None) and `order_hint`, the parameters in
        the order their gradients become final (buckets are cut along it).

        `exec_ctx`: the model's egovlp_amd.ops.ExecContext -- gradients are produced on up to three of ITS streams (main, text
        tower, wgrad side stream) and a bucket's pack kernel is ordered behind all of them first."""
        if exchange not in ("direct", "allreduce"):
            raise ValueError("Bf16GradSync: exchange is 'direct' or 'allreduce'")
        self.exchange = exchange
        self.slice_sum_fn = slice_sum_fn or _hip_slice_sum
        self._xstream = None            # the private stream of the direct exchange (created on first use, CUDA only)
        self.use_hooks = use_hooks
        self.exec_ctx = exec_ctx
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("Bf16GradSync: no trainable parameters")
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.bucket_elems = int(bucket_mb * (1 << 20) / 2)
        self.pack_fn = pack_fn or _hip_pack
        self.unpack_fn = unpack_fn or _hip_unpack
        dev = self.params[0].device
        if pack_fn is None and dev.type != "cuda":
            raise RuntimeError("Bf16GradSync packs gradients with HIP kernels: parameters must live on an MI355X")
        self.device = dev
        if broadcast:
            with torch.no_grad():
                for p in self.params:
                    dist.broadcast(p.data, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0,
                                   group=process_group)
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._ready_order: List[int] = []
        self._buckets: Optional[List[_Bucket]] = None       # built after the first backward, from its ready order
        self._bucket_of = {}
        self._seen = set()
        self.stats = {"buckets": 0, "collectives_last_step": 0, "bytes_last_step": 0, "launched_during_backward": 0}
        self._handles = []
        if not use_hooks:
            if order_hint is not None:
                hinted = [p for p in order_hint if p.requires_grad]
                if {id(p) for p in hinted} != {id(p) for p in self.params}:
                    raise ValueError("Bf16GradSync: order_hint must be a permutation of the trainable parameters")
                self._ready_order = [self._index[id(p)] for p in hinted]
            else:
                self._ready_order = list(range(len(self.params)))[::-1]
            self._seen = set(self._ready_order)
            self._build_buckets()
            self._next = 0
        for p in (self.params if use_hooks else []):
            st = stream_of(p) if stream_of is not None else None
            if st is not None and p.is_cuda:
                with torch.cuda.stream(st):
                    self._handles.append(p.register_post_accumulate_grad_hook(self._on_ready))
            else:
                self._handles.append(p.register_post_accumulate_grad_hook(self._on_ready))

    # ---- hooks (run on the autograd engine thread, in the order gradients become final) ----------------------------
    def _on_ready(self, p):
        i = self._index[id(p)]
        if self._buckets is None:
            if i not in self._seen:
                self._seen.add(i)
                self._ready_order.append(i)
            return
        b = self._bucket_of.get(i)
        if b is None:
            return
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    _DIAG = __import__("os").environ.get("EGV_SYNC_DIAG", "")     # diagnostics: "hooks" = no pack / reduce / unpack, "noreduce"

    def _launch(self, b: _Bucket):
        if self._DIAG == "hooks":
            b.work = False
            return
        direct = self.exchange == "direct" and self._DIAG != "noreduce"
        if self.pack_fn is _hip_pack and not (direct and self.device.type == "cuda"):
            from . import ops
            # gradients are produced on up to three streams (main, text tower, wgrad side stream) of the model's context: the
            # all-reduce is issued behind the CURRENT stream, which therefore has to wait for the other two (the direct
            # exchange orders its own private stream behind all three instead and leaves the compute streams alone)
            (self.exec_ctx or ops.DEFAULT).join_streams_for_gradient_hook()
        grads = [p.grad for p in b.params]
        for g in grads:
            if g is None or g.dtype != torch.float32 or not g.is_contiguous():
                raise RuntimeError("Bf16GradSync needs dense contiguous fp32 gradients")
        if direct:
            self._launch_direct(b, grads)
        else:
            self.pack_fn(grads, b.flat, b.offsets, 1.0 / self.world)
            if self._DIAG == "noreduce":
                b.work = False
            else:
                b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.stats["collectives_last_step"] += 1
        self.stats["bytes_last_step"] += b.numel * 2

    def _launch_direct(self, b: _Bucket, grads):
        """pack -> all-to-all (slice r of every peer's bucket comes to rank r, all links at once) -> fp32 sum of the W slices ->
        all-gather of the reduced slices, all on the private exchange stream: nothing here makes the compute stream or the host
        wait (a collective issued inside `torch.cuda.stream(xs)` orders RCCL's stream behind xs and xs behind the collective)."""
        def chain():
            self.pack_fn(grads, b.flat, b.offsets, 1.0 / self.world)
            dist.all_to_all_single(b.recv, b.flat, group=self.group)
            self.slice_sum_fn(b.recv, self.world, b.slice, b.red)
            dist.all_gather_into_tensor(b.flat, b.red, group=self.group)
        if self.device.type != "cuda":
            chain()
            b.work = False
            return
        if self._xstream is None:
            self._xstream = torch.cuda.Stream()
        xs = self._xstream
        if self.pack_fn is _hip_pack:
            from . import ops
            (self.exec_ctx or ops.DEFAULT).order_behind_gradient_streams(xs)      # xs waits; main / text / wgrad streams do not
        else:
            xs.wait_stream(torch.cuda.current_stream())
        for g in grads:
            g.record_stream(xs)
        with torch.cuda.stream(xs):
            chain()
            done = torch.cuda.Event()
            done.record(xs)
        b.work = done

    def poll(self):
        """Hook-free mode: launch every not-yet-launched bucket (in order) whose parameters all have their gradient."""
        if self.use_hooks or self._buckets is None:
            return
        while self._next < len(self._buckets):
            b = self._buckets[self._next]
            if any(p.grad is None for p in b.params):
                break
            self._launch(b)
            self._next += 1

    def _agree_on_order(self, order):
        """Every rank must cut identical buckets, or the all-reduces pair up unrelated gradients (or hang): adopt rank 0's
        order, as DDP does with its bucket rebuild.  (Hook mode observes the order locally; the hook-free order comes from the
        model and is the same everywhere, but costs one tiny broadcast to check.)"""
        if self.world <= 1:
            return order
        t = torch.tensor(order, dtype=torch.int64, device=self.device if self.device.type == "cuda" else "cpu")
        mine = t.clone()
        dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        if not torch.equal(t, mine):
            import warnings
            warnings.warn("Bf16GradSync: this rank observed a different gradient-ready order than rank 0; using rank 0's")
        return [int(i) for i in t.tolist()]

    def _build_buckets(self):
        order = self._ready_order + [i for i in range(len(self.params)) if i not in self._seen]
        order = self._agree_on_order(order)
        buckets, cur, cur_n = [], [], 0
        for i in order:
            p = self.params[i]
            if cur and cur_n + p.numel() > self.bucket_elems:
                buckets.append(cur)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            buckets.append(cur)
        self._buckets = [_Bucket(ps, self.device, self.world, self.exchange == "direct") for ps in buckets]
        self._bucket_of = {self._index[id(p)]: b for b in self._buckets for p in b.params}
        self.stats["buckets"] = len(self._buckets)

    # ---- between backward and optimizer.step -------------------------------------------------------------------------
    def finish(self):
        """Wait for the bucket all-reduces of this backward and leave the rank-mean gradient in every p.grad."""
        return timed("grad_sync_exposed", self._finish)

    def _finish(self):
        if self._buckets is not None:      # how many buckets had already left when backward returned (the rest is exposed)
            self.stats["launched_during_backward"] = self.stats["collectives_last_step"]
        if not self.use_hooks:
            self.poll()
            if self._next < len(self._buckets):
                raise RuntimeError("Bf16GradSync.finish(): a parameter received no gradient (unused parameters are not supported)")
            self._next = 0
        if self._buckets is None:
            # first step: the ready order is known only now -> build the buckets and reduce them all here (no overlap)
            self._build_buckets()
            self.stats["collectives_last_step"] = self.stats["bytes_last_step"] = 0
            for b in self._buckets:
                b.params = [p for p in b.params]
                if any(p.grad is None for p in b.params):
                    raise RuntimeError("Bf16GradSync: a parameter received no gradient (unused parameters are not supported; "
                                       "the EgoClip step uses every parameter, SURVEY 7)")
                self._launch(b)
        for b in self._buckets:
            if b.work is None:
                raise RuntimeError("Bf16GradSync.finish(): a bucket was never launched -- some parameter received no "
                                   "gradient in this backward")
            if isinstance(b.work, torch.cuda.Event):
                torch.cuda.current_stream().wait_event(b.work)      # direct exchange: the chain's last event
            elif b.work is not False:
                b.work.wait()
            if self._DIAG != "hooks":
                self.unpack_fn([p.grad for p in b.params], b.flat, b.offsets)
            b.work = None
            b.pending = len(b.params)
        out = dict(self.stats)
        self.stats["collectives_last_step"] = self.stats["bytes_last_step"] = 0
        return out

    def remove_hooks(self):
        for h in self._handles:
            h.remove()
        self._handles = []
