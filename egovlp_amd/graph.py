"""hipGraph replay of the dual-encoder forward for launch-bound callers (EgoMCQ validation, feature dumps, retrieval eval).

The training step at B = 32 is GPU-bound (21 ms of host enqueue under 41 ms of kernels), so it is launched eagerly.  The
evaluation paths are the opposite: one text query + five candidate clips per EgoMCQ question (trainer/trainer_egoclip.py:
197-214) is ~330 kernel launches for ~3 ms of GPU work -- the Python / ctypes enqueue is the critical path.  `GraphedForward`
captures `model(data)` once per input shape into a HIP graph (both streams: the text tower forks onto its side stream and
re-joins inside the capture) and replays it with ONE launch per call:

    fwd = GraphedForward(model)                      # model.eval(), torch.no_grad() semantics
    text_embeds, video_embeds = fwd(data)            # same tensors model(data) returns; valid until the next call

Inputs are copied into the graph's static buffers; outputs are the graph's static output tensors (clone them to keep them
across calls).  The kernels and their arguments are exactly the eager ones, so results are bit-identical to `model(data)`.
Weights are read through the cached bf16 planes, which are refreshed OUTSIDE the graph: call `invalidate()` (or make a new
object) after the parameters changed (optimizer step, load_state_dict).
"""
from __future__ import annotations

import torch

from . import weights


class GraphedForward:
    def __init__(self, model, warmup: int = 2, max_graphs: int = 16):
        self.model = model
        self.warmup = warmup
        self.max_graphs = max_graphs
        self._graphs = {}
        self._epoch = None
        self.stats = {"captures": 0, "replays": 0}

    def invalidate(self):
        self._graphs.clear()

    @staticmethod
    def _key(data, video_only):
        v = data["video"]
        key = [tuple(v.shape), v.dtype, bool(video_only)]
        if not video_only:
            for k in sorted(data["text"]):
                key.append((k, tuple(data["text"][k].shape), data["text"][k].dtype))
        return tuple(key)

    def _capture(self, data, video_only):
        model = self.model
        static = {"video": data["video"].clone()}
        if not video_only:
            static["text"] = {k: t.clone() for k, t in data["text"].items()}
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side), torch.no_grad():      # warm-up off the capture: plane caches, kernel attributes, allocator
            for _ in range(self.warmup):
                model(static, video_only=video_only)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(graph):
            out = model(static, video_only=video_only)
        self.stats["captures"] += 1
        return {"graph": graph, "static": static, "out": out}

    @torch.no_grad()
    def __call__(self, data, video_only: bool = False):
        if self.model.training:
            raise RuntimeError("GraphedForward replays the eval-mode forward: call model.eval() first")
        if not data["video"].is_cuda:
            raise RuntimeError("GraphedForward needs device-resident inputs")
        if self._epoch != weights.EPOCH:                     # an optimizer step bumped the plane epoch: graphs read stale planes
            self._graphs.clear()
            self._epoch = weights.EPOCH
        key = self._key(data, video_only)
        ent = self._graphs.get(key)
        if ent is None:
            if len(self._graphs) >= self.max_graphs:
                self._graphs.pop(next(iter(self._graphs)))
            ent = self._graphs[key] = self._capture(data, video_only)
        ent["static"]["video"].copy_(data["video"], non_blocking=True)
        if not video_only:
            for k, t in data["text"].items():
                ent["static"]["text"][k].copy_(t, non_blocking=True)
        ent["graph"].replay()
        self.stats["replays"] += 1
        return ent["out"]


class GraphedTrainStep:
    """The whole EgoClip optimisation step -- zero_grad, dual-encoder forward on two streams, EgoNCE, backward with the
    weight gradients on their side stream, AdamW -- captured ONCE into a HIP graph and replayed with one launch per step
    (round-2 verdict, weak #8: ~840 kernel launches cost 20-30 ms of Python / ctypes enqueue per 39-ms step).

        step = GraphedTrainStep(model, loss_fn, optimizer)        # world size 1 (see `grad_sync` below)
        loss = step(data)                                          # same contract as trainer_egoclip.egoclip_step

    The first `warmup` calls run eagerly (optimizer state, weight-plane cache, kernel attributes, allocator); the next call
    captures; every later call copies the batch into the graph's static input buffers and replays.  What changes from step to
    step cannot be a launch argument (those are frozen at capture), so it lives in device memory the host refreshes before each
    replay with one tiny H2D copy: AdamW's {lr, step_size} per parameter group (`egv_adamw_multi(hyper_dev)`) and the per-step
    word XOR-ed into every dropout seed of the text tower (`seed_dev` of egv_dropout / egv_text_attn_*).  Host bookkeeping the
    replays skip (optimizer step counters, the weight-plane epoch) is done by `__call__`.

    Results: the kernels and their arguments are the eager ones, so with text dropout off a replayed step is bit-identical to
    the eager step (tests/test_gpu_trainer.py); with dropout on, the masks are drawn from (host seed ^ device word) instead of
    (host seed, call counter) -- the same distribution, not the same stream.

    Shapes are static: a batch of another shape re-captures (at most `max_graphs` graphs are kept).  `grad_sync` (world size > 1)
    is passed through to egoclip_step inside the capture; RCCL collectives are graph-capturable, but that path has not run on
    hardware here -- it is opt-in."""

    def __init__(self, model, loss_fn, optimizer, world_size=1, rank=0, grad_sync=None, warmup=2, max_graphs=4):
        from .optim import AdamW
        if not isinstance(optimizer, AdamW):
            raise TypeError("GraphedTrainStep needs egovlp_amd.optim.AdamW (its step-dependent scalars must be device-resident)")
        self.model, self.loss_fn, self.opt = model, loss_fn, optimizer
        self.world_size, self.rank, self.grad_sync = world_size, rank, grad_sync
        if warmup < 1:
            raise ValueError("GraphedTrainStep: at least one eager step before the capture (optimizer moments, plane cache and "
                             "kernel attributes must exist; allocating and zero-filling them inside the graph would repeat at every replay)")
        self.warmup_left = warmup
        self.max_graphs = max_graphs
        self._graphs = {}
        self._scalars = None       # device: per group {lr, step_size} floats, then the dropout seed word (int64 view)
        self._host = None          # ring of pinned staging copies, each with the event of the H2D copy that last read it
        self._host_i = 0
        self.stats = {"eager": 0, "captures": 0, "replays": 0}

    # ---- device-resident step scalars ----------------------------------------------------------------------------------
    def _ensure_scalars(self, device):
        if self._scalars is not None:
            return
        ng = len(self.opt.param_groups)
        n32 = 2 * ng + (2 * ng) % 2 + 2                       # group floats, pad to 8 bytes, one int64
        # A ring, not one buffer: the host may run several replays ahead of the GPU (nothing in a timed loop synchronises), and a
        # single staging buffer would be overwritten with step n + k's scalars before the queued copy of step n has read it.
        self._host = [[torch.zeros(n32, dtype=torch.float32).pin_memory(), None] for _ in range(4)]
        self._scalars = torch.zeros(n32, dtype=torch.float32, device=device)
        self._seed_off = (2 * ng + (2 * ng) % 2)
        self.opt._graph_hyper = {id(g): self._scalars[2 * i: 2 * i + 2] for i, g in enumerate(self.opt.param_groups)}
        self.model.text_model.seed_device = self._scalars[self._seed_off: self._seed_off + 2].view(torch.int64)

    def _next_step_of(self, group):
        for p in group["params"]:
            st = self.opt.state.get(p)
            if st and "step" in st:
                return st["step"] + 1
        return 1

    def _push_scalars(self, seed_word):
        from .optim import adamw_step_size
        slot = self._host[self._host_i]
        self._host_i = (self._host_i + 1) % len(self._host)
        if slot[1] is not None:
            slot[1].synchronize()          # the copy that read this staging buffer four steps ago has executed
        h = slot[0]
        for i, g in enumerate(self.opt.param_groups):
            b1, b2 = g["betas"]
            h[2 * i] = g["lr"]
            h[2 * i + 1] = adamw_step_size(g["lr"], b1, b2, self._next_step_of(g), g["correct_bias"])
        h[self._seed_off: self._seed_off + 2].view(torch.int64)[0] = seed_word
        self._scalars.copy_(h, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        slot[1] = ev

    def _seed_word(self):
        tm = self.model.text_model
        self._replay_no = getattr(self, "_replay_no", 0) + 1
        x = (torch.initial_seed() * 0x9E3779B97F4A7C15 + self._replay_no * 0xD1B54A32D192ED03 + tm.seed_rank * 0xA24BAED4963EE407) & (2 ** 63 - 1)
        return x ^ (x >> 29)

    def disable(self):
        """Back to eager: detach the device scalars from the optimizer and the text tower."""
        self.opt._graph_hyper = None
        self.model.text_model.seed_device = None
        self._graphs.clear()

    # ---- one step --------------------------------------------------------------------------------------------------------
    @staticmethod
    def _key(data):
        key = [tuple(data["video"].shape), data["video"].dtype]
        for k in sorted(data["text"]):
            key.append((k, tuple(data["text"][k].shape)))
        key.append((tuple(data["noun_vec"].shape), tuple(data["verb_vec"].shape)))
        return tuple(key)

    def _eager(self, data):
        from .trainer.trainer_egoclip import egoclip_step
        return egoclip_step(self.model, self.loss_fn, self.opt, data, self.world_size, self.rank, grad_sync=self.grad_sync)

    def _capture(self, data):
        static = {"video": data["video"].clone(), "text": {k: t.clone() for k, t in data["text"].items()},
                  "noun_vec": data["noun_vec"].clone(), "verb_vec": data["verb_vec"].clone()}
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            loss = self._eager(static)
        # the capture RECORDED the step, it did not run it: take back the host-side bookkeeping optimizer.step() did
        for g in self.opt.param_groups:
            for p in g["params"]:
                s = self.opt.state.get(p)
                if s and "step" in s:
                    s["step"] -= 1
        self.stats["captures"] += 1
        return {"graph": graph, "static": static, "loss": loss}

    def __call__(self, data):
        if not self.model.training:
            raise RuntimeError("GraphedTrainStep: call model.train() first")
        dev = data["video"].device
        self._ensure_scalars(dev)
        self._push_scalars(self._seed_word())            # also in the eager warm-up steps: the kernels read the device words
        if self.warmup_left > 0:
            self.warmup_left -= 1
            self.stats["eager"] += 1
            return self._eager(data)
        key = self._key(data)
        ent = self._graphs.get(key)
        if ent is None:
            if len(self._graphs) >= self.max_graphs:
                self._graphs.pop(next(iter(self._graphs)))
            ent = self._graphs[key] = self._capture(data)       # the capture itself does not execute the step ...
        st = ent["static"]
        st["video"].copy_(data["video"], non_blocking=True)
        for k, t in data["text"].items():
            st["text"][k].copy_(t, non_blocking=True)
        st["noun_vec"].copy_(data["noun_vec"], non_blocking=True)
        st["verb_vec"].copy_(data["verb_vec"], non_blocking=True)
        ent["graph"].replay()                                    # ... this does
        self.stats["replays"] += 1
        # host bookkeeping of the step the replay has just enqueued
        for g in self.opt.param_groups:
            for p in g["params"]:
                s = self.opt.state.get(p)
                if s and "step" in s:
                    s["step"] += 1
        weights.bump_epoch()
        return ent["loss"].clone()        # the graph's static loss tensor is overwritten by the next replay (egoclip_step returns a fresh one)
