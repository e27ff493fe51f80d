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

(Rounds 3 - 4 also shipped `GraphedTrainStep`, the whole optimisation step replayed from one graph: correct, tested, and 6 - 9 %
SLOWER than the eager three-stream schedule on ROCm 7.2 -- ~820 kernel nodes cost the host as much as 68 C-ABI calls and the
replayed kernels overlap less (profiles/r04_fin3_bench_graph_replay.json: 792 vs 873 pairs/s).  Retired in round 5; the code is
`git show c3983e8:egovlp_amd/graph.py`.  The C ABI stays capture-safe -- AdamW's scalars and the dropout seed word can be read
from device memory -- which is all a caller needs to capture the step itself.)
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
