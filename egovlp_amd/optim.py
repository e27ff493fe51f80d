"""AdamW with transformers==4.2.1 semantics on the fused multi-tensor kernel (egv_adamw_multi).

The reference builds its optimizer with `config.initialize('optimizer', transformers, params)`
(run/train_egoclip.py:72-73) -> `transformers.AdamW(lr=3e-5)` with that version's defaults
betas (0.9, 0.999), eps 1e-6, weight_decay 0.0, correct_bias True.  transformers 5.x no longer
ships AdamW, so this module is what `optimizer.type == "AdamW"` resolves to.

`overlap_backward()` (opt-in, single-process training without gradient accumulation): the update of a parameter needs
nothing but its own final gradient, and the update kernel is pure HBM streaming (16 B read + 12 B written per parameter,
5 GB per step) while the backward GEMMs that are still running are matrix-pipe bound.  With the overlap armed (by
`zero_grad()`), grad-ready hooks collect the parameters whose gradient has just become final and, every `min_elems`
parameters, enqueue their multi-tensor update on a second HIP stream behind an event of the producing stream; `step()`
updates whatever is left, joins the stream and invalidates the bf16 weight planes.  The arithmetic and its order per
element are those of `step()` -- results are bit-identical.  Backward never reads an fp32 parameter again once its
gradient hook has fired (every parameter belongs to exactly one autograd node; GEMMs read the cached bf16 planes, which are
refreshed only after `step()`).
"""
import torch

from . import ops, weights


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        if lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameters: {}".format(betas))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {} - should be >= 0.0".format(eps))
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      correct_bias=correct_bias))
        self._ov = None     # state of overlap_backward()
        self._graph_hyper = None

    # ------------------------------------------------------------------------------------------ overlap with backward
    def overlap_backward(self, min_elems=16 << 20, stream_of=None, exec_ctx=None):
        """Arm-able overlap of the update with the backward pass (see the module docstring).  Returns self.
        `stream_of(param)`: the stream the parameter's gradient is produced on (FrozenInTime.gradient_stream_of) -- the hook is
        registered under it so that autograd does not serialise the two towers' streams (see Bf16GradSync)."""
        if self._ov is not None:
            return self
        self._exec_ctx = exec_ctx        # the model's ExecContext: its streams are what the early updates are ordered behind
        group_of = {}
        for group in self.param_groups:
            for p in group["params"]:
                group_of[id(p)] = group
        self._ov = {"armed": False, "done": set(), "ready": [], "elems": 0, "min": int(min_elems), "stream": None,
                    "group_of": group_of, "dirty": False, "launches": 0}
        self._ov["handles"] = []
        for group in self.param_groups:
            for p in group["params"]:
                if not p.requires_grad:
                    continue
                st = stream_of(p) if stream_of is not None else None
                if st is not None and p.is_cuda:
                    with torch.cuda.stream(st):
                        self._ov["handles"].append(p.register_post_accumulate_grad_hook(self._on_grad))
                else:
                    self._ov["handles"].append(p.register_post_accumulate_grad_hook(self._on_grad))
        return self

    def zero_grad(self, set_to_none=True):
        super().zero_grad(set_to_none=set_to_none)
        if self._ov is not None:
            # the step protocol zero_grad -> forward -> backward -> step: the NEXT backward may update early
            self._ov["armed"] = True
            self._ov["ready"], self._ov["elems"] = [], 0

    def _on_grad(self, p):
        ov = self._ov
        if ov is None or not ov["armed"] or not p.is_cuda:
            return
        if id(p) in ov["done"]:
            raise RuntimeError("AdamW.overlap_backward(): a parameter received a second gradient before step() -- gradient "
                               "accumulation over several backward passes needs the plain step() (do not arm the overlap)")
        ov["ready"].append(p)
        ov["elems"] += p.numel()
        if ov["elems"] >= ov["min"]:
            self._flush_ready()

    @torch.no_grad()
    def _flush_ready(self):
        ov = self._ov
        ps, ov["ready"], ov["elems"] = ov["ready"], [], 0
        if not ps:
            return
        if ov["stream"] is None:
            ov["stream"] = torch.cuda.Stream()
        side = ov["stream"]
        # the gradients of `ps` are final on the streams that produced them: this node's stream, the text tower's stream and
        # (opt-in) the wgrad side stream -- order the current stream behind those, then the update stream behind it
        (self._exec_ctx or ops.DEFAULT).join_streams_for_gradient_hook()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._update(ps, 1.0)
        ov["done"].update(id(p) for p in ps)
        ov["dirty"] = True
        ov["launches"] += 1

    # ------------------------------------------------------------------------------------------------------ the update
    def _update(self, params, grad_scale):
        """Multi-tensor update of `params` (all with a gradient), grouped by (param group, step count)."""
        buckets = {}
        for p in params:
            if p.grad.is_sparse:
                raise RuntimeError("AdamW does not support sparse gradients")
            group = self._ov["group_of"][id(p)]
            st = self.state[p]
            if len(st) == 0 or "exp_avg" not in st:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st["step"] += 1
            if not (p.is_contiguous() and p.grad.is_contiguous()):
                raise RuntimeError("AdamW (HIP) needs contiguous parameters and gradients")
            b = buckets.setdefault((id(group), st["step"]), (group, st["step"], [], [], [], []))
            b[2].append(p); b[3].append(p.grad); b[4].append(st["exp_avg"]); b[5].append(st["exp_avg_sq"])
        for group, step, ps, gs, ms, vs in buckets.values():     # tensors at different step counts get their own launch
            self._launch(group, ps, gs, ms, vs, step, grad_scale)

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = closure() if closure is not None else None
        ov = self._ov
        if ov is not None and ov["armed"]:
            if grad_scale != 1.0 and (ov["dirty"] or ov["ready"]):
                raise RuntimeError("AdamW.overlap_backward() applies updates during backward: grad_scale must be 1")
            rest = [p for group in self.param_groups for p in group["params"]
                    if p.grad is not None and id(p) not in ov["done"]]
            ov["ready"], ov["elems"] = [], 0
            if rest:
                self._update(rest, grad_scale)
            if ov["dirty"]:
                torch.cuda.current_stream().wait_stream(ov["stream"])
                ov["dirty"] = False
            ov["done"].clear()
            ov["armed"] = False
        else:
            for group in self.param_groups:
                self._update_group(group, [p for p in group["params"] if p.grad is not None], grad_scale)
        weights.bump_epoch()   # parameters were written through raw pointers: invalidate the bf16 planes
        return loss

    def _update_group(self, group, params, grad_scale):
        ps, gs, ms, vs = [], [], [], []
        step = None
        for p in params:
            if p.grad.is_sparse:
                raise RuntimeError("AdamW does not support sparse gradients")
            st = self.state[p]
            if len(st) == 0 or "exp_avg" not in st:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st["step"] += 1
            if step is None:
                step = st["step"]
            elif step != st["step"]:
                # tensors at different step counts get their own launch group
                self._launch(group, ps, gs, ms, vs, step, grad_scale)
                ps, gs, ms, vs, step = [], [], [], [], st["step"]
            if not (p.is_contiguous() and p.grad.is_contiguous()):
                raise RuntimeError("AdamW (HIP) needs contiguous parameters and gradients")
            ps.append(p); gs.append(p.grad); ms.append(st["exp_avg"]); vs.append(st["exp_avg_sq"])
        self._launch(group, ps, gs, ms, vs, step, grad_scale)

    def _launch(self, group, ps, gs, ms, vs, step, grad_scale):
        if not ps:
            return
        b1, b2 = group["betas"]
        # graph_hyper: per param group, device floats {lr, step_size} the host refreshes before every HIP-graph replay
        # (egovlp_amd/graph.py GraphedTrainStep); None in eager mode
        hyper = self._graph_hyper.get(id(group)) if self._graph_hyper else None
        ops.adamw_multi(ps, gs, ms, vs, group["lr"], b1, b2, group["eps"], group["weight_decay"], step,
                        group["correct_bias"], grad_scale, hyper_dev=hyper)


def adamw_step_size(lr, beta1, beta2, step, correct_bias=True):
    """The step-dependent scalar of transformers-4.2.1 AdamW: lr * sqrt(1 - beta2^t) / (1 - beta1^t) (double precision, as the
    C entry point computes it)."""
    if not correct_bias:
        return float(lr)
    import math
    return float(lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step))
