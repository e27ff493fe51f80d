"""AdamW with transformers==4.2.1 semantics on the fused multi-tensor kernel (egv_adamw_multi).

The reference builds its optimizer with `config.initialize('optimizer', transformers, params)`
(run/train_egoclip.py:72-73) -> `transformers.AdamW(lr=3e-5)` with that version's defaults
betas (0.9, 0.999), eps 1e-6, weight_decay 0.0, correct_bias True.  transformers 5.x no longer
ships AdamW, so this module is what `optimizer.type == "AdamW"` resolves to.

(An `overlap_backward()` mode -- updates enqueued from grad-ready hooks under the rest of backward -- was built in round 2,
measured 1.8 % SLOWER (the persistent GEMM workgroups own every CU, a 5-GB HBM stream squeezed into their tails only delays
them; profiles/r02_w_*) and removed in round 4.)
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
        self._graph_hyper = None
        self._plans = {}        # id(param group) -> (params, states, exp_avg, exp_avg_sq, argument tables) of the steady-state launch

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            self._update_group(group, [p for p in group["params"] if p.grad is not None], grad_scale)
        weights.bump_epoch()   # parameters were written through raw pointers: invalidate the bf16 planes
        return loss

    def load_state_dict(self, state_dict):
        self._plans.clear()
        return super().load_state_dict(state_dict)

    def _update_group(self, group, params, grad_scale):
        # steady state: the same parameters with the same moment buffers, all at the same step count -- one launch whose
        # parameter / moment tables were built at the first such step (the gradient table is rebuilt: gradients are new tensors)
        plan = self._plans.get(id(group))
        if plan is not None and len(plan[0]) == len(params) and all(a is b for a, b in zip(plan[0], params)):
            _, states, ms, vs, tables, ptrs = plan
            step = states[0]["step"] + 1
            ok = all(p.data_ptr() == q for p, q in zip(params, ptrs))       # `p.data = ...` / `.to()` since the plan was made
            for p_, st, m, v in zip(params, states, ms, vs):
                # the cached state dicts must still BE the optimizer's state (a caller may have replaced optimizer.state[p] or one of
                # the moment tensors without load_state_dict): otherwise the fast path would keep updating orphaned buffers
                ok = ok and st["step"] + 1 == step and st["exp_avg"] is m and st.get("exp_avg_sq") is v and self.state.get(p_) is st
                st["step"] += 1
            grads = [p.grad for p in params]
            if ok and not any(g.is_sparse or not g.is_contiguous() for g in grads):
                self._launch(group, params, grads, ms, vs, step, grad_scale, tables)
                return
            for st in states:          # something changed under the plan: undo, take the general path
                st["step"] -= 1
            del self._plans[id(group)]
        ps, gs, ms, vs, sts = [], [], [], [], []
        step = None
        uniform = True
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
                uniform = False
            if not (p.is_contiguous() and p.grad.is_contiguous()):
                raise RuntimeError("AdamW (HIP) needs contiguous parameters and gradients")
            ps.append(p); gs.append(p.grad); ms.append(st["exp_avg"]); vs.append(st["exp_avg_sq"]); sts.append(st)
        self._launch(group, ps, gs, ms, vs, step, grad_scale)
        if uniform and ps:
            self._plans[id(group)] = (list(ps), sts, list(ms), list(vs), ops.adamw_tables(ps, ms, vs), [p.data_ptr() for p in ps])

    def _launch(self, group, ps, gs, ms, vs, step, grad_scale, tables=None):
        if not ps:
            return
        b1, b2 = group["betas"]
        # graph_hyper: per param group, device floats {lr, step_size} the host refreshes before every HIP-graph replay
        # (egovlp_amd/graph.py GraphedTrainStep); None in eager mode
        hyper = self._graph_hyper.get(id(group)) if self._graph_hyper else None
        ops.adamw_multi(ps, gs, ms, vs, group["lr"], b1, b2, group["eps"], group["weight_decay"], step,
                        group["correct_bias"], grad_scale, hyper_dev=hyper, tables=tables)


def adamw_step_size(lr, beta1, beta2, step, correct_bias=True):
    """The step-dependent scalar of transformers-4.2.1 AdamW: lr * sqrt(1 - beta2^t) / (1 - beta1^t) (double precision, as the
    C entry point computes it)."""
    if not correct_bias:
        return float(lr)
    import math
    return float(lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step))
