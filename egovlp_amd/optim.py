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


class LossScaler:
    """Dynamic loss scale of the fp16 backward (`exec_ctx.set_precision('f16mix', 'f16')`): torch.cuda.amp.GradScaler's rule, decided
    entirely on the device.  The reference back-propagates in fp32 (trainer/trainer_egoclip.py:139-141) and needs none of this; fp16
    gradient planes need the loss multiplied by S (here: `scaler.scale(loss).backward()`), and

      * S, the found-inf flag, the good-step counter and the number of skipped steps live in ONE 32-byte device block (`state`);
      * `AdamW.step(scaler=...)` enqueues: a scan of every gradient for inf / NaN (egv_grad_nonfinite_multi), the one-thread decision
        kernel (egv_loss_scale_update: overflow -> skip + S *= backoff_factor; growth_interval good steps in a row -> S *= growth_factor)
        and the AdamW kernels, which read {lr, step size, 1 / S, skip} from the block and do NOTHING in a skipped step;
      * the host never reads any of it back during training (no synchronisation per step); `get_scale()` / `skipped_steps()` do, for
        logs and tests.  The bias correction counts APPLIED steps (host step count minus the device's skipped count).
    Overflow is detectable because gradient planes are written UN-clamped (csrc/f16x2.h f16_grad_piece8): a value beyond fp16's range
    becomes inf and poisons every gradient behind it."""

    def __init__(self, init_scale=2.0 ** 16, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, max_scale=2.0 ** 24, device="cuda"):
        if not (init_scale >= 1.0 and growth_factor >= 1.0 and 0.0 < backoff_factor <= 1.0 and growth_interval >= 1 and max_scale >= init_scale):
            raise ValueError("LossScaler: init_scale >= 1, growth_factor >= 1, 0 < backoff_factor <= 1, growth_interval >= 1, max_scale >= init_scale")
        self.growth_factor, self.backoff_factor = float(growth_factor), float(backoff_factor)
        self.growth_interval, self.max_scale = int(growth_interval), float(max_scale)
        host = torch.zeros(8, dtype=torch.int32)
        host.view(torch.float32)[0] = float(init_scale)
        host.view(torch.float32)[6] = 1.0 / float(init_scale)
        self.state = host.to(device)
        self._f = self.state.view(torch.float32)
        self.scale_tensor = self._f[0]          # 0-dim view: `loss * scale_tensor` reads S when the multiplication RUNS on the stream
        self._extra_hyper = {}                  # parameter-group index > 0 -> its own {lr, step_size, 1 / S, skip} block

    def scale(self, loss):
        """loss -> S * loss (a device-side multiply: backward() then carries S through every gradient of the step)."""
        return loss * self.scale_tensor

    def hyper_block(self, group_index):
        if group_index == 0:
            return self._f[4:8]
        blk = self._extra_hyper.get(group_index)
        if blk is None:
            blk = self._extra_hyper[group_index] = torch.zeros(4, dtype=torch.float32, device=self.state.device)
        return blk

    # ---- host readbacks (synchronise: logs, tests, checkpoints) ----
    def get_scale(self):
        return float(self._f[0].item())

    def skipped_steps(self):
        return int(self.state[3].item())

    def state_dict(self):
        st = self.state.cpu()
        return {"scale": float(st.view(torch.float32)[0]), "growth_tracker": int(st[1]), "skipped": int(st[3])}

    def load_state_dict(self, d):
        host = torch.zeros(8, dtype=torch.int32)
        host.view(torch.float32)[0] = float(d["scale"])
        host.view(torch.float32)[6] = 1.0 / float(d["scale"])
        host[1], host[3] = int(d.get("growth_tracker", 0)), int(d.get("skipped", 0))
        self.state.copy_(host)


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
        self._scaler = None     # the LossScaler of the step() in progress (None: un-scaled gradients, host-side hyper-parameters)
        self._scaler_first = False
        self._plans = {}        # id(param group) -> (params, states, exp_avg, exp_avg_sq, argument tables) of the steady-state launch

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0, scaler=None):
        """`scaler` (LossScaler): the gradients carry its scale S -- scan them for inf / NaN, let the device decide whether the step is
        applied (and with which S next), un-scale inside the update.  No host synchronisation either way."""
        loss = closure() if closure is not None else None
        groups = [(g, [p for p in g["params"] if p.grad is not None]) for g in self.param_groups]
        if scaler is not None:
            grads = [p.grad for _, ps in groups for p in ps]
            if any(g.is_sparse or not g.is_contiguous() for g in grads):
                raise RuntimeError("AdamW (HIP) needs contiguous dense gradients")
            ops.grad_nonfinite_multi(grads, scaler.state)
            self._scaler, self._scaler_first = scaler, True
        try:
            for gi, (group, params) in enumerate(groups):
                self._group_index = gi
                self._update_group(group, params, grad_scale)
        finally:
            self._scaler = None
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
        hyper = None
        sc = self._scaler
        if sc is not None:
            # the decision kernel of this step (first launch only: scale logic + found-inf reset), then this group's hyper block
            # {lr, step size at the number of APPLIED steps, 1 / S, skip} -- all on the device, read by the AdamW kernel below
            hyper = sc.hyper_block(getattr(self, "_group_index", 0))
            ops.loss_scale_update(sc.state, hyper, group["lr"], b1, b2, step, group["correct_bias"], sc.growth_factor, sc.backoff_factor,
                                  sc.growth_interval, sc.max_scale, advance=self._scaler_first)
            self._scaler_first = False
        ops.adamw_multi(ps, gs, ms, vs, group["lr"], b1, b2, group["eps"], group["weight_decay"], step,
                        group["correct_bias"], grad_scale, hyper_dev=hyper, tables=tables)


def adamw_step_size(lr, beta1, beta2, step, correct_bias=True):
    """The step-dependent scalar of transformers-4.2.1 AdamW: lr * sqrt(1 - beta2^t) / (1 - beta1^t) (double precision, as the
    C entry point computes it)."""
    if not correct_bias:
        return float(lr)
    import math
    return float(lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step))
