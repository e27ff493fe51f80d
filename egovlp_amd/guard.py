"""Runtime guard of the per-block precision policy ('f16mix').

`ops.single_product_policy` -- which Linears of which video blocks run ONE fp16 product -- was derived on Gaussian weights with a 2x
margin to the 1e-3 parity bar.  On weights with outlier channels (what trained ViTs look like; `egovlp_amd.synth.heavy_tensor` is a
synthetic stand-in) the same policy measures 1.0e-3 on the video embedding where the fp32-grade modes stay at 4e-4
(profiles/r06j_heavy_weights_forward.txt): the policy's error depends on the weights, so it has to be MEASURED on the weights at
hand, not assumed.  `PrecisionGuard.check(batch)` does that on the device:

  * reference: the video tower's forward of the same batch on three bf16 products everywhere ('bf16x3': fp32-grade, and bf16's
    exponent range -- a saturating fp16 operand shows up as an error here as well; fp16 encoders clamp at 65504);
  * candidate: the current policy; error = ||v - v_ref|| / ||v_ref|| over the batch (and its worst row);
  * over budget (default 6e-4: the policy was designed to spend 5e-4 and measures 4.2e-4 .. 4.7e-4 on Gaussian weights; with the
    fp32-grade formats' own 3e-5 .. 4e-4 that keeps the embedding inside ~7e-4 of the fp32 reference) or non-finite -> DEMOTE one rung and
    measure again:
        auto (qkv / fc1 / fc2 single from depth/4, proj from depth/2) -> (depth/2, 3 depth/4) -> (3 depth/4, never)
        -> 'none' (= 'f16x2': two fp16 products everywhere, fp32-grade) -> bf16x3 forward with the bf16 backward;
  * every demotion is logged; nothing is raised and nothing is silently clipped.

Cost: one bf16x3 forward plus one forward per rung tried, without gradients (~40 ms at B = 32) -- the trainer runs it on the first
batch and every `interval` steps (a policy that was fine at initialisation can stop being fine as the weights train), bench.py once
before the timed steps.  The reference path (fp32 everywhere, model/model.py:100-143) needs no such thing; this is the price of spending
the parity margin."""
import logging

import torch

from . import ops

log = logging.getLogger("egovlp_amd.guard")


def ladder(depth):
    """The rungs from the shipped policy down to the fp32-grade formats: `f16_single` settings, then the all-bf16x3 forward."""
    a = ops.single_product_policy(depth)
    q1, p1 = a["qkv"], a["proj"]
    rungs = ["auto",
             {"fc2": min(depth, 2 * q1), "fc1": min(depth, 2 * q1), "qkv": min(depth, 2 * q1), "proj": min(depth, (3 * depth + 3) // 4)},
             {"fc2": min(depth, 3 * q1), "fc1": min(depth, 3 * q1), "qkv": min(depth, 3 * q1)},
             "none", "bf16x3"]
    return rungs


class PrecisionGuard:
    def __init__(self, model, budget=6e-4, interval=1000):
        self.model = getattr(model, "module", model)
        self.budget, self.interval = float(budget), int(interval)
        self.rung = None                   # index into ladder(depth), set at the first check from the policy in force; never climbs back by itself
        self._start_rung = 0
        self.history = []                  # one report per check
        self._steps = 0

    def active(self):
        """Is there anything to guard?  (an fp16-product forward: single-product Linears, or -- 'f16x2' -- at least fp16's range)"""
        return self.model.exec_ctx.fwd_passes == 2

    def maybe_check(self, data):
        """Called once per training step: measures on the first call and every `interval` calls."""
        self._steps += 1
        if self.active() and (self._steps == 1 or (self.interval > 0 and self._steps % self.interval == 0)):
            return self.check(data)
        return None

    @torch.no_grad()
    def _video(self, video):
        self.model.exec_ctx.begin_step()
        return self.model.compute_video(video).float()

    def _apply(self, rung_spec, bwd):
        ec = self.model.exec_ctx
        if rung_spec == "bf16x3":
            ec.set_precision("bf16x3", "bf16")
        elif rung_spec == "none":
            ec.set_precision("f16x2", bwd)
        else:
            ec.set_precision("f16mix", bwd, f16_single=rung_spec)

    def check(self, data):
        """Measure the current policy on `data['video']` and demote until it is inside the budget.  -> report dict (also appended to
        .history): rungs tried with their errors, the policy in force afterwards."""
        ec = self.model.exec_ctx
        if not self.active():
            return None
        video = data["video"] if isinstance(data, dict) else data
        depth = len(self.model.video_model.blocks)
        rungs = ladder(depth)
        if self.rung is None:              # start where the caller's policy is: 'none' = the f16x2 rung, anything else = the top
            self.rung = self._start_rung = 3 if ec.get("f16_single") in (None, "none", "") else 0
            if self.rung == 0 and ec.get("f16_single") != "auto":
                rungs[0] = ec.get("f16_single")            # a custom policy is measured as it is (and demoted along the same ladder)
        bwd = ec.precision_name()[1]
        was_training = self.model.training
        self.model.eval()
        tried = []
        try:
            ec.set_precision("bf16x3", "bf16x3")
            ref = self._video(video)
            ref_norm = ref.double().norm()
            while True:
                spec = rungs[self.rung]
                self._apply(spec, bwd)
                if spec == "bf16x3":
                    tried.append({"policy": "bf16x3", "err": 0.0, "worst_row": 0.0, "finite": True})
                    break
                v = self._video(video)
                finite = bool(torch.isfinite(v).all())
                err = float((v.double() - ref.double()).norm() / ref_norm) if finite else float("inf")
                rows = ((v.double() - ref.double()).norm(dim=1) / ref.double().norm(dim=1)) if finite else None
                worst = float(rows.max()) if finite else float("inf")
                tried.append({"policy": spec if isinstance(spec, str) else dict(spec), "err": err, "worst_row": worst, "finite": finite})
                if finite and err <= self.budget:
                    break
                log.warning("precision guard: policy %s measures %.2e on the video embedding against the bf16x3 forward of the same batch "
                            "(budget %.1e, worst row %.2e%s): demoting to %s", spec, err, self.budget, worst,
                            "" if finite else ", NON-FINITE", rungs[self.rung + 1])
                self.rung += 1
        finally:
            self.model.train(was_training)
            # the measurement built operand planes of formats the training step does not use (split-bf16 planes of every video weight for
            # the reference forward, planes of the rungs tried): a cache entry refreshes everything it owns after every optimizer step, so
            # drop them all -- the next forward rebuilds exactly what the policy in force needs
            ec.join_side_stream()
            ec.wc.clear()
        # data-parallel ranks keep the same kernels: everyone takes the most conservative rung any rank chose
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            t = torch.tensor([self.rung], device=video.device, dtype=torch.int32)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            if int(t) != self.rung:
                self.rung = int(t)
                self._apply(rungs[self.rung], bwd)
        pol = rungs[self.rung]
        report = {"budget": self.budget, "tried": tried, "policy": dict(pol) if isinstance(pol, dict) else pol,
                  "precision": "/".join(ec.precision_name()), "demoted": self.rung > self._start_rung}
        self.history.append(report)
        return report
