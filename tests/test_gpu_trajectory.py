"""Does the BENCHMARKED precision mode train the same model?  (round-2 verdict, weak #1: every parity test was one step.)

The benchmarked mode ("mixed": three-product bf16x3 forward = embeddings / loss at 2.5e-5 of fp32, single-pass bf16 backward =
weight gradients 2e-3 .. 2.4e-2 off) is compared over a TRAJECTORY of optimisation steps
  (a) with the all-bf16x3 mode (fp32-grade gradients, < 1e-3 of the oracle): 20 AdamW steps at B = 8 from the same weights on
      the same batches -- loss curve, held-out loss and parameter drift (bench.py::trajectory_drift, which also prints these
      numbers in the benchmark line);
  (b) with the fp32 CPU oracle itself (torch autograd + the transformers-4.2.1 AdamW restatement), 4 steps at B = 4, both modes.
The reference trains fp32 end to end (trainer/trainer_egoclip.py:123-141)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from egovlp_amd.synth import synth_batch, synth_state_dict  # noqa: E402
from oracle import egovlp_oracle as O  # noqa: E402


def _build():
    from egovlp_amd.model.model import FrozenInTime
    m = FrozenInTime(video_params={"model": "SpaceTimeTransformer", "arch_config": "base_patch16_224", "num_frames": 16,
                                   "pretrained": True, "time_init": "rand"},
                     text_params={"model": "distilbert-base-uncased", "pretrained": True, "input": "text"},
                     projection="minimal", load_checkpoint="")
    sd = synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, seed=0)
    m.load_state_dict(sd, strict=True)
    m.text_model.set_dropout(0.0, 0.0)
    return m.cuda().train(), sd


@pytest.fixture(scope="module")
def built():
    return _build()


def test_mixed_mode_tracks_the_fp32_grade_backward_over_20_steps(built):
    import bench
    from egovlp_amd.model.loss import EgoNCE
    from egovlp_amd.optim import AdamW
    m, _ = built
    r = bench.trajectory_drift(m, EgoNCE(), lambda ps: AdamW(ps, lr=3e-5), steps=20, B=8)
    print("trajectory (20 steps, B=8): loss bf16x3 %s" % r["loss_bf16x3"])
    print("                            loss mixed  %s" % r["loss_mixed"])
    print("  max rel loss gap %.3e | held-out loss %.5f vs %.5f (rel %.3e) | parameter drift %.4f" % (
        r["max_rel_loss_gap_mixed"], r["held_out_loss_mixed"], r["held_out_loss_bf16x3"], r["held_out_rel_gap_mixed"],
        r["param_drift_mixed"]))
    for k, v in r["param_drift_per_tensor_mixed"].items():
        print("    drift %-60s %.4f" % (k, v))
    # (synthetic pairs carry nothing to generalise from: the held-out loss does not have to drop -- 3.326 at the initial weights,
    # 3.370 / 3.368 after 20 steps on MI355X -- what is asserted is that both modes end at the SAME place)
    print("  held-out loss at the initial weights %.5f" % r["held_out_loss_initial"])
    # the two loss curves agree step by step, and so do the end points on a batch neither has seen
    # (measured on MI355X, two boxes: 3.0e-3 .. 5.1e-3 / 0.7e-3 .. 1.4e-3 / 0.059 .. 0.063)
    assert r["max_rel_loss_gap_mixed"] < 2e-2, r["max_rel_loss_gap_mixed"]
    assert r["held_out_rel_gap_mixed"] < 1e-2, r["held_out_rel_gap_mixed"]
    # the end points are close in units of the distance travelled (AdamW's early updates are sign-like, see the note)
    assert r["param_drift_mixed"] < 0.2, r["param_drift_mixed"]


ORACLE_STEPS, ORACLE_B = 4, 4


def _oracle_batches():
    return [synth_batch(ORACLE_B, T=4, L=32, seed=900 + i, ragged=True) for i in range(ORACLE_STEPS)]


@pytest.fixture(scope="module")
def oracle_run(built):
    """The fp32 trajectory on the CPU: torch autograd through the oracle + its AdamW, once for both modes."""
    _, sd = built
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    theta = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    mom = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in sd.items()}
    ref_losses = []
    for i, b in enumerate(_oracle_batches()):
        for v in theta.values():
            v.grad = None
        te, ve = O.frozen_in_time(b, theta, O.VideoCfg(), O.TextCfg())
        loss, _ = O.egoclip_loss(te, ve, b["noun_vec"], b["verb_vec"])
        loss.backward()
        ref_losses.append(float(loss))
        with torch.no_grad():
            for k, v in theta.items():
                if v.grad is not None:
                    O.adamw_step(v, v.grad, mom[k][0], mom[k][1], i + 1, lr=3e-5)
    return theta, ref_losses


@pytest.mark.parametrize("mode", ["bf16x3", "mixed"])
def test_four_training_steps_against_the_fp32_oracle(built, oracle_run, mode):
    """4 x (forward, EgoNCE, backward, AdamW) at B = 4 on the device vs the same steps of the CPU oracle with torch autograd and
    the oracle's AdamW (transformers 4.2.1 semantics): the loss of EVERY step inside the parity bar in both modes (the forward
    is fp32-grade in both; what differs is how good the gradients that produced the weights of step k were)."""
    from egovlp_amd import weights
    from egovlp_amd.model.loss import EgoNCE
    from egovlp_amd.optim import AdamW
    from egovlp_amd.trainer.trainer_egoclip import egoclip_step
    m, sd = built
    steps = ORACLE_STEPS
    batches = _oracle_batches()
    theta, ref_losses = oracle_run
    # ---- device trajectory
    m.load_state_dict(sd, strict=True)
    weights.bump_epoch()
    ec = m.exec_ctx
    ec.set_precision("bf16x3", "bf16" if mode == "mixed" else "bf16x3")
    try:
        opt = AdamW(m.parameters(), lr=3e-5)
        got = []
        for b in batches:
            d = {"video": b["video"].cuda(), "text": {k: v.cuda() for k, v in b["text"].items()},
                 "noun_vec": b["noun_vec"].cuda(), "verb_vec": b["verb_vec"].cuda()}
            got.append(float(egoclip_step(m, EgoNCE(), opt, d)))
    finally:
        ec.unset("fwd_passes", "bwd_passes")
    rels = [abs(a - b) / abs(b) for a, b in zip(got, ref_losses)]
    print("%s vs oracle over %d steps: losses %s | oracle %s | rel %s" % (
        mode, steps, ["%.5f" % x for x in got], ["%.5f" % x for x in ref_losses], ["%.1e" % x for x in rels]))
    # parameters after 4 steps: distance to the oracle's, in units of the distance the oracle moved
    with torch.no_grad():
        num = sum(((p.detach().cpu().double() - theta[k].detach().double()) ** 2).sum() for k, p in m.named_parameters())
        den = sum(((theta[k].detach().double() - sd[k].double()) ** 2).sum() for k, _ in m.named_parameters())
    drift = float(num / den) ** 0.5
    print("  %s: parameter drift vs the oracle after %d steps: %.4f of the distance moved" % (mode, steps, drift))
    assert rels[0] < 1e-3                       # identical weights: the forward parity bar
    assert max(rels) < (2e-3 if mode == "bf16x3" else 1e-2), rels
    assert drift < (0.1 if mode == "bf16x3" else 0.5), drift
    m.load_state_dict(sd, strict=True)
    weights.bump_epoch()
