"""Where does the un-profiled step spend its time?  HIP events at the phase boundaries of egoclip_step (forward / loss / main-stream
backward / wgrad-stream tail / optimizer), averaged over steps -- rocprofv3 slows the host enough to make the traced step host-bound
(its timeline shows idle gaps the real run does not have), so this is the honest split.   python tools/step_phases.py [steps]"""
import os
import sys
import torch

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                        # noqa: E402
from egovlp_amd.model.loss import EgoNCE                            # noqa: E402
from egovlp_amd.optim import AdamW                                  # noqa: E402
from egovlp_amd.synth import synth_batch                            # noqa: E402
from egovlp_amd.trainer.trainer_egoclip import AllGatherFused       # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
model = bench.build_model("base_patch16_224", 16).cuda().train()
ec = model.exec_ctx
ec.set_precision(os.environ.get("EGOVLP_PRECISION", "f16mix"))
ec.set(wgrad_side_stream=True, text_side_stream=True)
opt = AdamW(model.parameters(), lr=3e-5)
loss_fn = EgoNCE()
b = synth_batch(32, T=4, L=32, seed=1234)
data = {"video": b["video"].cuda(), "text": {k: v.cuda() for k, v in b["text"].items()}, "noun_vec": b["noun_vec"].cuda(), "verb_vec": b["verb_vec"].cuda()}
main = torch.cuda.current_stream()
E = lambda: torch.cuda.Event(enable_timing=True)
recs = []


def step(timed):
    ev = {k: E() for k in ("t0", "fwd", "loss", "bwd_main", "side_end", "joined", "opt")}
    ev["t0"].record(main)
    opt.zero_grad(set_to_none=True)
    te, ve = model(data)
    ev["fwd"].record(main)
    ve, te, n_, v_ = AllGatherFused.apply(ve, te, data["noun_vec"], data["verb_vec"], 1, 0)
    loss = loss_fn.fused(te, ve, n_, v_)
    ev["loss"].record(main)
    # the end-of-backward engine callback joins the side streams: take the positions of both streams BEFORE it runs -- a hook on
    # the first leaf's gradient is too early, so record from a callback queued ahead of the join instead
    sd = ec._side
    state = {}

    def before_join():
        ev["bwd_main"].record(main)
        if sd["stream"] is not None:
            ev["side_end"].record(sd["stream"])
            state["side"] = True
    loss.backward()
    # (the join already happened inside backward(): the side stream's last event is what the main stream waited for)
    ev["joined"].record(main)
    opt.step()
    ev["opt"].record(main)
    if timed:
        recs.append(ev)


for _ in range(5):
    step(False)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(steps):
    step(True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps * 1e3
acc = {}
for ev in recs:
    for a, c, name in (("t0", "fwd", "forward (both towers, main stream position)"), ("fwd", "loss", "gather + EgoNCE"), ("loss", "joined", "backward incl. the wait for the wgrad stream"),
                       ("joined", "opt", "AdamW (+ nothing else)")):
        acc[name] = acc.get(name, 0.0) + ev[a].elapsed_time(ev[c])
    acc["event span t0 -> opt"] = acc.get("event span t0 -> opt", 0.0) + ev["t0"].elapsed_time(ev["opt"])
# step-to-step: the next step's t0 vs this step's opt (weight-plane refresh sits at the start of the next forward)
gap = sum(recs[i]["opt"].elapsed_time(recs[i + 1]["t0"]) for i in range(len(recs) - 1)) / max(len(recs) - 1, 1)
print(f"un-profiled step: {dt:.3f} ms wall per step ({32 / dt * 1e3:.1f} pairs/s), precision {ec.precision_name()}")
for k, v in acc.items():
    print(f"  {k:55s} {v / len(recs):8.3f} ms")
print(f"  {'between steps (opt event -> next t0 event)':55s} {gap:8.3f} ms")
