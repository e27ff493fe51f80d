"""Host-side enqueue time of one step vs its GPU time (is the Python/ctypes layer on the critical path?)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from egovlp_amd import ops
from egovlp_amd.model.loss import EgoNCE
from egovlp_amd.optim import AdamW
from egovlp_amd.synth import synth_batch
from egovlp_amd.trainer.trainer_egoclip import egoclip_step
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
ops.Precision.set("bf16x3", "bf16") if prec == "mixed" else ops.Precision.set(prec)
model = bench.build_model("base_patch16_224", 16).cuda().train()
opt = AdamW(model.parameters(), lr=3e-5)
# the streams of the default bench.py run: text tower and weight gradients on side streams, main stream high priority
model.exec_ctx.set(wgrad_side_stream=os.environ.get("EGV_HOST_SIDE", "1") == "1", text_side_stream=os.environ.get("EGV_HOST_SIDE", "1") == "1")
torch.cuda.set_stream(torch.cuda.Stream(priority=-1))
HB, HT = int(os.environ.get("EGV_HOST_B", 32)), int(os.environ.get("EGV_HOST_T", 4))     # BASELINE config 4: 16 / 16
b = synth_batch(HB, T=HT, L=32, seed=1234)
data = {"video": b["video"].cuda(), "text": {k: v.cuda() for k, v in b["text"].items()}, "noun_vec": b["noun_vec"].cuda(), "verb_vec": b["verb_vec"].cuda()}
for _ in range(3):
    egoclip_step(model, EgoNCE(), opt, data)
torch.cuda.synchronize()
N = 8
t0 = time.perf_counter()
for _ in range(N):
    egoclip_step(model, EgoNCE(), opt, data)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{prec}: host enqueue {1e3*(t1-t0)/N:.2f} ms/step, total {1e3*(t2-t0)/N:.2f} ms/step (GPU drains {1e3*(t2-t1):.1f} ms after the last enqueue)")
# enqueue-only cost with the GPU idle-ish: time a step's enqueue after a sync
ts = []
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); egoclip_step(model, EgoNCE(), opt, data); ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
print("enqueue time of a single step from an idle stream: %.2f ms" % (1e3 * min(ts)))

# where the host time of one step goes (cProfile of three steps enqueued onto idle streams)
if os.environ.get("EGV_HOST_PROFILE", "1") == "1":
    import cProfile, pstats
    pr = cProfile.Profile()
    for _ in range(3):
        torch.cuda.synchronize()
        pr.enable(); egoclip_step(model, EgoNCE(), opt, data); pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    print("---- by internal time (3 steps)")
    st.sort_stats("tottime").print_stats(30)
    print("---- by cumulative time (3 steps)")
    st.sort_stats("cumtime").print_stats(40)
