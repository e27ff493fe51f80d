"""Forward parity on the hostile weight distribution (egovlp_amd.synth heavy_tensor): video / text embeddings of every precision mode
against the fp32 CPU oracle, B = 4, T = 4.  python tools/heavy_check.py [gauss|heavy]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egovlp_amd.model.model import FrozenInTime          # noqa: E402
from egovlp_amd.synth import synth_batch, synth_state_dict   # noqa: E402
from oracle import egovlp_oracle as O                   # noqa: E402

dist = sys.argv[1] if len(sys.argv) > 1 else "heavy"
m = FrozenInTime(video_params={"model": "SpaceTimeTransformer", "arch_config": "base_patch16_224", "num_frames": 4, "pretrained": True,
                               "time_init": "rand"},
                 text_params={"model": "distilbert-base-uncased", "pretrained": True, "input": "text"}, projection="minimal",
                 load_checkpoint="")
sd = synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, seed=0, dist=dist)
m.load_state_dict(sd)
m = m.cuda().eval()
b = synth_batch(4, T=4, L=32, seed=99, ragged=True)
dev = {"video": b["video"].cuda(), "text": {k: v.cuda() for k, v in b["text"].items()}}
torch.set_num_threads(min(os.cpu_count() or 1, 32))
with torch.no_grad():
    rt, rv = O.frozen_in_time(b, sd, O.VideoCfg(num_frames=4), O.TextCfg())


def rel(a, c):
    return float((a.double().cpu() - c.double()).norm() / c.double().norm())


print("dist", dist, "| oracle video embedding rms %.3g, max |x| %.3g" % (float(rv.pow(2).mean().sqrt()), float(rv.abs().max())))
for mode in (("bf16x3", "bf16x3"), ("f16x2", "f16"), ("f16mix", "f16"), ("bf16", "bf16")):
    m.exec_ctx.set_precision(*mode)
    for spec in ((None,) if mode[0] != "f16mix" else (None, "fc2:6,fc1:6,qkv:6,proj:9", "fc2:9,fc1:9,qkv:9,proj:12")):
        if spec:
            m.exec_ctx.set(f16_single=spec)
        with torch.no_grad():
            te, ve = m(dev)
        rows = [rel(ve[i], rv[i]) for i in range(4)]
        print("%-8s/%-6s %-28s video %.2e (rows %s) text %.2e finite %s" % (mode[0], mode[1], spec or "", rel(ve, rv), " ".join("%.1e" % r for r in rows), rel(te, rt), bool(torch.isfinite(ve).all())))
