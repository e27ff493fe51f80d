"""Attribute the small fill / copy kernels of one EgoClip step to their Python call sites (torch.profiler, with_stack).
python tools/find_fills.py [precision]"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from egovlp_amd.model.loss import EgoNCE  # noqa: E402
from egovlp_amd.ops import Precision  # noqa: E402
from egovlp_amd.optim import AdamW  # noqa: E402
from egovlp_amd.synth import synth_batch  # noqa: E402
from egovlp_amd.trainer.trainer_egoclip import egoclip_step  # noqa: E402

Precision.set(sys.argv[1] if len(sys.argv) > 1 else "bf16")
m = bench.build_model("base_patch16_224", 4).cuda().train()
opt = AdamW(m.parameters(), lr=3e-5)
b = synth_batch(32, T=4, L=32, seed=1)
batch = {"video": b["video"].cuda(), "text": {k: v.cuda() for k, v in b["text"].items()},
         "noun_vec": b["noun_vec"].cuda(), "verb_vec": b["verb_vec"].cuda()}
for _ in range(2):
    egoclip_step(m, EgoNCE(), opt, batch)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    egoclip_step(m, EgoNCE(), opt, batch)
    torch.cuda.synchronize()
sites = collections.Counter()
for ev in prof.events():
    n = ev.name
    if n in ("aten::zeros", "aten::zero_", "aten::fill_", "aten::zeros_like", "aten::copy_", "aten::cat", "aten::contiguous",
             "aten::clone", "aten::add_", "aten::mul", "aten::add", "aten::sum", "aten::index_select", "aten::embedding"):
        st = [s for s in (ev.stack or []) if "egovlp_amd" in s or "bench.py" in s or "autograd" in s]
        sites[(n, " <- ".join(s.split("/")[-1] for s in st[:3]))] += 1
for (n, st), c in sorted(sites.items(), key=lambda kv: -kv[1])[:60]:
    print("%4d  %-18s %s" % (c, n, st))
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
