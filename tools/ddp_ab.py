"""The one kept A/B of the gradient-exchange alternatives that measured slower (rounds 2-3): torch DistributedDataParallel (fp32
buckets, find_unused_parameters as the reference, base/base_trainer.py:258) and Bf16GradSync launched from grad-ready hooks,
against the shipped hook-free Bf16GradSync.  World size 1 under RCCL (what a 1-GPU box can run):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 tools/ddp_ab.py [steps]"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model  # noqa: E402
from egovlp_amd.dist import Bf16GradSync  # noqa: E402
from egovlp_amd.model.loss import EgoNCE  # noqa: E402
from egovlp_amd.optim import AdamW  # noqa: E402
from egovlp_amd.synth import synth_batch  # noqa: E402
from egovlp_amd.trainer.trainer_egoclip import egoclip_step  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
os.environ["EGV_FORCE_GATHER"] = "1"
b = synth_batch(32, T=4, L=32, seed=1234)
data = {"video": b["video"].cuda(), "text": {k: v.cuda() for k, v in b["text"].items()}, "noun_vec": b["noun_vec"].cuda(),
        "verb_vec": b["verb_vec"].cuda()}
for mode in ("hook-free", "hooks", "ddp"):
    model = build_model("base_patch16_224", 16, 0.1).cuda().train()
    ec = model.exec_ctx
    ec.set_precision("bf16x3", "bf16")
    ec.set(gemm_grid=248, wgrad_side_stream=(mode != "ddp"), text_side_stream=True)
    net, gs = model, None
    if mode == "ddp":
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], bucket_cap_mb=100, gradient_as_bucket_view=True)
    elif mode == "hooks":
        gs = Bf16GradSync(model.parameters(), stream_of=model.gradient_stream_of, exec_ctx=ec)
    else:
        gs = Bf16GradSync(model.parameters(), use_hooks=False, order_hint=model.gradient_ready_order(), exec_ctx=ec)
        ec.set(backward_poll=gs.poll)
    opt = AdamW(model.parameters(), lr=3e-5)
    for _ in range(3):
        egoclip_step(net, EgoNCE(), opt, data, 1, 0, grad_sync=gs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        egoclip_step(net, EgoNCE(), opt, data, 1, 0, grad_sync=gs)
    torch.cuda.synchronize()
    print("%-10s %.2f ms/step" % (mode, (time.perf_counter() - t0) / steps * 1e3), flush=True)
    del model, net, opt, gs
    torch.cuda.empty_cache()
dist.destroy_process_group()
