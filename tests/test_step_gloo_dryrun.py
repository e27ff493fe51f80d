"""The WHOLE data-parallel step on two (and three) real ranks, on CPU: `egoclip_step` of the real model (host code, autograd functions,
execution context) over the do-nothing C-ABI stand-in (tests/mock_hip.py), with a real gloo process group underneath -- the
embedding all-gather of forward, the hook-free bucket launches from the polls inside backward, the direct exchange
(all-to-all of slices -> fp32 slice sum -> all-gather), finish(), the optimizer call.  It is the closest this GPU-less container
gets to `bench.py --gpus 2`: what it pins is that the two ranks issue the SAME collectives in the SAME order (a mismatch hangs:
the test has a timeout), that buckets leave during backward on both ranks, that rank 0's initial weights arrive everywhere and
that every rank ends the step with bit-identical gradients.  Kernels compute nothing here (gradients are whatever torch.empty
left in them), so nothing is said about values -- tests/test_gradsync_gloo.py pins the arithmetic of the exchange, the -m gpu
tests the kernels."""
import hashlib
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _pack(grads, flat, offsets, scale):
    for g, o in zip(grads, offsets):
        flat[o:o + g.numel()] = (g.reshape(-1) * scale).to(torch.bfloat16)


def _unpack(grads, flat, offsets):
    for g, o in zip(grads, offsets):
        g.copy_(flat[o:o + g.numel()].float().view_as(g))


def _slice_sum(recv, world, slice_elems, out):
    out.copy_(recv.view(world, slice_elems).float().sum(0).to(torch.bfloat16))


def _digest(tensors):
    h = hashlib.sha1()
    for t in tensors:
        h.update(t.detach().contiguous().view(torch.uint8).numpy().tobytes())
    return h.hexdigest()


def _worker(rank, world, port, out, exchange):
    for p in (HERE, os.path.dirname(HERE)):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mock_hip import mock_hip
    from egovlp_amd.dist import Bf16GradSync
    from egovlp_amd.model.loss import EgoNCE
    from egovlp_amd.model.model import FrozenInTime
    from egovlp_amd.optim import AdamW
    from egovlp_amd.synth import synth_batch
    from egovlp_amd.trainer.trainer_egoclip import egoclip_step
    torch.manual_seed(100 + rank)                         # different initial weights per rank: the broadcast must fix that
    model = FrozenInTime(video_params={"model": "SpaceTimeTransformer", "arch_config": "base_patch16_224", "num_frames": 4,
                                       "pretrained": True, "time_init": "rand"},
                         text_params={"model": "distilbert-base-uncased", "pretrained": True, "input": "text"},
                         projection="minimal", load_checkpoint="").train()
    model.text_model.seed_rank = rank
    ec = model.exec_ctx
    sync = Bf16GradSync(model.parameters(), use_hooks=False, order_hint=model.gradient_ready_order(), exec_ctx=ec,
                        exchange=exchange, pack_fn=_pack, unpack_fn=_unpack, slice_sum_fn=_slice_sum)
    w0 = _digest(p for p in model.parameters())
    ec.set(backward_poll=sync.poll, gemm_grid=248)
    ec.set_precision("bf16x3", "bf16")
    opt = AdamW(model.parameters(), lr=3e-5)
    b = synth_batch(2, T=2, L=16, seed=3, rank=rank)
    data = {"video": b["video"], "text": b["text"], "noun_vec": b["noun_vec"], "verb_vec": b["verb_vec"]}
    steps = []
    with mock_hip() as calls:
        for step in range(2):
            calls.clear()
            # gradients come out of torch.empty (the kernels are stand-ins): give every rank DIFFERENT, finite ones, so that
            # "identical after the exchange" means something -- the wgrad buffers are allocated by the host code, fill them
            # right before the exchange reads them (the pack stand-in sees p.grad)
            def pack(grads, flat, offsets, scale, _step=step):
                for i, g in enumerate(grads):
                    g.copy_(torch.full_like(g, float(rank + 1) + 0.25 * _step + (i % 7)))
                _pack(grads, flat, offsets, scale)
            sync.pack_fn = pack
            egoclip_step(model, EgoNCE(), opt, data, world, rank, grad_sync=sync)
            steps.append({"during": sync.stats["launched_during_backward"], "buckets": sync.stats["buckets"],
                          "grads": _digest(p.grad for p in model.parameters()),
                          "first": float(next(model.parameters()).grad.reshape(-1)[0]),
                          "gemm_calls": calls.count("egv_gemm_nt"), "adamw": calls.count("egv_adamw_multi")})
    torch.save({"w0": w0, "steps": steps}, os.path.join(out, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("exchange,world", [("direct", 2), ("allreduce", 2), ("direct", 3)])
def test_multi_rank_step_on_the_real_model(tmp_path, exchange, world):
    port = 29671 + (exchange == "direct") + 2 * (world - 2)
    mp.spawn(_worker, args=(world, port, str(tmp_path), exchange), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), f"rank{i}.pt"), weights_only=False) for i in range(world)]
    assert all(x["w0"] == r[0]["w0"] for x in r)                      # rank 0's initial weights everywhere
    for step in range(2):
        a = r[0]["steps"][step]
        for x in r[1:]:
            b = x["steps"][step]
            assert a["buckets"] == b["buckets"] >= 5
            assert a["grads"] == b["grads"], step                     # bit-identical gradients on every rank after finish()
        if world == 2:
            # the mean of the two ranks' fills of a parameter: ((1 + s/4 + i%7) + (2 + s/4 + i%7)) / 2 -- exact in bf16
            assert abs(a["first"] - round(a["first"] * 4) / 4) < 1e-6 and abs((a["first"] % 1.0) - (0.5 + 0.25 * step) % 1.0) < 1e-6, a["first"]
        for x in (y["steps"][step] for y in r):
            assert x["during"] >= x["buckets"] - 1, x                 # only the tail bucket may be left to finish()
            assert x["gemm_calls"] == 12 * 18 + 2 + 6 * 12 + 2 * 3 and x["adamw"] >= 1
