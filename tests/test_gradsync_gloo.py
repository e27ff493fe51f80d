"""`Bf16GradSync` (the bf16 bucketed gradient all-reduce that replaces DDP, egovlp_amd/dist.py) on two gloo ranks.

The HIP pack / unpack kernels cannot run here, so the test injects torch restatements of them (scale by 1/W, round to bf16,
write at the bucket offsets; and back) -- what is exercised is the HOST logic the GPU path shares: bucket layout from the
observed ready order, hook accounting, one async all-reduce per bucket, unpack into p.grad, the first (bucket-building) step
and the hook-driven steps after it.  Gradients are chosen exactly representable, so (1/W) * sum_r grad_r must come back
BIT FOR BIT; a second case with random gradients checks the bf16 rounding model bf16(bf16(g0/2) + bf16(g1/2))."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _pack(grads, flat, offsets, scale):
    for g, o in zip(grads, offsets):
        flat[o:o + g.numel()] = (g.reshape(-1) * scale).to(torch.bfloat16)


def _unpack(grads, flat, offsets):
    for g, o in zip(grads, offsets):
        g.copy_(flat[o:o + g.numel()].float().view_as(g))


def _slice_sum(recv, world, slice_elems, out):
    out.copy_(recv.view(world, slice_elems).float().sum(0).to(torch.bfloat16))       # fp32 accumulation, one rounding


def _worker(rank, world, port, out, use_hooks=True, exchange="allreduce"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from egovlp_amd.dist import Bf16GradSync
    torch.manual_seed(100 + rank)                     # different initial weights per rank: the broadcast must fix that
    net = torch.nn.Sequential(torch.nn.Linear(24, 40), torch.nn.ReLU(), torch.nn.Linear(40, 33), torch.nn.Linear(33, 7))
    if use_hooks:
        sync = Bf16GradSync(net.parameters(), bucket_mb=0.004, pack_fn=_pack, unpack_fn=_unpack, exchange=exchange,
                            slice_sum_fn=_slice_sum)     # ~2 k elements per bucket
    else:
        # hook-free mode: buckets cut along the given ready order (last layer first); poll() is called when the gradient of the
        # FIRST layer's weight is being produced -- the gradients of the later layers are final by then -- and again by finish()
        order = [p for p in net.parameters()][::-1]
        sync = Bf16GradSync(net.parameters(), bucket_mb=0.004, pack_fn=_pack, unpack_fn=_unpack, use_hooks=False, order_hint=order,
                            exchange=exchange, slice_sum_fn=_slice_sum)
        polled = []

        def _poll(g):
            before = sync.stats["collectives_last_step"]
            sync.poll()
            polled.append(sync.stats["collectives_last_step"] - before)
            return g
        net[0].weight.register_hook(_poll)
    w0 = [p.detach().clone() for p in net.parameters()]
    results = []
    for step in range(3):
        for p in net.parameters():
            p.grad = None
        g = torch.Generator().manual_seed(1000 * step + rank)
        exact = step < 2
        # a loss whose gradient w.r.t. every parameter is a prescribed tensor: sum(p * G_p)
        Gs = []
        for p in net.parameters():
            if exact:
                G = torch.randint(-64, 64, p.shape, generator=g).float() / 8.0      # multiples of 1/8 below 8: exact in bf16, also halved and summed
            else:
                G = torch.randn(p.shape, generator=g)
            Gs.append(G)
        loss = sum((p * G).sum() for p, G in zip(net.parameters(), Gs))
        loss.backward()
        stats = sync.finish()
        results.append(([p.grad.clone() for p in net.parameters()], Gs, stats))
    if not use_hooks:
        assert polled and all(n >= 1 for n in polled), polled      # the mid-backward poll did launch buckets in every step
    torch.save({"w0": w0, "results": results}, os.path.join(out, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("use_hooks,exchange,world", [(True, "allreduce", 2), (False, "allreduce", 2), (True, "direct", 2),
                                                      (False, "direct", 2), (False, "direct", 3)])
def test_bf16_grad_sync_ranks(tmp_path, use_hooks, exchange, world):
    """Both exchanges (RCCL-style all-reduce of the bf16 bucket; the direct all-to-all -> fp32 slice sum -> all-gather), hook and
    hook-free launch, two ranks -- and three for the direct exchange (bucket length padded to 3 equal 16-byte-aligned slices)."""
    port = 29641 + (0 if use_hooks else 1) + (2 if exchange == "direct" else 0) + 4 * (world - 2)
    mp.spawn(_worker, args=(world, port, str(tmp_path), use_hooks, exchange), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), f"rank{i}.pt"), weights_only=False) for i in range(world)]
    for other in r[1:]:
        for a, b in zip(r[0]["w0"], other["w0"]):
            assert torch.equal(a, b)                  # rank 0's initial weights everywhere
    for step in range(3):
        gs = [r[k]["results"][step][0] for k in range(world)]
        Gs = [r[k]["results"][step][1] for k in range(world)]
        st0 = r[0]["results"][step][2]
        assert st0["buckets"] >= 2 and st0["collectives_last_step"] == st0["buckets"]
        for i in range(len(gs[0])):
            for k in range(1, world):
                assert torch.equal(gs[0][i], gs[k][i])          # every rank ends with the same gradient
            if step < 2 and world == 2:
                assert torch.equal(gs[0][i], (Gs[0][i] + Gs[1][i]) / world), (step, i)        # (1/W) * sum, bit for bit
            elif exchange == "direct":
                # fp32 accumulation of the pre-scaled bf16 values, ONE rounding at the end
                want = sum((G[i] / world).to(torch.bfloat16).float() for G in Gs).to(torch.bfloat16).float()
                assert torch.equal(gs[0][i], want), (step, i)
            else:
                want = ((Gs[0][i] / world).to(torch.bfloat16) + (Gs[1][i] / world).to(torch.bfloat16)).float()
                assert torch.equal(gs[0][i], want), (step, i)
