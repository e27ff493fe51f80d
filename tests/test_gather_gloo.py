"""Multi-rank semantics of the gathered contrastive step on CPU (gloo, world_size 2): the collective
plumbing (AllGather_multi / AllGatherFused) must reproduce the reference's behaviour recorded in
tests/golden/gather_w2.npz (reference AllGather_multi under gloo): rank-major order, identical loss on
both ranks, local-slice backward.  The loss math itself runs on the oracle here (no GPU in this tier)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, fused, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import types
    from egovlp_amd.trainer.trainer_egoclip import AllGather_multi, AllGatherFused
    from oracle import egovlp_oracle as O
    g = np.load(os.path.join(ROOT, "tests", "golden", "gather_w2.npz"))
    v = torch.from_numpy(g[f"v{rank}"]).requires_grad_(True)
    t = torch.from_numpy(g[f"t{rank}"]).requires_grad_(True)
    noun, verb = torch.from_numpy(g[f"noun{rank}"]), torch.from_numpy(g[f"verb{rank}"])
    if fused:
        va, ta, na, ba = AllGatherFused.apply(v, t, noun, verb, world, rank)
    else:
        args = types.SimpleNamespace(world_size=world, rank=rank)
        va, ta = AllGather_multi.apply(v, world, args), AllGather_multi.apply(t, world, args)
        na, ba = AllGather_multi.apply(noun, world, args), AllGather_multi.apply(verb, world, args)
    loss, _ = O.egoclip_loss(ta, va, na, ba)
    loss.backward()
    q.put((rank, float(loss), v.grad.numpy(), t.grad.numpy(), va.detach().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fused", [False, True])
def test_gather_world2_matches_reference(fused):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29620 + int(fused)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, fused, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda x: x[0])
    [p.join(60) for p in procs]
    g = np.load(os.path.join(ROOT, "tests", "golden", "gather_w2.npz"))
    for rank, loss, gv, gt, va in res:
        assert loss == pytest.approx(float(g["loss0"]), rel=1e-5)
        np.testing.assert_allclose(gv, g[f"gv{rank}"], rtol=2e-4, atol=1e-7)
        np.testing.assert_allclose(gt, g[f"gt{rank}"], rtol=2e-4, atol=1e-7)
        np.testing.assert_array_equal(va, np.concatenate([g["v0"], g["v1"]]))   # rank-major order


def test_world1_gather_is_identity():
    from egovlp_amd.trainer.trainer_egoclip import AllGatherFused
    a, b = torch.randn(3, 4, requires_grad=True), torch.randn(3, 4, requires_grad=True)
    n, v = torch.ones(3, 5), torch.ones(3, 2)
    va, ta, na, ba = AllGatherFused.apply(a, b, n, v, 1, 0)
    (va.sum() + 2 * ta.sum()).backward()
    assert torch.equal(a.grad, torch.ones(3, 4)) and torch.equal(b.grad, 2 * torch.ones(3, 4))
