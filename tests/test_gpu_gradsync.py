"""The arithmetic of the direct gradient exchange (egovlp_amd/dist.py::Bf16GradSync, csrc/gradsync.hip) at W > 1 on ONE GPU.

`gpurun` boxes have one GPU, so the W-rank exchange itself cannot run; what CAN run is every kernel of it on the data the
collectives would deliver.  For W in {2, 3, 8}: W independent fp32 gradient sets (one per simulated rank) are packed by
`egv_grad_pack_bf16` with scale 1/W into W flat buckets laid out exactly as `_Bucket` lays them out (W equal slices, each a whole
number of 16-byte pieces, zero padding), slice r of every rank's bucket is concatenated rank-major -- what `all_to_all_single`
hands to rank r --, `egv_slice_sum_bf16` reduces it, the W reduced slices are concatenated -- what `all_gather_into_tensor`
returns -- and `egv_grad_unpack_bf16` writes the fp32 gradients.  The result must equal, BIT FOR BIT,

    bf16( sum_p float( bf16( g_p / W ) ) )          (fp32 accumulation in rank order, ONE rounding at the end)

the expression tests/test_gradsync_gloo.py asserts for the Python stand-ins of these kernels.  This is what replaces the fp32
bucketed mean of DistributedDataParallel at base/base_trainer.py:258 of the reference."""
import pytest
import torch

pytestmark = pytest.mark.gpu

# odd sizes on purpose: tensors that are not multiples of 4 take the scalar path of the pack kernel, the others the 16-byte path
SHAPES = [(768, 768), (2304,), (3, 5, 7), (1,), (1000, 33), (256, 768), (17,), (4096,)]


def _layout(shapes, world):
    """Offsets as _Bucket.__init__ computes them: tensors back to back, each start 8-element (16-byte) aligned; total padded to
    `world` equal slices of whole 16-byte pieces."""
    offs, off = [], 0
    for s in shapes:
        offs.append(off)
        n = 1
        for d in s:
            n *= d
        off += (n + 7) // 8 * 8
    q = 8 * world
    total = (off + q - 1) // q * q
    return offs, total


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("dist_kind", ["exact", "normal", "wide"])
def test_direct_exchange_kernels_at_world_size(world, dist_kind):
    from egovlp_amd.dist import _hip_pack, _hip_slice_sum, _hip_unpack
    dev = torch.device("cuda")
    offs, total = _layout(SHAPES, world)
    slice_elems = total // world
    assert slice_elems % 8 == 0
    g = torch.Generator().manual_seed(77 + world)
    ranks = []
    for p in range(world):
        gs = []
        for s in SHAPES:
            if dist_kind == "exact":        # multiples of 1/8 below 8: g / W and the sum are exact in bf16 for W = 2
                t = torch.randint(-64, 64, s, generator=g).float() / 8.0
            elif dist_kind == "normal":
                t = torch.randn(s, generator=g)
            else:                           # eight decades of magnitude: sums of terms of very different size
                t = torch.randn(s, generator=g) * torch.pow(10.0, torch.randint(-6, 3, s, generator=g).float())
            gs.append(t)
        ranks.append(gs)
    # --- what every simulated rank does before the all-to-all
    flats = []
    for p in range(world):
        flat = torch.zeros(total, dtype=torch.bfloat16, device=dev)
        _hip_pack([t.to(dev) for t in ranks[p]], flat, offs, 1.0 / world)
        flats.append(flat)
    torch.cuda.synchronize()
    # the pack kernel alone: bf16(g * (1/W)) at the offsets, padding untouched (zero)
    for p in range(world):
        ref = torch.zeros(total, dtype=torch.bfloat16)
        for t, o in zip(ranks[p], offs):
            ref[o:o + t.numel()] = (t.reshape(-1) * (1.0 / world)).to(torch.bfloat16)
        assert torch.equal(flats[p].cpu().view(torch.int16), ref.view(torch.int16)), ("pack", world, p)
    # --- all-to-all: rank r receives slice r of every peer, rank-major; slice sum; all-gather of the reduced slices
    red = torch.empty(total, dtype=torch.bfloat16, device=dev)
    for r in range(world):
        recv = torch.cat([flats[p][r * slice_elems:(r + 1) * slice_elems] for p in range(world)]).contiguous()
        out = torch.empty(slice_elems, dtype=torch.bfloat16, device=dev)
        _hip_slice_sum(recv, world, slice_elems, out)
        red[r * slice_elems:(r + 1) * slice_elems] = out
    # --- unpack into fp32 gradients
    outs = [torch.full(s, float("nan"), device=dev) for s in SHAPES]
    _hip_unpack(outs, red, offs)
    torch.cuda.synchronize()
    for i, s in enumerate(SHAPES):
        acc = torch.zeros(s)
        for p in range(world):                                    # fp32 accumulation in rank order, as the kernel does
            acc = acc + (ranks[p][i] * (1.0 / world)).to(torch.bfloat16).float()
        want = acc.to(torch.bfloat16).float()
        got = outs[i].cpu()
        assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (world, dist_kind, i, float((got - want).abs().max()))
        if dist_kind == "exact" and world == 2:     # (W = 8: the sum of eight multiples of 1/64 below 1 needs 9 significant bits)
            assert torch.equal(got, sum(ranks[p][i] for p in range(world)) / world)     # the true mean, bit for bit


@pytest.mark.parametrize("world", [2, 8])
def test_pack_scale_is_one_over_world_not_one(world):
    """A wrong scale (1 instead of 1/W) must be visible: the gloo / RCCL world-size-1 tests cannot see it by construction."""
    from egovlp_amd.dist import _hip_pack
    dev = torch.device("cuda")
    t = torch.full((4096,), 3.0, device=dev)
    flat = torch.zeros(4096, dtype=torch.bfloat16, device=dev)
    _hip_pack([t], flat, [0], 1.0 / world)
    torch.cuda.synchronize()
    assert torch.equal(flat.float().cpu(), torch.full((4096,), 3.0 / world))
