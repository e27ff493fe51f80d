"""The C block calls (egv_block_fwd / egv_block_bwd: one C-ABI call per SpaceTimeBlock direction, csrc/block.hip) against the
per-kernel host path (_SpaceTimeBlockFn: one C-ABI call per kernel) on the SAME block, input and upstream gradient.

Both paths launch the same kernels with the same arguments in the same order on the main stream, so everything that is a pure
function of its inputs must come out BIT FOR BIT: the block output, the input gradient, every weight and bias gradient (TN GEMM,
un-split).  The LayerNorm affine gradients are sums of fp32 atomics (order varies from run to run of the SAME path): those are
compared at 1e-5 of their norm.  The reference has no counterpart (model/video_transformer.py:140-178 is one nn.Module forward);
what is pinned against the reference is the whole model (tests/test_gpu_model.py), which runs through this path by default."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _block(D=768, H=12, seed=0):
    from functools import partial
    from torch import nn
    from egovlp_amd.model.video_transformer import SpaceTimeBlock
    torch.manual_seed(seed)
    blk = SpaceTimeBlock(dim=D, num_heads=H, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), time_init='rand')
    with torch.no_grad():
        for name, p in blk.named_parameters():            # affine / biases away from their 1 / 0 initial values
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    return blk.cuda().train()


def _run(blk, ec, x, g, B, T, n, block_calls, side):
    from egovlp_amd.model import video_transformer as vt
    ec.set(block_calls=block_calls, wgrad_side_stream=side)
    for p in blk.parameters():
        p.grad = None
    xin = x.clone().requires_grad_(True)
    ec.begin_step()
    used = {"c": 0, "k": 0}
    y = blk(xin, B, T, n, ec)
    used["c" if isinstance(y.grad_fn, vt._SpaceTimeBlockCFn._backward_cls) else "k"] += 1
    y.backward(g)
    ec.join_side_stream()
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().clone() for k, p in blk.named_parameters()}
    return y.detach().clone(), xin.grad.detach().clone(), grads, used


@pytest.mark.parametrize("mode", [("bf16x3", "bf16x3"), ("bf16x3", "bf16"), ("bf16", "bf16")])
@pytest.mark.parametrize("side", [False, True])
@pytest.mark.parametrize("geom", [(4, 4, 196), (2, 16, 196)])
def test_block_calls_equal_the_per_kernel_path(mode, side, geom):
    from egovlp_amd import ops
    B, T, n = geom
    D = 768
    blk = _block(D)
    ec = ops.new_context()
    ec.set_precision(*mode)
    M = B * (1 + T * n)
    torch.manual_seed(5)
    x = torch.randn(B, 1 + T * n, D, device="cuda")
    g = torch.randn(B, 1 + T * n, D, device="cuda") * 0.1
    y_c, dx_c, gr_c, used_c = _run(blk, ec, x, g, B, T, n, True, side)
    y_k, dx_k, gr_k, used_k = _run(blk, ec, x, g, B, T, n, False, side)
    assert used_c == {"c": 1, "k": 0} and used_k == {"c": 0, "k": 1}, (used_c, used_k, M)
    assert torch.equal(y_c, y_k)
    assert torch.equal(dx_c, dx_k)
    for k in gr_k:
        a, b = gr_c[k], gr_k[k]
        assert a.shape == b.shape and a.is_contiguous()
        if "norm" in k:          # sums of fp32 atomics
            assert float((a.double() - b.double()).norm() / b.double().norm()) < 1e-5, k
        else:
            assert torch.equal(a, b), k


def test_block_calls_are_the_default_and_leave_no_copies():
    """The whole video tower steps through the block calls by default; the parameter gradients it hands to autograd are views of one
    buffer per block (stolen by AccumulateGrad, no copy): their storages coincide."""
    from egovlp_amd import ops
    B, T, n, D = 4, 4, 196, 768
    blk = _block(D)
    ec = ops.new_context()
    ec.set_precision("bf16x3", "bf16")
    assert ec.block_calls
    x = torch.randn(B, 1 + T * n, D, device="cuda", requires_grad=True)
    y = blk(x, B, T, n, ec)
    from egovlp_amd.model import video_transformer as vt
    assert isinstance(y.grad_fn, vt._SpaceTimeBlockCFn._backward_cls)
    y.backward(torch.randn_like(y))
    torch.cuda.synchronize()
    ptrs = {p.grad.untyped_storage().data_ptr() for p in blk.parameters()}
    assert len(ptrs) == 1, len(ptrs)
