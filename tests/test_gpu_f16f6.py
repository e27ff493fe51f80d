"""The f16f6 forward format on a real MI355X (`pytest -m gpu`), piece by piece through the C ABI: the encoders against the host
restatement BIT FOR BIT (tests/f16f6_ref.py), the LayerNorm and GELU-epilogue producers against fp64 math, and the fp16 + MXFP6
product of egv_gemm_nt(passes = 2) against (a) the exact product of the encoded operands and (b) the fp32 product it stands for.
End-to-end parity of the mode (embeddings / loss vs the reference goldens) is in tests/test_gpu_model.py."""
import pytest
import torch
import torch.nn.functional as F

import f16f6_ref as R

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def ops():
    from egovlp_amd import ops as _ops
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return _ops


def _inputs(rows, cols, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(rows, cols, generator=g) * scale
    x *= torch.logspace(-2, 1, rows).unsqueeze(1)[torch.randperm(rows, generator=g)]        # rows of very different magnitude
    return x


def _check_planes(pl, x):
    """device planes == host encoding of x, bit for bit (codes, scale bytes, fp16 and bf16 planes)."""
    h16, bf, slots = R.encode(x)
    rows, cols = x.shape
    assert torch.equal(pl.hi.cpu().view(torch.int16), h16.view(torch.int16))
    if pl.bf is not None:
        assert torch.equal(pl.bf.cpu().view(torch.int16), bf.view(torch.int16))
    got = R.unpack_slots(pl.lo.cpu(), rows, cols)
    assert torch.equal(got[..., :25], slots[..., :25])


def test_encode_is_bit_exact(ops):
    for rows, cols, seed in ((64, 256, 1), (333, 96, 2), (1000, 768, 3), (17, 32, 4)):
        x = _inputs(rows, cols, seed)
        if rows > 20:   # the corners: zeros, a saturating value, fp16 subnormals, exact block maxima at the scale boundary, ties
            x[0] = 0.0
            x[1, 5] = 1.0e5
            x[2] = x[2] * 1e-7
            x[3, :32] = 7.5
            x[4, :32] = torch.arange(32) * 0.0625
            x[5, :64] = -60.0
        pl = ops.f16f6_encode(x.cuda(), want_bf=True)
        torch.cuda.synchronize()
        _check_planes(pl, x)


def test_encode_multi_is_the_same_encoding(ops):
    xs = [_inputs(768, 768, 11, 0.05), _inputs(2304, 768, 12, 0.05), _inputs(96, 3072, 13, 0.02)]
    pls = [ops.empty_planes_f16f6(x.shape[0], x.shape[1], "cuda") for x in xs]
    dev = [x.cuda() for x in xs]
    ops.f16f6_encode_multi([(d, p.hi.data_ptr(), p.lo.data_ptr(), p.ld) for d, p in zip(dev, pls)])
    torch.cuda.synchronize()
    for x, p in zip(xs, pls):
        _check_planes(p, x)


@pytest.mark.parametrize("cols", [768, 1024, 64])
def test_layernorm_writes_the_format(ops, cols):
    rows = 777
    g = torch.Generator().manual_seed(cols)
    x = _inputs(rows, cols, 20 + cols)
    gamma, beta = 1.0 + 0.1 * torch.randn(cols, generator=g), 0.05 * torch.randn(cols, generator=g)
    pl, _, mean, rstd, _ = ops.layernorm_fwd(x.cuda(), gamma.cuda(), beta.cuda(), 1e-6, 2, want_bf=True)
    ref = F.layer_norm(x.double(), (cols,), gamma.double(), beta.double(), 1e-6)
    c6, l6 = R.decode_slots(R.unpack_slots(pl.lo.cpu(), rows, cols))
    assert rel(mean, x.double().mean(1)) < 1e-5 and rel(rstd, 1.0 / torch.sqrt(x.double().var(1, unbiased=False) + 1e-6)) < 1e-5
    assert rel(pl.hi.cpu().double() + l6, ref) < 2e-5            # fp16 + its MXFP6 residual: ~2^-16
    assert rel(pl.hi.cpu(), ref) < 4e-4 and rel(pl.bf.cpu(), ref) < 4e-3 and rel(c6, ref) < 5e-2
    # and the bytes are the encoding of what the kernel normalised: re-encode the fp32 LayerNorm of the same device
    _, yf, _, _, _ = ops.layernorm_fwd(x.cuda(), gamma.cuda(), beta.cuda(), 1e-6, 1, want_f32=True, want_planes=False)
    h16, _, slots = R.encode(yf.cpu())
    same = (pl.hi.cpu().view(torch.int16) == h16.view(torch.int16)).float().mean()
    assert float(same) > 0.99                                      # summation order of the statistics differs in the last ulp


def _f16f6(ops, x, want_bf=False):
    return ops.f16f6_encode(x.cuda().contiguous(), want_bf=want_bf)


@pytest.mark.parametrize("M,N,K", [(4200, 2304, 768), (3140, 768, 3072), (785, 256, 64), (25120, 768, 128)])
def test_gemm_f16f6_linear(ops, M, N, K):
    a, w = _inputs(M, K, 31), _inputs(N, K, 32, 0.03)
    g = torch.Generator().manual_seed(33)
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    pa, pw = _f16f6(ops, a), _f16f6(ops, w)
    exact = R.product(pa.hi.cpu(), R.unpack_slots(pa.lo.cpu(), M, K), pw.hi.cpu(), R.unpack_slots(pw.lo.cpu(), N, K))
    true = a.double() @ w.double().t()
    # bias + residual -> fp32 (proj / fc2 form)
    out = torch.empty(M, N, device="cuda")
    ops.gemm_nt(pa, pw, passes=2, bias=bias.cuda(), residual=res.cuda(), out_f32=out)
    got = out.cpu().double() - bias.double() - res.double()
    r_exact, r_true = rel(got, exact), rel(got, true)
    print("f16f6 gemm M=%d N=%d K=%d: vs exact product of the encoded operands %.2e, vs fp32 product %.2e" % (M, N, K, r_exact, r_true))
    assert r_exact < 4e-6 and r_true < 6e-5
    # bias -> split-bf16 planes (qkv form)
    pl = ops.empty_planes(M, N, 3, "cuda")
    ops.gemm_nt(pa, pw, passes=2, bias=bias.cuda(), out_planes=pl)
    assert rel(pl.float().cpu().double() - bias.double(), exact) < 2e-5


def test_gemm_f16f6_mlp_with_gelu_handover(ops):
    """fc1 -> GELU -> fc2 as the f16f6 mode runs it: the activation leaves the fc1 epilogue in the operand format (+ bf16 plane,
    + saved gelu' as bf16) and is consumed by fc2 directly."""
    M, D, Hd = 3140, 768, 3072
    x, w1, w2 = _inputs(M, D, 41), _inputs(Hd, D, 42, 0.03), _inputs(D, Hd, 43, 0.02)
    g = torch.Generator().manual_seed(44)
    b1, b2 = 0.1 * torch.randn(Hd, generator=g), 0.1 * torch.randn(D, generator=g)
    px, pw1, pw2 = _f16f6(ops, x), _f16f6(ops, w1), _f16f6(ops, w2)
    h = ops.empty_planes_f16f6(M, Hd, "cuda", want_bf=True)
    z = torch.empty(M, Hd, dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt(px, pw1, passes=2, bias=b1.cuda(), act=ops.ACT_GELU, aux_out=z, out_planes=h, aux_is_grad=True)
    pre = x.double() @ w1.double().t() + b1.double()
    act = F.gelu(pre)
    c6, l6 = R.decode_slots(R.unpack_slots(h.lo.cpu(), M, Hd))
    r_h = rel(h.hi.cpu().double() + l6, act)
    cdf = 0.5 * (1.0 + torch.erf(pre / 2 ** 0.5))
    dgelu = cdf + pre * torch.exp(-0.5 * pre * pre) / (2 * torch.pi) ** 0.5
    print("f16f6 gelu hand-over: h16 + l6 vs gelu %.2e, c6 %.2e, bf %.2e, saved gelu' %.2e" % (
        r_h, rel(c6, act), rel(h.bf.cpu(), act), rel(z.cpu(), dgelu)))
    assert r_h < 6e-5 and rel(c6, act) < 5e-2 and rel(h.bf.cpu(), act) < 4e-3 and rel(z.cpu(), dgelu) < 4e-3
    out = torch.empty(M, D, device="cuda")
    ops.gemm_nt(h, pw2, passes=2, bias=b2.cuda(), out_f32=out)
    ref = act @ w2.double().t() + b2.double()
    print("f16f6 mlp: vs fp64 %.2e" % rel(out, ref))
    assert rel(out, ref) < 1e-4


def test_gemm_f16f6_rejects_what_it_cannot_run(ops):
    from egovlp_amd._lib import EgovlpHipError
    a, w = _f16f6(ops, _inputs(512, 256, 51)), _f16f6(ops, _inputs(512, 256, 52))
    out = torch.empty(512, 512, device="cuda")
    with pytest.raises(ValueError):
        ops.gemm_nt(a, ops.split_f32(_inputs(512, 256, 53).cuda(), 3)[0], passes=2, out_f32=out)      # mixed operand formats
    small = _f16f6(ops, _inputs(128, 256, 54))
    with pytest.raises(EgovlpHipError):
        ops.gemm_nt(small, w, passes=2, out_f32=torch.empty(128, 512, device="cuda"))                  # below one big tile
    with pytest.raises(EgovlpHipError):
        ops.gemm_nt(a, w, passes=2, out_f32=out, ksplit=2)                                             # no split-K form
