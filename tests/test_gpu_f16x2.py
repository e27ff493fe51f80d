"""The f16x2 forward format on a real MI355X (`pytest -m gpu`), piece by piece through the C ABI: the encoders against the host
restatement BIT FOR BIT (tests/f16x2_ref.py), the LayerNorm and GELU-epilogue producers against fp64 math, and the two-fp16-product
GEMM of egv_gemm_nt(passes = 2) against (a) the exact product of the encoded operands and (b) the fp32 product it stands for -- next
to the three-product split-bf16 GEMM on the same operands.  End-to-end parity of the mode (embeddings / loss vs the reference goldens
and the oracle) is in tests/test_gpu_model.py."""
import pytest
import torch
import torch.nn.functional as F

import f16x2_ref as R

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


def _check_planes(pl, x, role):
    """device planes == host encoding of x, bit for bit (both fp16 planes and the bf16 plane)."""
    p1, p2, bf = R.encode(x, role)
    for name, got, want in (("plane 1", pl.hi, p1), ("plane 2", pl.lo, p2)) + ((("bf16 plane", pl.bf, bf),) if pl.bf is not None else ()):
        bad = (got.cpu().view(torch.int16) != want.view(torch.int16)).nonzero()
        assert bad.numel() == 0, (name, role, bad[:4].tolist(), [float(x[tuple(i)]) for i in bad[:4]])


@pytest.mark.parametrize("role", [0, 1])
def test_encode_is_bit_exact(ops, role):
    for rows, cols, seed in ((64, 256, 1), (333, 96, 2), (1000, 768, 3), (17, 8, 4)):
        x = _inputs(rows, cols, seed)
        if rows > 20:   # the corners: zeros, a saturating value, fp16 subnormals, ties
            x[0] = 0.0
            x[1, 5] = 1.0e5
            x[2] = x[2] * 1e-7
            x[3, :8] = torch.tensor([1.0, 1.0 + 2 ** -11, 1.0 + 2 ** -10, 0.5 + 2 ** -12, -1.0, 65504.0, -65520.0, 2 ** -14])
        pl = ops.f16x2_encode(x.cuda(), role, want_bf=True)
        torch.cuda.synchronize()
        _check_planes(pl, x, role)


def test_encode_multi_is_the_weight_encoding(ops):
    xs = [_inputs(768, 768, 11, 0.05), _inputs(2304, 768, 12, 0.05), _inputs(96, 3072, 13, 0.02)]
    pls = [ops.empty_planes_f16x2(x.shape[0], x.shape[1], "cuda") for x in xs]
    dev = [x.cuda() for x in xs]
    ops.f16x2_encode_multi([(d, p.hi.data_ptr(), p.lo.data_ptr(), p.ld) for d, p in zip(dev, pls)])
    torch.cuda.synchronize()
    for x, p in zip(xs, pls):
        _check_planes(p, x, 1)


@pytest.mark.parametrize("cols", [768, 1024, 64])
def test_layernorm_writes_the_format(ops, cols):
    rows = 777
    g = torch.Generator().manual_seed(cols)
    x = _inputs(rows, cols, 20 + cols)
    gamma, beta = 1.0 + 0.1 * torch.randn(cols, generator=g), 0.05 * torch.randn(cols, generator=g)
    pl, _, mean, rstd, _ = ops.layernorm_fwd(x.cuda(), gamma.cuda(), beta.cuda(), 1e-6, 2, want_bf=True)
    ref = F.layer_norm(x.double(), (cols,), gamma.double(), beta.double(), 1e-6)
    assert rel(mean, x.double().mean(1)) < 1e-5 and rel(rstd, 1.0 / torch.sqrt(x.double().var(1, unbiased=False) + 1e-6)) < 1e-5
    assert rel(pl.hi.cpu().double() + pl.lo.cpu().double(), ref) < 2e-5      # a1 + a2 = the value to ~2^-17
    assert rel(pl.hi.cpu().double() / (1.0 - R.E), ref) < 4e-4 and rel(pl.bf.cpu(), ref) < 4e-3
    # and the bytes are the encoding of what the kernel normalised: re-encode the fp32 LayerNorm of the same device
    _, yf, _, _, _ = ops.layernorm_fwd(x.cuda(), gamma.cuda(), beta.cuda(), 1e-6, 1, want_f32=True, want_planes=False)
    p1, _, _ = R.encode(yf.cpu(), 0)
    same = (pl.hi.cpu().view(torch.int16) == p1.view(torch.int16)).float().mean()
    assert float(same) > 0.99                                      # summation order of the statistics differs in the last ulp


@pytest.mark.parametrize("M,N,K", [(4200, 2304, 768), (3140, 768, 3072), (785, 256, 64), (25120, 768, 128), (25120, 2304, 768)])
def test_gemm_f16x2_linear(ops, M, N, K):
    a, w = _inputs(M, K, 31), _inputs(N, K, 32, 0.03)
    g = torch.Generator().manual_seed(33)
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    pa, pw = ops.f16x2_encode(a.cuda(), 0), ops.f16x2_encode(w.cuda(), 1)
    exact = R.product(pa.hi.cpu(), pa.lo.cpu(), pw.hi.cpu(), pw.lo.cpu())
    true = a.double() @ w.double().t()
    # bias + residual -> fp32 (proj / fc2 form)
    out = torch.empty(M, N, device="cuda")
    ops.gemm_nt(pa, pw, passes=2, bias=bias.cuda(), residual=res.cuda(), out_f32=out)
    got = out.cpu().double() - bias.double() - res.double()
    # the three-product split-bf16 GEMM on the same operands
    out3 = torch.empty(M, N, device="cuda")
    ops.gemm_nt(ops.split_f32(a.cuda(), 3)[0], ops.split_f32(w.cuda(), 3)[0], passes=3, out_f32=out3)
    r_exact, r_true, r3 = rel(got, exact), rel(got, true), rel(out3, true)
    print("f16x2 gemm M=%d N=%d K=%d: vs exact product of the encoded operands %.2e, vs fp32 product %.2e (bf16x3: %.2e)" % (M, N, K, r_exact, r_true, r3))
    assert r_exact < 4e-6 and r_true < 2e-5 and r_true < 4 * r3 + 5e-6
    # bias -> split-bf16 planes (qkv form)
    pl = ops.empty_planes(M, N, 3, "cuda")
    ops.gemm_nt(pa, pw, passes=2, bias=bias.cuda(), out_planes=pl)
    assert rel(pl.float().cpu().double() - bias.double(), exact) < 2e-5


def test_gemm_f16x2_mlp_with_gelu_handover(ops):
    """fc1 -> GELU -> fc2 as the f16x2 mode runs it: the activation leaves the fc1 epilogue in the operand format (+ bf16 plane,
    + saved gelu' as bf16) and is consumed by fc2 directly."""
    M, D, Hd = 3140, 768, 3072
    x, w1, w2 = _inputs(M, D, 41), _inputs(Hd, D, 42, 0.03), _inputs(D, Hd, 43, 0.02)
    g = torch.Generator().manual_seed(44)
    b1, b2 = 0.1 * torch.randn(Hd, generator=g), 0.1 * torch.randn(D, generator=g)
    px, pw1, pw2 = ops.f16x2_encode(x.cuda(), 0), ops.f16x2_encode(w1.cuda(), 1), ops.f16x2_encode(w2.cuda(), 1)
    h = ops.empty_planes_f16x2(M, Hd, "cuda", want_bf=True)
    z = torch.empty(M, Hd, dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt(px, pw1, passes=2, bias=b1.cuda(), act=ops.ACT_GELU, aux_out=z, out_planes=h, aux_is_grad=True)
    pre = x.double() @ w1.double().t() + b1.double()
    act = F.gelu(pre)
    r_h = rel(h.hi.cpu().double() + h.lo.cpu().double(), act)
    cdf = 0.5 * (1.0 + torch.erf(pre / 2 ** 0.5))
    dgelu = cdf + pre * torch.exp(-0.5 * pre * pre) / (2 * torch.pi) ** 0.5
    print("f16x2 gelu hand-over: a1 + a2 vs gelu %.2e, bf %.2e, saved gelu' %.2e" % (r_h, rel(h.bf.cpu(), act), rel(z.cpu(), dgelu)))
    assert r_h < 3e-5 and rel(h.bf.cpu(), act) < 4e-3 and rel(z.cpu(), dgelu) < 4e-3
    # the planes are the first-operand encoding of the activation the kernel computed (up to its last-ulp differences from fp64)
    p1, _, _ = R.encode(act.float(), 0)
    assert float((h.hi.cpu().view(torch.int16) == p1.view(torch.int16)).float().mean()) > 0.95
    out = torch.empty(M, D, device="cuda")
    ops.gemm_nt(h, pw2, passes=2, bias=b2.cuda(), out_f32=out)
    ref = act @ w2.double().t() + b2.double()
    print("f16x2 mlp: vs fp64 %.2e" % rel(out, ref))
    assert rel(out, ref) < 3e-5


def _f16_plane(ops, x):
    """ONE plane of plain fp16(x): the first operand of a single-fp16-product GEMM (egv_gemm_nt passes == 4)."""
    return ops.Planes(x.cuda().to(torch.float16).contiguous(), None, x.shape[0], x.shape[1], "f16")


@pytest.mark.parametrize("M,N,K", [(4200, 2304, 768), (3140, 768, 3072), (785, 256, 64), (25120, 768, 3072), (25120, 3072, 768)])
def test_gemm_single_fp16_product(ops, M, N, K):
    """egv_gemm_nt(passes = 4): fp16(A) x fp16(W) on the fp16 MFMA, fp32 accumulate -- equal to the exact product of the rounded
    operands, 2^-11-grade against the fp32 product.  The weight operand is the f16x2 encoding (its plane 1 IS fp16(W))."""
    a, w = _inputs(M, K, 31), _inputs(N, K, 32, 0.03)
    g = torch.Generator().manual_seed(33)
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    pa, pw = _f16_plane(ops, a), ops.f16x2_encode(w.cuda(), 1)
    exact = pa.hi.cpu().double() @ pw.hi.cpu().double().t()
    true = a.double() @ w.double().t()
    out = torch.empty(M, N, device="cuda")
    ops.gemm_nt(pa, pw, passes=4, bias=bias.cuda(), residual=res.cuda(), out_f32=out)           # fc2 form
    got = out.cpu().double() - bias.double() - res.double()
    r_exact, r_true = rel(got, exact), rel(got, true)
    print("single fp16 product M=%d N=%d K=%d: vs exact product of the rounded operands %.2e, vs fp32 product %.2e" % (M, N, K, r_exact, r_true))
    assert r_exact < 4e-6 and 5e-5 < r_true < 6e-4
    pl = ops.empty_planes(M, N, 3, "cuda")
    ops.gemm_nt(pa, pw, passes=4, bias=bias.cuda(), out_planes=pl)                              # qkv form: split-bf16 planes out
    assert rel(pl.float().cpu().double() - bias.double(), exact) < 2e-5


@pytest.mark.parametrize("cols", [768, 1024])
def test_layernorm_writes_one_plain_fp16_plane(ops, cols):
    rows = 777
    g = torch.Generator().manual_seed(cols)
    x = _inputs(rows, cols, 20 + cols)
    gamma, beta = 1.0 + 0.1 * torch.randn(cols, generator=g), 0.05 * torch.randn(cols, generator=g)
    pl, _, mean, rstd, _ = ops.layernorm_fwd(x.cuda(), gamma.cuda(), beta.cuda(), 1e-6, 2, want_bf=True, single=True)
    assert pl.fmt == "f16" and pl.lo is None and pl.hi.dtype == torch.float16
    ref = F.layer_norm(x.double(), (cols,), gamma.double(), beta.double(), 1e-6)
    assert rel(pl.hi.cpu(), ref) < 4e-4 and rel(pl.bf.cpu(), ref) < 4e-3
    _, yf, _, _, _ = ops.layernorm_fwd(x.cuda(), gamma.cuda(), beta.cuda(), 1e-6, 1, want_f32=True, want_planes=False)
    same = (pl.hi.cpu().view(torch.int16) == yf.cpu().to(torch.float16).view(torch.int16)).float().mean()
    assert float(same) > 0.99


@pytest.mark.parametrize("fc1_single", [False, True])
def test_single_product_mlp_with_gelu_handover(ops, fc1_single):
    """fc1 -> GELU -> fc2 with fc2 (and optionally fc1) as ONE fp16 product: the fc1 epilogue (two- or one-product instance) writes
    the activation as ONE plain fp16 plane (out_fmt 2, + bf16 plane, + saved gelu'), fc2 consumes it."""
    M, D, Hd = 3140, 768, 3072
    x, w1, w2 = _inputs(M, D, 41), _inputs(Hd, D, 42, 0.03), _inputs(D, Hd, 43, 0.02)
    g = torch.Generator().manual_seed(44)
    b1, b2 = 0.1 * torch.randn(Hd, generator=g), 0.1 * torch.randn(D, generator=g)
    pw1, pw2 = ops.f16x2_encode(w1.cuda(), 1), ops.f16x2_encode(w2.cuda(), 1)
    px = _f16_plane(ops, x) if fc1_single else ops.f16x2_encode(x.cuda(), 0)
    h = ops.empty_planes_f16x2(M, Hd, "cuda", want_bf=True, single=True)
    z = torch.empty(M, Hd, dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt(px, pw1, passes=4 if fc1_single else 2, bias=b1.cuda(), act=ops.ACT_GELU, aux_out=z, out_planes=h, aux_is_grad=True)
    pre = x.double() @ w1.double().t() + b1.double()
    act = F.gelu(pre)
    cdf = 0.5 * (1.0 + torch.erf(pre / 2 ** 0.5))
    dgelu = cdf + pre * torch.exp(-0.5 * pre * pre) / (2 * torch.pi) ** 0.5
    r_h = rel(h.hi.cpu(), act)
    print("single-product mlp (fc1 single: %s): h vs gelu %.2e, bf %.2e, saved gelu' %.2e" % (fc1_single, r_h, rel(h.bf.cpu(), act), rel(z.cpu(), dgelu)))
    assert r_h < (8e-4 if fc1_single else 4e-4) and rel(h.bf.cpu(), act) < 4e-3 and rel(z.cpu(), dgelu) < 4e-3
    out = torch.empty(M, D, device="cuda")
    ops.gemm_nt(h, pw2, passes=4, bias=b2.cuda(), out_f32=out)
    exact = h.hi.cpu().double() @ pw2.hi.cpu().double().t() + b2.double()
    ref = act @ w2.double().t() + b2.double()
    print("single-product mlp: fc2 vs the exact product of its operands %.2e, vs fp64 %.2e" % (rel(out, exact), rel(out, ref)))
    assert rel(out, exact) < 4e-6 and rel(out, ref) < 1.5e-3


@pytest.mark.parametrize("mode,T", [(0, 4), (1, 4), (1, 16)])
def test_attention_writes_the_fp16_plane_for_a_single_product_proj(ops, mode, T):
    """egv_divided_attn_fwd mode bit 1: same attention, same bf16(value) first plane, but the second output plane holds fp16(value)
    (not the bf16 residual) -- every row incl. the CLS row (combine kernel) -- and a passes = 4 proj GEMM consumes it."""
    B, n, H = 2, 196, 12
    S = 1 + T * n
    g = torch.Generator().manual_seed(7 + mode + T)
    qkv = ops.split_f32((torch.randn(B * S, 3 * H * 64, generator=g) * 0.5).cuda(), 3)[0]
    ref, lse0 = ops.divided_attn_fwd(qkv, B, T, n, H, mode, 3)
    out, lse1 = ops.divided_attn_fwd(qkv, B, T, n, H, mode, 3, out_f16=True)
    torch.cuda.synchronize()
    assert out.fmt == "bf16+f16" and out.lo.dtype == torch.float16
    val = ref.hi.float() + ref.lo.float()
    assert torch.equal(out.hi, ref.hi) and torch.equal(lse0, lse1)
    same = (out.lo.view(torch.int16) == val.to(torch.float16).view(torch.int16)).float().mean()
    print("attention mode %d T=%d: fp16 plane == fp16(hi + lo) on %.4f of the elements, rel %.2e" % (mode, T, float(same), rel(out.lo, val)))
    assert float(same) > 0.98 and rel(out.lo, val) < 3e-4       # hi + lo is the value to 2^-17: the fp16 of it differs by a last-place tie at most
    w = _inputs(768, 768, 61, 0.03)
    pw = ops.f16x2_encode(w.cuda(), 1)
    o = torch.empty(B * S, 768, device="cuda")
    ops.gemm_nt(out, pw, passes=4, out_f32=o)
    exact = out.lo.cpu().double() @ pw.hi.cpu().double().t()
    assert rel(o, exact) < 4e-6
    # the backward takes delta from the bf16 plane alone
    d_out = ops.split_f32((torch.randn(B * S, H * 64, generator=g) * 0.1).cuda(), 1)[0]
    qkv1 = ops.Planes(qkv.hi, None, qkv.rows, qkv.cols)
    a = ops.divided_attn_bwd(qkv1, out, d_out, lse1, B, T, n, H, mode, 1)
    b = ops.divided_attn_bwd(qkv1, ops.Planes(ref.hi, None, ref.rows, ref.cols), d_out, lse1, B, T, n, H, mode, 1)
    torch.cuda.synchronize()
    assert rel(a.hi.float(), b.hi.float()) < 1e-3          # identical inputs; the CLS rows accumulate with fp32 atomics (bf16 flips)


def test_gemm_f16x2_rejects_what_it_cannot_run(ops):
    from egovlp_amd._lib import EgovlpHipError
    a, w = ops.f16x2_encode(_inputs(512, 256, 51).cuda(), 0), ops.f16x2_encode(_inputs(512, 256, 52).cuda(), 1)
    out = torch.empty(512, 512, device="cuda")
    with pytest.raises(ValueError):
        ops.gemm_nt(a, ops.split_f32(_inputs(512, 256, 53).cuda(), 3)[0], passes=2, out_f32=out)      # mixed operand formats
    small = ops.f16x2_encode(_inputs(128, 256, 54).cuda(), 0)
    with pytest.raises(EgovlpHipError):
        ops.gemm_nt(small, w, passes=2, out_f32=torch.empty(128, 512, device="cuda"))                  # below one big tile
    with pytest.raises(EgovlpHipError):
        ops.gemm_nt(a, w, passes=2, out_f32=out, ksplit=2)                                             # no split-K form
    one = _f16_plane(ops, _inputs(512, 256, 55))
    with pytest.raises(ValueError):
        ops.gemm_nt(one, w, passes=2, out_f32=out)                                                     # one plane is not an f16x2 operand
    with pytest.raises(EgovlpHipError):
        ops.gemm_nt(one, w, passes=4, out_f32=out, ksplit=2)                                           # single product: un-split only
    with pytest.raises(EgovlpHipError):
        ops.gemm_nt(one, w, passes=4, act=ops.ACT_GELU, out_f32=out)                                   # its GELU form writes fp16 operand planes only
