"""The fp16 backward (`set_precision('f16mix', 'f16')`), piece by piece through the C ABI on a real MI355X (`pytest -m gpu`).

The reference back-propagates in fp32 (trainer/trainer_egoclip.py:139-141).  Here every backward GEMM of the video blocks multiplies ONE
fp16 product on scaled gradients; what has to hold for that to be the reference's gradient to ~2^-11 per operand instead of bf16's 2^-8:
  * the kernels multiply exactly what they are given (TN weight gradient with its alpha and column sums, dgrad with the GELU' epilogue on
    an fp16 saved derivative, fp32 / bf16-plane / fp16-plane outputs) -- checked against fp64 products of the ROUNDED operands;
  * the producers of gradient planes write fp16 WITHOUT saturation (inf on overflow: LayerNorm backward, the GELU' epilogue, the attention
    backward, the cast of an fp32 gradient) -- that is what makes an overflow detectable;
  * the attention forward's fp16 output formats decode to the same O in the backward;
  * the device-side loss scale: overflow -> the step is skipped (parameters AND moments untouched) and S halves; `growth_interval` good
    steps -> S doubles; the bias correction counts applied steps; nothing synchronises with the host.
Whole-model gradient accuracy of the mode is asserted in tests/test_gpu_model.py."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

E = 2.0 ** -6


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def ops():
    from egovlp_amd import ops as _ops
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return _ops


def _f16_planes(ops, x):
    """host fp32 [rows, cols] -> device Planes fmt 'f16' holding fp16(x) (torch's RNE cast: the values are chosen in range)."""
    t = x.to(torch.float16).cuda().contiguous()
    return ops.Planes(t, None, x.shape[0], x.shape[1], "f16")


@pytest.mark.parametrize("M,N,K,ks", [(2304, 768, 25120, None), (768, 3072, 6280, 1), (768, 768, 3140, 3), (256, 256, 785, None)])
def test_fp16_weight_gradient_tn(ops, M, N, K, ks):
    """dW[M, N] = alpha * dY^T X on fp16 planes stored k-major (token-major), bias gradient = column sums of dY (NOT scaled)."""
    g = torch.Generator().manual_seed(M + N + K)
    dy = torch.randn(K, M, generator=g) * torch.logspace(-3, 1, K).unsqueeze(1)[torch.randperm(K, generator=g)]
    x = torch.randn(K, N, generator=g)
    a, b = _f16_planes(ops, dy), _f16_planes(ops, x)
    want = a.hi.cpu().double().t() @ b.hi.cpu().double()
    for alpha in (1.0, 1.0 / (1.0 - E)):
        out = torch.empty(M, N, device="cuda")
        cs = ops.gemm_tn(a, b, passes=4, out_f32=out, want_colsum=True, ksplit=ks, alpha=alpha)
        r, rc = rel(out, want * alpha), rel(cs, a.hi.cpu().double().sum(0))
        print("fp16 wgrad M=%d N=%d K=%d ksplit=%s alpha=%.4f: vs the exact product of the fp16 operands %.2e, column sums %.2e" % (M, N, K, ks, alpha, r, rc))
        assert r < 3e-6 and rc < 3e-6
    # and against the fp32 gradient it stands for: 2^-11 per operand, where the bf16 backward had 2^-8
    true = dy.double().t() @ x.double()
    out1 = torch.empty(M, N, device="cuda")
    ops.gemm_tn(ops.split_f32(dy.cuda(), 1)[0], ops.split_f32(x.cuda(), 1)[0], passes=1, out_f32=out1)
    r16, rbf = rel(out, true * (1.0 / (1.0 - E))), rel(out1, true)
    print("  vs the fp32 product: fp16 operands %.2e, bf16 operands %.2e" % (r16, rbf))
    assert r16 < 6e-4 and r16 < rbf / 4


def test_fp16_dgrad_outputs_and_gelu_bwd(ops):
    """dX = dY W (NT, W^T as an fp16 plane): fp32 output (LayerNorm backward reads it), a bf16 plane (the attention backward reads dO so),
    and the fc2 dgrad with the GELU' epilogue on an fp16 saved derivative -> dZ as one un-clamped fp16 plane."""
    M, N, K = 3140, 768, 3072                 # dY [M, K=3072] . W^T: rows of W^T = N outputs... here C[M, N] = A[M, K] B[N, K]^T
    g = torch.Generator().manual_seed(5)
    a = torch.randn(M, K, generator=g) * 30.0
    w = torch.randn(N, K, generator=g) * 0.03
    pa, pw = _f16_planes(ops, a), _f16_planes(ops, w)
    want = pa.hi.cpu().double() @ pw.hi.cpu().double().t()
    out = torch.empty(M, N, device="cuda")
    ops.gemm_nt(pa, pw, passes=4, out_f32=out)
    assert rel(out, want) < 3e-6
    pl = ops.empty_planes(M, N, 1, "cuda")
    ops.gemm_nt(pa, pw, passes=4, out_planes=pl)
    assert torch.equal(pl.hi.cpu().view(torch.int16), out.cpu().to(torch.bfloat16).view(torch.int16))      # bf16(RNE) of the same accumulators
    # GELU': C[M, Hd] = (G[M, D] W2t[Hd, D]^T) * gelu'(z), z saved as fp16 by the forward epilogue
    M, D, Hd = 3140, 768, 3072
    G = torch.randn(M, D, generator=g) * 100.0
    w2t = torch.randn(Hd, D, generator=g) * 0.03
    gz = (torch.rand(M, Hd, generator=g) * 1.26 - 0.13).to(torch.float16)      # the range of gelu'
    pg, pw2 = _f16_planes(ops, G), _f16_planes(ops, w2t)
    dz = ops.empty_planes_f16x2(M, Hd, "cuda", single=True)
    ops.gemm_nt(pg, pw2, passes=4, act=ops.ACT_GELU_BWD, aux_in=gz.cuda(), out_planes=dz, aux_is_grad=True)
    wantz = (pg.hi.cpu().double() @ pw2.hi.cpu().double().t()) * gz.double()
    r = rel(dz.hi.cpu(), wantz)
    print("fp16 fc2 dgrad + GELU' (fp16 derivative) -> fp16 plane: %.2e" % r)
    assert r < 4e-4                          # one fp16 rounding of the result
    # a gradient beyond fp16's range leaves as inf, not as 65504
    big = ops.Planes(torch.full((M, D), 60000.0, dtype=torch.float16, device="cuda"), None, M, D, "f16")
    ops.gemm_nt(big, ops.Planes(torch.ones((Hd, D), dtype=torch.float16, device="cuda"), None, Hd, D, "f16"), passes=4,
                act=ops.ACT_GELU_BWD, aux_in=torch.ones((M, Hd), dtype=torch.float16, device="cuda"), out_planes=dz, aux_is_grad=True)
    assert bool(torch.isinf(dz.hi).all())


def test_gelu_epilogue_saves_the_derivative_as_fp16(ops):
    M, D, Hd = 3140, 768, 3072
    g = torch.Generator().manual_seed(9)
    x, w1 = torch.randn(M, D, generator=g), torch.randn(Hd, D, generator=g) * 0.05
    b1 = 0.1 * torch.randn(Hd, generator=g)
    px = ops.layernorm_fwd(x.cuda(), torch.ones(D).cuda(), torch.zeros(D).cuda(), 1e-6, 2, single=True)[0]      # one plain fp16 plane
    pw1 = ops.f16x2_encode(w1.cuda(), 1)
    h = ops.empty_planes_f16x2(M, Hd, "cuda", single=True)            # no bf16 copy: the fp16 backward reads this very plane
    z = torch.empty(M, Hd, dtype=torch.float16, device="cuda")
    ops.gemm_nt(px, pw1, passes=4, bias=b1.cuda(), act=ops.ACT_GELU, aux_out=z, out_planes=h, aux_is_grad=True)
    pre = px.hi.cpu().double() @ pw1.hi.cpu().double().t() + b1.double()
    cdf = 0.5 * (1.0 + torch.erf(pre / math.sqrt(2.0)))
    dgelu = cdf + pre * torch.exp(-0.5 * pre * pre) / math.sqrt(2.0 * math.pi)
    assert rel(h.hi.cpu(), pre * cdf) < 4e-4 and rel(z.cpu(), dgelu) < 4e-4
    assert h.bf is None


def test_layernorm_backward_writes_unclamped_fp16_planes(ops):
    rows, cols = 3140, 768
    g = torch.Generator().manual_seed(3)
    x, dy = torch.randn(rows, cols, generator=g), torch.randn(rows, cols, generator=g) * 50.0
    gamma = 1.0 + 0.1 * torch.randn(cols, generator=g)
    add = torch.randn(rows, cols, generator=g)
    add[7, 11] = 1.0e6                       # a residual gradient beyond fp16's range
    mean, var = x.mean(1), x.var(1, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-6)
    dx, dg, db, pl = ops.layernorm_bwd(dy.cuda(), x.cuda(), gamma.cuda(), mean.cuda(), rstd.cuda(), add1=add.cuda(), planes_passes=4)
    assert pl.fmt == "f16" and pl.lo is None
    want = dx.cpu().to(torch.float16)
    assert torch.equal(pl.hi.cpu().view(torch.int16), want.view(torch.int16))          # fp16(RNE) of the fp32 output, bit for bit
    assert bool(torch.isinf(pl.hi[7, 11])) and float(dx[7, 11]) > 9e5                    # NOT saturated to 65504
    xd = x.double().requires_grad_(True)
    F.layer_norm(xd, (cols,), gamma.double(), None, 1e-6).backward(dy.double())
    assert rel(dx.cpu().double() - add.double(), xd.grad) < 1e-5
    # dy handed over as ONE plane of un-clamped fp16 (what the dgrad GEMM in front writes in the fp16 backward, egv_layernorm_bwd_fmt
    # dx_fmt bit 1): bit-identical to the fp32 path fed the fp16-rounded values -- the plane is decoded exactly, nothing else changes --,
    # and an inf in the plane reaches dx (the overflow check downstream sees it)
    dy16 = ops.f16_cast(dy.cuda())
    dx_a, dg_a, db_a, pl_a = ops.layernorm_bwd(dy16, x.cuda(), gamma.cuda(), mean.cuda(), rstd.cuda(), add1=add.cuda(), planes_passes=4)
    dx_b, dg_b, db_b, pl_b = ops.layernorm_bwd(dy.to(torch.float16).float().cuda(), x.cuda(), gamma.cuda(), mean.cuda(), rstd.cuda(),
                                               add1=add.cuda(), planes_passes=4)
    assert torch.equal(dx_a, dx_b) and torch.equal(pl_a.hi.view(torch.int16), pl_b.hi.view(torch.int16))
    assert rel(dg_a, dg_b) < 1e-6 and rel(db_a, db_b) < 1e-6                # (the column reduce adds its partial sums with atomics: order)
    assert rel(dx_a.cpu().double() - add.double(), xd.grad) < 5e-4          # one fp16 rounding of dy: 2^-12 rms
    big = dy.clone()
    big[5, 9] = 1.0e6
    dx_c = ops.layernorm_bwd(ops.f16_cast(big.cuda()), x.cuda(), gamma.cuda(), mean.cuda(), rstd.cuda())[0]
    assert not bool(torch.isfinite(dx_c[5]).all()) and bool(torch.isfinite(dx_c[6]).all())


def test_cast_and_transposed_weight_plane(ops):
    from egovlp_amd.weights import WeightCache
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1000, 768, generator=g) * 1000.0
    x[3, 5] = 7.0e4
    x[4, 6] = -1.0e-9
    pl = ops.f16_cast(x.cuda())
    assert torch.equal(pl.hi.cpu().view(torch.int16), x.to(torch.float16).view(torch.int16))
    assert bool(torch.isinf(pl.hi[3, 5]))
    w = torch.nn.Parameter((torch.randn(2304, 768, generator=g) * 0.05).cuda())
    wc = WeightCache()
    p2, t16 = wc.get(w, need_t=True, fmt="f16x2", t_fmt="f16")
    assert t16.fmt == "f16" and (t16.rows, t16.cols) == (768, 2304) and t16.lo is None
    assert torch.equal(t16.hi[:, :2304].cpu().view(torch.int16), w.detach().cpu().t().contiguous().to(torch.float16).view(torch.int16))
    assert torch.equal(p2.hi.cpu().view(torch.int16), w.detach().cpu().to(torch.float16).view(torch.int16))     # plane 1 of the f16x2 encoding IS fp16(W)
    # refreshed in place by the multi-tensor launch after an optimizer step
    from egovlp_amd import weights
    with torch.no_grad():
        w.mul_(2.0)
    weights.bump_epoch()
    wc.begin_step()
    _, t16b = wc.get(w, need_t=True, fmt="f16x2", t_fmt="f16")
    assert t16b is t16
    assert torch.equal(t16.hi[:, :2304].cpu().view(torch.int16), w.detach().cpu().t().contiguous().to(torch.float16).view(torch.int16))


@pytest.mark.parametrize("B,T,n,H", [(2, 4, 196, 12), (2, 3, 49, 4), (7, 4, 196, 12)])     # the last: 336 groups on 256 persistent workgroups
@pytest.mark.parametrize("mode", [0, 1])
def test_attention_fp16_output_formats_and_fp16_gradient_planes(ops, B, T, n, H, mode):
    """The forward's fp16 output formats ('f16x2': a1 / a2 planes; 'f16': one plane) hold the same O as the split-bf16 planes, the backward
    decodes them for delta = rowsum(dO o O), and dqkv as an un-clamped fp16 plane is the bf16-plane result to one rounding."""
    S, D = 1 + T * n, H * 64
    g = torch.Generator().manual_seed(100 * mode + B + T + n)
    qkv = ops.split_f32((torch.randn(B * S, 3 * D, generator=g) * 1.5).cuda(), 3)[0]
    ref, lse = ops.divided_attn_fwd(qkv, B, T, n, H, mode, 3)
    o_ref = ref.float().cpu().double()
    outs = {}
    for fmt in ("f16x2", "f16", "bf16+f16"):
        o, l2 = ops.divided_attn_fwd(qkv, B, T, n, H, mode, 3, out_fmt=fmt)
        assert torch.equal(l2, lse)
        outs[fmt] = o
    assert rel(outs["f16x2"].hi.cpu().double() + outs["f16x2"].lo.cpu().double(), o_ref) < 2e-5       # a1 + a2 = O to ~2^-17
    assert rel(outs["f16x2"].hi.cpu().double() / (1.0 - E), o_ref) < 4e-4
    assert rel(outs["f16"].hi.cpu(), o_ref) < 4e-4 and outs["f16"].lo is None
    assert torch.equal(outs["bf16+f16"].lo.cpu().view(torch.int16), outs["f16"].hi.cpu().view(torch.int16))
    d_out = ops.split_f32((torch.randn(B * S, D, generator=g) * 200.0).cuda(), 1)[0]           # a "scaled" gradient
    base = ops.divided_attn_bwd(qkv, outs["bf16+f16"], d_out, lse, B, T, n, H, mode, 1)          # bf16 O, bf16 dqkv (round 5's path)
    for fmt in ("f16x2", "f16"):
        got = ops.divided_attn_bwd(qkv, outs[fmt], d_out, lse, B, T, n, H, mode, 1, grad_f16=True)
        assert got.fmt == "f16" and got.lo is None
        r = rel(got.hi.cpu(), base.hi.cpu().float())
        print("attention backward mode %d, O as %s, dqkv as fp16 vs the bf16-plane result: %.2e" % (mode, fmt, r))
        assert r < 4e-3                       # the bf16 rounding of the baseline's output dominates
    # overflow -> inf
    huge = ops.split_f32((torch.randn(B * S, D, generator=g) * 3.0e6).cuda(), 1)[0]
    got = ops.divided_attn_bwd(qkv, outs["f16"], huge, lse, B, T, n, H, mode, 1, grad_f16=True)
    assert bool(torch.isinf(got.hi.float()).any())


def test_loss_scaler_skips_on_overflow_and_grows(ops):
    from egovlp_amd.optim import AdamW, LossScaler
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(300, 40, device="cuda")), torch.nn.Parameter(torch.randn(17, device="cuda"))]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt, opt_ref = AdamW(ps, lr=1e-2), AdamW(ref, lr=1e-2)
    sc = LossScaler(init_scale=1024.0, growth_interval=3)
    S = 1024.0
    applied = good = 0
    for step in range(9):
        gs = [torch.randn_like(p) for p in ps]
        overflow = step in (2, 6)
        for p, r, g in zip(ps, ref, gs):
            p.grad = g * S                              # what backward() of the scaled loss leaves
            r.grad = g.clone()
        if overflow:
            ps[0].grad[5, 7] = float("inf") if step == 2 else float("nan")
        before = [p.detach().clone() for p in ps]
        m_before = [opt.state[p]["exp_avg"].clone() for p in ps] if applied else None
        opt.step(scaler=sc)
        if overflow:
            for p, b in zip(ps, before):
                assert torch.equal(p.detach(), b)                              # parameters untouched ...
            if m_before is not None:
                for p, m in zip(ps, m_before):
                    assert torch.equal(opt.state[p]["exp_avg"], m)             # ... and the moments too
            S *= 0.5
            good = 0
        else:
            opt_ref.step()
            applied += 1
            good += 1
            if good == 3:
                S *= 2.0
                good = 0
            for p, r in zip(ps, ref):
                # un-scaled inside the kernel, bias correction at the number of APPLIED steps: the un-scaled optimizer's trajectory
                assert rel(p, r) < 2e-6, (step, rel(p, r))
        assert sc.get_scale() == S, (step, sc.get_scale(), S)
    assert sc.skipped_steps() == 2 and applied == 7
    st = sc.state_dict()
    sc2 = LossScaler()
    sc2.load_state_dict(st)
    assert sc2.get_scale() == sc.get_scale() and sc2.skipped_steps() == 2


@pytest.mark.parametrize("B,T,n,H", [(2, 4, 196, 12), (2, 3, 49, 4), (1, 16, 196, 2), (2, 4, 256, 4)])
@pytest.mark.parametrize("mode", [0, 1])
def test_fp16_attention_matches_the_split_bf16_attention_and_beats_the_bf16_backward(ops, B, T, n, H, mode):
    """The attention of the fp16 backward mode multiplies fp16 in BOTH directions: qkv planes are an fp16 split (hi, lo) -- the
    three-product forward is fp32-grade like the split-bf16 one --, the backward reads q / k / v (hi plane) and dO as fp16, rounds P and
    dS to fp16, writes dqkv as fp16.  Reference: the three-product split-bf16 backward (fp32-grade); the single-product bf16 backward of
    round 5 on the same inputs is printed next to it -- it is what kept the weight gradients of the first blocks at 2e-2."""
    S, D = 1 + T * n, H * 64
    g = torch.Generator().manual_seed(1000 * mode + B + T + n)
    x = torch.randn(B * S, 3 * D, generator=g) * 1.5
    qkv3 = ops.split_f32(x.cuda(), 3)[0]
    hi = x.to(torch.float16)
    qkv16 = ops.Planes(hi.cuda(), (x - hi.float()).to(torch.float16).cuda(), B * S, 3 * D, "f16s")
    ref, lse = ops.divided_attn_fwd(qkv3, B, T, n, H, mode, 3)
    o_ref = ref.float().cpu().double()
    got, lse16 = ops.divided_attn_fwd(qkv16, B, T, n, H, mode, 3, out_fmt="f16x2")
    r_fwd = rel(got.hi.cpu().double() + got.lo.cpu().double(), o_ref)
    assert r_fwd < 3e-5 and rel(lse16, lse) < 1e-5, (r_fwd, rel(lse16, lse))
    one, _ = ops.divided_attn_fwd(qkv16, B, T, n, H, mode, 3, out_fmt="f16")
    # backward: a scaled gradient
    dy = torch.randn(B * S, D, generator=g) * 200.0
    d3 = ops.split_f32(dy.cuda(), 3)[0]
    want = ops.divided_attn_bwd(qkv3, ref, d3, lse, B, T, n, H, mode, 3).float().cpu().double()
    d1 = ops.split_f32(dy.cuda(), 1)[0]
    bf = ops.divided_attn_bwd(qkv3, ref, d1, lse, B, T, n, H, mode, 1).hi.cpu().float()
    d16 = ops.f16_cast(dy.cuda())
    for o in (got, one):
        f16 = ops.divided_attn_bwd(qkv16, o, d16, lse16, B, T, n, H, mode, 1, grad_f16=True)
        r16, rbf = rel(f16.hi.cpu(), want), rel(bf, want)
        print("attention backward %s B=%d T=%d n=%d H=%d (O as %s): fp16 operands %.2e, bf16 operands %.2e from the three-product result" % (
            "time" if mode else "space", B, T, n, H, o.fmt, r16, rbf))
        assert r16 < 1.2e-3 and r16 < rbf / 3
    huge = ops.f16_cast((dy * 1.0e4).cuda())
    assert bool(torch.isinf(huge.hi).any())
    bad = ops.divided_attn_bwd(qkv16, one, huge, lse16, B, T, n, H, mode, 1, grad_f16=True)
    assert not bool(torch.isfinite(bad.hi.float()).all())             # inf / NaN out, never a clamped finite gradient


def test_overflowing_steps_are_skipped_and_the_scale_finds_its_range():
    """The whole EgoClip step in the benchmarked mode ('f16mix' forward, fp16 backward) with a loss scale that starts absurdly high
    (2^30): fp16 gradient planes overflow to inf, `AdamW.step(scaler=...)` sees it on the device and does NOT touch parameters or
    moments, the scale halves -- step after step, without any host synchronisation -- until a backward fits fp16's range; from then on
    the steps are applied, the loss stays finite, and the bias correction has counted only the applied steps."""
    from egovlp_amd.model.loss import EgoNCE
    from egovlp_amd.model.model import FrozenInTime
    from egovlp_amd.optim import AdamW, LossScaler
    from egovlp_amd.synth import synth_batch, synth_state_dict
    from egovlp_amd.trainer.trainer_egoclip import egoclip_step
    m = FrozenInTime(video_params={"model": "SpaceTimeTransformer", "arch_config": "base_patch16_224", "num_frames": 4, "pretrained": True,
                                   "time_init": "rand"},
                     text_params={"model": "distilbert-base-uncased", "pretrained": True, "input": "text"}, projection="minimal",
                     load_checkpoint="")
    m.load_state_dict(synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, seed=5))
    m.text_model.set_dropout(0.0, 0.0)
    m = m.cuda().train()
    m.exec_ctx.set_precision("f16mix", "f16")
    b = synth_batch(4, T=4, L=16, seed=21)
    dev = {"video": b["video"].cuda(), "text": {k: v.cuda() for k, v in b["text"].items()}, "noun_vec": b["noun_vec"].cuda(),
           "verb_vec": b["verb_vec"].cuda()}
    opt = AdamW(m.parameters(), lr=3e-5)
    sc = LossScaler(init_scale=2.0 ** 30, growth_interval=1000, max_scale=2.0 ** 30)
    watch = [m.video_model.blocks[0].attn.qkv.weight, m.video_model.blocks[11].mlp.fc2.weight, m.text_model.transformer.layer[0].ffn.lin1.weight]
    before = [w.detach().clone() for w in watch]
    losses, scales, skipped = [], [], []
    for step in range(24):
        losses.append(egoclip_step(m, EgoNCE(), opt, dev, scaler=sc))
        scales.append(sc.get_scale())                    # (host readbacks: this is a test)
        skipped.append(sc.skipped_steps())
        if skipped[-1] == step + 1:                      # every step so far overflowed: nothing may have moved
            assert all(torch.equal(w.detach(), b0) for w, b0 in zip(watch, before)), step
    n_skip = skipped[-1]
    print("loss scale from 2^30: %d skipped steps, scale settles at 2^%d; losses %s" % (n_skip, int(math.log2(scales[-1])), ["%.4f" % float(x) for x in losses[-4:]]))
    assert 4 <= n_skip <= 20, (n_skip, scales)
    assert scales[-1] == 2.0 ** (30 - n_skip)                           # one halving per skipped step, no growth yet
    assert skipped == sorted(skipped) and all(skipped[i + 1] - skipped[i] in (0, 1) for i in range(len(skipped) - 1))
    assert all(torch.isfinite(x) for x in losses)
    assert all(not torch.equal(w.detach(), b0) for w, b0 in zip(watch, before))      # the applied steps did move the weights
    assert all(bool(torch.isfinite(w).all()) for w in watch)
    assert float(losses[-1]) < float(losses[0])                          # and in the right direction (same batch every step)
    # the optimizer's moments saw only finite, un-scaled gradients
    st = opt.state[watch[0]]
    assert bool(torch.isfinite(st["exp_avg"]).all()) and float(st["exp_avg"].abs().max()) < 1.0
