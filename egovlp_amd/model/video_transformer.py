"""SpaceTimeTransformer on MI355X kernels -- drop-in for the reference's model/video_transformer.py.

Same constructor, same parameter names and shapes (state_dict compatible), same
arithmetic (reference lines cited inline), different execution: one autograd node per
SpaceTimeBlock whose forward and hand-written backward are sequences of enqueues of the gfx950
kernels behind include/egovlp_hip.h.  The residual stream is fp32; every GEMM operand is a
split-bf16 plane pair produced by the epilogue of the kernel before it (LayerNorm, attention, GELU),
so no activation is ever re-formatted in a separate pass in forward.

nn.Linear / nn.LayerNorm / nn.Conv2d are used as PARAMETER CONTAINERS only (names + default init
identical to the reference); their own forward is never called.
"""
from __future__ import annotations

from functools import partial


import torch
from torch import nn

from .. import ops
from ..ops import ACT_GELU, ACT_GELU_BWD, ExecContext, Planes


# The dgrad GEMMs that feed LayerNorm-backward write fp32: handing dy over as bf16 planes instead halves those bytes but measured
# 0.5 ms/step SLOWER in round 2 and within noise in round 3 (the split epilogue's VALU work and 8-byte stores / loads cost what the
# bytes save; profiles/r03_stream_ab.txt) -- the switch is gone, egv_layernorm_bwd still accepts planes (tests).


def _lin_bwd(dy, x_pl: Planes, wt: Planes, Pb, need_dx=True, dx_planes=False, params=(), ec: ExecContext = None, allow_side=True):
    """Backward of y = x W^T + b.  `dy` is fp32 [M,N] (split to bf16 planes here, one pass, no transpose) or
    already-split row-major Planes.  The SAME row-major planes feed both gradients: dgrad contracts over N
    (dy . W, weights cached transposed) and wgrad contracts over the M token rows with the TN kernel
    (dy^T x via the CDNA4 transpose read, bias gradient from the same pass).
    `params`: the parameters (weight, bias) the returned dW / db will be accumulated into by autograd; `ec`: the model's
    execution context (side stream, grid cap); `allow_side=False`: the caller reads dW / db right away on ITS stream (slices of a
    padded head), so the weight gradient stays on the current stream.
    -> (dx fp32 [M,K] | None, dW fp32 [N,K], db [N])."""
    ec = ops.DEFAULT if ec is None else ec
    alpha = 1.0
    if Pb == 4:
        # the fp16 backward: dy = ONE plane of un-clamped fp16 (a scaled gradient), X = plane 1 of the forward's own fp16 operand (the
        # weight gradient is rescaled when that plane is a1 = fp16((1 - 2^-6) x) of an f16x2 encoding), W^T an fp16 plane
        if not isinstance(dy, Planes):
            dy = ops.f16_cast(dy)
        x_pl, alpha = x_pl.bwd16()
    else:
        if not isinstance(dy, Planes):
            dy = ops.split_f32(dy, Pb)[0]
        x_pl = x_pl.bwd()              # an f16x2 forward operand hands over its bf16 plane
    M, K = x_pl.rows, x_pl.cols
    N = dy.cols
    dev = x_pl.hi.device
    # off the critical path: fills the CUs the dgrad chain leaves idle (ops.side_stream); the text tower's own backward is
    # already off the video tower's stream (ops.TEXT_SIDE_STREAM) and keeps its small wgrads where they are.
    # The side stream is only safe while autograd's AccumulateGrad STEALS dW (parameter.grad is None: zero_grad(set_to_none=True),
    # one backward per step): with a gradient already in place it enqueues `grad += dW` on the node's stream, which is not
    # ordered behind the side stream -- such wgrads (gradient accumulation, set_to_none=False) stay on the main stream.
    accumulating = any(p_ is not None and p_.grad is not None for p_ in params)
    if allow_side and ec.wgrad_side_stream and not ec.on_text_stream() and not accumulating:
        with ec.side_stream(dy.hi, dy.lo, x_pl.hi, x_pl.lo, cost=float(M) * N * K):
            dW = torch.empty((N, K), dtype=torch.float32, device=dev)
            db = ops.gemm_tn(dy, x_pl, passes=Pb, out_f32=dW, want_colsum=True, ec=ec, alpha=alpha)
    else:
        dW = torch.empty((N, K), dtype=torch.float32, device=dev)
        db = ops.gemm_tn(dy, x_pl, passes=Pb, out_f32=dW, want_colsum=True, ec=ec, alpha=alpha)
    dx = None
    if need_dx and dx_planes:      # dx feeds a kernel that consumes planes (attention backward): no fp32 copy at all
        if Pb == 4:                # dO of the fp16 attention backward: one plane of un-clamped fp16
            dx = ops.empty_planes_f16x2(M, K, dev, single=True)
            ops.gemm_nt(dy, wt, passes=4, out_planes=dx, K=N, ec=ec, grad_out=True)
        else:
            dx = ops.empty_planes(M, K, Pb, dev)
            ops.gemm_nt(dy, wt, passes=Pb, out_planes=dx, K=N, ec=ec)
    elif need_dx:
        dx = torch.empty((M, K), dtype=torch.float32, device=dev)
        ops.gemm_nt(dy, wt, passes=Pb, out_f32=dx, K=N, ec=ec)
    return dx, dW, db


# Gradient hand-off between consecutive blocks' backward passes: the LayerNorm-backward kernel that produces a block's
# input gradient d_x also emits it as split-bf16 planes (the format the previous block's GEMMs consume).  autograd only
# carries the fp32 tensor, so the planes ride along ON that tensor object (`_egv_planes`: PyTorch preserves a tensor's Python
# object, attributes included, across the engine), stamped with the tensor's version counter.  Anything that replaces the
# tensor (gradient accumulation from a second consumer, hooks that return a new tensor) drops the attribute; anything that
# modifies it in place bumps the version -- either way the consumer falls back to one egv_split_f32 pass of the real values.
PLANE_HANDOFF = {"hit": 0, "miss": 0}     # diagnostics / tests


def _attach_grad_planes(g, Pb, planes):
    g._egv_planes = (Pb, planes, g._version)
    return g


def _take_grad_planes(g_out, g2d, Pb):
    ent = getattr(g_out, "_egv_planes", None)
    if ent is not None:
        try:
            del g_out._egv_planes
        except AttributeError:
            pass
        if ent[0] == Pb and ent[2] == g_out._version and ent[1].rows == g2d.shape[0] and ent[1].cols == g2d.shape[1]:
            PLANE_HANDOFF["hit"] += 1
            return ent[1]
    PLANE_HANDOFF["miss"] += 1
    return ops.f16_cast(g2d) if Pb == 4 else ops.split_f32(g2d, Pb)[0]


def f16x2_block_ok(M, D, Hd, train):
    """Can a block of this size run its qkv / fc1 / fc2 Linears in the f16x2 format?  (the big-tile kernel for the three products, and
    -- the fc1 epilogue of that format saves gelu' as bf16 -- for the single-pass fc2 dgrad that reads it back; smaller blocks run
    split-bf16 x3)"""
    if not (ops.f16x2_gemm_ok(M, 3 * D, D) and ops.f16x2_gemm_ok(M, Hd, D) and ops.f16x2_gemm_ok(M, D, Hd)):
        return False
    return not train or ops.uses_big_gemm(M, Hd, D, 1)


class _SpaceTimeBlockFn(torch.autograd.Function):
    """SpaceTimeBlock.forward, model/video_transformer.py:163-177:
         t  = timeattn(norm3(x));  tr = x + t
         s  = attn(norm1(tr));     sr = x + s          (residual from x, NOT tr -- :171)
         out = sr + mlp(norm2(sr))
    """

    @staticmethod
    def forward(ctx, x, geom, ec: ExecContext,
                n3w, n3b, tqkv_w, tqkv_b, tproj_w, tproj_b,
                n1w, n1b, sqkv_w, sqkv_b, sproj_w, sproj_b,
                n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b):
        B, T, n, H, eps = geom[:5]
        single = geom[5] if len(geom) > 5 else 0     # ops.F16_SINGLE_BITS: Linears of THIS block that run ONE fp16 product (f16x2 mode)
        S = 1 + T * n
        D = x.shape[-1]
        M = B * S
        P = ec.fwd_passes
        wc = ec.wc
        dev = x.device
        x2 = x.contiguous().view(M, D)
        train = any(ctx.needs_input_grad)   # grad mode is off inside Function.forward; this is the reliable signal
        Hd = fc1_w.shape[0]
        # 'f16x2': the LayerNorm -> qkv / fc1 and fc1 -> fc2 hand-overs are in the f16x2 operand format (two fp16 products instead of
        # three bf16 ones, big-tile kernel only); attention and the proj Linears keep split-bf16 three-product operands (Pa).  Token
        # counts too small for the big-tile kernel (toy geometries) run the block in bf16x3.
        if P == 2 and not f16x2_block_ok(M, D, Hd, train):
            P = 3
        fx2 = P == 2
        Pa = 3 if fx2 else P
        wf = "f16x2" if fx2 else "bf16"
        if not fx2:
            single = 0
        # the fp16 backward (ec.bwd_passes == 4) reads the forward's own fp16 planes: no bf16 copies; the attention output is fp16 planes only
        # ('f16': one plane for a one-product proj; 'f16x2': the proj runs TWO fp16 products where it ran three bf16 ones)
        h16 = fx2 and ec.bwd_passes == 4
        want_bf = train and not h16
        s_fc1, s_fc2, s_qkv, s_proj = bool(single & 1), bool(single & 2), bool(single & 4), bool(single & 8)
        P_qkv, P_fc1, P_fc2, P_proj = (4 if s_qkv else P), (4 if s_fc1 else P), (4 if s_fc2 else P), (4 if s_proj else (2 if h16 else Pa))
        wp = "f16x2" if (s_proj or h16) else "bf16"       # a single-product proj multiplies fp16(W) (plane 1 of the f16x2 encoding)
        afmt = ("f16" if s_proj else "f16x2") if h16 else ("bf16+f16" if s_proj else "bf16")

        def W(p, fmt="bf16"):
            return wc.get(p, need_t=False, fmt=fmt)[0]

        # ---- temporal attention branch (:166-167)
        n3, _, mean3, rstd3, _ = ops.layernorm_fwd(x2, n3w, n3b, eps, P, want_bf=want_bf, single=s_qkv)
        # qkv never exists in fp32: the attention kernels read planes (split-bf16; an fp16 split when the backward is fp16: the attention
        # then multiplies fp16 in both directions)
        qkv_t = ops.empty_planes_f16x2(M, 3 * D, dev, split=True) if h16 else ops.empty_planes(M, 3 * D, Pa, dev)
        ops.gemm_nt(n3, W(tqkv_w, wf), passes=P_qkv, bias=tqkv_b, out_planes=qkv_t, ec=ec)
        a_t, lse_t = ops.divided_attn_fwd(qkv_t, B, T, n, H, 1, Pa, out_fmt=afmt)
        tr = torch.empty((M, D), dtype=torch.float32, device=dev)
        ops.gemm_nt(a_t, W(tproj_w, wp), passes=P_proj, bias=tproj_b, residual=x2, out_f32=tr, ec=ec)
        # ---- spatial attention branch (:168-171)
        n1, _, mean1, rstd1, _ = ops.layernorm_fwd(tr, n1w, n1b, eps, P, want_bf=want_bf, single=s_qkv)
        qkv_s = ops.empty_planes_f16x2(M, 3 * D, dev, split=True) if h16 else ops.empty_planes(M, 3 * D, Pa, dev)
        ops.gemm_nt(n1, W(sqkv_w, wf), passes=P_qkv, bias=sqkv_b, out_planes=qkv_s, ec=ec)
        a_s, lse_s = ops.divided_attn_fwd(qkv_s, B, T, n, H, 0, Pa, out_fmt=afmt)
        sr = torch.empty((M, D), dtype=torch.float32, device=dev)
        ops.gemm_nt(a_s, W(sproj_w, wp), passes=P_proj, bias=sproj_b, residual=x2, out_f32=sr, ec=ec)
        # ---- MLP (:175, Mlp.forward :46-52), exact-erf GELU fused into the fc1 epilogue
        n2, _, mean2, rstd2, _ = ops.layernorm_fwd(sr, n2w, n2b, eps, P, want_bf=want_bf, single=s_fc1)
        h = ops.empty_planes_f16x2(M, Hd, dev, want_bf=want_bf, single=s_fc2) if fx2 else ops.empty_planes(M, Hd, P, dev)
        # saved for backward: the fp32 pre-activation z in the all-bf16x3 parity mode; when backward runs single-pass bf16
        # anyway, gelu'(z) itself as bf16 -- the epilogue has Phi(z) and phi(z) in registers, the buffer is half the bytes,
        # and the fc2-dgrad epilogue becomes one multiply instead of a second erf evaluation over 77 M elements
        # (the fp16 backward keeps gelu' as fp16: bf16's 2^-9 on the derivative would cap dZ's accuracy)
        z_dtype = torch.float16 if h16 else (torch.bfloat16 if (ec.bwd_passes in (1, 4) and ops.uses_big_gemm(M, Hd, D, P)) else torch.float32)
        z = torch.empty((M, Hd), dtype=z_dtype, device=dev) if train else None
        ops.gemm_nt(n2, W(fc1_w, wf), passes=P_fc1, bias=fc1_b, act=ACT_GELU, aux_out=z, out_planes=h,
                    aux_is_grad=z is not None and z_dtype != torch.float32, ec=ec)
        out = torch.empty((M, D), dtype=torch.float32, device=dev)
        ops.gemm_nt(h, W(fc2_w, wf), passes=P_fc2, bias=fc2_b, residual=sr, out_f32=out, ec=ec)

        if train:
            ctx.geom, ctx.ec, ctx.P, ctx.h16 = geom, ec, P, h16
            ctx.planes = (n3, a_t, n1, a_s, n2, h, qkv_t, qkv_s)
            ctx.save_for_backward(x2, mean3, rstd3, lse_t, tr, mean1, rstd1, lse_s, sr, mean2, rstd2, z,
                                  n3w, tqkv_w, tproj_w, n1w, sqkv_w, sproj_w, n2w, fc1_w, fc2_w)
        return out.view(B, S, D)

    @staticmethod
    def backward(ctx, g_out):
        (x2, mean3, rstd3, lse_t, tr, mean1, rstd1, lse_s, sr, mean2, rstd2, z,
         n3w, tqkv_w, tproj_w, n1w, sqkv_w, sproj_w, n2w, fc1_w, fc2_w) = ctx.saved_tensors
        n3, a_t, n1, a_s, n2, h, qkv_t, qkv_s = ctx.planes
        B, T, n, H, eps = ctx.geom[:5]
        ec = ctx.ec
        wc = ec.wc
        Pb = ec.bwd_passes
        ec.poll_backward()              # gradients of the blocks behind this one are final: the data-parallel exchange may start
        if Pb == 4 and not ctx.h16:
            if ec.fwd_passes == 2 and ctx.P == 2:
                raise RuntimeError("the backward precision changed to 'f16' after this block's forward (which wrote bf16 copies, not fp16 planes)")
            Pb = min(ec.bwd_passes_split, ctx.P)       # a block too small for the fp16 forward format (toy geometries) ran split-bf16
        if Pb == 3 and ctx.P != 3:
            raise RuntimeError("backward precision bf16x3 needs a bf16x3 forward (the saved activation planes carry no lo part)")
        h16 = Pb == 4
        M, D = x2.shape
        G = g_out.contiguous().view(M, D)

        def Wt(p):
            if h16:
                return wc.get(p, need_t=True, fmt="f16x2", t_fmt="f16")[1]
            return wc.get(p, need_t=True)[1]

        # ---- MLP backward.  dZ = (G . W2) * gelu'(z) comes out of the fc2-dgrad epilogue already split.
        G_pl = _take_grad_planes(g_out, G, Pb)
        Hd = fc1_w.shape[0]
        dZ = ops.empty_planes_f16x2(M, Hd, G.device, single=True) if h16 else ops.empty_planes(M, Hd, Pb, G.device)
        ops.gemm_nt(G_pl, Wt(fc2_w), passes=Pb, act=ACT_GELU_BWD, aux_in=z, out_planes=dZ, K=D,
                    aux_is_grad=z.dtype != torch.float32, ec=ec)
        _, d_fc2_w, d_fc2_b = _lin_bwd(G_pl, h, None, Pb, need_dx=False, params=(fc2_w,), ec=ec)
        ln16 = Pb == 4          # fp16 backward: the dgrads in front of a LayerNorm backward hand it ONE plane of un-clamped fp16 (no fp32 copy)
        d_n2, d_fc1_w, d_fc1_b = _lin_bwd(dZ, n2, Wt(fc1_w), Pb, dx_planes=ln16, params=(fc1_w,), ec=ec)
        # d_sr = G + LN2'(d_n2)
        d_sr, d_n2w, d_n2b, d_sr_pl = ops.layernorm_bwd(d_n2, sr, n2w, mean2, rstd2, add1=G, planes_passes=Pb)
        # ---- spatial attention backward
        d_as, d_sproj_w, d_sproj_b = _lin_bwd(d_sr_pl, a_s, Wt(sproj_w), Pb, dx_planes=True, params=(sproj_w,), ec=ec)
        d_qkv_s = ops.divided_attn_bwd(qkv_s, a_s, d_as, lse_s, B, T, n, H, 0, 1 if h16 else Pb, grad_f16=h16)
        d_n1, d_sqkv_w, d_sqkv_b = _lin_bwd(d_qkv_s, n1, Wt(sqkv_w), Pb, dx_planes=ln16, params=(sqkv_w,), ec=ec)
        d_tr, d_n1w, d_n1b, d_tr_pl = ops.layernorm_bwd(d_n1, tr, n1w, mean1, rstd1, planes_passes=Pb)
        # ---- temporal attention backward
        d_at, d_tproj_w, d_tproj_b = _lin_bwd(d_tr_pl, a_t, Wt(tproj_w), Pb, dx_planes=True, params=(tproj_w,), ec=ec)
        d_qkv_t = ops.divided_attn_bwd(qkv_t, a_t, d_at, lse_t, B, T, n, H, 1, 1 if h16 else Pb, grad_f16=h16)
        d_n3, d_tqkv_w, d_tqkv_b = _lin_bwd(d_qkv_t, n3, Wt(tqkv_w), Pb, dx_planes=ln16, params=(tqkv_w,), ec=ec)
        # x feeds norm3, the tr residual and the sr residual: dx = d_tr + d_sr + LN3'(d_n3)
        d_x, d_n3w, d_n3b, d_x_pl = ops.layernorm_bwd(d_n3, x2, n3w, mean3, rstd3, add1=d_tr, add2=d_sr,
                                                       planes_passes=Pb)
        S = 1 + T * n
        return (_attach_grad_planes(d_x.view(B, S, D), Pb, d_x_pl), None, None,
                d_n3w, d_n3b, d_tqkv_w, d_tqkv_b, d_tproj_w, d_tproj_b,
                d_n1w, d_n1b, d_sqkv_w, d_sqkv_b, d_sproj_w, d_sproj_b,
                d_n2w, d_n2b, d_fc1_w.view_as(fc1_w), d_fc1_b, d_fc2_w.view_as(fc2_w), d_fc2_b)


# ---------------------------------------------------------------------------------------------- one C call per block
# The same block through egv_block_fwd / egv_block_bwd (csrc/block.hip): the C side enqueues the block's kernels with pointers into
# one workspace arena per direction.  What stays in Python is policy: which precision, which stream each weight gradient goes to,
# how many k-slices it gets, the gradient-plane hand-over between blocks, the backward poll of the gradient exchange.
_BLOCK_CACHE = {}      # geometry -> (forward arena bytes, gradient offsets, gradient floats)
_BLOCK_BWD_BYTES = {}  # (geometry, k-slices) -> backward arena bytes
_W_ORDER = ("tqkv", "tproj", "sqkv", "sproj", "fc1", "fc2")


def block_calls_ok(ec: ExecContext, M, D, Hd):
    """May this block run through the C block calls?  (split-bf16 / bf16 precision, no per-kernel timer attached, every GEMM of
    the block un-split and at least one 256-wide tile: the per-kernel path covers the toy shapes)"""
    if not ec.block_calls or ec.kernel_timer is not None or (ec.bwd_passes == 3 and ec.fwd_passes != 3) or \
            (ec.bwd_passes == 4 and ec.fwd_passes != 2):
        return False
    if ec.fwd_passes == 2 and not f16x2_block_ok(M, D, Hd, True):
        return False
    if D < 256 or Hd < 256 or D % 64 or Hd % 64 or M < 256:
        return False
    return all(ops.auto_ksplit_nt(*sh) == 1 for sh in ((M, 3 * D, D), (M, D, D), (M, Hd, D), (M, D, Hd), (M, D, 3 * D)))


def _block_geom(B, T, n, H, D, Hd, P, Pb, train, z_bf16, single, eps, grid):
    from .._lib import BlockGeom
    return BlockGeom(B, T, n, H, D, Hd, P, Pb, int(train), int(z_bf16), float(eps), int(grid), int(single))


_X2_FMTS = ("f16x2", "bf16", "f16x2", "bf16", "f16x2", "f16x2")     # f16x2 mode: qkv / fc1 / fc2 weights in the f16x2 format, proj split-bf16


def _block_params(wc, ln, biases, weights, need_t, x2=False, proj_x2=False, t16=False):
    """egv_block_params from the parameter tensors: LayerNorm affine (n3w, n3b, n1w, n1b, n2w, n2b), the six biases and the
    cached operand planes of the six weights (W^T planes too when `need_t`).  The planes are refreshed IN PLACE after an optimizer
    step, so the struct stays the same from step to step: it is kept on the model's weight cache, keyed by the block's first weight
    and the direction, and reused while the cache still holds the very same plane objects and the small parameters have not moved."""
    import ctypes as C
    from .._lib import BlockParams
    P6, L6 = C.c_void_p * 6, C.c_int64 * 6
    # proj_x2: the proj Linears run one fp16 product in this block -- their forward weights are f16x2 encodings too
    # t16 (the fp16 backward): the transposed weights are single fp16 planes (wt_lo unused)
    fmts = [("f16x2" if (proj_x2 and i in (1, 3)) else _X2_FMTS[i]) if x2 else "bf16" for i in range(6)]
    pls = [wc.get(w, need_t=need_t, fmt=fmts[i], t_fmt="f16" if t16 else "bf16") for i, w in enumerate(weights)]
    small = tuple(t.data_ptr() for t in ln) + tuple(b.data_ptr() for b in biases)
    key = (id(weights[0]), need_t, x2, proj_x2, t16)
    hit = wc.param_structs.get(key)
    if hit is not None and hit[1] == small and all(a[0] is b[0] and a[1] is b[1] for a, b in zip(hit[0], pls)):
        return hit[2]
    whi, wlo, ldw = P6(*[p.hi.data_ptr() for p, _ in pls]), P6(*[p.lo.data_ptr() for p, _ in pls]), L6(*[p.ld for p, _ in pls])
    if need_t:
        thi, tlo, ldt = P6(*[t.hi.data_ptr() for _, t in pls]), P6(*[(t.lo.data_ptr() if t.lo is not None else None) for _, t in pls]), \
            L6(*[t.ld for _, t in pls])
    else:
        thi, tlo, ldt = P6(), P6(), L6()
    prm = BlockParams(*[t.data_ptr() for t in ln], P6(*[b.data_ptr() for b in biases]), whi, wlo, ldw, thi, tlo, ldt)
    wc.param_structs[key] = (pls, small, prm)
    return prm


class _SpaceTimeBlockCFn(torch.autograd.Function):
    """SpaceTimeBlock.forward / backward as ONE C-ABI call each (see _SpaceTimeBlockFn for the arithmetic)."""

    @staticmethod
    def forward(ctx, x, geom, ec: ExecContext,
                n3w, n3b, tqkv_w, tqkv_b, tproj_w, tproj_b,
                n1w, n1b, sqkv_w, sqkv_b, sproj_w, sproj_b,
                n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b):
        import ctypes as C
        from .. import _lib
        B, T, n, H, eps = geom[:5]
        S = 1 + T * n
        D = x.shape[-1]
        M = B * S
        Hd = fc1_w.shape[0]
        P, Pb = ec.fwd_passes, ec.bwd_passes
        single = (geom[5] if len(geom) > 5 else 0) if P == 2 else 0     # ops.F16_SINGLE_BITS of this block
        dev = x.device
        x2 = x.contiguous().view(M, D)
        train = any(ctx.needs_input_grad)
        z_bf16 = Pb in (1, 4) and ops.uses_big_gemm(M, Hd, D, P)      # fc1 saves gelu' in 16 bits (bf16; fp16 for the fp16 backward)
        key = (B, T, n, H, D, Hd, P, Pb, train, z_bf16, single)
        g = _block_geom(*key, eps, ec.gemm_grid)
        ent = _BLOCK_CACHE.get(key)
        if ent is None:
            off, tot = (C.c_int64 * 18)(), C.c_int64()
            nb = int(_lib.lib().egv_block_fwd_arena_bytes(C.byref(g)))
            _lib.check(_lib.lib().egv_block_grad_layout(C.byref(g), off, C.byref(tot)), "egv_block_grad_layout")
            if nb <= 0:
                raise _lib.EgovlpHipError("egv_block_fwd_arena_bytes: unsupported block geometry")
            ent = _BLOCK_CACHE[key] = (nb, tuple(int(o) for o in off), int(tot.value))
        arena = torch.empty(ent[0], dtype=torch.uint8, device=dev)
        out = torch.empty((M, D), dtype=torch.float32, device=dev)
        ln = (n3w, n3b, n1w, n1b, n2w, n2b)
        biases = (tqkv_b, tproj_b, sqkv_b, sproj_b, fc1_b, fc2_b)
        weights = (tqkv_w, tproj_w, sqkv_w, sproj_w, fc1_w, fc2_w)
        prm = _block_params(ec.wc, ln, biases, weights, need_t=False, x2=P == 2, proj_x2=bool(single & 8) or Pb == 4)
        _lib.check(_lib.lib().egv_block_fwd(C.byref(g), C.byref(prm), x2.data_ptr(), out.data_ptr(), arena.data_ptr(), ops._stream(x2)),
                   "egv_block_fwd")
        if train:
            ctx.key, ctx.eps, ctx.ec, ctx.arena, ctx.sizes = key, eps, ec, arena, ent
            ctx.save_for_backward(x2, *ln, *biases, *weights)
        return out.view(B, S, D)

    @staticmethod
    def backward(ctx, g_out):
        import ctypes as C
        from .. import _lib
        from .._lib import BlockBwdIO
        saved = ctx.saved_tensors
        x2, ln, biases, weights = saved[0], saved[1:7], saved[7:13], saved[13:19]
        if ctx.arena is None:
            raise RuntimeError("the C block calls release their forward workspace after the first backward: a second backward through "
                               "the same graph (retain_graph=True) needs the per-kernel path (exec_ctx.set(block_calls=False))")
        B, T, n, H, D, Hd, P, Pb, train, z_bf16, single = ctx.key
        ec = ctx.ec
        ec.poll_backward()              # gradients of the blocks behind this one are final: the data-parallel exchange may start
        Pb_now = ec.bwd_passes
        if Pb_now != Pb:
            raise RuntimeError("the backward precision changed between this block's forward and its backward")
        S = 1 + T * n
        M = B * S
        dev = x2.device
        G = g_out.contiguous().view(M, D)
        # the gradient-plane hand-over of the block behind this one (see _attach_grad_planes)
        g_hi = g_lo = None
        ent = getattr(g_out, "_egv_planes", None)
        if ent is not None:
            try:
                del g_out._egv_planes
            except AttributeError:
                pass
            if ent[0] == Pb and ent[2] == g_out._version and ent[1].rows == M and ent[1].cols == D:
                PLANE_HANDOFF["hit"] += 1
                g_pl = ent[1]
                g_hi, g_lo = g_pl.hi.data_ptr(), (g_pl.lo.data_ptr() if g_pl.lo is not None else None)
        if g_hi is None:
            PLANE_HANDOFF["miss"] += 1
        # weight gradients in the order the C side enqueues them (fc2, fc1, attn.proj, attn.qkv, timeattn.proj, timeattn.qkv): stream,
        # event and k-slices of each -- the policy of _lin_bwd / ops.gemm_tn
        shapes = [(w.shape[0], w[0].numel()) for w in weights]
        order = (5, 4, 3, 2, 1, 0)
        side_ok = ec.wgrad_side_stream and not ec.on_text_stream()
        use = [side_ok and weights[i].grad is None for i in range(6)]
        deal = ec.assign_side_streams([float(M) * shapes[i][0] * shapes[i][1] for i in order if use[i]]) if any(use) else []
        streams, events = [None] * 6, [None] * 6
        it = iter(deal)
        for i in order:
            if use[i]:
                streams[i], events[i] = next(it)
        ks = [ops.wgrad_ksplit(shapes[i][0], shapes[i][1], M, ec, use[i]) for i in range(6)]
        g = _block_geom(*ctx.key, ctx.eps, ec.gemm_grid)
        bkey = (ctx.key, tuple(ks))
        nb = _BLOCK_BWD_BYTES.get(bkey)
        if nb is None:
            nb = _BLOCK_BWD_BYTES[bkey] = int(_lib.lib().egv_block_bwd_arena_bytes(C.byref(g), (C.c_int32 * 6)(*ks)))
        barena = torch.empty(nb, dtype=torch.uint8, device=dev)
        _, goff, gtot = ctx.sizes
        grads = torch.empty(gtot, dtype=torch.float32, device=dev)
        d_x = torch.empty((M, D), dtype=torch.float32, device=dev)
        dx_pl = ops.empty_planes_f16x2(M, D, dev, single=True) if Pb == 4 else ops.empty_planes(M, D, Pb, dev)
        used = {s_ for s_ in streams if s_ is not None}          # the side streams read / write these allocations of the main stream
        if used:
            # held until the side streams are joined (end of backward) instead of record_stream-ed: the multi-GB arenas (see
            # hold_until_join), the gradient buffer, and the gradient planes handed over by the block behind this one -- BOTH planes (a
            # bf16x3 backward reads the lo plane in the fc2 weight gradient as well).  record_stream leaves an event per block and
            # stream with the caching allocator, which polls every outstanding event on every allocation: with 24 blocks and a host
            # that runs two steps ahead that was ~10 ms of host time per ViT-L/14 step (host_enqueue 26 ms against 16 from idle).
            ec.hold_until_join(ctx.arena, barena, grads, *((g_pl.hi, g_pl.lo) if g_hi is not None else ()))
        prm = _block_params(ec.wc, ln, biases, weights, need_t=True, x2=P == 2, proj_x2=bool(single & 8) or Pb == 4, t16=Pb == 4)
        P6 = C.c_void_p * 6
        io = BlockBwdIO(G.data_ptr(), g_hi, g_lo, x2.data_ptr(), ctx.arena.data_ptr(), barena.data_ptr(),
                        d_x.data_ptr(), dx_pl.hi.data_ptr(), dx_pl.lo.data_ptr() if dx_pl.lo is not None else None, grads.data_ptr(),
                        P6(*[s_.cuda_stream if s_ is not None else None for s_ in streams]),
                        P6(*[e.cuda_event if e is not None else None for e in events]), (C.c_int32 * 6)(*ks))
        _lib.check(_lib.lib().egv_block_bwd(C.byref(g), C.byref(prm), C.byref(io), ops._stream(x2)), "egv_block_bwd")
        ctx.arena = None

        sizes = [goff[i + 1] - goff[i] for i in range(17)] + [gtot - goff[17]]
        parts = grads.split_with_sizes(sizes)                 # 18 views of the one buffer (the layout is back to back)
        dW = [parts[i].view(weights[i].shape) for i in range(6)]
        db = parts[6:12]
        dln = parts[12:18]                                  # norm3 g/b, norm1 g/b, norm2 g/b
        return (_attach_grad_planes(d_x.view(B, S, D), Pb, dx_pl), None, None,
                dln[0], dln[1], dW[0], db[0], dW[1], db[1],
                dln[2], dln[3], dW[2], db[2], dW[3], db[3],
                dln[4], dln[5], dW[4], db[4], dW[5], db[5])


class _PatchTokensFn(torch.autograd.Function):
    """VideoPatchEmbed (:72-77) + flatten/CLS/pos/temporal (:305-320): patch gather -> MFMA GEMM (+bias)
    -> token assembly.  No gradient flows to the input frames."""

    @staticmethod
    def forward(ctx, video, geom, ec, proj_w, proj_b, cls_token, pos_embed, temporal_embed):
        B, T, n, P_, D, T_model = geom[:6]
        Pp = ec.fwd_passes_split
        wc = ec.wc
        mean, std = geom[6] if len(geom) > 6 else (ops.IMAGENET_MEAN, ops.IMAGENET_STD)
        aug = geom[7] if len(geom) > 7 else None
        # uint8 frames: /255 + Normalize (and, with `aug`, the train transform's crop / resize / flip) inside the gather
        a = ops.patch_gather(video.contiguous(), P_, Pp, mean, std, aug=aug)
        K = proj_w[0].numel()
        if a.cols == K:
            w_pl = wc.get(proj_w, need_t=False)[0]
        else:   # ViT-L/14: K = 588 is zero-padded to the 64-deep k-tile on both operands
            w_pl = ops.split_f32(torch.nn.functional.pad(proj_w.detach().reshape(D, K), (0, a.cols - K)), Pp)[0]
        pe = torch.empty((a.rows, D), dtype=torch.float32, device=video.device)
        ops.gemm_nt(a, w_pl, passes=Pp, bias=proj_b, out_f32=pe, ec=ec)
        x = ops.assemble_tokens(pe, cls_token, pos_embed, temporal_embed, B, T, n, D)
        ctx.geom, ctx.a, ctx.Pp, ctx.ec = geom, a, Pp, ec
        ctx.wshape, ctx.proj_w = proj_w.shape, proj_w
        return x

    @staticmethod
    def backward(ctx, dx):
        B, T, n, P_, D, T_model = ctx.geom[:6]
        ec = ctx.ec
        # next to an fp16 backward of the blocks: ONE bf16 product.  This wgrad is the last GEMM of backward (nothing left to hide it under),
        # its dY carries the blocks' 2.5e-3 already, and three products would cost 0.12 ms on the tail for 3.5e-3 -> 2.5e-3 on this one tensor
        Pb = 1 if ec.bwd_passes == 4 else ec.bwd_passes
        ec.poll_backward()              # every block's gradients are final here
        d_pe, d_cls, d_pos, d_tmp = ops.assemble_tokens_bwd(dx.contiguous(), B, T, n, D, T_model)
        K = ctx.wshape[1] * ctx.wshape[2] * ctx.wshape[3]
        # a zero-padded K (ViT-L/14: 588 -> 640) is cut off dW right here, on THIS stream: that wgrad must not run on a side stream
        # (bench.py's grad_rel_err had this gradient 100 % off in config 5 with the wgrad side streams on, rounds 3 - 4)
        _, d_w, d_b = _lin_bwd(d_pe, ctx.a, None, Pb, need_dx=False, params=(ctx.proj_w,), ec=ec, allow_side=(ctx.a.cols == K))
        if d_w.shape[1] != K:
            d_w = d_w[:, :K].contiguous()      # drop the zero-padded k columns
        return None, None, None, d_w.view(ctx.wshape), d_b, d_cls, d_pos, d_tmp


class _ClsNormFn(torch.autograd.Function):
    """`self.norm(x)[:, 0]` (:330): LayerNorm is per token, so only the B CLS rows are normalised."""

    @staticmethod
    def forward(ctx, x, w, b, eps, ec):
        B, S, D = x.shape
        xc = x.contiguous()
        _, y, mean, rstd, _ = ops.layernorm_fwd(xc.view(B * S, D), w, b, eps, 1, want_f32=True, want_planes=False,
                                                rows=B, ldx=S * D)
        ctx.save_for_backward(xc, w, mean, rstd)
        ctx.ec = ec
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, w, mean, rstd = ctx.saved_tensors
        B, S, D = xc.shape
        ctx.ec.poll_backward()          # first node of the video tower's backward: what ran before it (the text tower) is final
        dx = ops.zeros(tuple(xc.shape), device=xc.device)     # only the B CLS rows receive a gradient
        _, dg, db = ops.layernorm_bwd(dy.contiguous(), xc.view(B * S, D), w, mean, rstd, rows=B, ldx=S * D,
                                      dx=dx.view(B * S, D), lddx=S * D)
        return dx, dg, db, None, None


def to_2tuple(x):
    return x if isinstance(x, tuple) else (x, x)


class Mlp(nn.Module):
    """Parameter container for model/video_transformer.py:36-52 (fc1 -> GELU -> fc2)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        if drop != 0.:
            raise NotImplementedError("dropout > 0 is not on the EgoClip hot path (video drop rates are all 0)")
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, out_features)


class VideoPatchEmbed(nn.Module):
    """model/video_transformer.py:55-77."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, num_frames=8):
        super().__init__()
        img_size = to_2tuple(img_size)
        patch_size = to_2tuple(patch_size)
        self.img_size = img_size
        self.patch_size = patch_size
        self.num_patches = (img_size[1] // patch_size[1]) * (img_size[0] // patch_size[0]) * num_frames
        self.num_frames = num_frames
        self.embed_dim = embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)


class VarAttention(nn.Module):
    """Parameter container for model/video_transformer.py:80-98 (incl. the 'zeros' initialisation :90-96)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.,
                 initialize='random'):
        super().__init__()
        if attn_drop != 0. or proj_drop != 0.:
            raise NotImplementedError("attention dropout is 0 on the EgoClip hot path")
        if dim // num_heads != 64:
            raise NotImplementedError("the gfx950 attention kernels are built for head_dim 64 (ViT-B/16, ViT-L/14)")
        if qk_scale is not None and qk_scale != 64 ** -0.5:
            raise NotImplementedError("qk_scale override is not supported")
        self.num_heads = num_heads
        self.scale = 64 ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        if initialize == 'zeros':
            self.qkv.weight.data.fill_(0)
            self.qkv.bias.data.fill_(0)
            self.proj.weight.data.fill_(1)
            self.proj.bias.data.fill_(0)


class SpaceTimeBlock(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm, time_init='zeros',
                 attention_style='frozen-in-time'):
        super().__init__()
        if drop_path != 0.:
            raise NotImplementedError("stochastic depth is 0 on the EgoClip hot path")
        if attention_style != 'frozen-in-time':
            raise NotImplementedError  # model/video_transformer.py:173
        self.norm1 = norm_layer(dim)
        self.attn = VarAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                 attn_drop=attn_drop, proj_drop=drop)
        self.timeattn = VarAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                     attn_drop=attn_drop, proj_drop=drop, initialize=time_init)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.norm3 = norm_layer(dim)
        self.num_heads = num_heads
        self.attention_style = attention_style

    def forward(self, x, B, T, n, ec):
        # which Linears of THIS block run one fp16 product in the f16x2 mode: the model's precision policy (ops.single_product_policy)
        single = ec.f16_single_mask(getattr(self, "layer_index", None), getattr(self, "depth", None)) if ec.fwd_passes == 2 else 0
        geom = (B, T, n, self.num_heads, self.norm1.eps, single)
        fn = _SpaceTimeBlockCFn if block_calls_ok(ec, B * (1 + T * n), x.shape[-1], self.mlp.fc1.weight.shape[0]) else _SpaceTimeBlockFn
        return fn.apply(
            x, geom, ec,
            self.norm3.weight, self.norm3.bias, self.timeattn.qkv.weight, self.timeattn.qkv.bias,
            self.timeattn.proj.weight, self.timeattn.proj.bias,
            self.norm1.weight, self.norm1.bias, self.attn.qkv.weight, self.attn.qkv.bias,
            self.attn.proj.weight, self.attn.proj.bias,
            self.norm2.weight, self.norm2.bias, self.mlp.fc1.weight, self.mlp.fc1.bias,
            self.mlp.fc2.weight, self.mlp.fc2.bias)


class SpaceTimeTransformer(nn.Module):
    """Drop-in for model/video_transformer.py:180-338 (same ctor signature :196-199)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=True, qk_scale=None, representation_size=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0., hybrid_backbone=None, norm_layer=None,
                 num_frames=8, time_init='rand', attention_style='frozen-in-time'):
        super().__init__()
        if hybrid_backbone is not None:
            raise NotImplementedError('hybrid backbone not implemented')       # :231
        if drop_rate != 0. or attn_drop_rate != 0. or drop_path_rate != 0.:
            raise NotImplementedError("non-zero video drop rates are not on the EgoClip hot path (model/model.py:49-51)")
        if representation_size:
            raise NotImplementedError("representation_size (pre_logits) is not on the EgoClip hot path")
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.num_frames = num_frames
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)             # :228
        self.patch_embed = VideoPatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans,
                                           embed_dim=embed_dim, num_frames=num_frames)
        num_patches = self.patch_embed.num_patches
        self.patches_per_frame = num_patches // num_frames
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patches_per_frame + 1, embed_dim))
        self.temporal_embed = nn.Parameter(torch.zeros(1, num_frames, embed_dim))
        self.blocks = nn.ModuleList([
            SpaceTimeBlock(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                           qk_scale=qk_scale, norm_layer=norm_layer, time_init=time_init,
                           attention_style=attention_style)
            for _ in range(depth)])
        for i, blk in enumerate(self.blocks):
            blk.layer_index, blk.depth = i, depth     # the precision policy is per block (ExecContext.f16_single_mask)
        self.norm = norm_layer(embed_dim)
        self.pre_logits = nn.Identity()
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
        nn.init.trunc_normal_(self.pos_embed, std=.02, a=-2., b=2.)            # :264-265
        nn.init.trunc_normal_(self.cls_token, std=.02, a=-2., b=2.)
        if num_frames == 1:                                                    # :272-273
            self.apply(self._init_weights)
        self.exec_ctx = ops.new_context()     # FrozenInTime replaces it with the dual encoder's shared context

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02, a=-2., b=2.)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def no_weight_decay(self):
        return {'pos_embed', 'cls_token'}

    def set_input_augmentation(self, boxes, out_res=None):
        """Fuse the loader's train transform into the NEXT forward: `boxes` int32 [B, 5] (top, left, h, w, flip; see
        egovlp_amd.data_loader.transforms.train_transform_params) select, per clip, the region of the decoded uint8 frames
        that is resized to `out_res` (default: the model's img_size), flipped and normalised inside the patch gather.
        One-shot: consumed by the next forward_features call."""
        host = boxes.detach().to(dtype=torch.int64).cpu() if not boxes.is_cuda else None     # checked against the frame size in forward
        if host is not None and (host.dim() != 2 or host.shape[1] != 5 or bool((host[:, :2] < 0).any()) or bool((host[:, 2:4] < 1).any())):
            raise ValueError("set_input_augmentation: boxes are int [B, 5] rows (top >= 0, left >= 0, h >= 1, w >= 1, flip)")
        self._input_aug = (boxes.to(device=self.cls_token.device, dtype=torch.int32).contiguous(),
                           int(out_res or self.patch_embed.img_size[0]), host)

    def forward_features(self, x):
        b, curr_frames, channels, Hh, Ww = x.shape
        assert curr_frames <= self.num_frames                                  # :74
        P_ = self.patch_embed.patch_size[0]
        aug = getattr(self, "_input_aug", None)
        self._input_aug = None
        if aug is not None:
            if x.dtype != torch.uint8:
                raise ValueError("set_input_augmentation expects decoded uint8 frames")
            host = aug[2]
            if host is not None and (host.shape[0] != b or bool((host[:, 0] + host[:, 2] > Hh).any())
                                     or bool((host[:, 1] + host[:, 3] > Ww).any())):
                raise ValueError(f"set_input_augmentation: a crop box leaves the {Hh} x {Ww} frame (or the batch size changed)")
            aug = aug[:2]
            Hh = Ww = aug[1]                                                   # the resized crop is what gets patched
        n = (Hh // P_) * (Ww // P_)
        if n != self.patches_per_frame:
            raise NotImplementedError("input resolution must match the positional embedding")
        # `input_norm` = (mean, std) of the loader's Normalize (data_loader/transforms.py:34-39); only used when the frames
        # arrive as decoded uint8 (then x / 255 and the normalisation are fused into the patch gather on the device)
        geom = (b, curr_frames, n, P_, self.embed_dim, self.num_frames,
                getattr(self, "input_norm", (ops.IMAGENET_MEAN, ops.IMAGENET_STD)), aug)
        ec = self.exec_ctx
        x = _PatchTokensFn.apply(x, geom, ec, self.patch_embed.proj.weight, self.patch_embed.proj.bias,
                                 self.cls_token, self.pos_embed, self.temporal_embed)
        for blk in self.blocks:                                                # :325-328
            x = blk(x, b, curr_frames, n, ec)
        x = _ClsNormFn.apply(x, self.norm.weight, self.norm.bias, self.norm.eps, ec)   # :330
        return self.pre_logits(x)

    def forward(self, x):
        x = self.forward_features(x)
        if not isinstance(self.head, nn.Identity):
            raise NotImplementedError("classification head: FrozenInTime replaces it with Identity (model/model.py:55)")
        return x
