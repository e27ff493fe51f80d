"""Operand-quantisation emulator for the CPU oracle  --  TEST INFRASTRUCTURE ONLY.

Answers, without a GPU, the question "what does a forward product scheme cost in parity?": every matrix product of the
oracle's forward (the Linears of `model/video_transformer.py:41-50,103,135`, the attention products of `:29-33`, DistilBERT's
Linears and attention products, the patch-embed conv of `:70-77`, the projections of `model/model.py:72-79`) is replaced by an
fp32 emulation of what the MFMA path would compute from ROUNDED operands:

  bf16      : A_hi B_hi                                   (one bf16 product)
  bf16x3    : A_hi B_hi + A_hi B_lo + A_lo B_hi           (three bf16 products; lo = bf16(x - hi))
  fp16      : fp16(A) fp16(B)
  fp16+F    : as bf16+F with hi = fp16(x) (11 significant bits; saturating) and the residual taken against it
  fp16x2[:S]: A1 B1 + A2 B2, A1 = fp16((1 - e) A), A2 = fp16(A - A1), B1 = fp16(B), B2 = fp16(B1 + (B - B1) / e), e = 2^-S   (two fp16 products)
  fp16~F[:S]: fp16(A) fp16(B) + F(A 2^-S) F(B_lo 2^S) + F(A_lo 2^S) F(B 2^-S)   with F a PLAIN fp8 format (e5m2 / e4m3), fixed scales
  bf16+F    : A_hi B_hi + q_F(A) q_F(B_lo) + q_F(A_lo) q_F(B)   with F an MX element format (e4m3 / e5m2 / e2m3 / e3m2 / e2m1):
              q_F = OCP-MX block quantisation, one E8M0 scale per 32 consecutive k-elements -- the operand form of gfx950's
              `v_mfma_scale_f32_16x16x128_f8f6f4`; B_lo = B - B_hi is taken in fp32 BEFORE quantising.

Products of the rounded operands are exact in fp32 and the accumulation is fp32 on both sides, so the emulation differs from the
hardware only by summation order.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

# element formats: (exponent bits, mantissa bits, bias, largest finite value)
FORMATS = {
    "e4m3": (4, 3, 7, 448.0),      # OCP e4m3fn
    "e5m2": (5, 2, 15, 57344.0),
    "e2m3": (2, 3, 1, 7.5),        # fp6
    "e3m2": (3, 2, 3, 28.0),       # bf6
    "e2m1": (2, 1, 1, 6.0),        # fp4
}
BLOCK = 32


def bf16_hi(x):
    return x.to(torch.bfloat16).to(torch.float32)


def minifloat(x, fmt):
    """Round-to-nearest-even onto the value grid of `fmt` (subnormals kept, saturating)."""
    ebits, mbits, bias, vmax = FORMATS[fmt]
    ax = x.abs()
    _, e = torch.frexp(ax)                       # ax = m * 2^e, m in [0.5, 1)
    e = (e - 1).clamp(min=1 - bias)              # exponent of the leading bit, floor at the subnormal binade
    step = torch.ldexp(torch.ones_like(ax), e - mbits)
    q = (torch.round(ax / step) * step).clamp(max=vmax)
    return torch.where(x < 0, -q, q)


def mx_quant(x, fmt, scaling="ceil"):
    """OCP-MX block quantisation along the LAST dim (must be a multiple of 32 or is zero-padded).  Returns the dequantised
    fp32 tensor (element * 2^scale).  `scaling`: 'ceil' picks the smallest power of two that avoids saturation,
    'floor' is the OCP reference (shared exponent = floor(log2(amax)) - emax_elem, saturating elements), 'none' = no scale."""
    _, _, _, vmax = FORMATS[fmt]
    if scaling == "none":
        return minifloat(x, fmt)
    K = x.shape[-1]
    pad = (-K) % BLOCK
    xp = F.pad(x, (0, pad)) if pad else x
    blk = xp.reshape(*xp.shape[:-1], -1, BLOCK)
    amax = blk.abs().amax(dim=-1, keepdim=True).clamp(min=2.0 ** -120)
    if scaling == "ceil":
        _, e = torch.frexp(amax / vmax)          # amax / vmax = m 2^e, m in [0.5, 1)  ->  2^e >= amax / vmax
        e = torch.where(amax / vmax == torch.ldexp(torch.ones_like(amax), e - 1), e - 1, e)
    else:
        _, ea = torch.frexp(amax)
        _, em = torch.frexp(torch.tensor(vmax))
        e = (ea - 1) - (int(em) - 1)
    e = e.clamp(min=-127, max=127)
    scale = torch.ldexp(torch.ones_like(amax), e)
    out = minifloat(blk / scale, fmt) * scale
    out = out.reshape(xp.shape)
    return out[..., :K] if pad else out


class Scheme:
    """`name` as in the module docstring; `bmm(a, b)` = a @ b^T over the last dims: a [..., M, K], b [..., N, K]."""

    def __init__(self, name, scaling="ceil"):
        self.name = name
        self.scaling = scaling

    def bmm(self, a, b):
        n = self.name
        mm = lambda x, y: torch.matmul(x, y.transpose(-1, -2))
        if n == "fp32":
            return mm(a, b)
        if n == "fp16":
            return mm(a.to(torch.float16).float(), b.to(torch.float16).float())
        ah, bh = bf16_hi(a), bf16_hi(b)
        if n == "bf16":
            return mm(ah, bh)
        al, bl = a - ah, b - bh
        if n == "bf16x3":
            return mm(ah, bh) + (mm(ah, bf16_hi(bl)) + mm(bf16_hi(al), bh))
        if n.startswith("bf16x2"):               # the same two-product construction on bf16 planes ("bf16x2:5": e = 2^-5)
            _, _, sh = n.partition(":")
            e = 2.0 ** -int(sh or 5)
            a1 = bf16_hi((1.0 - e) * a)
            a2 = bf16_hi(a - a1)
            b1 = bf16_hi(b)
            b2 = bf16_hi(b1 + (b - b1) / e)
            return mm(a1, b1) + mm(a2, b2)
        if n.startswith("fp16x2"):               # TWO fp16 products: A1 B1 + A2 B2 with A1 = fp16((1 - e) A), A2 = fp16(A - A1),
            # B1 = fp16(B), B2 = fp16(B1 + (B - B1) / e), e = 2^-S ("fp16x2", "fp16x2:6"): the second product carries e A B1 (which A1
            # left out) AND A (B - B1); every rounding is either compensated or attenuated by e -> ~2^-17 relative per product
            _, _, sh = n.partition(":")
            e = 2.0 ** -int(sh or 6)
            h = lambda x: x.clamp(-65504.0, 65504.0).to(torch.float16).float()
            a1 = h((1.0 - e) * a)
            a2 = h(a - a1)
            b1 = h(b)
            b2 = h(b1 + (b - b1) / e)
            return mm(a1, b1) + mm(a2, b2)
        if n.startswith("fp16~"):                # fp16 main product + two PLAIN (unscaled) fp8 correction products with fixed
            # power-of-two scales: c = F(x 2^-S), l = F((x - fp16(x)) 2^S), so that c(A) l(B) + l(A) c(B) needs no rescaling and
            # accumulates straight into the main product's accumulator ("fp16~e5m2", "fp16~e5m2:6": S = 6 by default)
            fmt, _, sh = n[5:].partition(":")
            S = float(2 ** int(sh or 6))
            ah, bh = a.to(torch.float16).float(), b.to(torch.float16).float()
            ca, cb = minifloat(a / S, fmt), minifloat(b / S, fmt)
            la, lb = minifloat((a - ah) * S, fmt), minifloat((b - bh) * S, fmt)
            return mm(ah, bh) + (mm(ca, lb) + mm(la, cb))
        if n.startswith("fp16+"):                # fp16 main product, MX corrections of the fp16 residual
            ah, bh = a.to(torch.float16).float(), b.to(torch.float16).float()
            al, bl = a - ah, b - bh
        if n.startswith("bf16+") or n.startswith("fp16+"):
            fmt = n[5:]
            coarse, fine = fmt, fmt
            if "/" in fmt:                        # "bf16+e2m1/e4m3": format of the full-value copy / of the residual
                coarse, fine = fmt.split("/")
            q = lambda x, f: mx_quant(x, f, self.scaling)
            return mm(ah, bh) + (mm(q(a, coarse), q(bl, fine)) + mm(q(al, fine), q(b, coarse)))
        raise ValueError(n)


class Policy:
    """Maps an op kind ('qkv', 'proj', 'fc1', 'fc2', 'qk', 'pv', 'patch', 'text_lin', 'text_qk', 'text_pv', 'head')
    to a Scheme; `default` for the rest."""

    def __init__(self, default="fp32", scaling="ceil", **per_kind):
        self.default = Scheme(default, scaling)
        self.per_kind = {k: Scheme(v, scaling) for k, v in per_kind.items()}

    def __call__(self, kind, block=None):
        """`block`: index of the video block the Linear belongs to -- a per-block entry "fc1@7" overrides the kind's."""
        if block is not None:
            s = self.per_kind.get(f"{kind}@{block}")
            if s is not None:
                return s
        return self.per_kind.get(kind, self.default)


def _kind_of(name):
    if name is None:
        return "head"
    if "text_model" in name:
        return "text_lin"
    for k in ("qkv", "proj", "fc1", "fc2"):
        if f".{k}." in name or name.endswith(k + ".weight"):
            return "patch" if "patch_embed" in name else k
    return "head"


class _TorchProxy:
    def __init__(self, hook):
        self._hook = hook

    def __getattr__(self, k):
        return getattr(torch, k)

    def einsum(self, eq, a, b):
        pol = self._hook.policy
        if eq == "bid,bjd->bij":                 # QK^T  (model/video_transformer.py:30)
            return pol("qk").bmm(a, b)
        if eq == "bij,bjd->bid":                 # P V   (:32): contraction over keys
            return pol("pv").bmm(a, b.transpose(-1, -2))
        return torch.einsum(eq, a, b)

    def matmul(self, a, b):                      # DistilBERT attention (oracle distilbert()): q k^T, then w v
        pol = self._hook.policy
        self._hook.text_mm += 1
        kind = "text_qk" if self._hook.text_mm % 2 == 1 else "text_pv"
        return pol(kind).bmm(a, b.transpose(-1, -2))


class _FProxy:
    def __init__(self, hook):
        self._hook = hook

    def __getattr__(self, k):
        return getattr(F, k)

    def linear(self, x, w, b=None):
        name = self._hook.names.get(id(w))
        kind = _kind_of(name)
        blk = None
        if name is not None and ".blocks." in name:
            blk = int(name.split(".blocks.")[1].split(".")[0])
        y = self._hook.policy(kind, blk).bmm(x, w)
        return y if b is None else y + b

    def conv2d(self, x, w, b=None, stride=1):    # the patch embed (kernel = stride): a GEMM over unfolded patches
        cols = F.unfold(x, kernel_size=w.shape[-1], stride=stride).transpose(1, 2)      # [BT, n, C*p*p]
        y = self._hook.policy("patch").bmm(cols, w.reshape(w.shape[0], -1))
        if b is not None:
            y = y + b
        side = x.shape[-1] // stride
        return y.transpose(1, 2).reshape(x.shape[0], w.shape[0], side, side)


class QuantisedOracle:
    """Context manager: inside it, `oracle.egovlp_oracle` computes every product under `policy`."""

    def __init__(self, oracle_module, sd, policy):
        self.O = oracle_module
        self.names = {id(v): k for k, v in sd.items()}
        self.policy = policy
        self.text_mm = 0

    def __enter__(self):
        self._saved = (self.O.F, self.O.torch)
        self.O.F = _FProxy(self)
        self.O.torch = _TorchProxy(self)
        return self

    def __exit__(self, *exc):
        self.O.F, self.O.torch = self._saved
        return False
