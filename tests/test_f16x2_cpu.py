"""The arithmetic of the f16x2 operand format on the CPU (tests/f16x2_ref.py = the device encoders restated; tests/quant_emul.py =
the product scheme inside the oracle): the claim the kernels rest on -- two fp16 products give the accuracy class of three bf16
products -- holds on random operands, on badly scaled ones, and end to end on the oracle's forward."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (HERE, os.path.dirname(HERE)):
    if p not in sys.path:
        sys.path.insert(0, p)
import f16x2_ref as R                      # noqa: E402
from quant_emul import Scheme              # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def test_two_fp16_products_match_three_bf16_products():
    g = torch.Generator().manual_seed(0)
    a = torch.randn(384, 768, generator=g) * 3.0
    b = torch.randn(256, 768, generator=g) * 0.02
    ref = a.double() @ b.double().t()
    a1, a2, _ = R.encode(a, 0)
    b1, b2, _ = R.encode(b, 1)
    x2 = rel(R.product(a1, a2, b1, b2), ref)
    x3 = rel(Scheme("bf16x3").bmm(a, b), ref)
    one = rel(Scheme("fp16").bmm(a, b), ref)
    assert x2 < 1e-5 and x2 < 2 * x3 and one > 20 * x2, (x2, x3, one)
    # the emulator's scheme is the same arithmetic
    assert rel(Scheme("fp16x2:6").bmm(a, b), R.product(a1, a2, b1, b2)) < 1e-6
    # the roles matter: the construction is not symmetric
    a1w, a2w, _ = R.encode(a, 1)
    assert rel(R.product(a1w, a2w, b1, b2), ref) > 10 * x2


def test_planes_carry_the_value():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(64, 256, generator=g) * torch.logspace(-3, 2, 64).unsqueeze(1)
    a1, a2, bf = R.encode(x, 0)
    assert rel(a1.double() + a2.double(), x) < 2e-5                      # a1 + a2 = x up to fp16's rounding of the small part
    assert rel(a1.double() / (1.0 - R.E), x) < 5e-4
    b1, b2, _ = R.encode(x, 1)
    assert rel(b1.double() + (b2.double() - b1.double()) * R.E, x) < 2e-5   # b1 + e (b2 - b1) = x
    assert torch.equal(bf, x.to(torch.bfloat16))
    # saturation: the fp16 planes clamp at 65504, the bf16 copy keeps the value
    big = torch.tensor([[1.0e5, -2.0e5, 65504.0, 1.0]])
    p1, p2, pb = R.encode(big, 1)
    assert float(p1[0, 0]) == 65504.0 and float(p1[0, 1]) == -65504.0 and float(pb[0, 0]) == float(torch.tensor(1.0e5).to(torch.bfloat16))


def test_small_magnitudes_lose_relative_not_absolute_accuracy():
    g = torch.Generator().manual_seed(2)
    a = torch.randn(128, 512, generator=g) * 1e-4         # activations far below the format's comfortable range
    b = torch.randn(64, 512, generator=g) * 0.02
    ref = a.double() @ b.double().t()
    a1, a2, _ = R.encode(a, 0)
    b1, b2, _ = R.encode(b, 1)
    err = (R.product(a1, a2, b1, b2) - ref).abs().max()
    assert float(err) < 1e-7                               # measured 3e-8 at every scale from 1e-2 down to 1e-5: the absolute error does not grow
    r = rel(R.product(a1, a2, b1, b2), ref)
    assert 5e-5 < r < 5e-4, r                              # ... the relative accuracy degrades (1.7e-4 here, 5e-6 for O(1) activations)
