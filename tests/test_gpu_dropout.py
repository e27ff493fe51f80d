"""DistilBERT's train-mode dropout in the HIP path (HF defaults 0.1 / 0.1; reference: `self.text_model.train()`, model/model.py:36).
The masks are counter-based (csrc/common.h), not PyTorch's Philox stream, so what is pinned is: p = 0 is bit-identical to the
dropout-free path, the keep rate and the 1 / (1 - p) scaling, determinism in (p, seed), and -- by finite differences through a
whole TransformerBlock with all three sites active -- that backward uses exactly the masks of the forward."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_elementwise_dropout_statistics_and_determinism():
    from egovlp_amd import ops
    x = torch.rand(1024, 768, device="cuda") + 0.5
    assert torch.equal(ops.dropout(x, 0.0, 123), x)
    p = 0.1
    y = ops.dropout(x, p, 123)
    kept = y != 0
    n = x.numel()
    rate = float(kept.float().mean())
    assert abs(rate - (1 - p)) < 4 * (p * (1 - p) / n) ** 0.5, rate
    assert torch.allclose(y[kept], x[kept] / (1 - p), rtol=1e-6, atol=0)
    assert torch.equal(ops.dropout(x, p, 123), y)                       # same (p, seed) -> same mask (what backward relies on)
    assert not torch.equal(ops.dropout(x, p, 124) != 0, kept)           # another seed -> another mask
    add = torch.randn_like(x)
    assert torch.equal(ops.dropout(x, p, 123, add=add), y + add)
    # rows / columns are not correlated: keep rate per row and per column within 5 sigma
    assert float((kept.float().mean(1) - (1 - p)).abs().max()) < 5 * (p * (1 - p) / 768) ** 0.5
    assert float((kept.float().mean(0) - (1 - p)).abs().max()) < 5 * (p * (1 - p) / 1024) ** 0.5


@pytest.mark.parametrize("passes", [1, 3])
def test_attention_probability_dropout_mask(passes):
    """q = k = 0 -> uniform weights 1 / (#unmasked keys); V = one-hot rows -> the output row IS the (dropped) weight row."""
    from egovlp_amd import ops
    B, L, H = 3, 32, 12
    D = H * 64
    q = torch.zeros(B * L, D, device="cuda")
    k = torch.zeros_like(q)
    v = torch.zeros(B * L, D, device="cuda")
    for j in range(L):
        v.view(B, L, H, 64)[:, j, :, j] = 1.0
    mask = torch.ones(B, L, dtype=torch.long, device="cuda")
    mask[1, 20:] = 0
    nkeys = mask.sum(1).float()
    ref, _ = ops.text_attn_fwd(q, k, v, mask, B, L, H, passes)
    same, _ = ops.text_attn_fwd(q, k, v, mask, B, L, H, passes, 0.0, 77)
    assert torch.equal(ref.hi, same.hi)
    p = 0.25
    out, _ = ops.text_attn_fwd(q, k, v, mask, B, L, H, passes, p, 77)
    w = out.float().view(B, L, H, 64)[..., :L].permute(0, 2, 1, 3)            # [B, H, query, key]
    base = (1.0 / nkeys).view(B, 1, 1, 1) * mask.view(B, 1, 1, L).float()
    kept = w > 0
    valid = mask.view(B, 1, 1, L).bool().expand_as(kept)
    rate = float(kept[valid].float().mean())
    nv = int(valid.sum())
    assert abs(rate - (1 - p)) < 4 * (p * (1 - p) / nv) ** 0.5, rate
    assert not bool(kept[~valid].any())
    want = base.expand_as(w) / (1 - p)
    assert torch.allclose(w[kept], want[kept], rtol=2e-2 if passes == 1 else 1e-4)
    out2, _ = ops.text_attn_fwd(q, k, v, mask, B, L, H, passes, p, 77)
    assert torch.equal(out.hi, out2.hi)


def test_transformer_block_backward_uses_the_masks_of_the_forward():
    """Directional finite differences through a whole DistilBERT block with attention and FFN dropout on (fixed seeds)."""
    from egovlp_amd.model.text_transformer import DistilBertConfig, TransformerBlock
    from egovlp_amd.ops import Precision
    from egovlp_amd.ops import new_context as WeightCache   # a fresh execution context (private weight-plane cache) per evaluation
    Precision.set("bf16x3")
    torch.manual_seed(3)
    cfg = DistilBertConfig()
    blk = TransformerBlock(cfg).cuda()
    for m_ in blk.modules():
        if isinstance(m_, torch.nn.Linear):
            torch.nn.init.normal_(m_.weight, std=0.05)
    B, L, D = 2, 16, 768
    x = torch.randn(B, L, D, device="cuda")
    mask = torch.ones(B, L, dtype=torch.long, device="cuda")
    mask[1, 11:] = 0
    w = torch.randn(B, L, D, device="cuda")
    drop = (0.3, 1234567, 0.2, 7654321)

    def f(xx, wc):
        return (blk(xx, mask, wc, drop).double() * w.double()).sum()

    xg = x.clone().requires_grad_(True)
    f(xg, WeightCache()).backward()
    g = xg.grad
    gw = blk.ffn.lin1.weight.grad.clone()
    for trial in range(3):
        d = torch.randn_like(x)
        d /= d.norm()
        eps = 0.3          # the bf16x3 products carry ~1e-5 relative noise: the step must lift the difference well above it
        with torch.no_grad():
            fd = (f(x + eps * d, WeightCache()) - f(x - eps * d, WeightCache())) / (2 * eps)
        an = (g * d).sum()
        assert abs(float(fd) - float(an)) < 2e-2 * max(abs(float(an)), 1.0), (float(fd), float(an))
    # a weight direction too (the wgrad path sees the masked gradient)
    dW = torch.randn_like(blk.ffn.lin1.weight)
    dW /= dW.norm()
    eps = 0.3
    with torch.no_grad():
        W0 = blk.ffn.lin1.weight.data.clone()
        blk.ffn.lin1.weight.data = W0 + eps * dW
        fp = f(x, WeightCache())
        blk.ffn.lin1.weight.data = W0 - eps * dW
        fm = f(x, WeightCache())
        blk.ffn.lin1.weight.data = W0
    fd = float(fp - fm) / (2 * eps)
    an = float((gw * dW).sum())
    assert abs(fd - an) < 2e-2 * max(abs(an), 1.0), (fd, an)
    # and with other seeds the function is a different one
    with torch.no_grad():
        a = blk(x, mask, WeightCache(), drop)
        b = blk(x, mask, WeightCache(), (0.3, 1, 0.2, 2))
    assert not torch.equal(a, b)


def test_text_model_train_mode_is_stochastic_and_eval_is_not():
    from egovlp_amd.model.text_transformer import DistilBertModel
    from egovlp_amd.ops import Precision
    Precision.set("bf16x3")
    m = DistilBertModel().cuda()
    assert m.config.dropout == 0.1 and m.config.attention_dropout == 0.1        # HF defaults (the reference trains with them)
    ids = torch.randint(1000, 30000, (4, 16), device="cuda")
    m.train()
    a = m(input_ids=ids).last_hidden_state
    b = m(input_ids=ids).last_hidden_state
    assert not torch.equal(a, b)
    m.eval()
    c = m(input_ids=ids).last_hidden_state
    d = m(input_ids=ids).last_hidden_state
    assert torch.equal(c, d)
    m.train().set_dropout(0.0, 0.0)
    e = m(input_ids=ids).last_hidden_state
    assert torch.equal(e, c)
