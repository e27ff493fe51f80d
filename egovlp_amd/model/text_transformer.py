"""DistilBERT text encoder on MI355X kernels -- drop-in for the HF `DistilBertModel` the reference
builds at model/model.py:31-36 and calls at :122 (`text_model(**text).last_hidden_state`).

The arithmetic follows HF transformers' modeling_distilbert.py (pinned 4.2.1 in the reference's
environment.yml:60; container copy 5.15: Embeddings :82-118, eager attention :122-147,
DistilBertSelfAttention :150-203, FFN :206-224, TransformerBlock :227-259, post-LN, eps 1e-12,
exact-erf GELU).  Parameter names/shapes equal HF's, so `text_model.*` checkpoint keys load unchanged.

Deviation (documented in DESIGN.md): dropout / attention_dropout are 0 here; HF's default 0.1 makes the
reference's train-mode forward stochastic and is not part of the parity contract.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch
from torch import nn

from .. import ops
from ..ops import ACT_GELU, ACT_GELU_BWD, Precision
from ..weights import WeightCache
from .video_transformer import _lin_bwd


class DistilBertConfig:
    def __init__(self, vocab_size=30522, max_position_embeddings=512, dim=768, n_layers=6, n_heads=12,
                 hidden_dim=3072, dropout=0.0, attention_dropout=0.0, pad_token_id=0):
        self.vocab_size = vocab_size
        self.max_position_embeddings = max_position_embeddings
        self.dim = dim
        self.hidden_size = dim          # model/model.py:74 reads config.hidden_size
        self.n_layers = n_layers
        self.n_heads = n_heads
        self.hidden_dim = hidden_dim
        self.dropout = dropout
        self.attention_dropout = attention_dropout
        self.pad_token_id = pad_token_id


class _EmbedFn(torch.autograd.Function):
    """Embeddings.forward: LN(word[ids] + pos[:L]), eps 1e-12."""

    @staticmethod
    def forward(ctx, ids, word, pos, ln_w, ln_b, eps, pad_id):
        B, L = ids.shape
        D = word.shape[1]
        e = ops.embed_fwd(ids.contiguous(), word, pos, D)
        _, y, mean, rstd, _ = ops.layernorm_fwd(e, ln_w, ln_b, eps, 1, want_f32=True, want_planes=False)
        ctx.save_for_backward(ids, e, ln_w, mean, rstd)
        ctx.shapes = (word.shape, pos.shape, pad_id)
        return y.view(B, L, D)

    @staticmethod
    def backward(ctx, dy):
        ids, e, ln_w, mean, rstd = ctx.saved_tensors
        D = e.shape[1]
        de, dg, db = ops.layernorm_bwd(dy.contiguous().view(-1, D), e, ln_w, mean, rstd)
        d_word, d_pos = ops.embed_bwd(ids.contiguous(), de, ctx.shapes[0], ctx.shapes[1], ctx.shapes[2])
        return None, d_word, d_pos, dg, db, None, None


class _TextLayerFn(torch.autograd.Function):
    """TransformerBlock.forward (post-LN):  sa = LN(out_lin(MHA(x)) + x);  out = LN(lin2(gelu(lin1(sa))) + sa)."""

    @staticmethod
    def forward(ctx, x, mask, geom, wc: WeightCache,
                q_w, q_b, k_w, k_b, v_w, v_b, o_w, o_b, ln1_w, ln1_b, f1_w, f1_b, f2_w, f2_b, ln2_w, ln2_b):
        B, L, H, eps = geom
        D = x.shape[-1]
        M = B * L
        P = Precision.fwd_passes
        dev = x.device
        x2 = x.contiguous().view(M, D)
        train = any(ctx.needs_input_grad)

        def W(p):
            return wc.get(p, need_t=False)[0]

        x_pl, _, _ = ops.split_f32(x2, P)
        # q_lin / k_lin / v_lin as ONE [M, 768] x [2304, 768]^T GEMM (M = B*L = 1024 is latency-bound: three launches -> one)
        qkv = torch.empty((M, 3 * D), dtype=torch.float32, device=dev)
        ops.gemm_nt(x_pl, wc.get_cat((q_w, k_w, v_w), need_t=False)[0], passes=P, bias=torch.cat((q_b, k_b, v_b)),
                    out_f32=qkv)
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        c_pl, lse = ops.text_attn_fwd(q, k, v, mask, B, L, H, P)
        s1 = torch.empty((M, D), dtype=torch.float32, device=dev)
        ops.gemm_nt(c_pl, W(o_w), passes=P, bias=o_b, residual=x2, out_f32=s1)
        sa_pl, sa, mean1, rstd1, _ = ops.layernorm_fwd(s1, ln1_w, ln1_b, eps, P, want_f32=True)
        Hd = f1_w.shape[0]
        h = ops.empty_planes(M, Hd, P, dev)
        z = torch.empty((M, Hd), dtype=torch.float32, device=dev) if train else None
        ops.gemm_nt(sa_pl, W(f1_w), passes=P, bias=f1_b, act=ACT_GELU, aux_out=z, out_planes=h)
        s2 = torch.empty((M, D), dtype=torch.float32, device=dev)
        ops.gemm_nt(h, W(f2_w), passes=P, bias=f2_b, residual=sa, out_f32=s2)
        _, out, mean2, rstd2, _ = ops.layernorm_fwd(s2, ln2_w, ln2_b, eps, P, want_f32=True, want_planes=False)
        if train:
            ctx.geom, ctx.wc, ctx.P = geom, wc, P
            ctx.planes = (x_pl, c_pl, sa_pl, h)
            ctx.save_for_backward(mask, qkv, lse, s1, mean1, rstd1, z, s2, mean2, rstd2,
                                  q_w, k_w, v_w, o_w, ln1_w, f1_w, f2_w, ln2_w)
        return out.view(B, L, D)

    @staticmethod
    def backward(ctx, g_out):
        (mask, qkv, lse, s1, mean1, rstd1, z, s2, mean2, rstd2,
         q_w, k_w, v_w, o_w, ln1_w, f1_w, f2_w, ln2_w) = ctx.saved_tensors
        x_pl, c_pl, sa_pl, h = ctx.planes
        B, L, H, eps = ctx.geom
        wc = ctx.wc
        Pb = Precision.bwd_passes
        if Pb > ctx.P:
            raise RuntimeError("backward precision bf16x3 needs a bf16x3 forward")
        M, D = s1.shape
        G = g_out.contiguous().view(M, D)

        def Wt(p):
            return wc.get(p, need_t=True)[1]

        d_s2, d_ln2w, d_ln2b = ops.layernorm_bwd(G, s2, ln2_w, mean2, rstd2)
        # FFN
        g_pl = ops.split_f32(d_s2, Pb)[0]
        Hd = f1_w.shape[0]
        dZ = ops.empty_planes(M, Hd, Pb, G.device)
        ops.gemm_nt(g_pl, Wt(f2_w), passes=Pb, act=ACT_GELU_BWD, aux_in=z, out_planes=dZ, K=D)
        _, d_f2w, d_f2b = _lin_bwd(g_pl, h, None, Pb, need_dx=False)
        _, d_f1w, d_f1b = _lin_bwd(dZ, sa_pl, None, Pb, need_dx=False)
        d_sa = torch.empty((M, D), dtype=torch.float32, device=G.device)     # = d_s2 + dZ . W1
        ops.gemm_nt(dZ, Wt(f1_w), passes=Pb, residual=d_s2, out_f32=d_sa, K=Hd)
        d_s1, d_ln1w, d_ln1b = ops.layernorm_bwd(d_sa, s1, ln1_w, mean1, rstd1)
        # attention output projection
        d_ctx, d_ow, d_ob = _lin_bwd(d_s1, c_pl, Wt(o_w), Pb)
        D3 = 3 * D
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        _, _, _, dqkv = ops.text_attn_bwd(q, k, v, mask, d_ctx, lse, B, L, H, Pb, fused_out=True)
        # fused q/k/v projection backward: one wgrad (dW [2304, 768] + bias grads) and one dgrad chained onto d_s1
        dqkv_pl = ops.split_f32(dqkv, Pb)[0]
        _, dW3, db3 = _lin_bwd(dqkv_pl, x_pl, None, Pb, need_dx=False)
        acc = torch.empty((M, D), dtype=torch.float32, device=G.device)
        ops.gemm_nt(dqkv_pl, wc.get_cat((q_w, k_w, v_w), need_t=True)[1], passes=Pb, residual=d_s1, out_f32=acc, K=D3)
        grads = [dW3[:D], db3[:D], dW3[D:2 * D], db3[D:2 * D], dW3[2 * D:], db3[2 * D:]]
        return (acc.view(B, L, D), None, None, None, *grads, d_ow, d_ob, d_ln1w, d_ln1b,
                d_f1w, d_f1b, d_f2w, d_f2b, d_ln2w, d_ln2b)


class Embeddings(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.dim, padding_idx=config.pad_token_id)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.dim)
        self.LayerNorm = nn.LayerNorm(config.dim, eps=1e-12)


class MultiHeadSelfAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        if config.dim // config.n_heads != 64:
            raise NotImplementedError("head_dim must be 64")
        self.n_heads = config.n_heads
        self.q_lin = nn.Linear(config.dim, config.dim)
        self.k_lin = nn.Linear(config.dim, config.dim)
        self.v_lin = nn.Linear(config.dim, config.dim)
        self.out_lin = nn.Linear(config.dim, config.dim)


class FFN(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.lin1 = nn.Linear(config.dim, config.hidden_dim)
        self.lin2 = nn.Linear(config.hidden_dim, config.dim)


class TransformerBlock(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.attention = MultiHeadSelfAttention(config)
        self.sa_layer_norm = nn.LayerNorm(config.dim, eps=1e-12)
        self.ffn = FFN(config)
        self.output_layer_norm = nn.LayerNorm(config.dim, eps=1e-12)

    def forward(self, x, mask, wc):
        B, L, D = x.shape
        a, f = self.attention, self.ffn
        geom = (B, L, a.n_heads, self.sa_layer_norm.eps)
        return _TextLayerFn.apply(
            x, mask, geom, wc,
            a.q_lin.weight, a.q_lin.bias, a.k_lin.weight, a.k_lin.bias, a.v_lin.weight, a.v_lin.bias,
            a.out_lin.weight, a.out_lin.bias, self.sa_layer_norm.weight, self.sa_layer_norm.bias,
            f.lin1.weight, f.lin1.bias, f.lin2.weight, f.lin2.bias,
            self.output_layer_norm.weight, self.output_layer_norm.bias)


class Transformer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.layer = nn.ModuleList([TransformerBlock(config) for _ in range(config.n_layers)])


class DistilBertModel(nn.Module):
    def __init__(self, config=None):
        super().__init__()
        self.config = config or DistilBertConfig()
        if self.config.dropout != 0.0 or self.config.attention_dropout != 0.0:
            raise NotImplementedError("dropout is not implemented in the gfx950 text encoder (set to 0)")
        self.embeddings = Embeddings(self.config)
        self.transformer = Transformer(self.config)
        self._wc = WeightCache()
        # HF init (initializer_range 0.02) so random-init statistics match `DistilBertModel(DistilBertConfig())`
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Embedding):
                nn.init.normal_(m.weight, std=0.02)

    def forward(self, input_ids=None, attention_mask=None, **kw):
        if input_ids is None:
            raise NotImplementedError("inputs_embeds path is not on the EgoClip hot path")
        if input_ids.shape[1] > self.config.max_position_embeddings:
            raise ValueError("sequence longer than max_position_embeddings")
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        e = self.embeddings
        x = _EmbedFn.apply(input_ids, e.word_embeddings.weight, e.position_embeddings.weight,
                           e.LayerNorm.weight, e.LayerNorm.bias, e.LayerNorm.eps,
                           -1 if e.word_embeddings.padding_idx is None else e.word_embeddings.padding_idx)
        mask = attention_mask.to(torch.int64).contiguous()
        for blk in self.transformer.layer:
            x = blk(x, mask, self._wc)
        return SimpleNamespace(last_hidden_state=x)
