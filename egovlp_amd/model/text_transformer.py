"""DistilBERT text encoder on MI355X kernels -- drop-in for the HF `DistilBertModel` the reference
builds at model/model.py:31-36 and calls at :122 (`text_model(**text).last_hidden_state`).

The arithmetic follows HF transformers' modeling_distilbert.py (pinned 4.2.1 in the reference's
environment.yml:60; container copy 5.15: Embeddings :82-118, eager attention :122-147,
DistilBertSelfAttention :150-203, FFN :206-224, TransformerBlock :227-259, post-LN, eps 1e-12,
exact-erf GELU).  Parameter names/shapes equal HF's, so `text_model.*` checkpoint keys load unchanged.

Dropout: HF's defaults `dropout = attention_dropout = 0.1` are implemented and active in `train()` mode, as in the reference
(`self.text_model.train()`, model/model.py:36): embedding output, attention probabilities and FFN output, each with a
counter-based mask that the backward regenerates from (p, seed) instead of storing (csrc/common.h).  The masks are not
PyTorch's Philox stream -- parity tests run in `eval()` or with `set_dropout(0, 0)`; tests/test_gpu_dropout.py pins the keep
rate, the 1 / (1 - p) scaling and the forward / backward consistency.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch
from torch import nn

from .. import ops
from ..ops import ACT_GELU, ACT_GELU_BWD, ExecContext
from .video_transformer import _lin_bwd


class DistilBertConfig:
    def __init__(self, vocab_size=30522, max_position_embeddings=512, dim=768, n_layers=6, n_heads=12,
                 hidden_dim=3072, dropout=0.1, attention_dropout=0.1, pad_token_id=0):
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
    """Embeddings.forward: dropout(LN(word[ids] + pos[:L])), eps 1e-12."""

    @staticmethod
    def forward(ctx, ids, word, pos, ln_w, ln_b, eps, pad_id, drop):
        B, L = ids.shape
        D = word.shape[1]
        e = ops.embed_fwd(ids.contiguous(), word, pos, D)
        _, y, mean, rstd, _ = ops.layernorm_fwd(e, ln_w, ln_b, eps, 1, want_f32=True, want_planes=False)
        if drop[0] > 0:
            y = ops.dropout(y, drop[0], drop[1], seed_dev=drop[2])
        ctx.save_for_backward(ids, e, ln_w, mean, rstd)
        ctx.shapes = (word.shape, pos.shape, pad_id)
        ctx.drop = drop
        return y.view(B, L, D)

    @staticmethod
    def backward(ctx, dy):
        ids, e, ln_w, mean, rstd = ctx.saved_tensors
        D = e.shape[1]
        dy = dy.contiguous().view(-1, D)
        if ctx.drop[0] > 0:
            dy = ops.dropout(dy, ctx.drop[0], ctx.drop[1], seed_dev=ctx.drop[2])
        de, dg, db = ops.layernorm_bwd(dy, e, ln_w, mean, rstd)
        d_word, d_pos = ops.embed_bwd(ids.contiguous(), de, ctx.shapes[0], ctx.shapes[1], ctx.shapes[2])
        return None, d_word, d_pos, dg, db, None, None, None


class _TextLayerFn(torch.autograd.Function):
    """TransformerBlock.forward (post-LN):  sa = LN(out_lin(MHA(x)) + x);  out = LN(lin2(gelu(lin1(sa))) + sa)."""

    @staticmethod
    def forward(ctx, x, mask, geom, ec: ExecContext,
                q_w, q_b, k_w, k_b, v_w, v_b, o_w, o_b, ln1_w, ln1_b, f1_w, f1_b, f2_w, f2_b, ln2_w, ln2_b):
        B, L, H, eps, drop = geom          # drop = (attention p, attention seed, ffn p, ffn seed, device seed words | None); p = 0 outside train()
        D = x.shape[-1]
        M = B * L
        P = ec.fwd_passes_split
        wc = ec.wc
        dev = x.device
        x2 = x.contiguous().view(M, D)
        train = any(ctx.needs_input_grad)

        def W(p):
            return wc.get(p, need_t=False)[0]

        x_pl, _, _ = ops.split_f32(x2, P)
        # q_lin / k_lin / v_lin as ONE [M, 768] x [2304, 768]^T GEMM (M = B*L = 1024 is latency-bound: three launches -> one)
        qkv = torch.empty((M, 3 * D), dtype=torch.float32, device=dev)
        ops.gemm_nt(x_pl, wc.get_cat((q_w, k_w, v_w), need_t=False)[0], passes=P, bias=wc.get_bias_cat((q_b, k_b, v_b)),
                    out_f32=qkv, ec=ec)
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        c_pl, lse = ops.text_attn_fwd(q, k, v, mask, B, L, H, P, drop[0], drop[1], seed_dev=drop[4])
        s1 = torch.empty((M, D), dtype=torch.float32, device=dev)
        ops.gemm_nt(c_pl, W(o_w), passes=P, bias=o_b, residual=x2, out_f32=s1, ec=ec)
        sa_pl, sa, mean1, rstd1, _ = ops.layernorm_fwd(s1, ln1_w, ln1_b, eps, P, want_f32=True)
        Hd = f1_w.shape[0]
        h = ops.empty_planes(M, Hd, P, dev)
        z = torch.empty((M, Hd), dtype=torch.float32, device=dev) if train else None
        ops.gemm_nt(sa_pl, W(f1_w), passes=P, bias=f1_b, act=ACT_GELU, aux_out=z, out_planes=h, ec=ec)
        s2 = torch.empty((M, D), dtype=torch.float32, device=dev)
        if drop[2] > 0:        # FFN.forward: dropout(lin2(gelu(lin1(x)))), then the block's residual: s2 = drop(y) + sa
            ops.gemm_nt(h, W(f2_w), passes=P, bias=f2_b, out_f32=s2, ec=ec)
            s2 = ops.dropout(s2, drop[2], drop[3], add=sa, seed_dev=drop[4])
        else:
            ops.gemm_nt(h, W(f2_w), passes=P, bias=f2_b, residual=sa, out_f32=s2, ec=ec)
        _, out, mean2, rstd2, _ = ops.layernorm_fwd(s2, ln2_w, ln2_b, eps, P, want_f32=True, want_planes=False)
        if train:
            ctx.geom, ctx.ec, ctx.P = geom, ec, P
            ctx.planes = (x_pl, c_pl, sa_pl, h)
            ctx.save_for_backward(mask, qkv, lse, s1, mean1, rstd1, z, s2, mean2, rstd2,
                                  q_w, k_w, v_w, o_w, ln1_w, f1_w, f2_w, ln2_w)
        return out.view(B, L, D)

    @staticmethod
    def backward(ctx, g_out):
        (mask, qkv, lse, s1, mean1, rstd1, z, s2, mean2, rstd2,
         q_w, k_w, v_w, o_w, ln1_w, f1_w, f2_w, ln2_w) = ctx.saved_tensors
        x_pl, c_pl, sa_pl, h = ctx.planes
        B, L, H, eps, drop = ctx.geom
        ec = ctx.ec
        wc = ec.wc
        Pb = ec.bwd_passes_split
        if Pb > ctx.P:
            raise RuntimeError("backward precision bf16x3 needs a bf16x3 forward")
        M, D = s1.shape
        G = g_out.contiguous().view(M, D)

        def Wt(p):
            return wc.get(p, need_t=True)[1]

        d_s2, d_ln2w, d_ln2b = ops.layernorm_bwd(G, s2, ln2_w, mean2, rstd2)
        # FFN (d_s2 reaches lin2 through the dropout mask of the forward; the residual branch takes it as it is)
        g_pl = ops.split_f32(ops.dropout(d_s2, drop[2], drop[3], seed_dev=drop[4]) if drop[2] > 0 else d_s2, Pb)[0]
        Hd = f1_w.shape[0]
        dZ = ops.empty_planes(M, Hd, Pb, G.device)
        ops.gemm_nt(g_pl, Wt(f2_w), passes=Pb, act=ACT_GELU_BWD, aux_in=z, out_planes=dZ, K=D, ec=ec)
        _, d_f2w, d_f2b = _lin_bwd(g_pl, h, None, Pb, need_dx=False, params=(f2_w,), ec=ec)
        _, d_f1w, d_f1b = _lin_bwd(dZ, sa_pl, None, Pb, need_dx=False, params=(f1_w,), ec=ec)
        d_sa = torch.empty((M, D), dtype=torch.float32, device=G.device)     # = d_s2 + dZ . W1
        ops.gemm_nt(dZ, Wt(f1_w), passes=Pb, residual=d_s2, out_f32=d_sa, K=Hd, ec=ec)
        d_s1, d_ln1w, d_ln1b = ops.layernorm_bwd(d_sa, s1, ln1_w, mean1, rstd1)
        # attention output projection
        d_ctx, d_ow, d_ob = _lin_bwd(d_s1, c_pl, Wt(o_w), Pb, params=(o_w,), ec=ec)
        D3 = 3 * D
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        _, _, _, dqkv = ops.text_attn_bwd(q, k, v, mask, d_ctx, lse, B, L, H, Pb, fused_out=True, dropout_p=drop[0],
                                          seed=drop[1], seed_dev=drop[4])
        # fused q/k/v projection backward: one wgrad (dW [2304, 768] + bias grads) and one dgrad chained onto d_s1
        dqkv_pl = ops.split_f32(dqkv, Pb)[0]
        _, dW3, db3 = _lin_bwd(dqkv_pl, x_pl, None, Pb, need_dx=False, params=(q_w, k_w, v_w), ec=ec)
        acc = torch.empty((M, D), dtype=torch.float32, device=G.device)
        ops.gemm_nt(dqkv_pl, wc.get_cat((q_w, k_w, v_w), need_t=True)[1], passes=Pb, residual=d_s1, out_f32=acc, K=D3, ec=ec)
        grads = [dW3[:D], db3[:D], dW3[D:2 * D], db3[D:2 * D], dW3[2 * D:], db3[2 * D:]]
        return (acc.view(B, L, D), None, None, None, *grads, d_ow, d_ob, d_ln1w, d_ln1b,
                d_f1w, d_f1b, d_f2w, d_f2b, d_ln2w, d_ln2b)


# ---------------------------------------------------------------------------------------------- one C call per layer
_TEXT_CACHE = {}       # geometry key -> (forward arena bytes, backward arena bytes, gradient offsets, gradient floats)


def text_calls_ok(ec: ExecContext, M, D, Hd, H):
    """May this layer run through the C layer calls?  (split-bf16 / bf16 precision, no per-kernel timer, widths the TN kernel takes
    without an explicit transpose, and the weight gradients on the layer's own stream: the text tower on its side stream -- the
    default -- or no wgrad side streams at all; everything else keeps the per-kernel path)"""
    if not ec.block_calls or ec.kernel_timer is not None or ec.fwd_passes_split not in (1, 3) or ec.bwd_passes_split > ec.fwd_passes_split:
        return False
    if D < 256 or Hd < 256 or D % 64 or Hd % 64 or D != H * 64:
        return False
    return ec.on_text_stream() or not ec.wgrad_side_stream


def _text_geom(B, L, H, D, Hd, P, Pb, train, eps, drop, ec):
    from .._lib import TextGeom
    import ctypes as C
    M = B * L
    I4 = C.c_int32 * 4
    ksf = I4(*[ops.auto_ksplit_nt(M, n_, k_) for n_, k_ in ((3 * D, D), (D, D), (Hd, D), (D, Hd))])
    ksb = I4(*[ops.auto_ksplit_nt(M, n_, k_) for n_, k_ in ((Hd, D), (D, Hd), (D, D), (D, 3 * D))])
    ksw = I4(*[ops.wgrad_ksplit(n_, k_, M, ec, False) for n_, k_ in ((3 * D, D), (D, D), (Hd, D), (D, Hd))])
    sdev = drop[4]
    return TextGeom(B, L, H, D, Hd, P, Pb, int(train), float(eps), float(drop[0]), float(drop[2]), int(ec.gemm_grid),
                    int(drop[1]) & (2 ** 64 - 1), int(drop[3]) & (2 ** 64 - 1), sdev.data_ptr() if sdev is not None else None,
                    ksf, ksb, ksw), (tuple(ksf), tuple(ksb), tuple(ksw))


def _text_params(wc, ln, qkv_w, qkv_b, others_w, others_b, need_t):
    """egv_text_params: LayerNorm affine (sa_layer_norm w, b, output_layer_norm w, b), the fused q/k/v weight planes and bias (the
    weight cache keeps the concatenation), out_lin / lin1 / lin2.  Built once per layer and direction and kept on the model's weight
    cache (see video_transformer._block_params)."""
    import ctypes as C
    from .._lib import TextParams
    pls = [wc.get_cat(qkv_w, need_t=need_t)] + [wc.get(w, need_t=need_t) for w in others_w]
    bias = [wc.get_bias_cat(qkv_b)] + list(others_b)
    small = tuple(t.data_ptr() for t in ln) + tuple(b.data_ptr() for b in bias)
    key = (id(qkv_w[0]), need_t)
    hit = wc.param_structs.get(key)
    if hit is not None and hit[1] == small and all(a[0] is b[0] and a[1] is b[1] for a, b in zip(hit[0], pls)):
        return hit[2]
    P4, L4 = C.c_void_p * 4, C.c_int64 * 4

    def ptr(t):
        return t.data_ptr() if t is not None else None
    whi, wlo, ldw = P4(*[p.hi.data_ptr() for p, _ in pls]), P4(*[ptr(p.lo) for p, _ in pls]), L4(*[p.ld for p, _ in pls])
    if need_t:
        thi, tlo, ldt = P4(*[t.hi.data_ptr() for _, t in pls]), P4(*[ptr(t.lo) for _, t in pls]), L4(*[t.ld for _, t in pls])
    else:
        thi, tlo, ldt = P4(), P4(), L4()
    prm = TextParams(*[t.data_ptr() for t in ln], P4(*[b.data_ptr() for b in bias]), whi, wlo, ldw, thi, tlo, ldt)
    wc.param_structs[key] = (pls, small, prm)
    return prm


class _TextLayerCFn(torch.autograd.Function):
    """TransformerBlock.forward / backward as ONE C-ABI call each (see _TextLayerFn for the arithmetic)."""

    @staticmethod
    def forward(ctx, x, mask, geom, ec: ExecContext,
                q_w, q_b, k_w, k_b, v_w, v_b, o_w, o_b, ln1_w, ln1_b, f1_w, f1_b, f2_w, f2_b, ln2_w, ln2_b):
        import ctypes as C
        from .. import _lib
        B, L, H, eps, drop = geom
        D = x.shape[-1]
        M = B * L
        Hd = f1_w.shape[0]
        P, Pb = ec.fwd_passes_split, ec.bwd_passes_split
        dev = x.device
        x2 = x.contiguous().view(M, D)
        mask = mask.contiguous()
        train = any(ctx.needs_input_grad)
        g, ks = _text_geom(B, L, H, D, Hd, P, Pb, train, eps, drop, ec)
        key = (B, L, H, D, Hd, P, Pb, train, drop[2] > 0, ks)
        ent = _TEXT_CACHE.get(key)
        if ent is None:
            off, tot = (C.c_int64 * 12)(), C.c_int64()
            nf = int(_lib.lib().egv_text_layer_fwd_arena_bytes(C.byref(g)))
            nb = int(_lib.lib().egv_text_layer_bwd_arena_bytes(C.byref(g)))
            _lib.check(_lib.lib().egv_text_layer_grad_layout(C.byref(g), off, C.byref(tot)), "egv_text_layer_grad_layout")
            if nf <= 0 or nb <= 0:
                raise _lib.EgovlpHipError("egv_text_layer_fwd_arena_bytes: unsupported layer geometry")
            ent = _TEXT_CACHE[key] = (nf, nb, tuple(int(o) for o in off), int(tot.value))
        arena = torch.empty(ent[0], dtype=torch.uint8, device=dev)
        out = torch.empty((M, D), dtype=torch.float32, device=dev)
        ln = (ln1_w, ln1_b, ln2_w, ln2_b)
        prm = _text_params(ec.wc, ln, (q_w, k_w, v_w), (q_b, k_b, v_b), (o_w, f1_w, f2_w), (o_b, f1_b, f2_b), need_t=False)
        _lib.check(_lib.lib().egv_text_layer_fwd(C.byref(g), C.byref(prm), x2.data_ptr(), mask.data_ptr(), out.data_ptr(),
                                                 arena.data_ptr(), ops._stream(x2)), "egv_text_layer_fwd")
        if train:
            ctx.g, ctx.ec, ctx.arena, ctx.sizes, ctx.P = g, ec, arena, ent, P
            ctx.seed_dev = drop[4]           # keeps the device seed word alive (the geometry holds its address)
            ctx.save_for_backward(mask, q_w, k_w, v_w, q_b, k_b, v_b, o_w, f1_w, f2_w, o_b, f1_b, f2_b, *ln)
        return out.view(B, L, D)

    @staticmethod
    def backward(ctx, g_out):
        import ctypes as C
        from .. import _lib
        saved = ctx.saved_tensors
        mask, qkv_w, qkv_b, others_w, others_b, ln = saved[0], saved[1:4], saved[4:7], saved[7:10], saved[10:13], saved[13:17]
        g, ec = ctx.g, ctx.ec
        if ctx.arena is None:
            raise RuntimeError("the C layer calls release their forward workspace after the first backward: a second backward through "
                               "the same graph (retain_graph=True) needs the per-kernel path (exec_ctx.set(block_calls=False))")
        if ec.bwd_passes_split != g.bwd_passes:
            raise RuntimeError("the backward precision changed between this layer's forward and its backward")
        B, L, D = g.B, g.L, g.D
        M = B * L
        dev = mask.device
        G = g_out.contiguous().view(M, D)
        _, nb, goff, gtot = ctx.sizes
        barena = torch.empty(nb, dtype=torch.uint8, device=dev)
        grads = torch.empty(gtot, dtype=torch.float32, device=dev)
        d_x = torch.empty((M, D), dtype=torch.float32, device=dev)
        prm = _text_params(ec.wc, ln, qkv_w, qkv_b, others_w, others_b, need_t=True)
        _lib.check(_lib.lib().egv_text_layer_bwd(C.byref(g), C.byref(prm), G.data_ptr(), mask.data_ptr(), ctx.arena.data_ptr(),
                                                 barena.data_ptr(), d_x.data_ptr(), grads.data_ptr(), ops._stream(G)), "egv_text_layer_bwd")
        ctx.arena = None
        sizes = [goff[i + 1] - goff[i] for i in range(11)] + [gtot - goff[11]]
        parts = grads.split_with_sizes(sizes)
        dW3 = parts[0].view(3 * D, D)
        db3 = parts[4]
        dW = [parts[i].view(others_w[i - 1].shape) for i in (1, 2, 3)]
        return (d_x.view(B, L, D), None, None, None,
                dW3[:D], db3[:D], dW3[D:2 * D], db3[D:2 * D], dW3[2 * D:], db3[2 * D:],
                dW[0], parts[5], parts[8], parts[9], dW[1], parts[6], dW[2], parts[7], parts[10], parts[11])


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

    def forward(self, x, mask, ec, drop=(0.0, 0, 0.0, 0, None)):
        B, L, D = x.shape
        a, f = self.attention, self.ffn
        drop = tuple(drop) + (None,) * (5 - len(drop))       # (attention p, seed, ffn p, seed[, device seed word])
        geom = (B, L, a.n_heads, self.sa_layer_norm.eps, drop)
        fn = _TextLayerCFn if text_calls_ok(ec, B * L, D, f.lin1.weight.shape[0], a.n_heads) else _TextLayerFn
        return fn.apply(
            x, mask, geom, ec,
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
        for pr in (self.config.dropout, self.config.attention_dropout):
            if not 0.0 <= pr < 1.0:
                raise ValueError("dropout probabilities must be in [0, 1)")
        self._drop_calls = 0            # every train-mode forward draws fresh masks: seed = f(torch seed, call #, site)
        self.embeddings = Embeddings(self.config)
        self.transformer = Transformer(self.config)
        self.exec_ctx = ops.new_context()     # FrozenInTime replaces it with the dual encoder's shared context
        self.seed_rank = 0                    # mixed into the dropout seeds: data-parallel ranks must not draw the same masks
        # Capture-safe dropout seeds (part of the C ABI: `seed_dev` of egv_dropout / egv_text_attn_* / egv_text_geom): a device int64[1]
        # whose value is XOR-ed into every dropout seed by the kernels; when set, the host-side call counter stops advancing -- a caller
        # that replays a captured training step from a HIP graph changes the masks by writing this word, launch arguments being frozen
        # at capture.  Nothing in the package sets it since the graphed train step was retired in round 5 (eager is faster on ROCm 7.2,
        # DESIGN 6); egovlp_amd/graph.py::GraphedForward captures evaluation forwards, which draw no masks.
        self.seed_device = None
        # HF init (initializer_range 0.02) so random-init statistics match `DistilBertModel(DistilBertConfig())`
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Embedding):
                nn.init.normal_(m.weight, std=0.02)

    def set_dropout(self, dropout, attention_dropout):
        """Override HF's 0.1 / 0.1 (parity runs against the dropout-free oracle use 0, 0)."""
        self.config.dropout, self.config.attention_dropout = float(dropout), float(attention_dropout)
        return self

    def _seed(self, site):
        # 64-bit seed of one dropout site of one forward call: torch's seed, the call counter and the site id, mixed
        x = (torch.initial_seed() * 0x9E3779B97F4A7C15 + self._drop_calls * 0xD1B54A32D192ED03 + site * 0x94D049BB133111EB
             + self.seed_rank * 0xA24BAED4963EE407) & (2 ** 64 - 1)
        x ^= x >> 31
        return (x * 0xBF58476D1CE4E5B9) & (2 ** 64 - 1)

    def forward(self, input_ids=None, attention_mask=None, **kw):
        if input_ids is None:
            raise NotImplementedError("inputs_embeds path is not on the EgoClip hot path")
        if input_ids.shape[1] > self.config.max_position_embeddings:
            raise ValueError("sequence longer than max_position_embeddings")
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        e = self.embeddings
        pd = self.config.dropout if self.training else 0.0
        pa = self.config.attention_dropout if self.training else 0.0
        sdev = self.seed_device
        if (pd > 0 or pa > 0) and sdev is None:
            self._drop_calls += 1
        x = _EmbedFn.apply(input_ids, e.word_embeddings.weight, e.position_embeddings.weight,
                           e.LayerNorm.weight, e.LayerNorm.bias, e.LayerNorm.eps,
                           -1 if e.word_embeddings.padding_idx is None else e.word_embeddings.padding_idx,
                           (pd, self._seed(0), sdev))
        mask = attention_mask.to(torch.int64).contiguous()
        for li, blk in enumerate(self.transformer.layer):
            x = blk(x, mask, self.exec_ctx, (pa, self._seed(1 + 2 * li), pd, self._seed(2 + 2 * li), sdev))
        return SimpleNamespace(last_hidden_state=x)
