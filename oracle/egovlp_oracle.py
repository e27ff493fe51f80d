"""CPU oracle for the EgoClip dual-encoder hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain fp32 PyTorch-on-CPU *restatement* of the reference algorithm
(showlab/EgoVLP, paths relative to /root/reference).  It is the checker: only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
it.  Nothing under `egovlp_amd/` imports it, and the product path raises if the HIP
library is missing instead of falling back to this code.

Pinning: the reference has no tests or golden vectors, so the oracle is
pinned against OUTPUTS OF THE REFERENCE ITSELF, executed in the build container by
`tests/golden/make_golden.py` (which imports /root/reference with import stubs) and
committed under `tests/golden/*.npz`; `tests/test_oracle_golden.py` checks this file
against those fixtures.  Third-party arithmetic that is not under /root/reference:
HuggingFace DistilBERT (pinned transformers==4.2.1 in environment.yml:60); the
restatement follows its published algorithm (post-LN, eps 1e-12, exact-erf GELU,
scale d^-0.5, additive mask) and is pinned against the container's transformers 5.15
`DistilBertModel(attn_implementation='eager')` through the same fixtures.

Everything is functional: `sd` is a state_dict-like mapping key -> fp32 CPU tensor
using the reference's key names (SURVEY 8b schema).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class VideoCfg:
    img_size: int = 224
    patch_size: int = 16
    in_chans: int = 3
    embed_dim: int = 768
    depth: int = 12
    num_heads: int = 12
    mlp_ratio: float = 4.0
    num_frames: int = 16          # the MODEL's num_frames (temporal_embed rows), egoclip.json:10
    ln_eps: float = 1e-6          # model/video_transformer.py:228

    @property
    def patches_per_frame(self):
        return (self.img_size // self.patch_size) ** 2


@dataclass
class TextCfg:
    vocab_size: int = 30522
    max_pos: int = 512
    dim: int = 768
    n_layers: int = 6
    n_heads: int = 12
    hidden_dim: int = 3072
    ln_eps: float = 1e-12


# ----------------------------------------------------------------------------- video

def attn(q, k, v):
    """model/video_transformer.py:29-33 -- softmax(q k^T) v, no mask, no dropout."""
    sim = torch.einsum("bid,bjd->bij", q, k)
    a = sim.softmax(dim=-1)
    return torch.einsum("bij,bjd->bid", a, v)


def var_attention_core(qkv, num_heads, mode, n, f):
    """The attention part of VarAttention.forward (model/video_transformer.py:104-133) on the fused
    qkv projection [B, S, 3*D]: everything between the qkv Linear (:103) and the proj Linear (:135).

    mode 'space': patches regrouped '(b f) n d' (each frame attends within itself),
    mode 'time' : patches regrouped '(b n) f d' (each location attends across frames);
    in both the CLS key/value is prepended to every group (:117-121) and the CLS query
    attends over ALL keys (:112).  q is scaled BEFORE the CLS split (:106).
    """
    B, S, D3 = qkv.shape
    D = D3 // 3
    h = num_heads
    d = D // h
    q, k, v = qkv.chunk(3, dim=-1)

    def heads(t):                                                               # :104 'b n (h d) -> (b h) n d'
        return t.reshape(B, S, h, d).permute(0, 2, 1, 3).reshape(B * h, S, d)

    q, k, v = heads(q), heads(k), heads(v)
    q = q * (d ** -0.5)                                                         # :106
    cls_q, q_ = q[:, 0:1], q[:, 1:]                                             # :109
    cls_k, k_ = k[:, 0:1], k[:, 1:]
    cls_v, v_ = v[:, 0:1], v[:, 1:]
    cls_out = attn(cls_q, k, v)                                                 # :112

    def regroup(t):                                                             # :114
        t = t.reshape(B * h, f, n, d)                                           # 'b (f n) d'
        if mode == "space":
            return t.reshape(B * h * f, n, d)                                   # '(b f) n d'
        return t.permute(0, 2, 1, 3).reshape(B * h * n, f, d)                   # '(b n) f d'

    q_, k_, v_ = regroup(q_), regroup(k_), regroup(v_)
    r = q_.shape[0] // cls_k.shape[0]
    ck = cls_k.repeat_interleave(r, dim=0)                                      # :118 'b () d -> (b r) () d'
    cv = cls_v.repeat_interleave(r, dim=0)
    k_ = torch.cat((ck, k_), dim=1)                                             # :120-121
    v_ = torch.cat((cv, v_), dim=1)
    out = attn(q_, k_, v_)                                                      # :124
    if mode == "space":                                                         # :127 inverse rearrange
        out = out.reshape(B * h, f * n, d)
    else:
        out = out.reshape(B * h, n, f, d).permute(0, 2, 1, 3).reshape(B * h, f * n, d)
    out = torch.cat((cls_out, out), dim=1)                                      # :130
    return out.reshape(B, h, S, d).permute(0, 2, 1, 3).reshape(B, S, D)         # :133


def var_attention(x, sd, prefix, num_heads, mode, n, f):
    """VarAttention.forward, model/video_transformer.py:100-137: qkv Linear -> attention core -> proj."""
    qkv = F.linear(x, sd[prefix + "qkv.weight"], sd[prefix + "qkv.bias"])       # :103
    out = var_attention_core(qkv, num_heads, mode, n, f)
    return F.linear(out, sd[prefix + "proj.weight"], sd[prefix + "proj.bias"])  # :135


def text_attention_core(q, k, v, attention_mask, n_heads):
    """DistilBERT eager attention (modeling_distilbert.py:122-147) on projected q,k,v [B, L, D]."""
    B, L, D = q.shape
    d = D // n_heads
    sh = lambda t: t.view(B, L, n_heads, d).transpose(1, 2)
    neg = torch.finfo(q.dtype).min
    add_mask = torch.zeros(B, 1, 1, L, dtype=q.dtype).masked_fill(attention_mask[:, None, None, :] == 0, neg)
    w = torch.matmul(sh(q), sh(k).transpose(2, 3)) * (d ** -0.5) + add_mask
    w = F.softmax(w, dim=-1)
    return torch.matmul(w, sh(v)).transpose(1, 2).reshape(B, L, D)


def space_time_block(x, sd, p, cfg: VideoCfg, n, f, taps=None):
    """SpaceTimeBlock.forward, model/video_transformer.py:163-177.  NOTE the quirk: the
    spatial residual is added to the block INPUT x, not to time_residual (:171)."""
    D = cfg.embed_dim
    ln = lambda t, name: F.layer_norm(t, (D,), sd[p + name + ".weight"], sd[p + name + ".bias"], cfg.ln_eps)
    time_output = var_attention(ln(x, "norm3"), sd, p + "timeattn.", cfg.num_heads, "time", n, f)   # :166
    time_residual = x + time_output                                                                # :167
    space_output = var_attention(ln(time_residual, "norm1"), sd, p + "attn.", cfg.num_heads, "space", n, f)  # :168
    space_residual = x + space_output                                                              # :171
    hdn = F.linear(ln(space_residual, "norm2"), sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])  # :47
    hdn = F.gelu(hdn)                                                                              # exact erf, :37
    mlp = F.linear(hdn, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])                          # :50
    out = space_residual + mlp                                                                     # :175
    if taps is not None:
        taps.update(time_output=time_output, space_output=space_output, block_out=out)
    return out


def video_tokens(video, sd, cfg: VideoCfg, prefix="video_model."):
    """Patch embed + CLS + positional/temporal embeds: model/video_transformer.py:72-77,302-321."""
    B, T, C, H, W = video.shape
    assert T <= cfg.num_frames                                                  # :74
    x = video.reshape(B * T, C, H, W)
    x = F.conv2d(x, sd[prefix + "patch_embed.proj.weight"], sd[prefix + "patch_embed.proj.bias"],
                 stride=cfg.patch_size)                                         # :70,76
    x = x.flatten(2).transpose(2, 1).reshape(B, -1, cfg.embed_dim)              # :305-306 frame-major tokens
    cls = sd[prefix + "cls_token"].expand(B, -1, -1)
    x = torch.cat((cls, x), dim=1)                                              # :309-310
    pos = sd[prefix + "pos_embed"]
    n = cfg.patches_per_frame
    cls_embed = pos[:, 0, :].unsqueeze(1)                                       # :312
    tile_pos = pos[:, 1:, :].repeat(1, cfg.num_frames, 1)                       # :313 (the MODEL's num_frames)
    tile_tmp = sd[prefix + "temporal_embed"].repeat_interleave(n, 1)            # :315
    total = torch.cat([cls_embed, tile_pos + tile_tmp], dim=1)                  # :316-317
    return x + total[:, : x.shape[1]]                                           # :319-320


def video_encoder(video, sd, cfg: VideoCfg, prefix="video_model.", taps=None):
    """SpaceTimeTransformer.forward_features -> [B, D]; head/pre_logits are Identity
    (model/model.py:55-56)."""
    B, T = video.shape[:2]
    x = video_tokens(video, sd, cfg, prefix)
    if taps is not None:
        taps["tokens"] = x
    n, f = cfg.patches_per_frame, T
    for i in range(cfg.depth):                                                  # :325-328
        t = {} if (taps is not None and i == 0) else None
        x = space_time_block(x, sd, f"{prefix}blocks.{i}.", cfg, n, f, t)
        if t is not None:
            taps.update({f"block0_{k}": v for k, v in t.items()})
    x = F.layer_norm(x, (cfg.embed_dim,), sd[prefix + "norm.weight"], sd[prefix + "norm.bias"], cfg.ln_eps)
    return x[:, 0]                                                              # :330


# ----------------------------------------------------------------------------- text

def distilbert(input_ids, attention_mask, sd, cfg: TextCfg, prefix="text_model.", taps=None):
    """HF DistilBertModel.forward -> last_hidden_state [B,L,dim] (eval mode: dropout off).
    Embeddings modeling_distilbert.py:82-118, attention :122-203, FFN :206-224,
    TransformerBlock :227-259 (post-LN)."""
    B, L = input_ids.shape
    D, H = cfg.dim, cfg.n_heads
    d = D // H
    x = sd[prefix + "embeddings.word_embeddings.weight"][input_ids] \
        + sd[prefix + "embeddings.position_embeddings.weight"][:L][None]
    x = F.layer_norm(x, (D,), sd[prefix + "embeddings.LayerNorm.weight"],
                     sd[prefix + "embeddings.LayerNorm.bias"], cfg.ln_eps)
    if taps is not None:
        taps["text_embed"] = x
    neg = torch.finfo(torch.float32).min
    add_mask = torch.zeros(B, 1, 1, L).masked_fill(attention_mask[:, None, None, :] == 0, neg)
    for i in range(cfg.n_layers):
        p = f"{prefix}transformer.layer.{i}."
        lin = lambda t, n: F.linear(t, sd[p + n + ".weight"], sd[p + n + ".bias"])
        sh = lambda t: t.view(B, L, H, d).transpose(1, 2)
        q, k, v = sh(lin(x, "attention.q_lin")), sh(lin(x, "attention.k_lin")), sh(lin(x, "attention.v_lin"))
        w = torch.matmul(q, k.transpose(2, 3)) * (d ** -0.5) + add_mask
        w = F.softmax(w, dim=-1)
        ctx = torch.matmul(w, v).transpose(1, 2).reshape(B, L, D)
        sa = lin(ctx, "attention.out_lin")
        sa = F.layer_norm(sa + x, (D,), sd[p + "sa_layer_norm.weight"], sd[p + "sa_layer_norm.bias"], cfg.ln_eps)
        ff = lin(F.gelu(lin(sa, "ffn.lin1")), "ffn.lin2")
        x = F.layer_norm(ff + sa, (D,), sd[p + "output_layer_norm.weight"], sd[p + "output_layer_norm.bias"], cfg.ln_eps)
        if taps is not None and i == 0:
            taps["text_layer0"] = x
    return x


# ----------------------------------------------------------------------------- model / loss

def frozen_in_time(data, sd, vcfg: VideoCfg, tcfg: TextCfg, taps=None):
    """FrozenInTime.forward(return_embeds=True), model/model.py:100-143, projection='minimal':
    txt_proj = ReLU -> Linear(768,256) (:73-75), vid_proj = Linear(768,256) (:77-79)."""
    t = distilbert(data["text"]["input_ids"], data["text"]["attention_mask"], sd, tcfg, taps=taps)[:, 0, :]  # :122
    text_embeds = F.linear(F.relu(t), sd["txt_proj.1.weight"], sd["txt_proj.1.bias"])                        # :125
    v = video_encoder(data["video"], sd, vcfg, taps=taps)
    video_embeds = F.linear(v, sd["vid_proj.0.weight"], sd["vid_proj.0.bias"])                               # :142
    if taps is not None:
        taps.update(text_cls=t, video_cls=v)
    return text_embeds, video_embeds


def sim_matrix(a, b, eps=1e-8):
    """model/model.py:189-197."""
    a_n, b_n = a.norm(dim=1)[:, None], b.norm(dim=1)[:, None]
    a_norm = a / torch.max(a_n, eps * torch.ones_like(a_n))
    b_norm = b / torch.max(b_n, eps * torch.ones_like(b_n))
    return torch.mm(a_norm, b_norm.transpose(0, 1))


def norm_softmax_loss(x, temperature=0.05):
    """NormSoftmaxLoss.forward, model/loss.py:13-25."""
    i_logsm = F.log_softmax(x / temperature, dim=1)
    j_logsm = F.log_softmax(x.t() / temperature, dim=1)
    idiag = torch.diag(i_logsm)
    jdiag = torch.diag(j_logsm)
    return -idiag.sum() / len(idiag) - jdiag.sum() / len(jdiag)


def egonce(x, mask_v, mask_n, temperature=0.05, noun=True, verb=True):
    """EgoNCE.forward, model/loss.py:34-53, restated with device=x.device because the
    reference hard-codes `torch.eye(n).cuda()` (:35) and cannot run on CPU unmodified."""
    mask_diag = torch.eye(x.shape[0], device=x.device, dtype=x.dtype)
    if noun and verb:
        mask = mask_v * mask_n + mask_diag
    elif noun:
        mask = mask_n + mask_diag
    else:
        mask = mask_v + mask_diag
    i_sm = F.softmax(x / temperature, dim=1)
    j_sm = F.softmax(x.t() / temperature, dim=1)
    mask_bool = mask > 0
    idiag = torch.log(torch.sum(i_sm * mask_bool, dim=1))
    jdiag = torch.log(torch.sum(j_sm * mask_bool, dim=1))
    return -idiag.sum() / len(idiag) - jdiag.sum() / len(jdiag)


def egoclip_loss(text_embeds, video_embeds, noun_vec, verb_vec, loss="EgoNCE"):
    """The loss part of the train step, trainer/trainer_egoclip.py:130-137 (single rank:
    the gathered tensors are the local ones)."""
    output = sim_matrix(text_embeds, video_embeds)
    if loss == "EgoNCE":
        sim_v = sim_matrix(verb_vec, verb_vec)
        sim_n = sim_matrix(noun_vec, noun_vec)
        return egonce(output, sim_v, sim_n), output
    return norm_softmax_loss(output), output


def adamw_step(p, g, m, v, step, lr=3e-5, beta1=0.9, beta2=0.999, eps=1e-6, weight_decay=0.0,
               correct_bias=True):
    """transformers==4.2.1 `AdamW.step` (optimization.py), the optimizer named by
    run/train_egoclip.py:73 + configs/pt/egoclip.json:49-54.  `step` is 1-based.  In place."""
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    denom = v.sqrt().add_(eps)
    step_size = lr
    if correct_bias:
        step_size = step_size * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    p.addcdiv_(m, denom, value=-step_size)
    if weight_decay > 0.0:
        p.add_(p, alpha=-lr * weight_decay)
    return p


# ----------------------------------------------------------------------------- fine-tune ranking losses

def max_margin_ranking_loss(x, margin=0.2, fix_norm=True, weight=None):
    """MaxMarginRankingLoss.forward (model/loss.py:63-89) and, with `weight`, AdaptiveMaxMarginRankingLoss.forward
    (:100-132), restated without the index_select gymnastics: the two halves of the reference's concatenation are the
    row-direction terms relu(w_i m - (x_ii - x_ij)) and the column-direction terms relu(w_i m - (x_ii - x_ji)); fix_norm
    removes the diagonal pairs before the mean (:77-87, :119-130)."""
    n = x.shape[0]
    d = torch.diag(x).unsqueeze(1)                                              # x1: x_ii expanded along j (:66-69)
    m = margin if weight is None else weight.unsqueeze(1) * margin             # w1 (:106-109)
    rows = F.relu(m - (d - x))                                                  # x2 = x.view(-1)          (:71)
    cols = F.relu(m - (d - x.t()))                                              # x3 = x.t().view(-1)      (:72)
    if fix_norm:
        keep = 1.0 - torch.eye(n, dtype=x.dtype)
        return ((rows + cols) * keep).sum() / (2 * n * (n - 1))
    return (rows + cols).sum() / (2 * n * n)


def dual_softmax_similarity(text_embeds, vid_embeds, temp=500.0):
    """run/test_epic.py:31-38,137-143 (`--dual_softmax`): sim = text @ video^T; sim = softmax(sim / 500, dim=1) * sim;
    sim = softmax(sim, dim=0).  [texts, videos]."""
    sim = torch.mm(text_embeds, vid_embeds.transpose(0, 1))
    sim = F.softmax(sim / temp, dim=1) * sim
    return F.softmax(sim, dim=0)


def text_token_embeds(input_ids, attention_mask, sd, cfg: "TextCfg"):
    """FrozenInTime.compute_text_tokens, model/model.py:128-138: txt_proj (ReLU -> Linear) applied to EVERY token of
    DistilBERT's last hidden state -> [B, L, 256] (the NLQ / MQ feature dumps, run/test_nlq.py:107-110)."""
    hidden = distilbert(input_ids, attention_mask, sd, cfg)
    return F.linear(F.relu(hidden), sd["txt_proj.1.weight"], sd["txt_proj.1.bias"])
