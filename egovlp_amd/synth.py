"""Deterministic synthetic weights and inputs for the EgoClip hot path.

There is no network here for checkpoints or Ego4D data, so every parity test, the
benchmark and the golden-vector generator draw weights and inputs from this module.
Each tensor is a pure function of (its state_dict key, its shape, a seed): the values
do not depend on module construction order or on which implementation (the reference,
the oracle, the HIP model) asks for them, so "same key + same shape" == "same weights".

Input contract (reference: data_loader/EgoClip_EgoMCQ_dataset.py:87-97,
data_loader/transforms.py:38-39, trainer/trainer_egoclip.py:115-121):
  video  float32 [B,T,3,H,W]  ImageNet-normalised frames
  text   input_ids int64 [B,L] (token 101 = [CLS] first), attention_mask int64 [B,L]
  noun_vec float32 [B,582] multi-hot, verb_vec float32 [B,118] multi-hot
"""
import zlib
from collections import OrderedDict

import torch

NOUN_DIM = 582
VERB_DIM = 118
VOCAB = 30522


def _gen(name: str, seed: int) -> torch.Generator:
    return torch.Generator(device="cpu").manual_seed((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def synth_tensor(name: str, shape, seed: int = 0) -> torch.Tensor:
    """fp32 CPU tensor for state_dict key `name`. Scales are chosen so that every
    sub-op is exercised: attention logits have O(1) spread (qkv std .06), LayerNorm
    affine is non-trivial, temporal/positional embeddings are non-zero (the shipped
    time_init='zeros' would hide temporal bugs)."""
    g = _gen(name, seed)
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    parent = name.rsplit(".", 2)[-2] if name.count(".") >= 1 else ""
    is_ln = ("norm" in parent.lower()) or ("layernorm" in parent.lower())
    if is_ln and leaf == "weight":
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if is_ln and leaf == "bias":
        return 0.05 * torch.randn(shape, generator=g)
    if leaf == "bias":
        return 0.02 * torch.randn(shape, generator=g)
    if any(k in name for k in ("qkv", "q_lin", "k_lin")):
        return 0.06 * torch.randn(shape, generator=g)
    if "word_embeddings" in name or "position_embeddings" in name:
        return 0.05 * torch.randn(shape, generator=g)
    if leaf in ("cls_token", "pos_embed", "temporal_embed"):
        return 0.05 * torch.randn(shape, generator=g)
    return 0.03 * torch.randn(shape, generator=g)


HEAVY_GAIN, HEAVY_COMP = 30.0, 10.0       # (fully compensated, 30 / 30, the function is LESS stable: all-bf16x3 4e-3 from the oracle; 30 / 10: 4e-4)


def heavy_tensor(name: str, shape, seed: int = 0) -> torch.Tensor:
    """A HOSTILE variant of synth_tensor: what trained transformers look like where iid Gaussians do not (the reference trains from
    pretrained ViT / DistilBERT weights, model/model.py:31-36,45-63), while staying a well-conditioned function (the all-bf16x3 mode
    still meets the parity bar on it -- a distribution on which fp32-grade arithmetic itself is off says nothing about a precision policy):
      * per-output-channel log-normal scales on every Linear of both towers (sigma 0.5);
      * LayerNorm gains log-normal, with three x30 OUTLIER channels in every LayerNorm of the video tower; the Linear that consumes them
        (qkv, fc1) has those INPUT columns divided by 10 -- trained networks balance such channels, not exactly: the operands have a 30x
        dynamic range inside a row and the outlier channels still weigh 3x in every product;
      * the same three residual channels carry a large token-independent offset in the positional / class embeddings ("massive
        activations": huge, nearly constant over tokens -- softmax and LayerNorm cancel them analytically, a 2^-11 operand rounding
        does not);
      * three x8 rows in fc1 (|h| of several hundred after GELU) with the matching fc2 columns divided by 8.
    Same key + shape + seed -> same tensor, as synth_tensor."""
    base = synth_tensor(name, shape, seed)
    g = _gen("heavy:" + name, seed)
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    parent = name.rsplit(".", 2)[-2] if name.count(".") >= 1 else ""
    is_ln = ("norm" in parent.lower()) or ("layernorm" in parent.lower())
    D = shape[-1]
    outliers = [(7 * 31) % D, (7 * 131) % D, (7 * 257) % D] if D >= 64 else []        # the same residual channels everywhere
    video_block = name.startswith("video_model.blocks")
    if is_ln and leaf == "weight":
        w = torch.exp(0.5 * torch.randn(shape, generator=g))
        if video_block:
            for c in outliers:
                w[c] *= HEAVY_GAIN
        return w
    if is_ln:
        return base
    if leaf == "weight" and len(shape) >= 2 and not any(k in name for k in ("embeddings", "patch_embed")):
        w = base * torch.exp(0.5 * torch.randn(shape[0], generator=g)).view(-1, *([1] * (len(shape) - 1)))
        if video_block and (name.endswith("qkv.weight") or name.endswith("mlp.fc1.weight")):
            for c in outliers:
                w[:, c] /= HEAVY_COMP
        if name.endswith("mlp.fc1.weight"):
            for r in (11, 977, 2222):
                if r < shape[0]:
                    w[r] *= 8.0
        if name.endswith("mlp.fc2.weight"):
            for r in (11, 977, 2222):
                if r < shape[1]:
                    w[:, r] /= 8.0
        return w
    if leaf in ("pos_embed", "cls_token") and name.startswith("video_model"):
        w = base.clone()
        for c in outliers:
            w[..., c] += 1.0              # 20 sigma of the ordinary entries, the same for every token
        return w
    return base


def synth_state_dict(schema, seed: int = 0, dist: str = "gauss") -> "OrderedDict[str, torch.Tensor]":
    """schema: mapping key -> shape (e.g. {k: v.shape for k, v in model.state_dict().items()}).  dist: 'gauss' (synth_tensor) or
    'heavy' (heavy_tensor: outlier channels, log-normal scales -- the distribution the precision policy was NOT tuned on)."""
    if dist not in ("gauss", "heavy"):
        raise ValueError("synth_state_dict: dist is 'gauss' or 'heavy'")
    out = OrderedDict()
    for k, shp in schema.items():
        out[k] = (heavy_tensor if dist == "heavy" else synth_tensor)(k, shp, seed)
    return out


def synth_batch(B: int, T: int = 4, L: int = 32, res: int = 224, seed: int = 1234, rank: int = 0,
                ragged: bool = False, nouns: int = 24, verbs: int = 8):
    """One synthetic EgoClip batch. `ragged` draws caption lengths in [8,L]
    so that attention_mask handling is pinned; nouns/verbs restrict the active vocabulary
    so that off-diagonal EgoNCE positives actually occur; row 1 gets all-zero noun/verb
    vectors to pin the eps path of sim_matrix (model/model.py:193-195)."""
    g = torch.Generator(device="cpu").manual_seed(seed + rank)
    video = torch.rand(B, T, 3, res, res, generator=g)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 1, 3, 1, 1)
    video = ((video - mean) / std).contiguous()
    ids = torch.randint(1000, VOCAB, (B, L), generator=g)
    ids[:, 0] = 101
    mask = torch.ones(B, L, dtype=torch.long)
    if ragged:
        lens = torch.randint(8, L + 1, (B,), generator=g)
        lens[0] = L
        mask = (torch.arange(L)[None, :] < lens[:, None]).long()
        ids = ids * mask  # [PAD] = 0
    noun = torch.zeros(B, NOUN_DIM)
    verb = torch.zeros(B, VERB_DIM)
    n_idx = torch.randint(0, nouns, (B, 2), generator=g)
    v_idx = torch.randint(0, verbs, (B, 1), generator=g)
    noun.scatter_(1, n_idx, 1.0)
    verb.scatter_(1, v_idx, 1.0)
    if B > 1:
        noun[1].zero_()
        verb[1].zero_()
    if B > 3:                      # guarantee one off-diagonal EgoNCE positive pair (0,3)
        noun[3] = noun[0]
        verb[3] = verb[0]
    return {"video": video, "text": {"input_ids": ids, "attention_mask": mask},
            "noun_vec": noun, "verb_vec": verb}
