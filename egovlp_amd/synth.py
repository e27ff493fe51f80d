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


def synth_state_dict(schema, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """schema: mapping key -> shape (e.g. {k: v.shape for k, v in model.state_dict().items()})."""
    out = OrderedDict()
    for k, shp in schema.items():
        out[k] = synth_tensor(k, shp, seed)
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
