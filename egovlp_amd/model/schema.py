"""state_dict schema of the reference FrozenInTime (model/model.py:14-95): key -> shape.

The key names and shapes ARE the checkpoint-compatibility contract: the
drop-in modules in this package expose exactly these keys so `EgoVLP_PT_BEST`-style
checkpoints load with strict=True.  327 tensors, 180.93 M parameters at the defaults.
"""
from collections import OrderedDict


def video_schema(embed_dim=768, depth=12, patch_size=16, in_chans=3, img_size=224, num_frames=16,
                 mlp_ratio=4.0, prefix="video_model."):
    D = embed_dim
    Hd = int(D * mlp_ratio)
    n = (img_size // patch_size) ** 2
    s = OrderedDict()
    s[prefix + "cls_token"] = (1, 1, D)
    s[prefix + "pos_embed"] = (1, n + 1, D)
    s[prefix + "temporal_embed"] = (1, num_frames, D)
    s[prefix + "patch_embed.proj.weight"] = (D, in_chans, patch_size, patch_size)
    s[prefix + "patch_embed.proj.bias"] = (D,)
    for i in range(depth):
        p = f"{prefix}blocks.{i}."
        s[p + "norm1.weight"] = (D,); s[p + "norm1.bias"] = (D,)
        s[p + "attn.qkv.weight"] = (3 * D, D); s[p + "attn.qkv.bias"] = (3 * D,)
        s[p + "attn.proj.weight"] = (D, D); s[p + "attn.proj.bias"] = (D,)
        s[p + "timeattn.qkv.weight"] = (3 * D, D); s[p + "timeattn.qkv.bias"] = (3 * D,)
        s[p + "timeattn.proj.weight"] = (D, D); s[p + "timeattn.proj.bias"] = (D,)
        s[p + "norm2.weight"] = (D,); s[p + "norm2.bias"] = (D,)
        s[p + "mlp.fc1.weight"] = (Hd, D); s[p + "mlp.fc1.bias"] = (Hd,)
        s[p + "mlp.fc2.weight"] = (D, Hd); s[p + "mlp.fc2.bias"] = (D,)
        s[p + "norm3.weight"] = (D,); s[p + "norm3.bias"] = (D,)
    s[prefix + "norm.weight"] = (D,); s[prefix + "norm.bias"] = (D,)
    return s


def text_schema(vocab=30522, max_pos=512, dim=768, n_layers=6, hidden=3072, prefix="text_model."):
    s = OrderedDict()
    s[prefix + "embeddings.word_embeddings.weight"] = (vocab, dim)
    s[prefix + "embeddings.position_embeddings.weight"] = (max_pos, dim)
    s[prefix + "embeddings.LayerNorm.weight"] = (dim,); s[prefix + "embeddings.LayerNorm.bias"] = (dim,)
    for i in range(n_layers):
        p = f"{prefix}transformer.layer.{i}."
        for lin in ("q_lin", "k_lin", "v_lin", "out_lin"):
            s[p + f"attention.{lin}.weight"] = (dim, dim); s[p + f"attention.{lin}.bias"] = (dim,)
        s[p + "sa_layer_norm.weight"] = (dim,); s[p + "sa_layer_norm.bias"] = (dim,)
        s[p + "ffn.lin1.weight"] = (hidden, dim); s[p + "ffn.lin1.bias"] = (hidden,)
        s[p + "ffn.lin2.weight"] = (dim, hidden); s[p + "ffn.lin2.bias"] = (dim,)
        s[p + "output_layer_norm.weight"] = (dim,); s[p + "output_layer_norm.bias"] = (dim,)
    return s


def state_dict_schema(projection_dim=256, **video_kw):
    """Full FrozenInTime schema in the reference's registration order (model/model.py:31-86)."""
    s = OrderedDict()
    s.update(text_schema())
    s.update(video_schema(**video_kw))
    dim = video_kw.get("embed_dim", 768)
    s["txt_proj.1.weight"] = (projection_dim, 768); s["txt_proj.1.bias"] = (projection_dim,)
    s["vid_proj.0.weight"] = (projection_dim, dim); s["vid_proj.0.bias"] = (projection_dim,)
    return s
