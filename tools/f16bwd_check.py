"""Round-6 bring-up check of the fp16 backward on one GPU: python tools/f16bwd_check.py [B]
(1) the C block calls against the per-kernel path in the ('f16mix', 'f16') mode (forward embeddings bit-identical, gradients within
    the fp32-atomics noise); (2) the gradients of ('f16mix', 'bf16') and ('f16mix', 'f16') against the all-bf16x3 backward of the same
    step, per watched tensor and over ALL parameters (max / median rel-L2)."""
import sys

import torch

sys.path.insert(0, ".")
from egovlp_amd.model.loss import EgoNCE          # noqa: E402
from egovlp_amd.model.model import FrozenInTime   # noqa: E402
from egovlp_amd.optim import LossScaler           # noqa: E402
from egovlp_amd.synth import synth_batch, synth_state_dict   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
m = FrozenInTime(video_params={"model": "SpaceTimeTransformer", "arch_config": "base_patch16_224", "num_frames": 4, "pretrained": True,
                               "time_init": "rand"},
                 text_params={"model": "distilbert-base-uncased", "pretrained": True, "input": "text"}, projection="minimal",
                 load_checkpoint="")
m.load_state_dict(synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, seed=3))
m.text_model.set_dropout(0.0, 0.0)
m = m.cuda().train()
b = synth_batch(B, T=4, L=32, seed=11)
dev = {"video": b["video"].cuda(), "text": {k: v.cuda() for k, v in b["text"].items()}}
nv, vv = b["noun_vec"].cuda(), b["verb_vec"].cuda()


def rel(a, c):
    return float((a.double() - c.double()).norm() / (c.double().norm() + 1e-30))


def run(fwd, bwd, block_calls=True, scale=None):
    ec = m.exec_ctx
    ec.set_precision(fwd, bwd)
    ec.set(block_calls=block_calls)
    for p in m.parameters():
        p.grad = None
    te, ve = m(dev)
    loss = EgoNCE().fused(te, ve, nv, vv)
    sc = LossScaler(init_scale=scale) if scale else None
    (sc.scale(loss) if sc else loss).backward()
    ec.join_side_stream()
    torch.cuda.synchronize()
    k = 1.0 / scale if scale else 1.0
    return te.detach().clone(), ve.detach().clone(), float(loss), {n: (p.grad * k).clone() for n, p in m.named_parameters()}


ref = run("bf16x3", "bf16x3")
watch = ["video_model.blocks.0.timeattn.qkv.weight", "video_model.blocks.0.attn.proj.weight", "video_model.blocks.5.attn.proj.weight",
         "video_model.blocks.11.mlp.fc2.weight", "video_model.blocks.3.mlp.fc1.weight", "video_model.patch_embed.proj.weight",
         "video_model.blocks.6.norm1.weight", "video_model.blocks.0.attn.qkv.bias",
         "text_model.transformer.layer.0.attention.q_lin.weight", "vid_proj.0.weight"]
for name, args in (("f16mix/bf16", ("f16mix", "bf16", True, None)), ("f16mix/f16 S=2^16", ("f16mix", "f16", True, 65536.0)),
                   ("f16mix/f16 S=2^16 per-kernel", ("f16mix", "f16", False, 65536.0)), ("f16mix/f16 S=2^8", ("f16mix", "f16", True, 256.0)),
                   ("f16x2/f16 S=2^16", ("f16x2", "f16", True, 65536.0))):
    te, ve, loss, g = run(*args)
    errs = {n: rel(g[n], ref[3][n]) for n in g}
    bad = [n for n in g if not torch.isfinite(g[n]).all()]
    srt = sorted(errs.values())
    print("%-30s video %.2e text %.2e loss %.2e | grads vs bf16x3: max %.2e median %.2e  nonfinite %d" % (
        name, rel(ve, ref[1]), rel(te, ref[0]), abs(loss - ref[2]) / abs(ref[2]), srt[-1], srt[len(srt) // 2], len(bad)))
    print("    worst:", [(n, "%.1e" % e) for n, e in sorted(errs.items(), key=lambda kv: -kv[1])[:4]])
    print("    " + "  ".join("%s %.1e" % (w.replace("video_model.", "").replace("text_model.transformer.", "t."), errs[w]) for w in watch))
    if name == "f16mix/f16 S=2^16":
        keep = (te, ve, g)
    if name == "f16mix/f16 S=2^16 per-kernel":
        print("    per-kernel vs block calls: video bit-identical %s, text %s; grads max rel %.2e" % (
            torch.equal(ve, keep[1]), torch.equal(te, keep[0]), max(rel(g[n], keep[2][n]) for n in g)))
