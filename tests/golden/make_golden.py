"""Generate golden vectors by RUNNING THE REFERENCE ITSELF (showlab/EgoVLP, /root/reference).

Run once in the build container (the GPU box has no /root/reference):
    python tests/golden/make_golden.py
Writes small .npz fixtures next to this file.  Weights and inputs are NOT stored for the
full-size model (181 M params): they are pure functions of (key, shape, seed) from
egovlp_amd.synth, so any implementation regenerates them bit-identically; the tiny-config
fixture stores everything.

Fixtures
  full_b4.npz   FrozenInTime (ViT-B/16 T=4 + DistilBERT L=32 ragged mask), B=4, eval():
                text/video embeds, CLS features, sim_matrix, NormSoftmaxLoss, EgoNCE (reference
                class with `.cuda()` neutralised), block-0 taps (sub-sampled rows),
                gradient slices of sentinel weights and of the embeddings.
  tiny_video.npz  reference SpaceTimeTransformer(img 32, patch 16, dim 128, depth 2, heads 2 (head_dim 64),
                num_frames 4) on [3,3,3,32,32] input (curr_frames 3 < num_frames 4): all tensors.
  gather_w2.npz reference AllGather_multi under gloo, world_size 2: loss + local embedding grads.
  heads.npz     the OSCC / PNR classification head: the reference's CrossEntropy class on random scores, and the reference
                FrozenInTime(projection_dim = 2 | 17) video_only train step (scores, loss, head + encoder gradient slices).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))

from egovlp_amd.synth import synth_state_dict, synth_batch  # noqa: E402
from oracle import ref_import  # noqa: E402

SENTINELS = [
    "video_model.blocks.0.timeattn.qkv.weight",
    "video_model.blocks.11.mlp.fc2.weight",
    "video_model.patch_embed.proj.weight",
    "video_model.temporal_embed",
    "video_model.blocks.5.norm1.weight",
    "text_model.transformer.layer.0.attention.q_lin.weight",
    "text_model.embeddings.position_embeddings.weight",
    "txt_proj.1.weight",
    "vid_proj.0.bias",
]


def np32(t):
    return t.detach().cpu().float().numpy()


def make_full(mm, ml):
    torch.manual_seed(0)
    net = mm.FrozenInTime(
        video_params={"model": "SpaceTimeTransformer", "arch_config": "base_patch16_224", "num_frames": 16,
                      "pretrained": True, "time_init": "zeros"},
        text_params={"model": "distilbert-base-uncased", "pretrained": True, "input": "text"},
        projection="minimal", load_checkpoint="")
    net.text_model.config._attn_implementation = "eager"
    schema = {k: v.shape for k, v in net.state_dict().items()}
    net.load_state_dict(synth_state_dict(schema, seed=0), strict=True)
    net.eval()
    B = 4
    batch = synth_batch(B, T=4, L=32, seed=1234, ragged=True)
    data = {"video": batch["video"], "text": batch["text"]}

    taps = {}
    blk0 = net.video_model.blocks[0]
    h1 = blk0.timeattn.register_forward_hook(lambda m, i, o: taps.__setitem__("block0_time_output", o))
    h2 = blk0.attn.register_forward_hook(lambda m, i, o: taps.__setitem__("block0_space_output", o))
    h3 = blk0.register_forward_hook(lambda m, i, o: taps.__setitem__("block0_block_out", o))
    h4 = net.text_model.embeddings.register_forward_hook(lambda m, i, o: taps.__setitem__("text_embed", o))
    text_embeds, video_embeds = net(data)
    for h in (h1, h2, h3, h4):
        h.remove()
    text_embeds.retain_grad()
    video_embeds.retain_grad()
    sim = mm.sim_matrix(text_embeds, video_embeds)
    infonce = ml.NormSoftmaxLoss()(sim)
    # EgoNCE: the reference hard-codes torch.eye(n).cuda() (model/loss.py:35) -> neutralise .cuda()
    _cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        sim_v = mm.sim_matrix(batch["verb_vec"], batch["verb_vec"])
        sim_n = mm.sim_matrix(batch["noun_vec"], batch["noun_vec"])
        ego = ml.EgoNCE()(sim, sim_v, sim_n)
    finally:
        torch.Tensor.cuda = _cuda
    ego.backward()
    out = {
        "text_embeds": np32(text_embeds), "video_embeds": np32(video_embeds), "sim": np32(sim),
        "infonce": np32(infonce), "egonce": np32(ego), "sim_v": np32(sim_v), "sim_n": np32(sim_n),
        "grad_text_embeds": np32(text_embeds.grad), "grad_video_embeds": np32(video_embeds.grad),
        # sub-sampled taps: tokens 0 (CLS), 1, 197 (frame 1 first patch), 784 (last); first 64 channels
        "tap_rows": np.array([0, 1, 197, 784]),
    }
    rows = [0, 1, 197, 784]
    for k in ("block0_time_output", "block0_space_output", "block0_block_out"):
        out[k] = np32(taps[k][:, rows, :64])
    out["text_embed"] = np32(taps["text_embed"][:, :4, :64])
    params = dict(net.named_parameters())
    for name in SENTINELS:
        g = params[name].grad
        g2 = g.reshape(g.shape[0], -1) if g.dim() > 1 else g.reshape(1, -1)
        out["grad:" + name] = np32(g2[:8, :64])
        out["gradnorm:" + name] = np32(g.norm())
    np.savez_compressed(os.path.join(HERE, "full_b4.npz"), **out)
    print("full_b4: infonce %.6f egonce %.6f sim[%.4f,%.4f]" % (infonce.item(), ego.item(), sim.min().item(), sim.max().item()))


def make_tiny(mv):
    torch.manual_seed(0)
    net = mv.SpaceTimeTransformer(img_size=32, patch_size=16, embed_dim=128, depth=2, num_heads=2,
                                  num_frames=4, time_init="rand", num_classes=0)
    schema = {("video_model." + k): v.shape for k, v in net.state_dict().items()}
    sd = synth_state_dict(schema, seed=7)
    net.load_state_dict({k[len("video_model."):]: v for k, v in sd.items()}, strict=True)
    net.eval()
    g = torch.Generator().manual_seed(99)
    video = torch.randn(3, 3, 3, 32, 32, generator=g)          # curr_frames 3 < model num_frames 4
    video.requires_grad_(False)
    taps = {}
    h = net.blocks[0].register_forward_hook(lambda m, i, o: taps.__setitem__("block0", o))
    feats = net(video)
    h.remove()
    feats.square().sum().backward()
    out = {"video": np32(video), "feats": np32(feats), "block0": np32(taps["block0"])}
    for k, v in sd.items():
        out["w:" + k] = np32(v)
    for n, p in net.named_parameters():
        if p.grad is not None:
            out["g:video_model." + n] = np32(p.grad)
    np.savez_compressed(os.path.join(HERE, "tiny_video.npz"), **out)
    print("tiny_video: feats", tuple(feats.shape), float(feats.abs().mean()))


def _gather_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mm, ml, te, mv = ref_import.load_reference()
    import types
    args = types.SimpleNamespace(world_size=world, rank=rank)
    B = 4
    g = torch.Generator().manual_seed(500 + rank)
    v = torch.randn(B, 256, generator=g, requires_grad=True)
    t = torch.randn(B, 256, generator=g, requires_grad=True)
    b = synth_batch(B, T=1, L=4, res=2, seed=900, rank=rank)
    va = te.AllGather_multi.apply(v, world, args)
    ta = te.AllGather_multi.apply(t, world, args)
    na = te.AllGather_multi.apply(b["noun_vec"], world, args)
    vba = te.AllGather_multi.apply(b["verb_vec"], world, args)
    sim = mm.sim_matrix(ta, va)
    _cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    loss = ml.EgoNCE()(sim, mm.sim_matrix(vba, vba), mm.sim_matrix(na, na))
    torch.Tensor.cuda = _cuda
    loss.backward()
    q.put((rank, np32(v), np32(t), np32(b["noun_vec"]), np32(b["verb_vec"]), np32(loss), np32(v.grad), np32(t.grad)))
    dist.barrier()
    dist.destroy_process_group()


def make_gather():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, 29611, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda x: x[0])
    [p.join() for p in procs]
    out = {}
    for r, v, t, n, vb, loss, gv, gt in res:
        out.update({f"v{r}": v, f"t{r}": t, f"noun{r}": n, f"verb{r}": vb, f"loss{r}": loss, f"gv{r}": gv, f"gt{r}": gt})
    np.savez_compressed(os.path.join(HERE, "gather_w2.npz"), **out)
    print("gather_w2: loss", out["loss0"], out["loss1"])


def make_losses(ml):
    """The reference's own MaxMarginRankingLoss / AdaptiveMaxMarginRankingLoss (model/loss.py:55-133) on seeded similarity
    matrices: loss values and d loss / d x from torch autograd."""
    out = {}
    g = torch.Generator().manual_seed(4242)
    for n in (5, 48, 200):
        x = (torch.rand(n, n, generator=g) * 2 - 1).requires_grad_(True)
        w = torch.rand(n, generator=g) * 1.5 + 0.1
        for fix in (True, False):
            for name, loss, args in (("mm", ml.MaxMarginRankingLoss(margin=0.2, fix_norm=fix), ()),
                                     ("amm", ml.AdaptiveMaxMarginRankingLoss(margin=0.4, fix_norm=fix), (w,))):
                x.grad = None
                v = loss(x, *args)
                v.backward()
                key = f"{name}_n{n}_fix{int(fix)}"
                out["loss_" + key] = np32(v)
                out["grad_" + key] = np32(x.grad)
        out[f"x_n{n}"] = np32(x)
        out[f"w_n{n}"] = np32(w)
    np.savez_compressed(os.path.join(HERE, "losses.npz"), **out)
    print("losses:", {k: float(v) for k, v in out.items() if k.startswith("loss_") and "n48" in k})


def make_retrieval(mm):
    """(a) the `--dual_softmax` similarity of run/test_epic.py (the script cannot be imported -- it pulls the data loaders -- so
    its three helper functions `sim_matrix_mm`, `softmax_numpy` are EXECUTED from the reference's own source text and the
    expression of :140-143 is applied to them); (b) FrozenInTime.compute_text_tokens of the reference model on the seeded
    B = 4 ragged batch (sub-sampled rows)."""
    import ast
    import torch.nn.functional as F
    src = open("/root/reference/run/test_epic.py").read()
    tree = ast.parse(src)
    want = {"sim_matrix_mm", "softmax_numpy"}
    ns = {"torch": torch, "F": F, "np": np}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in want:
            exec(compile(ast.Module(body=[node], type_ignores=[]), "run/test_epic.py", "exec"), ns)
    out = {}
    g = torch.Generator().manual_seed(99)
    for nt, nv in ((7, 5), (150, 96), (300, 410)):
        t = torch.randn(nt, 256, generator=g) * 2.0
        v = torch.randn(nv, 256, generator=g) * 2.0
        sim = ns["sim_matrix_mm"](t, v)                                  # run/test_epic.py:141
        sim = ns["softmax_numpy"](sim / 500, dim=1) * sim                # :142
        sim = ns["softmax_numpy"](sim, dim=0)                            # :143
        out[f"text_{nt}x{nv}"], out[f"video_{nt}x{nv}"], out[f"dual_{nt}x{nv}"] = np32(t), np32(v), np.asarray(sim, np.float32)
    torch.manual_seed(0)
    net = mm.FrozenInTime(
        video_params={"model": "SpaceTimeTransformer", "arch_config": "base_patch16_224", "num_frames": 16,
                      "pretrained": True, "time_init": "zeros"},
        text_params={"model": "distilbert-base-uncased", "pretrained": True, "input": "text"},
        projection="minimal", load_checkpoint="")
    net.text_model.config._attn_implementation = "eager"
    net.load_state_dict(synth_state_dict({k: v.shape for k, v in net.state_dict().items()}, seed=0), strict=True)
    net.eval()
    batch = synth_batch(4, T=4, L=32, seed=1234, ragged=True)
    with torch.no_grad():
        tok = net.compute_text_tokens(batch["text"])                    # [4, 32, 256]
        vid = net(batch, video_only=True)                               # model/model.py:100-103
    out["text_tokens"] = np32(tok)
    out["video_only"] = np32(vid)
    np.savez_compressed(os.path.join(HERE, "retrieval.npz"), **out)
    print("retrieval:", {k: v.shape for k, v in out.items()})


def heads_inputs(classes, B=3, T=4):
    gg = torch.Generator().manual_seed(8 + classes)
    video = torch.randn(B, T, 3, 224, 224, generator=gg)
    state = torch.randint(0, classes, (B,), generator=gg)
    return video, state


def make_heads(mm, ml):
    """The classification fine-tune head as the reference runs it (configs/ft/oscc.json, trainer/trainer_oscc.py:335-338):
    (a) the reference's own `CrossEntropy` class (model/loss.py:135-141) on random scores / labels incl. ignored rows: loss and
    d loss / d scores; (b) the reference FrozenInTime with `projection_dim` = 2 (OSCC) and 17 (PNR), `model(data, video_only=True)`
    in train() mode, its CrossEntropy, backward: scores, loss, the head's gradients and slices of two encoder gradients.
    Weights are egovlp_amd.synth (seed 21) so that the drop-in regenerates them; `heads_inputs` below is shared with the test."""
    out = {}
    g = torch.Generator().manual_seed(4)
    for rows, cols, ign in ((64, 2, 0), (37, 17, 5)):
        x = (3.0 * torch.randn(rows, cols, generator=g)).requires_grad_(True)
        t = torch.randint(0, cols, (rows,), generator=g)
        if ign:
            t[torch.randperm(rows, generator=g)[:ign]] = -100
        loss = ml.CrossEntropy()(x, t)
        loss.backward()
        key = f"ce_{rows}x{cols}"
        out["x_" + key], out["t_" + key], out["loss_" + key], out["grad_" + key] = np32(x), t.numpy(), np32(loss), np32(x.grad)
    for classes in (2, 17):
        torch.manual_seed(0)
        net = mm.FrozenInTime(
            video_params={"model": "SpaceTimeTransformer", "arch_config": "base_patch16_224", "num_frames": 4,
                          "pretrained": True, "time_init": "rand"},
            text_params={"model": "distilbert-base-uncased", "pretrained": True, "input": "text"},
            projection="minimal", projection_dim=classes, load_checkpoint="")
        net.load_state_dict(synth_state_dict({k: v.shape for k, v in net.state_dict().items()}, seed=21), strict=True)
        net.train()
        B, T = 3, 4
        video, state = heads_inputs(classes, B, T)
        scores = net({"video": video}, video_only=True)                     # trainer/trainer_oscc.py:335
        loss = ml.CrossEntropy()(scores, state)                             # :338
        loss.backward()
        params = dict(net.named_parameters())
        k = f"head{classes}"
        # the clip itself is NOT stored (2.4 MB each): the test redraws it from the same CPU generator and checks this corner
        out["video_corner_" + k], out["state_" + k] = np32(video[:, :, :, :2, :2]), state.numpy()
        out["scores_" + k], out["loss_" + k] = np32(scores), np32(loss)
        out["g_vid_proj_w_" + k], out["g_vid_proj_b_" + k] = np32(params["vid_proj.0.weight"].grad), np32(params["vid_proj.0.bias"].grad)
        for name in ("video_model.blocks.11.mlp.fc2.weight", "video_model.blocks.0.attn.qkv.weight"):
            gr = params[name].grad
            out[f"g:{name}:" + k] = np32(gr[:8, :64])
            out[f"gn:{name}:" + k] = np32(gr.norm())
    np.savez_compressed(os.path.join(HERE, "heads.npz"), **out)
    print("heads:", {k_: (v.shape if hasattr(v, "shape") else v) for k_, v in out.items() if k_.startswith(("loss_", "scores_"))})


if __name__ == "__main__":
    assert ref_import.available(), "needs /root/reference (build container only)"
    which = sys.argv[1:] or ["tiny", "gather", "full", "losses", "retrieval", "heads"]
    if "gather" in which:
        make_gather()
    mm, ml, te, mv = ref_import.load_reference()
    if "tiny" in which:
        make_tiny(mv)
    if "full" in which:
        make_full(mm, ml)
    if "losses" in which:
        make_losses(ml)
    if "retrieval" in which:
        make_retrieval(mm)
    if "heads" in which:
        make_heads(mm, ml)
