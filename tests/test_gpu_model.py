"""Model-level parity on a real MI355X (`pytest -m gpu`): the drop-in modules (egovlp_amd.model.*) driven
through the C ABI vs (a) golden vectors produced by the REFERENCE itself (tests/golden/*.npz) and (b) the CPU
oracle on identical seeded weights / inputs.  The north-star tolerance is 1e-3 relative (rel-L2) on the
embeddings and the loss in the parity mode ("bf16x3"); gradients get the same bar.  The single-pass "bf16"
mode is measured too and bounded loosely (it is the fast mode, not the parity mode)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# Gradient bound of the BENCHMARKED mode ("mixed": bf16x3 forward, single-pass bf16 backward).  The forward -- embeddings and
# loss, what north_star puts the 1e-3 bar on -- is bit-identical to the parity mode; the backward rounds dY, W and the saved
# activations to bf16 once per GEMM (2^-9 per operand) and that error compounds through the 12 blocks: measured rel-L2 on
# MI355X from ~2e-3 (last block, head) to ~2e-2 (block 0's temporal-attention qkv weight, the deepest gradient of the model);
# whole-tensor gradient norms stay within 1e-3.  The tests print every value; the bound below is what they assert.
MIXED_GRAD = 5e-2

from egovlp_amd.synth import synth_batch, synth_state_dict  # noqa: E402
from oracle import egovlp_oracle as O  # noqa: E402

PARITY = 1e-3
# The f16x2 forward (two fp16 products in the video blocks' qkv / fc1 / fc2 Linears) is fp32-grade like bf16x3
# (tests/quant_emul.py: 3.2e-5 on the embeddings where bf16x3 gives 2.7e-5); it is held to a FIFTH of the bar.
X2_BAR = 2e-4
# 'f16mix' (the benchmarked mode of round 5): ONE fp16 product in the qkv / fc1 / fc2 Linears of the last three quarters of the video
# blocks (ops.single_product_policy).  The per-op / per-block table on the CPU oracle (profiles/r05_precision_table.txt) predicts
# 4.4e-4 (ViT-B/16, T = 4), 4.0e-4 (T = 16), 3.6e-4 (ViT-L/14) on the video embedding over a batch; asserted: 7e-4 on batches, the
# north-star bar itself (1e-3) on single rows.
MIX_BAR = 7e-4


# The fp16 backward (round 6: 'f16x2' / 'f16mix' pair with it by default; '.../bf16' names round 5's pairing): ONE fp16 product per
# backward GEMM and per attention product on loss-scaled gradients -- 2^-11 per operand.  Measured on MI355X against the bf16x3 backward:
# 2.6e-3 on the deepest tensors (block 0, patch embedding) where the bf16 backward had 2.0e-2 (profiles/r06_fp16_backward_bringup.txt).
F16_GRAD = 1e-2


def _fbar(mode, per_row=False):
    return {"f16x2": X2_BAR, "f16mix": PARITY if per_row else MIX_BAR}.get(mode.split("/")[0], PARITY)


def _gbar(mode):
    if mode == "bf16x3":
        return 3 * PARITY
    return MIXED_GRAD if (mode == "mixed" or mode.endswith("/bf16")) else F16_GRAD


def _set_mode(mode):
    """'bf16x3', 'mixed' (bf16x3 forward, bf16 backward), 'f16x2' / 'f16mix' (fp16 backward), 'f16x2/bf16' / 'f16mix/bf16'."""
    from egovlp_amd.ops import Precision
    if mode == "mixed":
        Precision.set("bf16x3", "bf16")
    elif "/" in mode:
        Precision.set(*mode.split("/"))
    elif mode in ("f16x2", "f16mix"):
        Precision.set(mode, "f16")
    else:
        Precision.set(mode)


def _backward(m, loss, retained=()):
    """loss.backward() in the model's backward precision: the fp16 backward runs on the loss multiplied by the model's device-side loss
    scale (what egoclip_step does); the gradients -- parameters and the `retained` non-leaf tensors -- are un-scaled here, as AdamW does
    inside its update, so that callers compare plain gradients."""
    ec = m.exec_ctx
    if ec.bwd_passes != 4:
        loss.backward()
        return
    sc = ec.loss_scaler()
    S = sc.get_scale()
    sc.scale(loss).backward()
    ec.join_side_stream()
    with torch.no_grad():
        for p_ in m.parameters():
            if p_.grad is not None:
                p_.grad.mul_(1.0 / S)
        for t in retained:
            t.grad.mul_(1.0 / S)


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def to_dev(batch):
    return {"video": batch["video"].cuda(), "text": {k: v.cuda() for k, v in batch["text"].items()},
            "noun_vec": batch["noun_vec"].cuda(), "verb_vec": batch["verb_vec"].cuda()}


def build_full(time_init="zeros"):
    from egovlp_amd.model.model import FrozenInTime
    m = FrozenInTime(video_params={"model": "SpaceTimeTransformer", "arch_config": "base_patch16_224", "num_frames": 16,
                                   "pretrained": True, "time_init": time_init},
                     text_params={"model": "distilbert-base-uncased", "pretrained": True, "input": "text"},
                     projection="minimal", load_checkpoint="")
    sd = synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, seed=0)
    m.load_state_dict(sd, strict=True)
    m.text_model.set_dropout(0.0, 0.0)      # the oracle is the deterministic path; dropout has its own tests
    return m.cuda(), sd


@pytest.fixture(scope="module")
def full():
    from egovlp_amd.ops import Precision
    Precision.set("bf16x3")
    m, sd = build_full()
    return m, sd


def test_state_dict_schema_matches_reference(full):
    from egovlp_amd.model.schema import state_dict_schema
    m, _ = full
    ours = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    ref = {k: tuple(v) for k, v in state_dict_schema().items()}
    assert ours == ref
    assert len(ours) == 327
    assert abs(sum(int(np.prod(s)) for s in ours.values()) / 1e6 - 180.93) < 0.01


def test_tiny_video_encoder_matches_reference_golden(golden_dir):
    """fwd + all parameter gradients of the reference SpaceTimeTransformer (tiny config, T=3 < num_frames=4)."""
    from egovlp_amd.model.video_transformer import SpaceTimeTransformer
    from egovlp_amd.ops import Precision
    Precision.set("bf16x3")
    g = np.load(os.path.join(golden_dir, "tiny_video.npz"))
    net = SpaceTimeTransformer(img_size=32, patch_size=16, embed_dim=128, depth=2, num_heads=2, num_frames=4,
                               time_init="rand", num_classes=0)
    sd = {k[len("w:video_model."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w:")}
    net.load_state_dict(sd, strict=True)
    net = net.cuda()
    feats = net(torch.from_numpy(g["video"]).cuda())
    assert rel(feats, g["feats"]) < PARITY
    feats.square().sum().backward()
    worst = 0.0
    for name, p in net.named_parameters():
        key = "g:video_model." + name
        if key in g.files:
            r = rel(p.grad, g[key])
            worst = max(worst, r)
            assert r < PARITY, (name, r)
    print("tiny video: feats rel %.2e, worst grad rel %.2e" % (rel(feats, g["feats"]), worst))


def test_full_model_matches_reference_golden(full, golden_dir):
    m, sd = full
    g = np.load(os.path.join(golden_dir, "full_b4.npz"))
    batch = synth_batch(4, T=4, L=32, seed=1234, ragged=True)
    m.eval()
    te, ve = m(to_dev(batch))
    r_t, r_v = rel(te, g["text_embeds"]), rel(ve, g["video_embeds"])
    print("full B=4 bf16x3: text rel %.2e video rel %.2e" % (r_t, r_v))
    assert r_t < PARITY and r_v < PARITY
    from egovlp_amd.model.loss import EgoNCE, NormSoftmaxLoss
    d = to_dev(batch)
    ego = EgoNCE().fused(te, ve, d["noun_vec"], d["verb_vec"])
    nce = NormSoftmaxLoss().fused(te, ve)
    assert abs(float(ego) - float(g["egonce"])) < PARITY * abs(float(g["egonce"]))
    assert abs(float(nce) - float(g["infonce"])) < PARITY * abs(float(g["infonce"]))
    te.retain_grad(); ve.retain_grad()
    ego.backward()
    assert rel(te.grad, g["grad_text_embeds"]) < PARITY and rel(ve.grad, g["grad_video_embeds"]) < PARITY
    params = dict(m.named_parameters())
    for key in g.files:
        if key.startswith("grad:"):
            name = key[5:]
            gr = params[name].grad
            g2 = gr.reshape(gr.shape[0], -1) if gr.dim() > 1 else gr.reshape(1, -1)
            r1 = rel(g2[:8, :64], g[key])
            r2 = abs(float(gr.norm()) / float(g["gradnorm:" + name]) - 1)
            print("  grad %-55s slice rel %.2e norm rel %.2e" % (name, r1, r2))
            assert r1 < 3 * PARITY and r2 < PARITY, name
    for p in m.parameters():
        p.grad = None


def test_full_model_golden_in_the_benchmarked_mixed_mode(full, golden_dir):
    """bench.py's default precision (bf16x3 forward / single-pass bf16 backward) against the reference's golden vectors:
    embeddings and losses at the parity bar, every gradient the fixture holds inside MIXED_GRAD."""
    from egovlp_amd.model.loss import EgoNCE
    from egovlp_amd.ops import Precision
    m, sd = full
    g = np.load(os.path.join(golden_dir, "full_b4.npz"))
    batch = synth_batch(4, T=4, L=32, seed=1234, ragged=True)
    try:
        Precision.set("bf16x3", "bf16")
        m.eval()
        d = to_dev(batch)
        te, ve = m(d)
        assert rel(te, g["text_embeds"]) < PARITY and rel(ve, g["video_embeds"]) < PARITY
        ego = EgoNCE().fused(te, ve, d["noun_vec"], d["verb_vec"])
        assert abs(float(ego) - float(g["egonce"])) < PARITY * abs(float(g["egonce"]))
        te.retain_grad(); ve.retain_grad()
        ego.backward()
        assert rel(te.grad, g["grad_text_embeds"]) < PARITY and rel(ve.grad, g["grad_video_embeds"]) < PARITY   # the head is fp32
        params = dict(m.named_parameters())
        worst = 0.0
        for key in g.files:
            if key.startswith("grad:"):
                name = key[5:]
                gr = params[name].grad
                g2 = gr.reshape(gr.shape[0], -1) if gr.dim() > 1 else gr.reshape(1, -1)
                r1 = rel(g2[:8, :64], g[key])
                r2 = abs(float(gr.norm()) / float(g["gradnorm:" + name]) - 1)
                worst = max(worst, r1)
                print("  mixed grad %-55s slice rel %.2e norm rel %.2e" % (name, r1, r2))
                assert r1 < MIXED_GRAD and r2 < MIXED_GRAD, name
        print("full B=4 mixed: worst sentinel-gradient rel %.2e" % worst)
    finally:
        Precision.set("bf16x3")
        for p_ in m.parameters():
            p_.grad = None
        m.train()


@pytest.mark.parametrize("x2mode", ["f16x2", "f16mix", "f16x2/bf16", "f16mix/bf16"])
def test_full_model_golden_in_the_f16x2_mode(full, golden_dir, x2mode):
    """The fp16-product forwards against the reference's golden vectors at B = 4 (M = 3140 tokens: every qkv / fc1 / fc2 GEMM of the 12
    blocks runs the two-fp16-product kernel -- 'f16mix': the one-product kernel from block 3 on): embeddings and losses inside X2_BAR /
    MIX_BAR; gradients inside F16_GRAD with the fp16 backward (the default pairing: fp16 attention in both directions, scaled loss) and
    inside MIXED_GRAD with round 5's bf16 backward ('.../bf16')."""
    from egovlp_amd import ops
    from egovlp_amd.model.loss import EgoNCE, NormSoftmaxLoss
    from egovlp_amd.ops import Precision
    m, sd = full
    g = np.load(os.path.join(golden_dir, "full_b4.npz"))
    batch = synth_batch(4, T=4, L=32, seed=1234, ragged=True)
    assert ops.f16x2_gemm_ok(4 * 785, 2304, 768) and ops.f16x2_gemm_ok(4 * 785, 768, 3072)
    try:
        _set_mode(x2mode)
        assert Precision.name() == (x2mode.split("/")[0], "bf16" if x2mode.endswith("/bf16") else "f16")
        fbar, gbar = _fbar(x2mode), _gbar(x2mode)
        m.eval()
        d = to_dev(batch)
        te, ve = m(d)
        r_t, r_v = rel(te, g["text_embeds"]), rel(ve, g["video_embeds"])
        ego = EgoNCE().fused(te, ve, d["noun_vec"], d["verb_vec"])
        nce = NormSoftmaxLoss().fused(te, ve)
        r_e, r_n = abs(float(ego) - float(g["egonce"])) / abs(float(g["egonce"])), abs(float(nce) - float(g["infonce"])) / abs(float(g["infonce"]))
        print("full B=4 %s: text rel %.2e video rel %.2e egonce rel %.2e infonce rel %.2e" % (x2mode, r_t, r_v, r_e, r_n))
        assert r_t < X2_BAR and r_v < fbar and r_e < fbar and r_n < fbar         # the text tower is three-product in both
        te.retain_grad(); ve.retain_grad()
        m.train()
        for p_ in m.parameters():
            p_.grad = None
        _backward(m, ego, (te, ve))
        assert rel(te.grad, g["grad_text_embeds"]) < PARITY and rel(ve.grad, g["grad_video_embeds"]) < PARITY
        params = dict(m.named_parameters())
        worst = 0.0
        for key in g.files:
            if key.startswith("grad:"):
                name = key[5:]
                gr = params[name].grad
                g2 = gr.reshape(gr.shape[0], -1) if gr.dim() > 1 else gr.reshape(1, -1)
                r1 = rel(g2[:8, :64], g[key])
                r2 = abs(float(gr.norm()) / float(g["gradnorm:" + name]) - 1)
                worst = max(worst, r1)
                assert r1 < gbar and r2 < gbar, (name, r1, r2)
        print("full B=4 %s: worst sentinel-gradient rel %.2e" % (x2mode, worst))
    finally:
        Precision.set("bf16x3")
        for p_ in m.parameters():
            p_.grad = None
        m.train()


def test_full_model_fast_bf16_mode_error_is_bounded(full):
    """Single-pass bf16 operands: NOT the parity mode (~6e-3 drift on the reference itself)."""
    from egovlp_amd.ops import Precision
    m, sd = full
    batch = synth_batch(4, T=4, L=32, seed=1234, ragged=True)
    sdc = {k: v for k, v in sd.items()}
    with torch.no_grad():
        ref_t, ref_v = O.frozen_in_time(batch, sdc, O.VideoCfg(), O.TextCfg())
    try:
        Precision.set("bf16")
        m.eval()
        with torch.no_grad():
            te, ve = m(to_dev(batch))
    finally:
        Precision.set("bf16x3")
    r_t, r_v = rel(te, ref_t), rel(ve, ref_v)
    print("full B=4 bf16 (1 pass): text rel %.2e video rel %.2e" % (r_t, r_v))
    assert r_t < 5e-2 and r_v < 5e-2


@pytest.mark.parametrize("mode", ["bf16x3", "mixed"])
def test_train_step_matches_oracle(full, mode):
    """One full optimisation step (fwd, EgoNCE, bwd, AdamW) vs the oracle + torch autograd on the CPU, in the parity mode and in
    the benchmarked mixed mode (same forward; the AdamW update is sign-dominated, bounds in the body)."""
    from egovlp_amd.model.loss import EgoNCE
    from egovlp_amd.ops import Precision
    from egovlp_amd.optim import AdamW
    from egovlp_amd.trainer.trainer_egoclip import egoclip_step
    m, sd = full
    if mode == "mixed":
        Precision.set("bf16x3", "bf16")
    else:
        Precision.set("bf16x3")
    B = 4
    batch = synth_batch(B, T=4, L=32, seed=4321, ragged=True)
    m.load_state_dict(sd, strict=True)
    m.train()
    opt = AdamW(m.parameters(), lr=3e-5)
    watch = ["video_model.blocks.3.attn.qkv.weight", "text_model.transformer.layer.2.ffn.lin1.weight",
             "video_model.pos_embed", "vid_proj.0.weight", "video_model.blocks.7.norm3.bias"]
    try:
        loss = egoclip_step(m, EgoNCE(), opt, to_dev(batch))
        # oracle
        sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        te, ve = O.frozen_in_time(batch, sdo, O.VideoCfg(), O.TextCfg())
        ref, _ = O.egoclip_loss(te, ve, batch["noun_vec"], batch["verb_vec"])
        ref.backward()
        assert abs(float(loss) - float(ref)) < PARITY * abs(float(ref))
        new = dict(m.named_parameters())
        # Adam's first update is lr * g / (|g| + eps): sign-like, so an element whose gradient is within the backward's error of
        # zero flips by 2 lr.  bf16x3 backward (gradients at 1e-4): measured < 1e-3 of the update norm, bound 2e-2; single-pass
        # backward (gradients at 1e-2): measured 4e-2 .. 8e-2, bound 1.5e-1.
        bound = 2e-2 if mode == "bf16x3" else 1.5e-1
        for name in watch:
            p = sdo[name].detach().clone()
            O.adamw_step(p, sdo[name].grad, torch.zeros_like(p), torch.zeros_like(p), 1, lr=3e-5)
            upd_ref = p - sd[name]
            upd = new[name].detach().cpu() - sd[name]
            r = rel(upd, upd_ref)
            print("  %s update %-55s rel %.2e" % (mode, name, r))
            assert r < bound, name
    finally:
        m.load_state_dict(sd, strict=True)      # the module-scoped model is shared with the tests below
        from egovlp_amd import weights
        weights.bump_epoch()
    Precision.set("bf16x3")


@pytest.mark.parametrize("name,kw,T,B", [
    ("config4_T16", dict(img_size=224, patch_size=16, embed_dim=768, depth=12, num_heads=12, num_frames=16), 16, 1),
    ("config5_vitl14", dict(img_size=224, patch_size=14, embed_dim=1024, depth=24, num_heads=16, num_frames=4), 4, 1),
])
def test_other_baseline_configs_video_encoder_matches_oracle(name, kw, T, B):
    """BASELINE configs 4 (16 frames: temporal attention 16 x 17, S = 3137) and 5 (ViT-L/14: D = 1024, 24 blocks, 257 keys):
    video encoder forward + a few parameter gradients vs the fp32 CPU oracle at a small batch, parity mode."""
    from egovlp_amd.model.video_transformer import SpaceTimeTransformer
    from egovlp_amd.ops import Precision
    Precision.set("bf16x3")
    net = SpaceTimeTransformer(num_classes=0, time_init="rand", **kw)
    sd = synth_state_dict({k: v.shape for k, v in net.state_dict().items()}, seed=5)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().train()
    g = torch.Generator().manual_seed(77)
    video = torch.randn(B, T, 3, kw["img_size"], kw["img_size"], generator=g)
    feats = net(video.cuda())
    feats.square().sum().backward()
    cfg = O.VideoCfg(img_size=kw["img_size"], patch_size=kw["patch_size"], embed_dim=kw["embed_dim"], depth=kw["depth"],
                     num_heads=kw["num_heads"], num_frames=kw["num_frames"])
    sdo = {"video_model." + k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.video_encoder(video, sdo, cfg)
    ref.square().sum().backward()
    r = rel(feats, ref)
    print("%s: feats rel %.2e" % (name, r))
    assert r < PARITY
    last = kw["depth"] - 1
    for pname in ("blocks.0.timeattn.qkv.weight", "blocks.%d.mlp.fc2.weight" % last, "patch_embed.proj.weight",
                  "temporal_embed", "blocks.%d.attn.proj.bias" % (last // 2)):
        got = dict(net.named_parameters())[pname].grad
        want = sdo["video_model." + pname].grad
        rg = rel(got, want)
        print("  grad %-36s rel %.2e" % (pname, rg))
        assert rg < 3 * PARITY, (pname, rg)


def test_egomcq_validation_epoch_matches_oracle(full):
    """`Multi_Trainer_dist._valid_epoch` (reference trainer/trainer_egoclip.py:182-275): six synthetic EgoMCQ questions (one
    text query, five candidate clips each) through the drop-in in eval mode vs the CPU oracle on the same weights."""
    import types as _types
    from egovlp_amd.model.metric import egomcq_accuracy_metrics
    from egovlp_amd.trainer.trainer_egoclip import Multi_Trainer_dist
    m, sd = full
    g = torch.Generator().manual_seed(99)
    questions = []
    for q in range(6):
        video = torch.randn(1, 5, 4, 3, 224, 224, generator=g)
        ids = torch.randint(1000, 30000, (1, 16), generator=g)
        ids[:, 0] = 101
        mask = torch.ones(1, 16, dtype=torch.long)
        mask[:, 10 + q:] = 0
        questions.append({"video": video, "text": {"input_ids": ids, "attention_mask": mask},
                          "correct": torch.tensor([q % 5]), "type": torch.tensor([1 + q % 2])})

    class Loader(list):
        batch_size = 1
        dataset_name = "EgoMCQ-synthetic"

    import tempfile
    from egovlp_amd.utils.config import DictConfig
    args = _types.SimpleNamespace(world_size=1, rank=0, local_rank=0)
    config = DictConfig({"n_gpu": 1, "trainer": {"epochs": 1, "save_period": 1, "verbosity": 2, "monitor": "off",
                                                 "init_val": False}}, save_dir=tempfile.mkdtemp())
    tr = Multi_Trainer_dist(args, m, None, [egomcq_accuracy_metrics], None, config, [Loader()],
                            valid_data_loader=[Loader([dict(q) for q in questions])], len_epoch=1)
    res = tr._valid_epoch(1)
    pred = tr.last_val_predictions[0]
    assert pred.shape == (6, 5)
    ref = []
    with torch.no_grad():
        for q in questions:
            te, ve = O.frozen_in_time({"video": q["video"][0], "text": q["text"]}, sd, O.VideoCfg(), O.TextCfg())
            ref.append(O.sim_matrix(te, ve))
    ref = torch.cat(ref)
    assert rel(pred, ref) < PARITY
    want = egomcq_accuracy_metrics(ref, torch.cat([q["correct"] for q in questions]), torch.cat([q["type"] for q in questions]))
    assert res["nested_val_metrics"][0]["egomcq_accuracy_metrics"] == want
    # the same epoch through HIP-graph replay (args.graph_eval): one capture for the one input shape, same predictions
    args.graph_eval = True
    tr.valid_data_loader = [Loader([dict(q) for q in questions])]
    res_g = tr._valid_epoch(1)
    pred_g = tr.last_val_predictions[0]
    fwd = tr.last_graphed_forward
    print("graphed EgoMCQ epoch: %d captures, %d replays, max |pred - eager| %.2e"
          % (fwd.stats["captures"], fwd.stats["replays"], float((pred_g - pred).abs().max())))
    assert fwd.stats == {"captures": 1, "replays": 6}
    assert float((pred_g - pred).abs().max()) < 1e-6
    assert res_g["nested_val_metrics"][0]["egomcq_accuracy_metrics"] == want
    m.train()


def test_uint8_frames_give_the_same_video_embeddings(full):
    """Decoded uint8 frames in (x / 255 and ImageNet Normalize fused into the patch gather) == fp32 frames normalised on the host."""
    from egovlp_amd import ops
    m, _ = full
    g = torch.Generator().manual_seed(5)
    u8 = torch.randint(0, 256, (2, 4, 3, 224, 224), generator=g, dtype=torch.uint8)
    mean, std = torch.tensor(ops.IMAGENET_MEAN).view(1, 1, 3, 1, 1), torch.tensor(ops.IMAGENET_STD).view(1, 1, 3, 1, 1)
    host = (u8.float() / 255).sub(mean).div(std)
    m.eval()
    with torch.no_grad():
        a = m.compute_video(u8.cuda())
        b = m.compute_video(host.cuda())
    assert torch.equal(a, b)
    m.train()


def test_full_size_batch_is_consistent_with_oracle_rows_and_with_its_halves(full):
    """BASELINE configs[1] size (B = 32, T = 4: the 25 120-token GEMM shapes bench.py runs): the embeddings of the full batch
    must (a) equal the CPU oracle on a few of its rows computed one clip at a time (the encoders are per-sample), and (b) not
    depend on how the batch is cut (two halves of 16: other tile counts, same numbers up to summation order)."""
    m, sd = full
    batch = synth_batch(32, T=4, L=32, seed=2024, ragged=True)
    dev = to_dev(batch)
    m.eval()
    with torch.no_grad():
        te, ve = m(dev)
        halves = [m({"video": dev["video"][i:i + 16], "text": {k: v[i:i + 16] for k, v in dev["text"].items()}}) for i in (0, 16)]
    te2, ve2 = torch.cat([h[0] for h in halves]), torch.cat([h[1] for h in halves])
    assert rel(te2, te) < 1e-4 and rel(ve2, ve) < 1e-4      # other tile counts and split-K factors: bf16x3 noise level (~2e-5)
    rows = [0, 13, 31]
    with torch.no_grad():
        for r in rows:
            one = {"video": batch["video"][r:r + 1], "text": {k: v[r:r + 1] for k, v in batch["text"].items()}}
            rt, rv = O.frozen_in_time(one, sd, O.VideoCfg(), O.TextCfg())
            assert rel(te[r:r + 1], rt) < PARITY and rel(ve[r:r + 1], rv) < PARITY, r
    m.train()


def test_full_size_train_step_matches_oracle_on_the_whole_batch(full):
    """BASELINE configs[1] at full size (B = 32, T = 4: M = 25 120 tokens -- the only size where the 320-row tiles, split-K 7..28
    and the 25 120-row TN weight gradients all run together): loss, both embedding gradients and sentinel weight gradients of
    ONE backward against the CPU oracle on the whole batch (fp32 autograd, ~1 min on the box's cores), in the parity mode
    (everything at 1e-3 / 3e-3), in the bf16-backward modes (gradients inside MIXED_GRAD) and in the BENCHMARKED mode 'f16mix' with its
    fp16 backward (gradients inside F16_GRAD = 1e-2; round-5 verdict: "MIXED_GRAD tightened to <= 1e-2 at B = 32 vs the oracle")."""
    from egovlp_amd.model.loss import EgoNCE
    from egovlp_amd.ops import Precision
    m, sd = full
    B = 32
    batch = synth_batch(B, T=4, L=32, seed=777, ragged=True)
    sentinels = ["video_model.blocks.0.timeattn.qkv.weight", "video_model.blocks.11.mlp.fc2.weight",
                 "video_model.patch_embed.proj.weight", "text_model.transformer.layer.0.attention.q_lin.weight",
                 "video_model.blocks.5.attn.proj.weight", "video_model.blocks.6.mlp.fc1.bias", "video_model.blocks.2.norm2.weight",
                 "video_model.temporal_embed", "vid_proj.0.weight"]
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    sdo = {k: v.clone().requires_grad_(k in sentinels) for k, v in sd.items()}
    rt, rv = O.frozen_in_time(batch, sdo, O.VideoCfg(), O.TextCfg())
    rt.retain_grad(); rv.retain_grad()
    rl, _ = O.egoclip_loss(rt, rv, batch["noun_vec"], batch["verb_vec"])
    rl.backward()
    m.load_state_dict(sd, strict=True)
    m.train()
    dev = to_dev(batch)
    params = dict(m.named_parameters())
    try:
        for mode in ("bf16x3", "mixed", "f16x2", "f16mix", "f16mix/bf16"):
            _set_mode(mode)
            fbar, gbound = _fbar(mode), _gbar(mode)
            for p_ in m.parameters():
                p_.grad = None
            te, ve = m(dev)
            te.retain_grad(); ve.retain_grad()
            loss = EgoNCE().fused(te, ve, dev["noun_vec"], dev["verb_vec"])
            _backward(m, loss, (te, ve))
            r_t, r_v, r_l = rel(te, rt), rel(ve, rv), abs(float(loss) - float(rl)) / abs(float(rl))
            print("B=32 %s: text rel %.2e video rel %.2e loss rel %.2e | d_text %.2e d_video %.2e" % (
                mode, r_t, r_v, r_l, rel(te.grad, rt.grad), rel(ve.grad, rv.grad)))
            assert r_t < fbar and r_v < fbar and r_l < fbar
            assert rel(te.grad, rt.grad) < PARITY and rel(ve.grad, rv.grad) < PARITY      # the contrastive head is fp32 in every mode
            for name in sentinels:
                r = rel(params[name].grad, sdo[name].grad)
                print("  B=32 %s grad %-55s rel %.2e" % (mode, name, r))
                assert r < gbound, (mode, name, r)
    finally:
        Precision.set("bf16x3")
        for p_ in m.parameters():
            p_.grad = None


@pytest.mark.parametrize("mode", ["bf16x3", "f16x2", "f16mix"])
@pytest.mark.parametrize("name,arch,T,model_frames", [("config4_T16", "base_patch16_224", 16, 16), ("config5_vitl14", "large_patch14_224", 4, 4)])
def test_other_baseline_configs_full_model_and_egonce_match_oracle(name, arch, T, model_frames, mode, request):
    """BASELINE configs 4 (16 frames) and 5 (ViT-L/14) through the FULL dual encoder + EgoNCE at B = 2: embeddings, loss,
    embedding gradients and two weight gradients vs the CPU oracle, in the parity mode (1e-3 / 3e-3) and in the f16x2 mode
    (forward inside X2_BAR, gradients inside MIXED_GRAD)."""
    from egovlp_amd.model.loss import EgoNCE
    from egovlp_amd.model.model import FrozenInTime
    from egovlp_amd.ops import Precision
    _set_mode(mode)
    request.addfinalizer(lambda: Precision.set("bf16x3"))
    fbar, gbar = _fbar(mode), _gbar(mode)
    m = FrozenInTime(video_params={"model": "SpaceTimeTransformer", "arch_config": arch, "num_frames": model_frames,
                                   "pretrained": True, "time_init": "rand"},
                     text_params={"model": "distilbert-base-uncased", "pretrained": True, "input": "text"},
                     projection="minimal", load_checkpoint="")
    sd = synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, seed=9)
    m.load_state_dict(sd, strict=True)
    m.text_model.set_dropout(0.0, 0.0)
    m = m.cuda().train()
    batch = synth_batch(2, T=T, L=32, seed=31, ragged=True)
    dev = to_dev(batch)
    te, ve = m(dev)
    te.retain_grad(); ve.retain_grad()
    loss = EgoNCE().fused(te, ve, dev["noun_vec"], dev["verb_vec"])
    _backward(m, loss, (te, ve))
    large = arch == "large_patch14_224"
    cfg = O.VideoCfg(patch_size=14, embed_dim=1024, depth=24, num_heads=16, num_frames=model_frames) if large \
        else O.VideoCfg(num_frames=model_frames)
    last = cfg.depth - 1
    watch = ["video_model.blocks.%d.mlp.fc1.weight" % last, "video_model.blocks.0.attn.qkv.weight", "txt_proj.1.weight"]
    sdo = {k: v.clone().requires_grad_(k in watch) for k, v in sd.items()}
    rt, rv = O.frozen_in_time(batch, sdo, cfg, O.TextCfg())
    rt.retain_grad(); rv.retain_grad()
    rl, _ = O.egoclip_loss(rt, rv, batch["noun_vec"], batch["verb_vec"])
    rl.backward()
    errs = {"text": rel(te, rt), "video": rel(ve, rv), "loss": abs(float(loss) - float(rl)) / abs(float(rl)),
            "d_text": rel(te.grad, rt.grad), "d_video": rel(ve.grad, rv.grad)}
    print(name, mode, {k: "%.2e" % v for k, v in errs.items()})
    assert all(errs[k] < fbar for k in ("text", "video", "loss")) and errs["d_text"] < PARITY and errs["d_video"] < PARITY, errs
    params = dict(m.named_parameters())
    for w in watch:
        r = rel(params[w].grad, sdo[w].grad)
        print("  %s %s grad %-45s rel %.2e" % (name, mode, w, r))
        assert r < gbar, (w, r)


@pytest.mark.parametrize("which", ["wgrad", "text", "both"])
def test_weight_gradients_on_the_side_stream_are_the_same_gradients(full, which):
    """ops.WGRAD_SIDE_STREAM: every wgrad GEMM runs on a second HIP stream behind an event of the main stream and the main
    stream re-joins at the end of backward (autograd engine callback).  Same kernels, same inputs -> the same gradients up to
    the run-to-run noise of the few fp32-atomic reductions upstream (LayerNorm dgamma, CLS-token gradients: ~1e-6); a missing
    join or a recycled input buffer would show up as an O(1) error."""
    from egovlp_amd import ops
    m, _ = full
    m.train()
    batch = to_dev(synth_batch(4, T=4, L=32, seed=77))
    from egovlp_amd.model.loss import EgoNCE
    lossf = EgoNCE()

    def grads():
        for p in m.parameters():
            p.grad = None
        te, ve = m(batch)
        lossf.fused(te, ve, batch["noun_vec"], batch["verb_vec"]).backward()
        return {k: p.grad.clone() for k, p in m.named_parameters()}

    try:
        ref = grads()
        ops.WGRAD_SIDE_STREAM = which in ("wgrad", "both")
        ops.TEXT_SIDE_STREAM = which in ("text", "both")      # DistilBERT tower on its own stream under the video tower
        for _ in range(3):                       # repeated: a missing join shows up as a stale / half-written gradient
            got = grads()
            worst = max((rel(got[k], ref[k]), k) for k in ref)
            print("side streams (%s): worst gradient rel %.2e (%s)" % ((which,) + worst))
            assert worst[0] < 1e-4, worst
    finally:
        ops.WGRAD_SIDE_STREAM = ops.TEXT_SIDE_STREAM = False
        m.eval()


def test_padded_patch_embedding_gradient_with_the_wgrad_side_stream():
    """ViT-L/14's patch embedding contracts over K = 3 x 14 x 14 = 588, zero-padded to 640 for the k-tile; the padding is cut off
    the weight gradient by an ATen copy on the node's stream, so THAT wgrad must stay on it: with the wgrad side streams on (as
    bench.py and the trainer run) the copy raced with the side-stream GEMM and the gradient came out 100 % wrong (bench.py's
    grad_rel_err of config 5, rounds 3 - 4).  One ViT-L/14-shaped block, side stream off vs on."""
    from egovlp_amd.model.loss import EgoNCE
    from egovlp_amd.model.model import FrozenInTime
    torch.manual_seed(0)
    m = FrozenInTime(video_params={"model": "SpaceTimeTransformer", "arch_config": "custom", "num_frames": 4, "pretrained": True,
                                   "time_init": "rand", "arch_kwargs": dict(img_size=224, patch_size=14, embed_dim=1024, depth=1, num_heads=16)},
                     text_params={"model": "distilbert-base-uncased", "pretrained": True, "input": "text",
                                  "config": dict(vocab_size=30522, dim=768, n_layers=1, n_heads=12, hidden_dim=3072)},
                     projection="minimal", load_checkpoint="").cuda().train()
    m.text_model.set_dropout(0.0, 0.0)
    # the all-bf16x3 backward: no bf16 rounding sits behind the CLS rows' fp32 atomics, so the two runs agree to the atomics' own ~1e-6 and
    # the bar can be TIGHT (round-5 advisor: with a single-pass backward one bf16 flip was worth 4e-4 and the bar had been loosened to
    # 2e-3, which a partial race -- a few stale rows -- could have hidden under); the stream policy under test is the same in every mode
    m.exec_ctx.set_precision("bf16x3", "bf16x3")
    batch = to_dev(synth_batch(4, T=4, L=16, seed=5))
    w = m.video_model.patch_embed.proj.weight

    def grad(side):
        m.exec_ctx.set(wgrad_side_stream=side)
        for p in m.parameters():
            p.grad = None
        te, ve = m(batch)
        EgoNCE().fused(te, ve, batch["noun_vec"], batch["verb_vec"]).backward()
        torch.cuda.synchronize()
        return w.grad.clone()

    ref = grad(False)
    assert float(ref.abs().max()) > 0
    for _ in range(3):
        got = grad(True)
        r = rel(got, ref)
        worst = float((got - ref).abs().max() / ref.abs().max())
        # same kernels, same inputs: what differs is the order of the fp32 atomics of the CLS rows; the race this test guards against made
        # the gradient 100 % wrong, a partial one (stale rows) would show in the max-abs element error
        assert r < 1e-4 and worst < 1e-3, (r, worst)


def test_retrieval_heads_match_the_reference_golden(full, golden_dir):
    """§8(f4) consumers of the encoders: compute_text_tokens (NLQ / MQ feature dumps), forward(video_only=True) (OSCC / PNR
    heads, feature dumps) and the dual-softmax retrieval similarity of run/test_epic.py, against outputs of the reference."""
    from egovlp_amd.model.model import dual_softmax_similarity, sim_matrix_mm
    g = np.load(os.path.join(golden_dir, "retrieval.npz"))
    m, _ = full
    m.eval()
    batch = to_dev(synth_batch(4, T=4, L=32, seed=1234, ragged=True))
    with torch.no_grad():
        tok = m.compute_text_tokens(batch["text"])
        vid = m(batch, video_only=True)
    assert tok.shape == (4, 32, 256)
    print("text tokens rel %.2e, video_only rel %.2e" % (rel(tok, g["text_tokens"]), rel(vid, g["video_only"])))
    assert rel(tok, g["text_tokens"]) < PARITY and rel(vid, g["video_only"]) < PARITY
    for shape in ("7x5", "150x96", "300x410"):
        t, v = torch.from_numpy(g["text_" + shape]).cuda(), torch.from_numpy(g["video_" + shape]).cuda()
        assert rel(sim_matrix_mm(t, v), t.double().cpu() @ v.double().cpu().t()) < 2e-5
        got = dual_softmax_similarity(t, v)
        r = rel(got, g["dual_" + shape])
        print("dual softmax %s rel %.2e" % (shape, r))
        assert r < 1e-4, shape                               # exp() of O(100) logits amplifies the GEMM's 1e-5



@pytest.mark.parametrize("mode", ["bf16x3", "f16x2", "f16mix"])
@pytest.mark.parametrize("name,arch,T,model_frames", [("config4_T16_B16", "base_patch16_224", 16, 16),
                                                      ("config5_vitl14_B16", "large_patch14_224", 4, 4)])
def test_other_baseline_configs_at_full_size_match_oracle_rows_and_their_halves(name, arch, T, model_frames, mode, request):
    """BASELINE configs 4 (T = 16: M = 16 x 3137 = 50 192 tokens) and 5 (ViT-L/14: M = 16 x 1025 = 16 400 tokens, the 640-deep
    zero-padded patch GEMM at its full 16 384 rows) at the BENCHMARKED batch B = 16 -- the tile counts, quantisation and split-K
    factors bench.py really runs (round-2 verdict, weak #2: these configs were compared with the oracle at B = 1..2 only).
    As for configs[1]: (a) three rows of the batch equal the CPU oracle run one clip at a time (the encoders are per-sample),
    (b) the batch equals its two halves of 8 up to summation order, (c) the EgoNCE loss on the full batch equals the oracle's
    loss on the device embeddings' oracle counterparts for the checked rows' sub-batch."""
    from egovlp_amd.model.model import FrozenInTime
    from egovlp_amd.ops import Precision
    _set_mode(mode)
    request.addfinalizer(lambda: Precision.set("bf16x3"))
    fbar = _fbar(mode, per_row=True)
    m = FrozenInTime(video_params={"model": "SpaceTimeTransformer", "arch_config": arch, "num_frames": model_frames,
                                   "pretrained": True, "time_init": "rand"},
                     text_params={"model": "distilbert-base-uncased", "pretrained": True, "input": "text"},
                     projection="minimal", load_checkpoint="")
    sd = synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, seed=9)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    B = 16
    batch = synth_batch(B, T=T, L=32, seed=4100, ragged=True)
    dev = to_dev(batch)
    with torch.no_grad():
        te, ve = m(dev)
        halves = [m({"video": dev["video"][i:i + 8], "text": {k: v[i:i + 8] for k, v in dev["text"].items()}}) for i in (0, 8)]
    te2, ve2 = torch.cat([h[0] for h in halves]), torch.cat([h[1] for h in halves])
    r_ht, r_hv = rel(te2, te), rel(ve2, ve)
    large = arch == "large_patch14_224"
    cfg = O.VideoCfg(patch_size=14, embed_dim=1024, depth=24, num_heads=16, num_frames=model_frames) if large \
        else O.VideoCfg(num_frames=model_frames)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    worst = 0.0
    rows = [0, 7, 15]
    ref_t, ref_v = [], []
    with torch.no_grad():
        for r in rows:
            one = {"video": batch["video"][r:r + 1], "text": {k: v[r:r + 1] for k, v in batch["text"].items()}}
            rt, rv = O.frozen_in_time(one, sd, cfg, O.TextCfg())
            ref_t.append(rt); ref_v.append(rv)
            e = max(rel(te[r:r + 1], rt), rel(ve[r:r + 1], rv))
            worst = max(worst, e)
            assert e < fbar, (name, mode, r, e)
    # the contrastive head on those rows: device embeddings vs oracle embeddings through the same oracle loss
    idx = torch.tensor(rows)
    l_dev, _ = O.egoclip_loss(te[idx].cpu(), ve[idx].cpu(), batch["noun_vec"][idx], batch["verb_vec"][idx])
    l_ref, _ = O.egoclip_loss(torch.cat(ref_t), torch.cat(ref_v), batch["noun_vec"][idx], batch["verb_vec"][idx])
    r_l = abs(float(l_dev) - float(l_ref)) / abs(float(l_ref))
    print("%s %s: rows vs oracle worst rel %.2e | halves text %.2e video %.2e | loss rel %.2e" % (name, mode, worst, r_ht, r_hv, r_l))
    # other tile counts: summation order only in bf16x3 (~2e-5); f16x2 has no batch-dependent rounding either
    assert r_ht < 1e-4 and r_hv < 1e-4 and r_l < fbar


@pytest.mark.parametrize("dist", ["gauss", "heavy"])
def test_precision_guard_measures_the_policy_and_demotes_it_on_hostile_weights(dist):
    """The per-block single-product policy of 'f16mix' was tuned on Gaussian weights (round-5 verdict, weak #1 / advisor): on the
    HOSTILE distribution (egovlp_amd.synth.heavy_tensor: log-normal channel scales, x30 outlier LayerNorm gains on three residual
    channels that also carry a token-independent offset, x8 fc1 rows) the shipped policy measures ~1.0e-3 on the video embedding where
    the fp32-grade modes stay at 4e-4.  egovlp_amd.guard.PrecisionGuard measures the policy against the all-bf16x3 forward of the same
    batch ON THE DEVICE and demotes it rung by rung until it is inside its budget: on Gaussian weights nothing happens (4.7e-4 < 6e-4),
    on the hostile ones the demotion fires -- and what is left in force meets the north-star bar against the fp32 CPU oracle, per batch
    and per row, where the un-guarded policy does not.  Raises nothing, clips nothing silently (the demotion is logged)."""
    from egovlp_amd.guard import PrecisionGuard
    from egovlp_amd.model.model import FrozenInTime
    m = FrozenInTime(video_params={"model": "SpaceTimeTransformer", "arch_config": "base_patch16_224", "num_frames": 4, "pretrained": True,
                                   "time_init": "rand"},
                     text_params={"model": "distilbert-base-uncased", "pretrained": True, "input": "text"}, projection="minimal",
                     load_checkpoint="")
    sd = synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, seed=0, dist=dist)
    m.load_state_dict(sd)
    m = m.cuda().train()
    batch = synth_batch(4, T=4, L=32, seed=99, ragged=True)
    dev = to_dev(batch)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    with torch.no_grad():
        rt, rv = O.frozen_in_time(batch, sd, O.VideoCfg(num_frames=4), O.TextCfg())

    def video_err():
        m.eval()
        with torch.no_grad():
            _, ve = m(dev)
        m.train()
        return rel(ve, rv), max(rel(ve[i], rv[i]) for i in range(ve.shape[0]))

    ec = m.exec_ctx
    ec.set_precision("f16mix", "f16")
    unguarded = video_err()
    g = PrecisionGuard(m)
    rep = g.check(dev)
    guarded = video_err()
    print("precision guard on %s weights: un-guarded policy %.2e (worst row %.2e) from the oracle; guard tried %s -> policy %s, %.2e (worst row %.2e)" % (
        dist, unguarded[0], unguarded[1], [(str(t["policy"]), "%.2e" % t["err"]) for t in rep["tried"]], rep["policy"], guarded[0], guarded[1]))
    assert rep["tried"][0]["policy"] == "auto" and all(t["finite"] for t in rep["tried"])
    assert guarded[0] < 8e-4 and guarded[1] < PARITY                      # what is left in force holds the bar, rows included
    if dist == "gauss":
        assert not rep["demoted"] and rep["policy"] == "auto" and ec.precision_name() == ("f16mix", "f16")
        assert unguarded[0] == guarded[0]
    else:
        assert rep["demoted"] and rep["policy"] != "auto" and len(rep["tried"]) >= 2
        assert unguarded[1] > PARITY > guarded[1]                        # the demotion is what brings the rows back inside the bar
        assert rep["tried"][-1]["err"] <= g.budget < rep["tried"][0]["err"]
        assert ec.precision_name()[1] == "f16"                            # the backward precision is the caller's
    # a second check keeps the rung (no flapping), and a training step runs in the policy left in force
    again = g.check(dev)
    assert again["policy"] == rep["policy"] and len(again["tried"]) == 1
    from egovlp_amd.model.loss import EgoNCE
    from egovlp_amd.optim import AdamW
    from egovlp_amd.trainer.trainer_egoclip import egoclip_step
    loss = egoclip_step(m, EgoNCE(), AdamW(m.parameters(), lr=3e-5), dev)
    assert bool(torch.isfinite(loss))


def test_precision_guard_answers_fp16_saturation_with_the_bf16x3_forward():
    """fp16 operand planes are written saturating (csrc/f16x2.h f16x2_clamp: an activation beyond +-65504 is clipped, not inf) -- the
    advisor's round-5 point: nothing in the fp16-product forward itself notices.  The guard does: block 5's fc1 scaled by 1e5 (and fc2
    by 1e-5: the product is unchanged in exact arithmetic, model/video_transformer.py:41-52) puts |gelu(z)| near 1e5 and fc2's weights
    into fp16's subnormals; every fp16 rung of the ladder measures far outside the budget, the guard ends on the all-bf16x3 forward
    (bf16's exponent range), and what it leaves in force is inside the parity bar against the fp32 CPU oracle."""
    from egovlp_amd.guard import PrecisionGuard
    from egovlp_amd.model.model import FrozenInTime
    m = FrozenInTime(video_params={"model": "SpaceTimeTransformer", "arch_config": "base_patch16_224", "num_frames": 4, "pretrained": True,
                                   "time_init": "rand"},
                     text_params={"model": "distilbert-base-uncased", "pretrained": True, "input": "text"}, projection="minimal",
                     load_checkpoint="")
    sd = synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, seed=0)
    for k in list(sd):
        if k.startswith("video_model.blocks.5.mlp.fc1."):
            sd[k] = sd[k] * 1e5
        if k == "video_model.blocks.5.mlp.fc2.weight":
            sd[k] = sd[k] * 1e-5
    m.load_state_dict(sd)
    m = m.cuda().train()
    batch = synth_batch(4, T=4, L=32, seed=99, ragged=True)
    dev = to_dev(batch)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    with torch.no_grad():
        _, rv = O.frozen_in_time(batch, sd, O.VideoCfg(num_frames=4), O.TextCfg())

    def video_err():
        m.eval()
        with torch.no_grad():
            _, ve = m(dev)
        m.train()
        return rel(ve, rv)

    ec = m.exec_ctx
    ec.set_precision("f16mix", "f16")
    unguarded = video_err()
    g = PrecisionGuard(m)
    rep = g.check(dev)
    guarded = video_err()
    print("saturating fc1 (x 1e5): un-guarded %.2e from the oracle; guard tried %s -> %s, %.2e" % (
        unguarded, [(str(t["policy"]), "%.2e" % t["err"]) for t in rep["tried"]], rep["policy"], guarded))
    assert unguarded > 10 * PARITY                       # the clipped activations are a gross error, and a silent one
    assert rep["demoted"] and rep["policy"] == "bf16x3" and ec.precision_name() == ("bf16x3", "bf16")
    assert [str(t["policy"]) for t in rep["tried"]][0] == "auto" and len(rep["tried"]) == 5
    assert guarded < PARITY
