"""The classification fine-tune heads (the part round 2 left out): OSCC / PNR run FrozenInTime with
`projection_dim` = 2 / 17 on `video_only=True` features and train with CrossEntropy (configs/ft/oscc.json, pnr.json;
model/loss.py:135-141; trainer/trainer_oscc.py:335-338).  The loss kernel is compared with torch's own nn.CrossEntropyLoss (which
IS the reference's loss) and the whole head -- video encoder, 768 -> 2 projection, loss, backward -- with the CPU oracle."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from egovlp_amd.synth import synth_state_dict  # noqa: E402
from oracle import egovlp_oracle as O  # noqa: E402


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("rows,cols,ignored", [(64, 2, 0), (37, 17, 5), (1, 2, 0), (300, 1000, 17)])
def test_cross_entropy_matches_torch(rows, cols, ignored):
    from egovlp_amd.model.loss import CrossEntropy
    g = torch.Generator().manual_seed(rows * 131 + cols)
    x = (3.0 * torch.randn(rows, cols, generator=g)).requires_grad_(True)
    t = torch.randint(0, cols, (rows,), generator=g)
    if ignored:
        t[torch.randperm(rows, generator=g)[:ignored]] = -100
    ref = torch.nn.CrossEntropyLoss()(x, t)
    ref.backward()
    xd = x.detach().cuda().requires_grad_(True)
    loss = CrossEntropy()(xd, t.cuda())
    (2.0 * loss).backward()                                   # a non-unit upstream gradient
    assert abs(float(loss) - float(ref)) < 1e-5 * max(1.0, abs(float(ref)))
    assert rel(xd.grad, 2.0 * x.grad) < 1e-5
    if ignored:
        assert float(xd.grad[t.cuda() == -100].abs().max()) == 0.0


def test_cross_entropy_all_ignored_is_nan_like_torch():
    from egovlp_amd.model.loss import CrossEntropy
    x = torch.randn(4, 3).cuda()
    t = torch.full((4,), -100).cuda()
    assert torch.isnan(CrossEntropy()(x, t)) and torch.isnan(torch.nn.CrossEntropyLoss()(x.cpu(), t.cpu()))


@pytest.mark.parametrize("classes", [2, 17])
def test_oscc_pnr_head_train_step_matches_oracle(classes):
    """configs/ft/oscc.json / pnr.json: FrozenInTime(projection_dim = classes), `model(data, video_only=True)` scores,
    CrossEntropy, backward -- against the oracle's video encoder + a linear head + torch's cross-entropy on the CPU."""
    from egovlp_amd.model.loss import CrossEntropy
    from egovlp_amd.model.model import FrozenInTime
    from egovlp_amd.ops import Precision
    Precision.set("bf16x3")
    m = FrozenInTime(video_params={"model": "SpaceTimeTransformer", "arch_config": "base_patch16_224", "num_frames": 4,
                                   "pretrained": True, "time_init": "rand"},
                     text_params={"model": "distilbert-base-uncased", "pretrained": True, "input": "text"},
                     projection="minimal", projection_dim=classes, load_checkpoint="")
    sd = synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, seed=21)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    B, T = 6, 4
    g = torch.Generator().manual_seed(8)
    video = torch.randn(B, T, 3, 224, 224, generator=g)
    state = torch.randint(0, classes, (B,), generator=g)
    scores = m({"video": video.cuda()}, video_only=True)
    assert scores.shape == (B, classes)
    loss = CrossEntropy()(scores, state.cuda())
    loss.backward()
    watch = ["vid_proj.0.weight", "vid_proj.0.bias", "video_model.blocks.11.mlp.fc2.weight", "video_model.blocks.0.attn.qkv.weight"]
    sdo = {k: v.clone().requires_grad_(k in watch) for k, v in sd.items()}
    feats = O.video_encoder(video, sdo, O.VideoCfg(num_frames=4))
    ref_scores = F.linear(feats, sdo["vid_proj.0.weight"], sdo["vid_proj.0.bias"])
    ref_loss = F.cross_entropy(ref_scores, state)
    ref_loss.backward()
    errs = {"scores": rel(scores, ref_scores), "loss": abs(float(loss) - float(ref_loss)) / abs(float(ref_loss))}
    params = dict(m.named_parameters())
    for w in watch:
        errs["d " + w] = rel(params[w].grad, sdo[w].grad)
    print("head classes=%d:" % classes, {k: "%.2e" % v for k, v in errs.items()})
    assert errs["scores"] < 1e-3 and errs["loss"] < 1e-3
    assert all(v < 3e-3 for k, v in errs.items() if k.startswith("d ")), errs
    assert all(p.grad is None for k, p in m.named_parameters() if k.startswith("text_model.") or k.startswith("txt_proj."))
