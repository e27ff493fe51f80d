"""Import the REAL reference (showlab/EgoVLP at /root/reference) on CPU -- test infrastructure.

Only `tests/golden/make_golden.py` uses this, and only inside the build container:
/root/reference does not exist on the GPU box, so nothing at test/bench/smoke run time
may call it.  The reference is pure Python but imports packages that are absent here
(timm, torchvision, sacred, tensorboardX, cv2, decord, av, ffmpeg, humanize, ipdb,
dominate); SURVEY 8c / Appendix B give the verified stub recipe reproduced here.
Order matters: transformers must be imported and a DistilBERT instantiated BEFORE any
stub exists, every stub needs a ModuleSpec, and the fake ViT checkpoint must be non-empty.
"""
import importlib.machinery as mach
import os
import sys
import types

import torch
import torch.nn as nn

REF = os.environ.get("EGOVLP_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "model"))


def load_reference():
    """Returns (model.model, model.loss, trainer.trainer_egoclip, model.video_transformer)."""
    import transformers
    from transformers import DistilBertConfig, DistilBertModel
    _ = DistilBertModel(DistilBertConfig(n_layers=1))

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__spec__ = mach.ModuleSpec(name, None)
        m.__path__ = []
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    for n in ["av", "cv2", "ffmpeg", "humanize", "ipdb", "sacred", "tensorboardX", "dominate"]:
        if n not in sys.modules:
            stub(n)
    if "decord" not in sys.modules:
        stub("decord", bridge=types.SimpleNamespace(set_bridge=lambda *_: None))
    if "torchvision" not in sys.modules:
        tv = stub("torchvision")
        tv.transforms = stub("torchvision.transforms")

    class DropPath(nn.Module):          # identity at p=0 (all video drop rates are 0)
        def __init__(self, p=0.0):
            super().__init__()

        def forward(self, x):
            return x

    if "timm" not in sys.modules:
        t = stub("timm")
        t.models = stub("timm.models")
        t.models.layers = stub(
            "timm.models.layers", DropPath=DropPath,
            to_2tuple=lambda x: x if isinstance(x, tuple) else (x, x),
            trunc_normal_=lambda w, std=1.0: nn.init.trunc_normal_(w, std=std, a=-2.0, b=2.0))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    transformers.AutoModel.from_pretrained = classmethod(
        lambda cls, *a, **k: DistilBertModel(DistilBertConfig(), ).eval())
    _load = torch.load

    def fake_load(p, *a, **k):
        if "jx_vit_base" in str(p):
            return {"_dummy": torch.zeros(1)}
        return _load(p, *a, **k)

    torch.load = fake_load
    import model.model as mm
    import model.loss as ml
    import model.video_transformer as mv
    import trainer.trainer_egoclip as te
    return mm, ml, te, mv
