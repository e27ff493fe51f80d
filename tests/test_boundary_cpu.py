"""The drop-in boundary on the CPU (no GPU needed: modules are parameter containers until `forward`).

* the reference's own ConfigParser, fed `configs/pt/egoclip.json` UNCHANGED, instantiates FrozenInTime / EgoNCE / AdamW
  from the drop-in modules by reflection, exactly as run/train_egoclip.py:63,69,73 do (327 keys, 180.93 M parameters);
* a checkpoint in the reference's format (`module.` prefix, pickled ConfigParser) loads through
  `FrozenInTime(load_checkpoint=...)` with strict=True in a process that has no `parse_config` module;
* `_inflate_positional_embeds` equals the reference's own method for 16 -> 4 and 4 -> 16 frames.
The reference-dependent halves run in a subprocess (tests/ref_boundary_probe.py) and are skipped where /root/reference is
absent (the GPU box); the rest runs everywhere."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("EGOVLP_REFERENCE", "/root/reference")
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "model")), reason="reference checkout not present")

VIDEO_PARAMS = {"model": "SpaceTimeTransformer", "arch_config": "base_patch16_224", "num_frames": 16, "pretrained": True,
                "time_init": "zeros"}
TEXT_PARAMS = {"model": "distilbert-base-uncased", "pretrained": True, "input": "text"}


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    out = tmp_path_factory.mktemp("probe")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ref_boundary_probe.py"), str(out)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("PROBE ")][-1]
    return json.loads(line[6:]), out


@needs_ref
def test_reference_configparser_builds_the_dropin_classes_from_egoclip_json(probe):
    res, _ = probe
    assert res["model"] == "FrozenInTime" and res["keys"] == 327 and abs(res["params_M"] - 180.93) < 0.01
    assert res["n_trainable"] == 327 and res["model_frames"] == 16 and res["time_init_zero"]
    assert res["loss"] == "EgoNCE" and res["loss_temperature"] == 0.05
    # transformers==4.2.1 AdamW defaults (run/train_egoclip.py:73)
    assert res["optimizer"] == "AdamW" and res["lr"] == 3e-5 and res["eps"] == 1e-6
    assert res["betas"] == [0.9, 0.999] and res["weight_decay"] == 0.0


@needs_ref
def test_reference_format_checkpoint_loads_strict_without_parse_config(probe):
    from egovlp_amd.model.model import FrozenInTime
    from egovlp_amd.synth import synth_state_dict
    _, out = probe
    assert "parse_config" not in sys.modules
    path = os.path.join(str(out), "ref_format_checkpoint.pth")
    with pytest.raises(Exception):                  # what a plain torch.load does with this file
        torch.load(path, map_location="cpu")
    m = FrozenInTime(video_params=dict(VIDEO_PARAMS), text_params=dict(TEXT_PARAMS), projection="minimal",
                     load_checkpoint=path)
    want = synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, seed=21)
    for k, v in m.state_dict().items():
        assert torch.equal(v, want[k]), k


@needs_ref
def test_inflate_positional_embeds_matches_the_reference(probe):
    from egovlp_amd.model.model import FrozenInTime
    _, out = probe
    g = np.load(os.path.join(str(out), "ref_inflate.npz"))
    for load_frames, curr_frames, fix in [(16, 4, "zeros"), (4, 16, "zeros"), (4, 16, "bilinear")]:
        m = FrozenInTime(video_params={**VIDEO_PARAMS, "num_frames": curr_frames}, text_params=dict(TEXT_PARAMS),
                         projection="minimal", load_checkpoint="", load_temporal_fix=fix)
        key = f"{load_frames}_{curr_frames}_{fix}"
        sd = {"video_model.temporal_embed": torch.from_numpy(g["in_" + key]).clone(),
              "video_model.pos_embed": torch.zeros(1, 197, 768)}
        new = m._inflate_positional_embeds(sd)["video_model.temporal_embed"]
        assert new.shape == (1, curr_frames, 768)
        assert torch.equal(new, torch.from_numpy(g["out_" + key])), key


def test_checkpoint_with_unimportable_config_object_and_module_prefix(tmp_path):
    """The same load path without the reference: the pickled `config` is an instance of a class whose module disappears
    before loading (as parse_config.ConfigParser does on a machine without the reference)."""
    import types
    from collections import OrderedDict
    from egovlp_amd.model.model import FrozenInTime
    from egovlp_amd.synth import synth_state_dict
    from egovlp_amd.utils.util import load_checkpoint_file
    m0 = FrozenInTime(video_params={**VIDEO_PARAMS, "num_frames": 4}, text_params=dict(TEXT_PARAMS), projection="minimal",
                      load_checkpoint="")
    vals = synth_state_dict({k: v.shape for k, v in m0.state_dict().items()}, seed=5)
    mod = types.ModuleType("parse_config_gone")

    class ConfigParser:                               # pickled by reference (module + qualname), not by value
        def __init__(self):
            self._config = {"arch": {"type": "FrozenInTime"}, "optimizer": {"type": "AdamW"}}
    ConfigParser.__module__ = "parse_config_gone"
    ConfigParser.__qualname__ = "ConfigParser"
    mod.ConfigParser = ConfigParser
    sys.modules["parse_config_gone"] = mod
    path = str(tmp_path / "checkpoint-epoch1.pth")
    try:
        torch.save({"arch": "FrozenInTime", "epoch": 1, "monitor_best": 0.0, "optimizer": {},
                    "state_dict": OrderedDict(("module." + k, v) for k, v in vals.items()), "config": ConfigParser()}, path)
    finally:
        del sys.modules["parse_config_gone"]
    import pickle
    with pytest.raises(pickle.UnpicklingError):                     # not silently: the fallback is opt-in (advisor finding)
        load_checkpoint_file(path, map_location="cpu")
    with pytest.raises(FileNotFoundError):                          # and only the safe unpickler's refusal triggers it
        load_checkpoint_file(path + ".missing", map_location="cpu", trusted=True)
    with pytest.warns(UserWarning, match="allow-list unpickler"):
        ck = load_checkpoint_file(path, map_location="cpu", trusted=True)
    assert ck["epoch"] == 1 and ck["config"]["optimizer"]["type"] == "AdamW"        # read access like _resume_checkpoint's
    m = FrozenInTime(video_params={**VIDEO_PARAMS, "num_frames": 4}, text_params=dict(TEXT_PARAMS), projection="minimal",
                     load_checkpoint=path)
    for k, v in m.state_dict().items():
        assert torch.equal(v, vals[k]), k
    # 16-frame checkpoint into a 4-frame model: the first 4 rows survive (model/model.py:158-159)
    vals16 = dict(vals)
    vals16["video_model.temporal_embed"] = torch.randn(1, 16, 768)
    torch.save({"state_dict": vals16}, path)
    m4 = FrozenInTime(video_params={**VIDEO_PARAMS, "num_frames": 4}, text_params=dict(TEXT_PARAMS), projection="minimal",
                      load_checkpoint=path)
    assert torch.equal(m4.video_model.temporal_embed, vals16["video_model.temporal_embed"][:, :4])


def test_lenient_unpickler_never_resolves_code_carrying_globals(tmp_path):
    """A checkpoint-shaped pickle that names os.system / builtins.eval as a reducer: with the allow-list unpickler the global
    becomes an inert placeholder class -- nothing is executed -- while tensors and containers next to it load normally."""
    import pickle
    from egovlp_amd.utils.util import _LenientUnpickler, _Placeholder

    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ("echo pwned > %s" % (tmp_path / "pwned"),))

    class Evil2:
        def __reduce__(self):
            return (eval, ("__import__('os').getcwd()",))
    blob = pickle.dumps({"epoch": 3, "state_dict": {"w": [1.0, 2.0]}, "config": Evil(), "x": Evil2()})
    import io
    out = _LenientUnpickler(io.BytesIO(blob)).load()
    assert out["epoch"] == 3 and out["state_dict"]["w"] == [1.0, 2.0]
    assert isinstance(out["config"], _Placeholder) and isinstance(out["x"], _Placeholder)
    assert not (tmp_path / "pwned").exists()


def test_lenient_unpickler_allow_list_is_exact_not_a_module_prefix(tmp_path):
    """Globals that live UNDER allowed packages but run code when called (torch.hub.load, torch.load, numpy.load,
    torch.utils.cpp_extension.load) are placeholders too: the allow-list is (module, name) pairs, not `torch.*` / `numpy.*`;
    tensors, parameters, numpy arrays / scalars, OrderedDict and Paths still round-trip."""
    import io
    import pickle
    from collections import OrderedDict
    import numpy as np
    from egovlp_amd.utils.util import _LenientUnpickler, _Placeholder
    seen = []

    def reducer(fn, *args):
        class R:
            def __reduce__(self):
                return (fn, args)
        return R()
    import torch.hub
    import torch.utils.cpp_extension
    evil = {"hub": reducer(torch.hub.load, "someone/repo", "model"), "tl": reducer(torch.load, str(tmp_path / "x.pt")),
            "npl": reducer(np.load, str(tmp_path / "x.npy")), "ext": reducer(torch.utils.cpp_extension.load, "m", ["a.cpp"])}
    out = _LenientUnpickler(io.BytesIO(pickle.dumps(evil))).load()
    assert all(isinstance(v, _Placeholder) for v in out.values()), out
    good = {"sd": OrderedDict(w=torch.arange(6.).view(2, 3), p=torch.nn.Parameter(torch.ones(2))), "arr": np.arange(4, dtype=np.float32),
            "scalar": np.float64(2.5), "step": 7, "path": __import__("pathlib").PurePosixPath("/a/b")}
    path = tmp_path / "good.pth"
    torch.save(good, path)
    from egovlp_amd.utils.util import _LenientPickle
    back = torch.load(path, weights_only=False, pickle_module=_LenientPickle)
    assert torch.equal(back["sd"]["w"], good["sd"]["w"]) and torch.equal(back["sd"]["p"], good["sd"]["p"])
    assert np.array_equal(back["arr"], good["arr"]) and float(back["scalar"]) == 2.5 and back["step"] == 7 and str(back["path"]) == "/a/b"


def test_eval_token_padding_for_graph_replay():
    """`_pad_tokens` (EgoMCQ validation with args.graph_eval): captions are right-padded to a multiple of 8 tokens with [PAD]
    ids and masked positions, so a handful of captured graphs covers every caption length; shorter-than-multiple inputs keep
    their content, exact multiples are returned untouched."""
    import torch
    from egovlp_amd.trainer.trainer_egoclip import _pad_tokens
    t = {"input_ids": torch.arange(1, 12).view(1, 11), "attention_mask": torch.ones(1, 11, dtype=torch.long)}
    p = _pad_tokens(t, 8)
    assert p["input_ids"].shape == (1, 16) and p["attention_mask"].shape == (1, 16)
    assert torch.equal(p["input_ids"][:, :11], t["input_ids"]) and int(p["input_ids"][:, 11:].abs().sum()) == 0
    assert int(p["attention_mask"].sum()) == 11
    t16 = {"input_ids": torch.ones(2, 16, dtype=torch.long), "attention_mask": torch.ones(2, 16, dtype=torch.long)}
    assert _pad_tokens(t16, 8) is t16


def test_gradient_ready_order_covers_every_parameter_once():
    """FrozenInTime.gradient_ready_order() (the bucket order of the hook-free gradient exchange): a permutation of the trainable
    parameters -- the tower forward() builds last comes first (default: the text tower, then the video tower from its last block
    to its first; tests/test_host_dryrun_cpu.py checks the order against what autograd really does) -- and
    gradient_stream_of() names no stream for host-resident parameters."""
    from egovlp_amd.model.model import FrozenInTime
    m = FrozenInTime(video_params={**VIDEO_PARAMS, "num_frames": 4}, text_params=dict(TEXT_PARAMS), projection="minimal",
                     load_checkpoint="")
    order = m.gradient_ready_order()
    params = [p for p in m.parameters() if p.requires_grad]
    assert len(order) == len(params) and {id(p) for p in order} == {id(p) for p in params}
    names = {id(p): k for k, p in m.named_parameters()}
    seq = [names[id(p)] for p in order]
    last_text = max(i for i, k in enumerate(seq) if k.startswith("text_model.") or k.startswith("txt_proj."))
    first_video = min(i for i, k in enumerate(seq) if k.startswith("video_model.") or k.startswith("vid_proj."))
    assert FrozenInTime.TEXT_TOWER_LAST and last_text < first_video  # the whole text tower is final before the video tower
    assert seq[0].startswith("txt_proj.") and seq[first_video].startswith("vid_proj.")
    b11 = min(i for i, k in enumerate(seq) if k.startswith("video_model.blocks.11."))
    b0 = min(i for i, k in enumerate(seq) if k.startswith("video_model.blocks.0."))
    assert b11 < b0                                                  # last block first
    assert all(m.gradient_stream_of(p) is None for p in params[:5])  # CPU parameters: no stream
