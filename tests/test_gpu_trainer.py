"""`Multi_Trainer_dist.train()` on a real MI355X: the boundary run/train_egoclip.py:88-98 calls (reference
base/base_trainer.py:313-480, trainer/trainer_egoclip.py:82-180).  Synthetic loaders stand in for the Ego4D data loader;
everything behind them -- epoch loop, tokenizer hand-off, the `neg_param` batch doubling (:109-113), egoclip_step, the LR rule
(:75-80,178), checkpoint files in the reference's format and resume -- is the product code."""
import os
import types

import pytest
import torch

pytestmark = pytest.mark.gpu

from egovlp_amd.synth import synth_batch, synth_state_dict  # noqa: E402

CFG = {"name": "EgoClip_4f", "n_gpu": 1,
       "arch": {"type": "FrozenInTime", "args": {
           "video_params": {"model": "SpaceTimeTransformer", "arch_config": "base_patch16_224", "num_frames": 4,
                            "pretrained": True, "time_init": "rand"},
           "text_params": {"model": "distilbert-base-uncased", "pretrained": True, "input": "text"},
           "projection": "minimal", "load_checkpoint": ""}},
       "optimizer": {"type": "AdamW", "args": {"lr": 3e-5}},
       "loss": {"type": "EgoNCE", "args": {}},
       "metrics": ["egomcq_accuracy_metrics"],
       "trainer": {"epochs": 2, "max_samples_per_epoch": 500000, "save_dir": "unused", "save_period": 1, "verbosity": 2,
                   "monitor": "off", "init_val": False, "neptune": False}}


class FakeTokenizer:
    """Deterministic stand-in for the HF tokenizer (run/train_egoclip.py:53): words -> ids, padded to the longest caption."""

    def __call__(self, texts, return_tensors='pt', padding=True, truncation=True):
        rows = [[101] + [1000 + (sum(map(ord, w)) * 7919) % 28000 for w in t.split()][:30] + [102] for t in texts]
        L = max(len(r) for r in rows)
        ids = torch.zeros(len(rows), L, dtype=torch.long)
        mask = torch.zeros(len(rows), L, dtype=torch.long)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = torch.tensor(r)
            mask[i, :len(r)] = 1
        return {"input_ids": ids, "attention_mask": mask}


WORDS = "c opens the drawer and picks a knife from it then cuts an onion on the board while the man looks".split()


class Loader:
    """Two batches per epoch of B clips (+ B scene-aware negatives when neg_param), captions as STRINGS."""
    dataset_name = "EgoClip-synthetic"

    def __init__(self, B, neg_param, n_batches=2):
        self.batch_size, self.neg, self.n_batches = B, neg_param, n_batches
        self.n_samples = B * n_batches

    def __len__(self):
        return self.n_batches

    def batch(self, i):
        b = synth_batch(self.batch_size, T=2, L=8, seed=50 + i)
        g = torch.Generator().manual_seed(900 + i)
        caps = [" ".join(WORDS[int(j)] for j in torch.randint(0, len(WORDS), (5 + k % 4,), generator=g)) for k in range(self.batch_size)]
        d = {"video": b["video"], "text": caps, "noun_vec": b["noun_vec"], "verb_vec": b["verb_vec"]}
        if self.neg:
            n = synth_batch(self.batch_size, T=2, L=8, seed=70 + i)
            d.update({"video_neg": n["video"], "text_neg": [c + " again" for c in caps], "noun_vec_neg": n["noun_vec"],
                      "verb_vec_neg": n["verb_vec"]})
        return d

    def __iter__(self):
        return (self.batch(i) for i in range(self.n_batches))


def build(tmp, resume=None):
    import egovlp_amd.model.loss as module_loss
    import egovlp_amd.model.model as module_arch
    import egovlp_amd.optim as module_optim
    from egovlp_amd.ops import Precision
    from egovlp_amd.utils.config import DictConfig
    Precision.set("bf16x3")
    config = DictConfig(CFG, save_dir=tmp, resume=resume)
    model = config.initialize('arch', module_arch)                          # run/train_egoclip.py:63
    model.load_state_dict(synth_state_dict({k: v.shape for k, v in model.state_dict().items()}, seed=2))
    model.text_model.set_dropout(0.0, 0.0)      # resumed runs restart the mask counter: keep the comparison deterministic
    loss = config.initialize(name="loss", module=module_loss)               # :69
    optimizer = config.initialize('optimizer', module_optim, filter(lambda p: p.requires_grad, model.parameters()))   # :73
    return config, model, loss, optimizer


def make_trainer(tmp, neg_param, resume=None):
    from egovlp_amd.model.metric import egomcq_accuracy_metrics
    from egovlp_amd.trainer.trainer_egoclip import Multi_Trainer_dist
    config, model, loss, optimizer = build(tmp, resume)
    args = types.SimpleNamespace(world_size=1, rank=0, local_rank=0, learning_rate1=2e-4, schedule=[60, 80])
    tr = Multi_Trainer_dist(args, model, loss, [egomcq_accuracy_metrics], optimizer, config=config,
                            data_loader=[Loader(2, neg_param)], valid_data_loader=None, tokenizer=FakeTokenizer(),
                            max_samples_per_epoch=CFG['trainer']['max_samples_per_epoch'])
    return tr


def test_train_checkpoint_resume(tmp_path):
    from egovlp_amd.utils.util import load_checkpoint_file
    tr = make_trainer(tmp_path / "run1", neg_param=False)
    tr.train()                                                              # run/train_egoclip.py:98
    files = sorted(os.listdir(tmp_path / "run1"))
    assert "checkpoint-epoch1.pth" in files and "checkpoint-epoch2.pth" in files
    ck1 = load_checkpoint_file(str(tmp_path / "run1" / "checkpoint-epoch1.pth"), map_location="cpu", trusted=True)
    assert set(ck1) == {"arch", "epoch", "state_dict", "optimizer", "monitor_best", "config"}      # base/base_trainer.py:407-414
    assert ck1["arch"] == "FrozenInTime" and ck1["epoch"] == 1 and len(ck1["state_dict"]) == 327
    assert ck1["optimizer"]["param_groups"][0]["lr"] == 2e-4               # the LR rule after epoch 1 (:75-80,178)
    final1 = {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}
    # ---- resume from the epoch-1 file: epoch 2 runs again from the saved weights + optimizer state
    tr2 = make_trainer(tmp_path / "run2", neg_param=False, resume=str(tmp_path / "run1" / "checkpoint-epoch1.pth"))
    assert tr2.start_epoch == 2
    for k, v in tr2.model.state_dict().items():
        assert torch.equal(v.cpu(), ck1["state_dict"][k]), k
    st = tr2.optimizer.state_dict()["state"]
    assert len(st) == 327 and all(s["step"] == 2 for s in st.values())     # two steps were taken in epoch 1
    assert tr2.optimizer.param_groups[0]["lr"] == 2e-4
    tr2.train()
    assert sorted(os.listdir(tmp_path / "run2")) == ["checkpoint-epoch2.pth"]
    # same data, same state -> the same epoch 2 (the kernels with fp32 atomics reorder sums: compare the UPDATE, loosely)
    num = den = 0.0
    for k, v in tr2.model.state_dict().items():
        d1 = final1[k] - ck1["state_dict"][k]
        d2 = v.cpu() - ck1["state_dict"][k]
        num += float((d1 - d2).double().pow(2).sum())
        den += float(d1.double().pow(2).sum())
    assert (num / den) ** 0.5 < 2e-2, (num / den) ** 0.5


def test_neg_param_doubles_the_batch(tmp_path):
    """trainer/trainer_egoclip.py:109-113: with scene-aware negatives the step runs on 2B clips / captions / noun-verb rows."""
    from egovlp_amd.trainer.trainer_egoclip import egoclip_step
    tr = make_trainer(tmp_path / "neg", neg_param=True)
    tr.epochs = 1
    seen = []
    orig = tr.model.forward

    def spy(data, *a, **k):
        seen.append((data["video"].shape[0], data["text"]["input_ids"].shape[0]))
        return orig(data, *a, **k)
    tr.model.forward = spy
    log = tr._train_epoch(1)
    assert seen == [(4, 4), (4, 4)]
    # the same two steps by hand on pre-concatenated batches give the same mean loss
    config, model, loss, optimizer = build(tmp_path / "neg2")
    model = model.cuda().train()
    tok, ld, tot = FakeTokenizer(), Loader(2, True), 0.0
    for i in range(2):
        d = ld.batch(i)
        data = {"video": torch.cat((d["video"], d["video_neg"])).cuda(),
                "text": {k: v.cuda() for k, v in tok(d["text"] + d["text_neg"]).items()},
                "noun_vec": torch.cat((d["noun_vec"], d["noun_vec_neg"])).cuda(),
                "verb_vec": torch.cat((d["verb_vec"], d["verb_vec_neg"])).cuda()}
        tot += float(egoclip_step(model, loss, optimizer, data))
    assert abs(log["loss_0"] - tot / 2) < 1e-4 * abs(tot / 2)
