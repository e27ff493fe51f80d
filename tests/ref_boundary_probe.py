"""Runs INSIDE a subprocess of tests/test_boundary_cpu.py, only where /root/reference exists (the build container): imports
the REAL reference (oracle/ref_import.py stub recipe) and
  1. builds the reference's own `parse_config.ConfigParser` on `configs/pt/egoclip.json` UNCHANGED, exactly as
     run/train_egoclip.py:142-165 does, and lets it instantiate the drop-in classes by reflection (:63,69,73);
  2. writes a checkpoint in the reference's format -- `module.`-prefixed state_dict + the live ConfigParser object
     (base/base_trainer.py:407-414) -- for the parent test to load WITHOUT a parse_config module;
  3. runs the reference's own `_inflate_positional_embeds` (model/model.py:145-187) on temporal embeddings of 16 -> 4 and
     4 -> 16 frames and saves the results.
Prints one JSON line.  usage: python tests/ref_boundary_probe.py <out_dir>"""
import argparse
import collections
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out_dir = sys.argv[1]

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import ref_import  # noqa: E402

mm, ml, te, mv = ref_import.load_reference()
from parse_config import ConfigParser  # noqa: E402  (the reference's)

import egovlp_amd.model.loss as module_loss  # noqa: E402
import egovlp_amd.model.model as module_arch  # noqa: E402
import egovlp_amd.optim as module_optim  # noqa: E402

os.chdir(out_dir)            # ConfigParser creates results/<name>/{models,log,tf}/<timestamp> under the cwd
sys.argv = ["train_egoclip.py", "-c", os.path.join(ref_import.REF, "configs/pt/egoclip.json")]
args = argparse.ArgumentParser(description='PyTorch Template')
args.add_argument('-c', '--config', default='configs/pt/egoclip.json', type=str)
args.add_argument('-r', '--resume', default=None, type=str)
args.add_argument('-d', '--device', default=None, type=str)
args.add_argument('-k', '--local_rank', type=int, default=0)
args.add_argument('-ws', '--world_size', type=int, default=1)
args.add_argument('-rk', '--rank', type=int, default=0)
args.add_argument('-lr1', '--learning_rate1', type=float, default=2e-4)
args.add_argument('-sc', '--schedule', default=[60, 80])
CustomArgs = collections.namedtuple('CustomArgs', 'flags type target')
options = [CustomArgs(['--lr', '--learning_rate'], type=float, target=('optimizer', 'args', 'lr')),
           CustomArgs(['--bs', '--batch_size'], type=int, target=('data_loader', 'args', 'batch_size'))]
config = ConfigParser(args, options)

model = config.initialize('arch', module_arch)                                    # run/train_egoclip.py:63
loss = config.initialize(name="loss", module=module_loss)                         # :69
trainable = list(filter(lambda p: p.requires_grad, model.parameters()))
optimizer = config.initialize('optimizer', module_optim, trainable)               # :73 (transformers.AdamW is gone in 5.x)
sd = model.state_dict()
res = {"model": type(model).__name__, "keys": len(sd), "params_M": sum(v.numel() for v in sd.values()) / 1e6,
       "loss": type(loss).__name__, "loss_temperature": loss.temperature, "optimizer": type(optimizer).__name__,
       "lr": optimizer.param_groups[0]["lr"], "eps": optimizer.param_groups[0]["eps"],
       "betas": list(optimizer.param_groups[0]["betas"]), "weight_decay": optimizer.param_groups[0]["weight_decay"],
       "n_trainable": len(trainable), "model_frames": model.video_params["num_frames"],
       "time_init_zero": bool((model.video_model.blocks[0].timeattn.qkv.weight == 0).all()
                              and (model.video_model.blocks[0].timeattn.proj.weight == 1).all())}

# 2. a reference-format checkpoint: DDP 'module.' prefix + the live ConfigParser (what _save_checkpoint pickles)
from egovlp_amd.synth import synth_state_dict  # noqa: E402
vals = synth_state_dict({k: v.shape for k, v in sd.items()}, seed=21)
state = {'arch': 'FrozenInTime', 'epoch': 3, 'state_dict': collections.OrderedDict(('module.' + k, v) for k, v in vals.items()),
         'optimizer': {'state': {}, 'param_groups': []}, 'monitor_best': 0.5, 'config': config}
torch.save(state, os.path.join(out_dir, "ref_format_checkpoint.pth"))

# 3. the reference's own temporal-embedding inflation
out = {}
for load_frames, curr_frames, fix in [(16, 4, 'zeros'), (4, 16, 'zeros'), (4, 16, 'bilinear')]:
    g = torch.Generator().manual_seed(load_frames * 100 + curr_frames)
    te_ = torch.randn(1, load_frames, 768, generator=g)
    fake = types.SimpleNamespace(
        state_dict=lambda cf=curr_frames: {'video_model.temporal_embed': torch.zeros(1, cf, 768),
                                           'video_model.pos_embed': torch.zeros(1, 197, 768)},
        video_params={'num_frames': curr_frames, 'model': 'SpaceTimeTransformer'}, load_temporal_fix=fix)
    new = mm.FrozenInTime._inflate_positional_embeds(fake, {'video_model.temporal_embed': te_.clone(),
                                                            'video_model.pos_embed': torch.zeros(1, 197, 768)})
    out[f"in_{load_frames}_{curr_frames}_{fix}"] = te_.numpy()
    out[f"out_{load_frames}_{curr_frames}_{fix}"] = new['video_model.temporal_embed'].numpy()
np.savez(os.path.join(out_dir, "ref_inflate.npz"), **out)
print("PROBE " + json.dumps(res))
