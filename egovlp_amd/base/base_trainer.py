"""Epoch loop, monitoring and checkpoint format of the reference's `Multi_BaseTrainer_dist`
(base/base_trainer.py:239-480) -- the part of the boundary `run/train_egoclip.py:88-98` calls (`trainer.train()`).

Kept as in the reference: constructor signature, `config` duck type (`config['trainer'][...]`, `config['n_gpu']`,
`config.get_logger`, `config.save_dir`, `config.resume` -- the reference's `ConfigParser` works unchanged, and so does
`egovlp_amd.utils.config.DictConfig`), `train()` bookkeeping (log dict, `monitor` modes, save_period), and the checkpoint
file: `{'arch', 'epoch', 'state_dict', 'optimizer', 'monitor_best', 'config'}` as `checkpoint-epoch{N}.pth` /
`model_best.pth`, rank 0 only; `_resume_checkpoint` accepts state_dicts with or without the DDP `module.` prefix.

Different by design: the reference wraps the model in `DistributedDataParallel(find_unused_parameters=True)`
(:258, fp32 bucketed all-reduce + a graph walk per step although no parameter is ever unused).  Here the model is NOT
wrapped: at world size > 1 gradients are averaged by `egovlp_amd.dist.Bf16GradSync` (bf16 buckets, exchanged over RCCL while
backward is still running: launched hook-free from the block-boundary polls of the video tower's backward, on a private stream).  `self.model` therefore has the same attribute surface and
`state_dict` keys (no `module.` prefix) on every world size.
"""
from __future__ import annotations

from abc import abstractmethod

import os

import torch
import torch.distributed as dist
from numpy import inf

from ..utils.util import load_checkpoint_file


class Multi_BaseTrainer_dist:
    def __init__(self, args, model, loss, metrics, optimizer, config, writer=None, init_val=False):
        self.config = config
        self.logger = config.get_logger('trainer', config['trainer']['verbosity'])
        self.init_val = init_val
        self.args = args
        if not torch.cuda.is_available():
            raise RuntimeError("egovlp_amd trains on an MI355X: no HIP device is visible (there is no CPU path)")
        local_rank = getattr(args, 'local_rank', 0)
        torch.cuda.set_device(local_rank)            # every egovlp_amd op launches on the CURRENT device's current stream
        self.device = torch.device('cuda', local_rank)
        self.model = model.to(self.device)
        self.model.device = self.device
        self.world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # this trainer owns every gradient hook of the run (Bf16GradSync joins the streams): weight gradients may go to the
        # side stream (egovlp_amd.ops.side_stream); EGV_WGRAD_SIDE=0 keeps them on the main stream
        # (safe with one backward per step and zero_grad(set_to_none=True), which egoclip_step does; a wgrad whose parameter
        # already holds a gradient stays on the main stream by itself -- model/video_transformer.py::_lin_bwd).
        # Settings go to the MODEL's execution context: nothing here is process-wide.
        from .. import ops
        ec = getattr(self.model, "exec_ctx", None) or ops.DEFAULT
        ec.set(wgrad_side_stream=os.environ.get("EGV_WGRAD_SIDE", "1") == "1")
        # precision mode of the run: `args.precision` ("bf16x3" = the library's parity-grade default, "f16mix" = the benchmarked mode:
        # fp16-product forward under the measured per-block policy + fp16 backward under the device-side loss scale, "f16x2", "mixed"),
        # or EGOVLP_PRECISION; the reference has no such switch (fp32 everywhere)
        prec = getattr(args, "precision", None) or os.environ.get("EGOVLP_PRECISION")
        if prec:
            ec.set_precision("bf16x3", "bf16") if prec == "mixed" else ec.set_precision(*prec.split("/"))
        self.grad_sync = None
        if self.world_size > 1:
            from ..dist import Bf16GradSync
            text = getattr(self.model, "text_model", None)
            if text is not None and hasattr(text, "seed_rank"):
                text.seed_rank = dist.get_rank()           # ranks must not draw identical dropout masks
            if hasattr(self.model, "gradient_ready_order"):
                # hook-free: buckets are launched from the polls of the video tower's backward (autograd grad-ready hooks
                # cost 0.6 ms per step more on this model, profiles/r02_d_dp_overhead.txt)
                self.grad_sync = Bf16GradSync(self.model.parameters(), use_hooks=False,
                                              order_hint=self.model.gradient_ready_order(), exec_ctx=ec,
                                              exchange=os.environ.get("EGV_GRAD_EXCHANGE", "direct"))
                ec.set(backward_poll=self.grad_sync.poll)
            else:
                self.grad_sync = Bf16GradSync(self.model.parameters(), stream_of=getattr(self.model, "gradient_stream_of", None),
                                              exec_ctx=ec, exchange=os.environ.get("EGV_GRAD_EXCHANGE", "direct"))
            # the persistent GEMM owns every CU for the length of a launch: leave one CU per XCD to the RCCL kernels of the
            # overlapped gradient exchange (bench.py does the same; the wgrad split-K policy follows the cap)
            ec.set(gemm_grid=int(os.environ.get("EGV_GEMM_GRID", "248")))
        self.loss = loss.to(self.device) if hasattr(loss, 'to') else loss
        self.metrics = metrics
        self.optimizer = optimizer

        cfg_trainer = config['trainer']
        self.epochs = cfg_trainer['epochs']
        self.save_period = cfg_trainer['save_period']
        self.monitor = cfg_trainer.get('monitor', 'off')
        self.init_val = cfg_trainer.get('init_val', True)
        if self.monitor == 'off':
            self.mnt_mode = 'off'
            self.mnt_best = 0
        else:
            self.mnt_mode, self.mnt_metric = self.monitor.split()
            assert self.mnt_mode in ['min', 'max']
            self.mnt_best = inf if self.mnt_mode == 'min' else -inf
            self.early_stop = cfg_trainer.get('early_stop', inf)
        self.start_epoch = 1
        self.checkpoint_dir = config.save_dir
        self.writer = writer
        if getattr(config, 'resume', None) is not None:
            self._resume_checkpoint(config.resume)

    @abstractmethod
    def _train_epoch(self, epoch):
        raise NotImplementedError

    @abstractmethod
    def _valid_epoch(self, epoch):
        raise NotImplementedError

    # ---- epoch loop ------------------------------------------------------------------------------------------------------
    # Contract with the reference's callers (run/train_egoclip.py:98 `trainer.train()`; base/base_trainer.py:313-380 is what
    # they expect to have happened afterwards), kept and tested in tests/test_gpu_trainer.py:
    #   * optional validation pass before the first epoch (`init_val`), then epochs start_epoch .. epochs;
    #   * rank 0 logs one flat dict per epoch: 'epoch', every scalar `_train_epoch` returned, the metric lists under their
    #     function names ('metrics' -> name, 'val_metrics' -> 'val_' + name) and the nested validation metrics as
    #     'val_{loader}_{metric}_{entry}';
    #   * `monitor = "<min|max> <key>"` tracks the best value of a logged key (a missing key switches monitoring off with a
    #     warning), `save_period` and a new best both trigger a checkpoint, written by rank 0 only;
    #   * returns the number of consecutive epochs without improvement.
    def _flat_epoch_log(self, epoch, result):
        log = {'epoch': epoch}
        names = [m.__name__ for m in (self.metrics or [])]
        for key, value in result.items():
            if key == 'metrics':
                log.update(zip(names, value))
            elif key == 'val_metrics':
                log.update(('val_' + n, v) for n, v in zip(names, value))
            elif key == 'nested_val_metrics':
                for loader, per_metric in value.items():
                    for metric, entries in per_metric.items():
                        for entry, v in entries.items():
                            log[f"val_{loader}_{metric}_{entry}"] = v
            else:
                log[key] = value
        return log

    def _is_new_best(self, log):
        """-> True / False, or None when monitoring is (or has just been switched) off."""
        if self.mnt_mode == 'off':
            return None
        if self.mnt_metric not in log:
            self.logger.warning("Warning: Metric '{}' is not found. Model performance monitoring is disabled.".format(self.mnt_metric))
            self.mnt_mode = 'off'
            return None
        value = log[self.mnt_metric]
        better = value <= self.mnt_best if self.mnt_mode == 'min' else value >= self.mnt_best
        if better:
            self.mnt_best = value
        return better

    def train(self):
        is_writer = self.args.rank == 0
        stale_epochs = 0
        if self.init_val:
            self._valid_epoch(-1)
        for epoch in range(self.start_epoch, self.epochs + 1):
            result = self._train_epoch(epoch)
            best = False
            if is_writer:
                log = self._flat_epoch_log(epoch, result)
                for key, value in log.items():
                    self.logger.info('    {:15s}: {}'.format(str(key), value))
                verdict = self._is_new_best(log)
                if verdict is not None:
                    best = verdict
                    stale_epochs = 0 if verdict else stale_epochs + 1
                if best or epoch % self.save_period == 0:
                    self._save_checkpoint(epoch, save_best=best)
        return stale_epochs

    # ---- checkpoints: the reference's FILE FORMAT (base/base_trainer.py:399-480), so that its checkpoints resume here and
    # ours load there: one dict {'arch', 'epoch', 'state_dict', 'optimizer', 'monitor_best', 'config'} per file,
    # 'checkpoint-epoch{N}.pth' every save_period epochs and a second copy 'model_best.pth' for a new best.
    def _checkpoint_state(self, epoch):
        state = {'arch': type(self.model).__name__, 'epoch': epoch, 'state_dict': self.model.state_dict(),
                 'optimizer': self.optimizer.state_dict(), 'monitor_best': self.mnt_best, 'config': self.config}
        # the dynamic loss scale of an fp16 backward (egovlp_amd.optim.LossScaler; an extra key the reference's loader ignores)
        ec = getattr(getattr(self.model, 'module', self.model), 'exec_ctx', None)
        if ec is not None and getattr(ec, '_scaler', None) is not None:
            state['loss_scaler'] = ec._scaler.state_dict()
        return state

    def _save_checkpoint(self, epoch, save_best=False):
        state = self._checkpoint_state(epoch)
        targets = [('checkpoint-epoch{}.pth'.format(epoch), "Saving checkpoint: {} ...")]
        if save_best:
            targets.append(('model_best.pth', "Saving current best: {} ..."))
        for name, message in targets:
            path = str(self.checkpoint_dir / name)
            torch.save(state, path)
            self.logger.info(message.format(path if name != 'model_best.pth' else name))

    @staticmethod
    def _match_module_prefix(state_dict, want_prefix):
        """Reference checkpoints come from a DDP-wrapped model ('module.' on every key); this trainer never wraps.  Re-key
        the loaded state_dict to what the live model uses."""
        has = next(iter(state_dict)).startswith('module.')
        if has == want_prefix:
            return state_dict
        if has:
            return type(state_dict)((k[len('module.'):], v) for k, v in state_dict.items())
        return type(state_dict)(('module.' + k, v) for k, v in state_dict.items())

    def _resume_checkpoint(self, resume_path):
        resume_path = str(resume_path)
        self.logger.info("Loading checkpoint: {} ...".format(resume_path))
        ckpt = load_checkpoint_file(resume_path, map_location=self.device, trusted=True)
        self.start_epoch = ckpt['epoch'] + 1
        self.mnt_best = ckpt['monitor_best']

        def cfg_of(section):            # the pickled config is a ConfigParser, a DictConfig or a placeholder: all optional
            try:
                return ckpt['config'][section], self.config[section]
            except (KeyError, TypeError):
                return None, None
        theirs, ours = cfg_of('arch')
        if theirs is not None and theirs != ours:
            self.logger.warning("Warning: Architecture configuration given in config file is different from that of "
                                "checkpoint. This may yield an exception while state_dict is being loaded.")
        live_keys = self.model.state_dict().keys()
        self.model.load_state_dict(self._match_module_prefix(ckpt['state_dict'], next(iter(live_keys)).startswith('module.')))
        from .. import weights
        weights.bump_epoch()        # parameters changed under the cached bf16 operand planes
        theirs, ours = cfg_of('optimizer')
        if theirs is not None and theirs['type'] != ours['type']:
            self.logger.warning("Warning: Optimizer type given in config file is different from that of checkpoint. "
                                "Optimizer parameters not being resumed.")
        else:
            self.optimizer.load_state_dict(ckpt['optimizer'])
        ec = getattr(getattr(self.model, 'module', self.model), 'exec_ctx', None)
        if ec is not None and isinstance(ckpt.get('loss_scaler'), dict):
            ec.loss_scaler().load_state_dict(ckpt['loss_scaler'])
        self.logger.info("Checkpoint loaded. Resume training from epoch {}".format(self.start_epoch))
