"""EgoClip pre-training step -- drop-in for the reference's trainer/trainer_egoclip.py.

`AllGather_multi` keeps the reference's autograd contract (trainer/trainer_egoclip.py:11-27): forward =
all-gather + rank-major concatenation, backward = the LOCAL rows of the incoming gradient, no reduction
(every rank computes the identical global loss; DDP's mean over ranks then yields (1/W) dL_global/dtheta,
SURVEY 3.2).  On MI355X the four per-step gathers of the reference (:126-129: video, text, noun, verb =
four latency-bound RCCL launches + 4W allocations + 4 cats) become ONE `all_gather_into_tensor` of a
packed [B, 256+256+582+118] fp32 row block (~152 KiB per rank at B=32) written straight into its final
place -- xGMI is point-to-point, so for a payload this small launch latency, not link bandwidth, is
what there is to save.  `backend='nccl'` on PyTorch-ROCm IS RCCL.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist

from ..base.base_trainer import Multi_BaseTrainer_dist
from ..model.model import sim_matrix


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _gather_rows(t: torch.Tensor, world: int) -> torch.Tensor:
    if world == 1 and os.environ.get("EGV_FORCE_GATHER") != "1":
        return t
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    from ..dist import timed
    timed("embedding_all_gather", lambda: dist.all_gather_into_tensor(out, t.contiguous()))
    return out


def _pad_tokens(text, multiple):
    """Right-pad input_ids (0 = [PAD]) and attention_mask (0 = masked) to a multiple of `multiple` tokens."""
    L = text['input_ids'].shape[1]
    Lp = (L + multiple - 1) // multiple * multiple
    if Lp == L:
        return text
    return {k: torch.nn.functional.pad(v, (0, Lp - L), value=0) for k, v in text.items()}


class AllGather_multi(torch.autograd.Function):
    """An autograd function that performs allgather on a tensor (reference signature kept:
    `AllGather_multi.apply(tensor, n_gpu, args)` with args.world_size / args.rank)."""

    @staticmethod
    def forward(ctx, tensor, n_gpu, args):
        ctx.rank = args.rank
        ctx.batch_size = tensor.shape[0]
        return _gather_rows(tensor, args.world_size)

    @staticmethod
    def backward(ctx, grad_output):
        return (grad_output[ctx.batch_size * ctx.rank: ctx.batch_size * (ctx.rank + 1)], None, None)


class AllGatherFused(torch.autograd.Function):
    """(video_embeds, text_embeds, noun_vec, verb_vec) -> their global-batch versions with ONE collective."""

    @staticmethod
    def forward(ctx, video, text, noun, verb, world_size, rank):
        ctx.rank, ctx.B = rank, video.shape[0]
        if world_size == 1 and os.environ.get("EGV_FORCE_GATHER") != "1":   # (forced: 1-GPU smoke test of the collective path)
            return video, text, noun, verb
        widths = [video.shape[1], text.shape[1], noun.shape[1], verb.shape[1]]
        packed = torch.cat([video, text, noun.to(video.dtype), verb.to(video.dtype)], dim=1)
        allp = _gather_rows(packed, world_size)
        v, t, n, b = torch.split(allp, widths, dim=1)
        return v.contiguous(), t.contiguous(), n.contiguous(), b.contiguous()

    @staticmethod
    def backward(ctx, gv, gt, gn, gb):
        lo, hi = ctx.B * ctx.rank, ctx.B * (ctx.rank + 1)
        return gv[lo:hi], gt[lo:hi], None, None, None, None


def egoclip_step(model, loss_fn, optimizer, data, world_size=1, rank=0, fused_head=True, grad_sync=None, scaler=None):
    """One optimisation step = trainer/trainer_egoclip.py:123-141 (zero_grad, forward, gathers,
    similarity + loss, backward, optimizer.step).  Returns the (device) loss tensor; no host sync.
    `grad_sync` (egovlp_amd.dist.Bf16GradSync, world size > 1) averages the gradients over the ranks -- its all-reduces
    are launched by grad-ready hooks during backward; `finish()` waits for them before the optimizer reads p.grad.
    `scaler` (egovlp_amd.optim.LossScaler; default when the model's backward precision is 'f16': exec_ctx.loss_scaler()): the loss is multiplied by the device-side
    loss scale before backward() and the optimizer un-scales, or skips the step after an overflow -- no host synchronisation."""
    ec0 = getattr(getattr(model, 'module', model), 'exec_ctx', None)
    if scaler is None and ec0 is not None and ec0.bwd_passes == 4:
        # fp16 gradient planes flush un-scaled gradients of 1e-6 to zero: the model's own scaler (on the device its parameters live on)
        scaler = ec0.loss_scaler(device=next(getattr(model, 'module', model).parameters()).device)
    optimizer.zero_grad(set_to_none=True)
    text_embeds, video_embeds = model(data)
    n_embeds, v_embeds = data['noun_vec'], data['verb_vec']
    video_embeds, text_embeds, n_embeds, v_embeds = AllGatherFused.apply(
        video_embeds, text_embeds, n_embeds, v_embeds, world_size, rank)
    is_ego = type(loss_fn).__name__ == 'EgoNCE'
    n, D = text_embeds.shape
    # the one-launch head covers global batches up to 1024 rows of <= 256 features (8 x 128 per GPU); beyond that the
    # API-compatible sim_matrix + loss.forward path takes over (n <= 4096)
    if fused_head and hasattr(loss_fn, 'fused') and n <= 1024 and D <= 256 and D % 4 == 0:
        loss = loss_fn.fused(text_embeds, video_embeds, n_embeds, v_embeds) if is_ego \
            else loss_fn.fused(text_embeds, video_embeds)
    else:
        output = sim_matrix(text_embeds, video_embeds)                      # :130
        if is_ego:
            sim_v = sim_matrix(v_embeds, v_embeds)                          # :133
            sim_n = sim_matrix(n_embeds, n_embeds)                          # :134
            loss = loss_fn(output, sim_v, sim_n)                            # :135
        else:
            loss = loss_fn(output)
    (loss if scaler is None else scaler.scale(loss)).backward()             # :139
    ec = getattr(getattr(model, 'module', model), 'exec_ctx', None)
    if ec is not None:
        ec.join_side_stream()       # idempotent; covers a backward whose end-of-pass callback did not run
    if grad_sync is not None:
        grad_sync.finish()
    if scaler is None:
        optimizer.step()                                                    # :141
    else:
        optimizer.step(scaler=scaler)
    return loss.detach()


def _to_device_async(data, device, stream):
    """Host batch -> device on `stream`: tensors go through pinned staging copies (a pageable source makes the copy synchronous);
    -> (device batch, event of the last copy).  Keys that are not tensors are passed through."""
    out = {}

    def put(t):
        if not torch.is_tensor(t) or t.device.type != 'cpu' or torch.device(device).type != 'cuda':
            return t.to(device) if torch.is_tensor(t) else t
        src = t if t.is_pinned() else t.contiguous().pin_memory()
        return src.to(device, non_blocking=True)
    if torch.device(device).type == 'cuda':
        with torch.cuda.stream(stream):
            for k, v in data.items():
                out[k] = {kk: put(vv) for kk, vv in v.items()} if isinstance(v, dict) or hasattr(v, 'items') else put(v)
            ev = torch.cuda.Event()
            ev.record(stream)
        return out, ev
    for k, v in data.items():
        out[k] = {kk: put(vv) for kk, vv in v.items()} if isinstance(v, dict) or hasattr(v, 'items') else put(v)
    return out, None


def _prefetched(host_iter, device):
    """Yield (batch_idx, dl_idx, device batch) with the copy of batch i + 1 in flight on a copy stream while batch i is consumed;
    ends with (None, None, None)."""
    cuda = torch.device(device).type == 'cuda'
    stream = torch.cuda.Stream() if cuda else None
    it = iter(host_iter)

    def start():
        try:
            bi, di, data = next(it)
        except StopIteration:
            return None
        dev, ev = _to_device_async(data, device, stream)
        return bi, di, dev, ev
    nxt = start()
    while nxt is not None:
        bi, di, dev, ev = nxt
        nxt = start()                    # the next batch's host work + copy start BEFORE this batch's step is enqueued
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            for v in dev.values():       # the tensors were allocated on the copy stream: tell the allocator who uses them
                for t in (v.values() if isinstance(v, dict) else [v]):
                    if torch.is_tensor(t) and t.is_cuda:
                        t.record_stream(torch.cuda.current_stream())
        yield bi, di, dev
    yield None, None, None


class Multi_Trainer_dist(Multi_BaseTrainer_dist):
    """Drop-in for the reference's trainer class (trainer/trainer_egoclip.py:29-275): same constructor, `train()` /
    checkpointing from the base class (egovlp_amd.base.Multi_BaseTrainer_dist == base/base_trainer.py:239-480), the
    training hot loop `_train_epoch` (:82-180) and the EgoMCQ validation `_valid_epoch` (:182-275)."""

    def __init__(self, args, model, loss, metrics, optimizer, config, data_loader, valid_data_loader=None,
                 lr_scheduler=None, len_epoch=None, writer=None, visualizer=None, tokenizer=None,
                 max_samples_per_epoch=50000):
        super().__init__(args, model, loss, metrics, optimizer, config, writer)
        self.config = config
        self.args = args
        self.data_loader = data_loader
        if len_epoch is None:
            self.len_epoch = min(len(x) for x in data_loader)               # epoch-based training (:44-47)
        else:
            self.len_epoch = len_epoch
        self.valid_data_loader = valid_data_loader
        self.do_validation = self.valid_data_loader is not None
        self.lr_scheduler = lr_scheduler
        self.visualizer = visualizer
        self.val_chunking = True
        self.metrics = metrics if metrics is not None else []
        self.batch_size = self.data_loader[0].batch_size
        self.log_step = int(np.sqrt(self.batch_size))
        self.total_batch_sum = sum(x.batch_size for x in self.data_loader)
        self.tokenizer = tokenizer
        self.max_samples_per_epoch = max_samples_per_epoch
        self.n_gpu = self.args.world_size
        self.allgather = AllGather_multi.apply

    def _host_batches(self):
        """(batch_idx, dl_idx, data on the HOST) in the reference's order and with its stopping rules (:104-108,158-159)."""
        for batch_idx, data_li in enumerate(zip(*self.data_loader)):
            if (batch_idx + 1) * self.total_batch_sum > self.max_samples_per_epoch:
                break
            for dl_idx, data in enumerate(data_li):
                if 'video_neg' in data.keys():                              # :109-113, scene-aware negatives: B -> 2B
                    data['text'] = data['text'] + data['text_neg']
                    data['video'] = torch.cat((data['video'], data['video_neg']), axis=0)
                    data['noun_vec'] = torch.cat((data['noun_vec'], data['noun_vec_neg']), axis=0)
                    data['verb_vec'] = torch.cat((data['verb_vec'], data['verb_vec_neg']), axis=0)
                    for k in ('text_neg', 'video_neg', 'noun_vec_neg', 'verb_vec_neg'):
                        data.pop(k, None)        # concatenated above: not staged / copied to the device a second time
                if self.tokenizer is not None:
                    data['text'] = self.tokenizer(data['text'], return_tensors='pt', padding=True, truncation=True)
                yield batch_idx, dl_idx, data
            if batch_idx == self.len_epoch:
                break

    def _adjust_learning_rate(self, optimizer, epoch, args):
        lr = args.learning_rate1                                            # :75-80
        for milestone in args.schedule:
            lr *= 0.1 if epoch >= milestone else 1.
        for param_group in optimizer.param_groups:
            param_group['lr'] = lr

    _guard = None

    def _train_epoch(self, epoch):
        self.model.train()
        total_loss = [torch.zeros((), device=self.device) for _ in self.data_loader]
        for loader in self.data_loader:
            if hasattr(loader, 'train_sampler'):
                loader.train_sampler.set_epoch(epoch)                       # :101-102
        # The reference moves every batch to the device with blocking `.to(device)` calls on the compute stream right before the
        # step (:115-121).  Here the NEXT batch is prepared (negatives concatenated, captions tokenised), staged in pinned host
        # memory and copied on a private copy stream while the current step runs; the step only waits for the copy's event.
        feed = _prefetched(self._host_batches(), self.device)
        for batch_idx, dl_idx, data in feed:
            if batch_idx is None:
                break
            # the per-block precision policy is measured on the weights at hand: first batch, then every `precision_guard_interval`
            # steps (egovlp_amd.guard.PrecisionGuard; a no-op unless the forward runs fp16 products)
            if self._guard is None:
                from ..guard import PrecisionGuard
                self._guard = PrecisionGuard(self.model, interval=int(getattr(self.args, 'precision_guard_interval', 1000)))
            self._guard.maybe_check(data)
            loss = egoclip_step(self.model, self.loss, self.optimizer, data, self.n_gpu, self.args.rank,
                                grad_sync=self.grad_sync)
            total_loss[dl_idx] += loss      # stays on the device: no per-step .item() sync (reference :148,150)
            if self.writer is not None and self.args.rank == 0 and batch_idx % self.log_step == 0:
                total = int(self.data_loader[dl_idx].n_samples / self.n_gpu) if hasattr(self.data_loader[dl_idx], 'n_samples') else 0
                current = batch_idx * self.data_loader[dl_idx].batch_size
                final_total = (epoch - 1) * total + current
                self.writer.add_scalar(f'Loss_training/loss_{dl_idx}', float(loss), final_total)   # :143-148
        log = {f'loss_{dl_idx}': float(total_loss[dl_idx]) / self.len_epoch for dl_idx in range(len(self.data_loader))}   # :162-164
        if self.writer is not None and self.args.rank == 0:
            for dl_idx in range(len(self.data_loader)):
                self.writer.add_scalar(f'Loss_training/loss_total_{dl_idx}', log[f'loss_{dl_idx}'], epoch - 1)
        if self.do_validation:                                              # :172-175
            val_log = self._valid_epoch(epoch)
            if self.args.rank == 0:
                log.update(val_log)
        self._adjust_learning_rate(self.optimizer, epoch, self.args)        # :178
        return log

    def _valid_epoch(self, epoch):
        """EgoMCQ validation = reference trainer/trainer_egoclip.py:182-275: for every question the text query and its five
        candidate clips go through the same encoders in eval mode (`model(data, return_embeds=True)`, :211), the prediction
        is `sim_matrix(text, video)` [1, 5] (:214), predictions / answers / types are all-gathered over the ranks (:225-235)
        and scored by the configured metrics (model/metric.py:218-234).  Differences: the gathers are one collective each
        (`all_gather_into_tensor`) and are skipped without a process group; results stay on the device until the end."""
        self.model.eval()
        # one query + five clips per question is ~330 launches for ~3 ms of GPU work: with `args.graph_eval` the forward is
        # captured once per input shape into a HIP graph and replayed (egovlp_amd/graph.py); queries are padded to a multiple
        # of 8 tokens (masked keys contribute exact zeros) so that a handful of graphs covers every caption length
        fwd = None
        if getattr(self.args, 'graph_eval', False):
            from ..graph import GraphedForward
            fwd = GraphedForward(self.model)
            self.last_graphed_forward = fwd
        n_loaders = len(self.valid_data_loader)
        gt_arr = {x: [] for x in range(n_loaders)}
        pred_arr = {x: [] for x in range(n_loaders)}
        type_arr = {x: [] for x in range(n_loaders)}
        world = _world()
        with torch.no_grad():
            for dl_idx, dl in enumerate(self.valid_data_loader):
                for data in dl:
                    data['video'] = data['video'][0]                                    # remove batch (:205)
                    if self.tokenizer is not None:
                        data['text'] = self.tokenizer(data['text'], return_tensors='pt', padding=True, truncation=True)
                    data['text'] = {key: val.to(self.device) for key, val in data['text'].items()}
                    data['video'] = data['video'].to(self.device)
                    if fwd is not None:
                        data['text'] = _pad_tokens(data['text'], 8)
                        text_embed, vid_embed = fwd(data)
                        text_embed, vid_embed = text_embed.clone(), vid_embed.clone()  # the graph's static outputs
                    else:
                        text_embed, vid_embed = self.model(data, return_embeds=True)   # :211
                    data_gt = data['correct'][0].to(self.device).unsqueeze(0)
                    data_pred = sim_matrix(text_embed, vid_embed)                        # :214
                    data_type = data['type'][0].to(self.device).unsqueeze(0)
                    gt_arr[dl_idx].append(_gather_rows(data_gt, world))
                    pred_arr[dl_idx].append(_gather_rows(data_pred, world))
                    type_arr[dl_idx].append(_gather_rows(data_type, world))
        nested_metrics = {x: {} for x in range(n_loaders)}
        for dl_idx in range(n_loaders):
            gt_cat = torch.cat(gt_arr[dl_idx]).cpu()
            pred_cat = torch.cat(pred_arr[dl_idx]).cpu()
            type_cat = torch.cat(type_arr[dl_idx]).cpu()
            for metric in self.metrics:
                nested_metrics[dl_idx][metric.__name__] = metric(pred_cat, gt_cat, type_cat)
        res_dict = {}
        if self.args.rank == 0:
            res_dict = {f'val_loss_{dl_idx}': 0.0 for dl_idx in range(n_loaders)}       # the reference never accumulates it (:192)
            res_dict['nested_val_metrics'] = nested_metrics
        self.last_val_predictions = {x: torch.cat(pred_arr[x]).cpu() for x in range(n_loaders)}
        return res_dict
