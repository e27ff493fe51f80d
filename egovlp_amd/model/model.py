"""FrozenInTime dual encoder + sim_matrix -- drop-in for the reference's model/model.py.

Same constructor / forward / attribute surface (model/model.py:14-143), so
`config.initialize('arch', egovlp_amd.model.model)` builds it from configs/pt/egoclip.json unchanged.
Differences from the reference are confined to what cannot exist in an offline MI355X container:
  * `AutoModel.from_pretrained('distilbert-base-uncased')` (:32) needs the HF hub -> the text encoder is
    our DistilBertModel with HF's init; real weights arrive through `load_checkpoint` / load_state_dict;
  * the timm ViT-B/16 checkpoint `pretrained/jx_vit_base_p16_224-80ecf9dd.pth` (:48) is loaded when the
    file exists and skipped (random init) otherwise;
  * `arch_config='large_patch14_224'` is an extension (BASELINE config 5); the reference only accepts
    'base_patch16_224' (:45-53).
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..ops import ACT_RELU_BWD, ExecContext
from ..utils.util import load_checkpoint_file, state_dict_data_parallel_fix
from .text_transformer import DistilBertModel
from .video_transformer import SpaceTimeTransformer, _lin_bwd


class _ProjFn(torch.autograd.Function):
    """y = [relu](x) W^T + b on rows of x (possibly strided: the CLS row of every caption).
    Narrow heads (projection_dim = 2 of the OSCC fine-tune, 17 of PNR: configs/ft/oscc.json, pnr.json) run with the output
    width zero-padded to a multiple of 32 -- the GEMM writes 4-column pieces and the dgrad contracts over the output width in
    32-deep MFMA steps; the pad rows of W are zeros, the pad columns of y / dy are dropped / zero."""

    @staticmethod
    def forward(ctx, x2d, w, b, relu, ec: ExecContext):
        P = ec.fwd_passes_split
        M, K = x2d.shape
        N = w.shape[0]
        Np = N if N % 32 == 0 else (N + 31) // 32 * 32
        if relu:
            a = ops.relu_split(x2d, P)
        else:
            a, _, _ = ops.split_f32(x2d, P)
        if Np == N:
            w_pl, bias = ec.wc.get(w, need_t=False)[0], b
        else:       # tiny (Np x K): padded planes are rebuilt per call instead of living in the weight cache
            w_pl = ops.split_f32(F.pad(w.detach(), (0, 0, 0, Np - N)), 3)[0]
            bias = None if b is None else F.pad(b.detach(), (0, Np - N))
        y = torch.empty((M, Np), dtype=torch.float32, device=x2d.device)
        ops.gemm_nt(a, w_pl, passes=P, bias=bias, out_f32=y, ec=ec)
        ctx.a, ctx.relu, ctx.ec, ctx.P, ctx.Np = a, relu, ec, P, Np
        ctx.save_for_backward(x2d, w)
        return y if Np == N else y[:, :N]

    @staticmethod
    def backward(ctx, dy):
        x2d, w = ctx.saved_tensors
        ec = ctx.ec
        Pb = ec.bwd_passes_split
        N, Np = w.shape[0], ctx.Np
        if not ec.on_text_stream():
            ec.poll_backward()          # vid_proj: the first node of the video tower's backward on the main stream
        dy = dy.contiguous() if Np == N else F.pad(dy, (0, Np - N))
        dy_pl = ops.split_f32(dy, Pb)[0]
        # a padded (narrow) head slices dW / db below, on this stream: its wgrad must not run on the side stream (the slice copy would
        # read dW before the side stream's GEMM has written it)
        _, dW, db = _lin_bwd(dy_pl, ctx.a, None, Pb, need_dx=False, params=(w,), ec=ec, allow_side=(Np == N))
        dx = torch.empty((x2d.shape[0], x2d.shape[1]), dtype=torch.float32, device=dy.device)
        if Np == N:
            wt = ec.wc.get(w, need_t=True)[1]
        else:
            wt = ops.split_f32(F.pad(w.detach(), (0, 0, 0, Np - N)), 3, want_rowmajor=False, want_transposed=True)[1]
            dW, db = dW[:N].contiguous(), db[:N].contiguous()
        if ctx.relu:
            ops.gemm_nt(dy_pl, wt, passes=Pb, act=ACT_RELU_BWD, aux_in=x2d, out_f32=dx, K=Np, ec=ec)
        else:
            ops.gemm_nt(dy_pl, wt, passes=Pb, out_f32=dx, K=Np, ec=ec)
        return dx, dW, db, None, None


class _ReluLinear(nn.Sequential):
    """nn.Sequential(nn.ReLU(), nn.Linear) container (keys `1.weight`, `1.bias`), model/model.py:73-75."""


class BaseModel(nn.Module):
    """base/base_model.py:7-25."""

    def __str__(self):
        n = sum(p.numel() for p in self.parameters() if p.requires_grad)
        return super().__str__() + '\nTrainable parameters: {}'.format(n)


class FrozenInTime(BaseModel):
    def __init__(self, video_params, text_params, projection_dim=256, load_checkpoint=None,
                 projection='minimal', load_temporal_fix='zeros'):
        super().__init__()
        self.video_params = video_params
        self.text_params = text_params
        self.load_temporal_fix = load_temporal_fix
        if not text_params['pretrained']:
            raise NotImplementedError("Huggingface text models require pretrained init.")       # :27-28
        if self.text_params['model'].startswith('distilbert'):
            # `text_params['config']` (extension, tests): DistilBertConfig overrides for toy-sized towers
            from .text_transformer import DistilBertConfig
            self.text_model = DistilBertModel(DistilBertConfig(**text_params['config']) if text_params.get('config') else None)
        else:
            raise NotImplementedError(f"{text_params['model']}: only distilbert is on the EgoClip hot path")
        self.text_model.train()

        if video_params['model'] == "SpaceTimeTransformer":
            num_frames = video_params.get('num_frames', 4)
            time_init = video_params.get('time_init', 'zeros')
            attention_style = video_params.get('attention_style', 'frozen-in-time')
            arch_config = video_params.get('arch_config', 'base_patch16_224')
            if arch_config == 'base_patch16_224':
                model = SpaceTimeTransformer(num_frames=num_frames, time_init=time_init,
                                             attention_style=attention_style)
                vit_path = "pretrained/jx_vit_base_p16_224-80ecf9dd.pth"
            elif arch_config == 'large_patch14_224':          # extension: BASELINE config 5
                model = SpaceTimeTransformer(patch_size=14, embed_dim=1024, depth=24, num_heads=16,
                                             num_frames=num_frames, time_init=time_init,
                                             attention_style=attention_style)
                vit_path = None
            elif arch_config == 'custom':                     # extension (tests): SpaceTimeTransformer(**video_params['arch_kwargs'])
                model = SpaceTimeTransformer(num_frames=num_frames, time_init=time_init, attention_style=attention_style,
                                             **video_params['arch_kwargs'])
                vit_path = None
            else:
                raise NotImplementedError                                                        # :53
            model.head = nn.Identity()
            model.pre_logits = nn.Identity()
            ftr_dim = model.embed_dim
            if load_checkpoint in ["", None] and vit_path and os.path.exists(vit_path):
                vit_checkpoint = load_checkpoint_file(vit_path, map_location="cpu")      # a plain state_dict: safe path only
                new_vit_dict = state_dict_data_parallel_fix(vit_checkpoint, model.state_dict())
                model.load_state_dict(new_vit_dict, strict=False)                                # :58-63
            self.video_model = model
        else:
            raise NotImplementedError(f"{video_params['model']} not implemented")               # :66
        self.video_model.fc = nn.Identity()

        if projection == 'minimal':
            txt_proj = _ReluLinear(nn.ReLU(), nn.Linear(self.text_model.config.hidden_size, projection_dim))
            vid_proj = nn.Sequential(nn.Linear(ftr_dim, projection_dim))
        elif projection == '':
            txt_proj = nn.Identity()
            vid_proj = nn.Identity()
        else:
            raise NotImplementedError                                                            # :84
        self.txt_proj = txt_proj
        self.vid_proj = vid_proj
        # ONE execution context for the dual encoder and its two towers: precision policy, side streams, grid cap, weight-plane
        # cache (egovlp_amd.ops.ExecContext).  Private to this model; unset settings follow ops.DEFAULT.
        self.exec_ctx = ops.new_context()
        self.video_model.exec_ctx = self.text_model.exec_ctx = self.exec_ctx

        if load_checkpoint not in ["", None]:
            local_rank = int(os.environ.get('LOCAL_RANK', 0))
            dev = 'cuda:{}'.format(local_rank) if torch.cuda.is_available() else 'cpu'
            # reference checkpoints pickle their ConfigParser next to the weights (base/base_trainer.py:407-414): read them
            # with the lenient unpickler of utils/util.py (torch >= 2.6 refuses the global under weights_only=True)
            checkpoint = load_checkpoint_file(load_checkpoint, map_location=dev, trusted=True)   # the file the config names
            state_dict = checkpoint['state_dict']
            new_state_dict = state_dict_data_parallel_fix(state_dict, self.state_dict())
            new_state_dict = self._inflate_positional_embeds(new_state_dict)
            self.load_state_dict(new_state_dict, strict=True)                                    # :88-95

    def gradient_stream_of(self, param):
        """The HIP stream `param`'s gradient is produced on: the text tower's stream for DistilBERT and txt_proj when the towers
        run on two streams (ops.TEXT_SIDE_STREAM), else None (= the stream backward() is called on).  For code that
        registers gradient hooks (egovlp_amd.dist.Bf16GradSync in hook mode)."""
        if not self.exec_ctx.text_side_stream or not param.is_cuda:
            return None
        ids = getattr(self, "_text_param_ids", None)
        if ids is None:
            ids = {id(q) for q in self.text_model.parameters()} | {id(q) for q in self.txt_proj.parameters()}
            self._text_param_ids = ids
        return self.exec_ctx.text_stream() if id(param) in ids else None

    # Which tower's autograd nodes are created LAST in forward() -- and therefore run FIRST in backward (the engine runs the
    # ready node with the highest sequence number).  True: video first, then text (on its own stream, forked from an event
    # recorded before the video tower), so that backward enqueues the small text tower first and its 66 M parameters (a third of
    # the gradient exchange) can leave at the first poll of the video tower's backward instead of from finish().
    TEXT_TOWER_LAST = os.environ.get("EGV_TEXT_LAST", "1") == "1"

    def gradient_ready_order(self):
        """Trainable parameters in the order their gradients become final in backward(): autograd runs the later-created nodes
        first, so the tower forward() builds LAST comes first (its projection, then its parameters from the last layer to the
        first), then the other tower the same way.  Bf16GradSync(use_hooks=False) cuts its buckets along this order and launches
        them strictly in order from the backward polls -- a wrong order is not an error, it only delays every bucket to finish()
        (tests/test_gpu_dist.py asserts buckets do leave during backward on the real model)."""
        video = [p for p in self.vid_proj.parameters()] + [p for p in self.video_model.parameters()][::-1]
        text = [p for p in self.txt_proj.parameters()] + [p for p in self.text_model.parameters()][::-1]
        out, seen = [], set()
        for p in (text + video if self.TEXT_TOWER_LAST else video + text):
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                out.append(p)
        for p in self.parameters():                 # anything not covered above (none today)
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                out.append(p)
        return out

    def set_device(self, device):
        self.device = device

    def forward(self, data, video_only=False, return_embeds=True):
        ec = self.exec_ctx
        ec.begin_step()
        if video_only:
            return self.compute_video(data['video'])
        if ec.text_side_stream and data['video'].is_cuda:
            # the two towers are independent until the loss: DistilBERT (M = B*L = 1024 token rows, latency-bound launches
            # that fill a fraction of the chip) runs on a second HIP stream under the video tower.  autograd replays each
            # tower's backward on the stream its forward ran on and orders them against the loss by itself.
            main, side = torch.cuda.current_stream(), ec.text_stream()      # (begin_step refreshed the weight planes on `main`)
            ec._text["main"] = main
            for t in data['text'].values():
                t.record_stream(side)
            if self.TEXT_TOWER_LAST:
                fork = torch.cuda.Event()
                fork.record(main)                   # the text stream starts from HERE, not from behind the video tower
                video_embeddings = self.compute_video(data['video'])
                side.wait_event(fork)
                with torch.cuda.stream(side):
                    text_embeddings = self.compute_text(data['text'])
            else:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    text_embeddings = self.compute_text(data['text'])
                video_embeddings = self.compute_video(data['video'])
            main.wait_stream(side)
            text_embeddings.record_stream(main)
        elif self.TEXT_TOWER_LAST:
            video_embeddings = self.compute_video(data['video'])
            text_embeddings = self.compute_text(data['text'])
        else:
            text_embeddings = self.compute_text(data['text'])
            video_embeddings = self.compute_video(data['video'])
        if return_embeds:
            return text_embeddings, video_embeddings
        return sim_matrix(text_embeddings, video_embeddings)

    def _txt(self, x2d):
        if isinstance(self.txt_proj, nn.Identity):
            return x2d
        lin = self.txt_proj[1]
        return _ProjFn.apply(x2d, lin.weight, lin.bias, True, self.exec_ctx)

    def compute_text(self, text_data):
        if not self.text_params['model'].startswith('distilbert'):
            raise NotImplementedError
        hidden = self.text_model(**text_data).last_hidden_state          # :122
        return self._txt(hidden[:, 0, :])

    def compute_text_tokens(self, text_data):
        hidden = self.text_model(**text_data).last_hidden_state          # :133
        B, L, D = hidden.shape
        y = self._txt(hidden.reshape(B * L, D))
        return y.view(B, L, -1)

    def compute_video(self, video_data):
        v = self.video_model(video_data)
        if isinstance(self.vid_proj, nn.Identity):
            return v
        lin = self.vid_proj[0]
        return _ProjFn.apply(v, lin.weight, lin.bias, False, self.exec_ctx)

    def _inflate_positional_embeds(self, new_state_dict):
        """model/model.py:145-187: adapt temporal_embed when the checkpoint has a different num_frames."""
        curr_keys = list(self.state_dict().keys())
        if 'video_model.temporal_embed' in new_state_dict and 'video_model.temporal_embed' in curr_keys:
            load_temporal_embed = new_state_dict['video_model.temporal_embed']
            load_num_frames = load_temporal_embed.shape[1]
            curr_num_frames = self.video_params['num_frames']
            embed_dim = load_temporal_embed.shape[2]
            if load_num_frames != curr_num_frames:
                if load_num_frames > curr_num_frames:
                    new_temporal_embed = load_temporal_embed[:, :curr_num_frames, :]
                else:
                    if self.load_temporal_fix == 'zeros':
                        new_temporal_embed = torch.zeros([load_temporal_embed.shape[0], curr_num_frames, embed_dim])
                        new_temporal_embed[:, :load_num_frames] = load_temporal_embed
                    elif self.load_temporal_fix in ['interp', 'bilinear']:
                        mode = 'bilinear' if self.load_temporal_fix == 'bilinear' else 'nearest'
                        new_temporal_embed = F.interpolate(load_temporal_embed.unsqueeze(0),
                                                           (curr_num_frames, embed_dim), mode=mode,
                                                           align_corners=True).squeeze(0)
                    else:
                        raise NotImplementedError
                new_state_dict['video_model.temporal_embed'] = new_temporal_embed
        if 'video_model.pos_embed' in new_state_dict and 'video_model.pos_embed' in curr_keys:
            if new_state_dict['video_model.pos_embed'].shape[1] != self.state_dict()['video_model.pos_embed'].shape[1]:
                raise NotImplementedError(
                    'Loading models with different spatial resolution / patch number not yet implemented, sorry.')
        return new_state_dict


class _SimMatrixFn(torch.autograd.Function):
    """sim_matrix(a, b, eps) = (a / max(|a|, eps)) @ (b / max(|b|, eps))^T, model/model.py:189-197, for the
    square case through the contrastive-head kernels (egv_egonce_fwd_bwd writes x and its row/col maths)."""

    @staticmethod
    def forward(ctx, a, b, eps):
        from ..loss_ops import sim_fwd
        sim, ctxdata = sim_fwd(a, b, eps)
        ctx.data = ctxdata
        return sim

    @staticmethod
    def backward(ctx, g):
        from ..loss_ops import sim_bwd
        da, db = sim_bwd(ctx.data, g)
        return da, db, None


def sim_matrix(a, b, eps=1e-8):
    """added eps for numerical stability (model/model.py:189-197)."""
    return _SimMatrixFn.apply(a, b, eps)


def sim_matrix_mm(a, b):
    """Plain inner-product similarity a @ b^T (run/test_epic.py:31-33), fp32-grade on the bf16x3 GEMM."""
    ops._need_cuda(a, b)
    n, m = a.shape[0], b.shape[0]
    m4 = (m + 3) // 4 * 4                     # the GEMM writes 4-column pieces: pad the video rows with zeros
    b = b.detach().contiguous().float()
    if m4 != m:
        b = torch.cat([b, b.new_zeros(m4 - m, b.shape[1])])
    out = torch.empty((n, m4), dtype=torch.float32, device=a.device)
    a_pl = ops.split_f32(a.detach().contiguous().float(), 3)[0]
    b_pl = ops.split_f32(b, 3)[0]
    ops.gemm_nt(a_pl, b_pl, passes=3, out_f32=out)      # stand-alone helper: the DEFAULT context
    return out[:, :m]


def dual_softmax_similarity(text_embeds, vid_embeds, temp=500.0):
    """The `--dual_softmax` retrieval similarity of run/test_epic.py:137-143: sim = text @ video^T (un-normalised),
    sim = softmax(sim / 500, dim=1) * sim, sim = softmax(sim, dim=0); [texts, videos] on the device."""
    from ..loss_ops import dual_softmax
    return dual_softmax(sim_matrix_mm(text_embeds, vid_embeds), temp)
