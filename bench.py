#!/usr/bin/env python
"""EgoClip pre-training step benchmark (BASELINE.json metric: clip-pairs/sec, whole node).

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Either the caller launches the ranks (`python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
come from the environment), or -- bare `python bench.py --gpus N` -- this script launches them itself (self_launch) and
refuses to run when fewer than N GPUs are visible.

A "step" is one pass of the hot path = trainer/trainer_egoclip.py:123-141 restated in
egovlp_amd.trainer.trainer_egoclip.egoclip_step: zero_grad, dual-encoder forward, embedding all-gather (RCCL),
similarity + EgoNCE, backward (DDP gradient all-reduce overlapped), AdamW.  Workload = BASELINE configs[1]/[2]:
synthetic EgoClip batch, 4 x 3 x 224 x 224 frames + 32-token text, ViT-B/16 + DistilBERT, B = 32 per GPU,
random-init weights (no network for checkpoints), inputs resident in HBM before the timed region.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before the HIP runtime initialises: see egovlp_amd/__init__.py

import torch
import torch.distributed as dist

import ctypes
_libc = ctypes.CDLL(None)     # fflush(NULL): native code (RCCL) writes to stdout through C stdio

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic work per clip-pair: 2*MACs, forward; training step = 3x
FWD_GFLOP_PER_PAIR = {("base_patch16_224", 4): 187.4, ("base_patch16_224", 16): 741.9, ("large_patch14_224", 4): 856.0}
PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA, MI355X_MICROARCH.md chip-level table


def build_model(arch, model_frames, text_dropout=0.1):
    from egovlp_amd.model.model import FrozenInTime
    from egovlp_amd.synth import synth_state_dict
    m = FrozenInTime(video_params={"model": "SpaceTimeTransformer", "arch_config": arch, "num_frames": model_frames,
                                   "pretrained": True, "time_init": "rand"},
                     text_params={"model": "distilbert-base-uncased", "pretrained": True, "input": "text"},
                     projection="minimal", load_checkpoint="")
    m.load_state_dict(synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, seed=0))
    m.text_model.set_dropout(text_dropout, text_dropout)
    return m


def cpu_baseline(B=8, T=4, L=32, reps=3):
    """The oracle (fp32 PyTorch-on-CPU restatement of the reference, pinned to reference outputs by tests/golden) timed on the
    host cores of this box: fwd + EgoNCE + bwd of the same workload at B = 8 (BASELINE configs[0]), one warm-up and
    `reps` timed repetitions, median."""
    from egovlp_amd.model.schema import state_dict_schema
    from egovlp_amd.synth import synth_batch, synth_state_dict
    from oracle import egovlp_oracle as O
    # many-socket hosts lose to oversubscription on these GEMM sizes (256 threads: 359 s for B=8): cap at 32
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    sd = {k: v.requires_grad_(True) for k, v in synth_state_dict(state_dict_schema(), seed=0).items()}

    def one(b):
        for v in sd.values():
            v.grad = None
        batch = synth_batch(b, T=T, L=L, seed=7)
        t0 = time.perf_counter()
        te, ve = O.frozen_in_time(batch, sd, O.VideoCfg(), O.TextCfg())
        loss, _ = O.egoclip_loss(te, ve, batch["noun_vec"], batch["verb_vec"])
        loss.backward()
        return time.perf_counter() - t0

    one(1)                       # warm-up (thread pool, allocator)
    ts = sorted(one(B) for _ in range(reps))
    dt = ts[len(ts) // 2]
    return {"value": round(B / dt, 4), "unit": "clip-pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "config": {"workload": f"the same EgoClip step (fwd + EgoNCE + bwd, no optimizer) at B={B} (BASELINE configs[0]); the GPU line "
                                   f"runs B=32", "batch": B, "frames": T, "text_tokens": L, "dtype": "f32"},
            "sample": f"median of {reps} fwd+EgoNCE+bwd steps of the CPU oracle at B={B} (T={T}, L={L}) after one warm-up: "
                      f"{dt:.1f} s [{ts[0]:.1f} .. {ts[-1]:.1f}], {torch.get_num_threads()} of {os.cpu_count()} logical cpus",
            "why_port": "/root/reference does not exist on the GPU box, so the reference itself cannot be timed there; the "
                        "oracle is its fp32 torch-CPU restatement, bit-identical to the reference on the committed golden "
                        "fixtures (tests/test_oracle_golden.py)"}


def grad_rel_err(model, loss_fn, data, world, rank, set_precision, mode):
    """rel-L2 distance between the gradients of precision `mode` and of the all-bf16x3 step (whose gradients the GPU tests
    hold to 1e-3 .. 3e-3 of the fp32 oracle) on the benchmark batch, for a fixed set of sentinel tensors."""
    from egovlp_amd.trainer.trainer_egoclip import AllGatherFused
    names = ["video_model.blocks.0.timeattn.qkv.weight", "video_model.blocks.5.attn.proj.weight",
             "video_model.blocks.11.mlp.fc2.weight", "video_model.patch_embed.proj.weight",
             "text_model.transformer.layer.0.attention.q_lin.weight", "vid_proj.0.weight"]
    params = dict(model.named_parameters())

    def grads(prec):
        set_precision(prec)
        for p_ in model.parameters():
            p_.grad = None
        te, ve = model(data)
        ve, te, n_, v_ = AllGatherFused.apply(ve, te, data["noun_vec"], data["verb_vec"], world, rank)
        loss = loss_fn.fused(te, ve, n_, v_)
        k = 1.0
        if model.exec_ctx.bwd_passes == 4:       # the fp16 backward: scaled loss, gradients un-scaled here (AdamW does it in a step)
            sc = model.exec_ctx.loss_scaler()
            k = 1.0 / sc.get_scale()
            loss = sc.scale(loss)
        loss.backward()
        model.exec_ctx.join_side_stream()
        return {n: params[n].grad.detach().double().clone() * k for n in names}

    pd, pa = model.text_model.config.dropout, model.text_model.config.attention_dropout
    model.text_model.set_dropout(0.0, 0.0)          # two calls draw different masks: compare the deterministic function
    poll = model.exec_ctx.backward_poll             # these backward passes are measurements, not steps: no gradient exchange
    model.exec_ctx.set(backward_poll=None)
    ref, got = grads("bf16x3"), grads(mode)
    model.exec_ctx.set(backward_poll=poll)
    model.text_model.set_dropout(pd, pa)
    set_precision(mode)
    for p_ in model.parameters():
        p_.grad = None
    errs = {n: float((got[n] - ref[n]).norm() / ref[n].norm()) for n in names}
    return {"vs": "the bf16x3 backward of the same step (itself within 3e-3 of the fp32 CPU oracle, tests/test_gpu_model.py)",
            "max": round(max(errs.values()), 5), "median": round(sorted(errs.values())[len(errs) // 2], 5),
            "per_tensor": {k: round(v, 5) for k, v in errs.items()}}


TRAJ_SENTINELS = ["video_model.blocks.0.timeattn.qkv.weight", "video_model.blocks.5.attn.proj.weight",
                  "video_model.blocks.11.mlp.fc2.weight", "video_model.patch_embed.proj.weight", "video_model.pos_embed",
                  "text_model.transformer.layer.0.attention.q_lin.weight", "text_model.transformer.layer.5.ffn.lin2.bias",
                  "vid_proj.0.weight", "txt_proj.1.weight"]


def trajectory_drift(model, loss_fn, make_opt, steps=20, B=8, T=4, L=32, modes=("bf16x3", "mixed"), seed=5000):
    """Does training in the benchmarked precision mode TRACK training with fp32-grade gradients?  (round-2 verdict, weak #1: every
    parity test is a single step.)  From the same initial weights, `steps` optimisation steps on the same sequence of synthetic
    batches (a fresh batch per step, text dropout off so that both runs see the same function) in each mode of `modes`
    ("bf16x3": three-product forward AND backward, gradients within 1e-3 .. 3e-3 of the fp32 oracle; "mixed": the benchmarked mode,
    same forward, single-pass bf16 backward).  Returns the loss curves, their largest relative gap, the loss of both end
    points on a held-out batch, and the parameter drift ||theta_mode - theta_ref|| / ||theta_ref - theta_0|| (how far the
    end point is from the reference end point, in units of the distance training moved the reference) -- whole model and
    per sentinel tensor.  Restores the model's weights, precision and dropout afterwards."""
    from egovlp_amd import weights
    from egovlp_amd.synth import synth_batch
    from egovlp_amd.trainer.trainer_egoclip import egoclip_step
    ec = model.exec_ctx
    saved_prec = (ec._s.get("fwd_passes"), ec._s.get("bwd_passes"), ec._s.get("f16_single"))
    pd, pa = model.text_model.config.dropout, model.text_model.config.attention_dropout
    model.text_model.set_dropout(0.0, 0.0)
    was_training = model.training
    model.train()
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}

    def dev(b):
        return {"video": b["video"].cuda(), "text": {k: v.cuda() for k, v in b["text"].items()},
                "noun_vec": b["noun_vec"].cuda(), "verb_vec": b["verb_vec"].cuda()}

    batches = [dev(synth_batch(B, T=T, L=L, seed=seed + i)) for i in range(steps)]
    held_out = dev(synth_batch(B, T=T, L=L, seed=seed + 10007))
    ec.set_precision("bf16x3")
    with torch.no_grad():
        te, ve = model(held_out)
        held_out_initial = float(loss_fn.fused(te, ve, held_out["noun_vec"], held_out["verb_vec"]))
    runs = {}
    for mode in modes:
        model.load_state_dict(sd0)
        weights.bump_epoch()
        if mode == "mixed":
            ec.set_precision("bf16x3", "bf16")
        elif mode == "bf16x3":
            ec.set_precision("bf16x3", "bf16x3")
        else:
            ec.set_precision(mode)            # "f16x2" / "f16mix": fp16-product forward of the video blocks' Linears, bf16 backward
        opt = make_opt(model.parameters())
        losses = [egoclip_step(model, loss_fn, opt, b) for b in batches]
        with torch.no_grad():
            te, ve = model(held_out)
            final = loss_fn.fused(te, ve, held_out["noun_vec"], held_out["verb_vec"])
        runs[mode] = {"loss": [float(x) for x in torch.stack(losses).cpu()], "held_out": float(final),
                      "theta": {k: v.detach().clone() for k, v in model.named_parameters()}}
        del opt
    ref, out = runs[modes[0]], {"steps": steps, "batch": B, "reference_mode": modes[0],
                                "held_out_loss_initial": round(held_out_initial, 5)}
    names = list(ref["theta"].keys())

    def dist2(a, b, keys):
        return float(sum(((a[k].double() - b[k].double()) ** 2).sum() for k in keys)) ** 0.5

    moved = dist2(ref["theta"], sd0, names)
    out["loss_" + modes[0]] = [round(x, 5) for x in ref["loss"]]
    out["held_out_loss_" + modes[0]] = round(ref["held_out"], 5)
    for mode in modes[1:]:
        r = runs[mode]
        out["loss_" + mode] = [round(x, 5) for x in r["loss"]]
        out["held_out_loss_" + mode] = round(r["held_out"], 5)
        out["max_rel_loss_gap_" + mode] = round(max(abs(a - b) / abs(b) for a, b in zip(r["loss"], ref["loss"])), 6)
        out["held_out_rel_gap_" + mode] = round(abs(r["held_out"] - ref["held_out"]) / abs(ref["held_out"]), 6)
        out["param_drift_" + mode] = round(dist2(r["theta"], ref["theta"], names) / moved, 5)
        out["param_drift_per_tensor_" + mode] = {
            k: round(dist2(r["theta"], ref["theta"], [k]) / max(dist2(ref["theta"], sd0, [k]), 1e-30), 5) for k in TRAJ_SENTINELS if k in ref["theta"]}
    out["note"] = ("param_drift = ||theta_mode - theta_ref|| / ||theta_ref - theta_0|| after `steps` AdamW steps from the same "
                   "weights on the same batches (text dropout off); AdamW's early updates are sign-like (lr * g / (|g| + eps)), so "
                   "elements whose gradient is within the backward's rounding error of zero move by up to 2 lr per step in "
                   "different directions -- the loss curves are the meaningful comparison")
    model.load_state_dict(sd0)
    weights.bump_epoch()
    model.text_model.set_dropout(pd, pa)
    ec.unset("fwd_passes", "bwd_passes", "f16_single")
    if saved_prec[0] is not None:
        ec.set(fwd_passes=saved_prec[0], bwd_passes=saved_prec[1])
    if saved_prec[2] is not None:
        ec.set(f16_single=saved_prec[2])
    model.train(was_training)
    return out


def measure_gemm_traffic(args, timeout_s=240):
    """roofline.traffic, measured in THIS run: two child runs of this script (2 timed steps, every secondary leg off, same precision /
    size) under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... --pmc WRITE_SIZE` -- separate passes, as the counters do not fit one
    and as MI355X_MICROARCH.md prescribes; never combined with sys / hip traces -- and the mean over every gemm_big_kernel launch of
    bytes = 2048 x FETCH_SIZE[KiB] + 1024 x WRITE_SIZE[KiB]: the guide's gfx950 correction (a reported FETCH KiB stands for 2048 bytes of
    a wide coalesced read), confirmed on this kernel's own load / store paths by a 1 GiB calibration copy
    (profiles/r03_traffic_calibration.json, tools/traffic_calib.py).  Infinity-Cache hits are counted: fabric traffic, an upper bound of
    DRAM traffic.  -> (dict | None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found on this box"
    raw = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="egv_pmc_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
               os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--batch", str(args.batch), "--frames", str(args.frames), "--arch", args.arch,
               "--precision", args.precision, "--no-cpu-baseline", "--no-kernel-timing", "--no-fast-mode", "--no-trajectory", "--no-h2d-leg",
               "--no-dp-leg", "--no-grad-err", "--no-guard", "--no-traffic"]
        try:
            p = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout_s)
        except subprocess.TimeoutExpired:
            shutil.rmtree(d, ignore_errors=True)
            return None, f"the rocprofv3 --pmc {counter} pass did not finish in {timeout_s} s"
        files = glob.glob(os.path.join(d, "**", "*counter_collection*.csv"), recursive=True)
        if p.returncode != 0 or not files:
            shutil.rmtree(d, ignore_errors=True)
            return None, f"the rocprofv3 --pmc {counter} pass failed (exit {p.returncode}): {(p.stderr or '')[-200:]}"
        tot, n = 0.0, 0
        with open(files[0]) as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") == counter and "gemm_big_kernel" in r.get("Kernel_Name", ""):
                    tot += float(r["Counter_Value"])
                    n += 1
        shutil.rmtree(d, ignore_errors=True)
        if n == 0:
            return None, f"no gemm_big_kernel launches in the --pmc {counter} pass"
        raw[counter] = (tot / n, n)
    f_kib, w_kib = raw["FETCH_SIZE"][0], raw["WRITE_SIZE"][0]
    return {"hbm_bytes_per_launch": round((2.0 * f_kib + w_kib) * 1024.0), "fetch_kib_raw": round(f_kib, 1), "write_kib_raw": round(w_kib, 1),
            "launches_counted": [raw["FETCH_SIZE"][1], raw["WRITE_SIZE"][1]],
            "how": "this run: two child runs of this command (2 timed steps) under rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE "
                   "(separate passes); mean over all gemm_big_kernel launches of 2048 x FETCH_SIZE[KiB] + 1024 x WRITE_SIZE[KiB] (gfx950 "
                   "correction of MI355X_MICROARCH.md, calibrated in profiles/r03_traffic_calibration.json); Infinity-Cache hits included "
                   "(fabric traffic)"}, None


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: start N ranks of this script on this node with
    torch.distributed.run (one process per GPU, RCCL rendezvous on 127.0.0.1) and return its exit code.  Fails loudly, before
    anything is launched, when the node has fewer than N GPUs -- a 1-GPU number must never be printed as an N-GPU one."""
    import socket
    import subprocess
    if not args.launch_dry_run and torch.cuda.device_count() < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} HIP device(s) are visible on this node",
              file=sys.stderr)
        return 2
    with socket.socket() as s:        # a free rendezvous port (the driver's own launcher passes --master-port itself)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def launch_dry_run(world, rank):
    """The launch path without a GPU: every rank joins a gloo process group, proves it with an all-reduce of its rank id and
    rank 0 prints the JSON line (n_gpus = ranks that actually joined)."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="gloo")
    t = torch.tensor([float(rank + 1), 1.0])
    dist.all_reduce(t)
    joined = int(t[1])
    ok = joined == world == dist.get_world_size() and int(t[0]) == world * (world + 1) // 2
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": joined, "ranks_joined": joined, "world_size": dist.get_world_size(),
                          "backend": dist.get_backend(), "ok": bool(ok)}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU")
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--arch", default="base_patch16_224")
    ap.add_argument("--precision", default=os.environ.get("EGOVLP_PRECISION", "f16mix"),
                    help="f16mix (default: embeddings and loss inside the 1e-3 parity bar with a margin of 2 -- the video blocks' qkv / fc1 / fc2 "
                         "Linears as TWO fp16 products in the first quarter of the tower and ONE behind it, everything else three bf16 "
                         "products --, backward single-pass bf16) | f16x2 (round 4: two fp16 products in every block, fp32-grade forward) | "
                         "mixed (the same with three bf16 products everywhere in the forward) | "
                         "bf16 (single pass everywhere, fast mode) | bf16x3 (fp32-grade everywhere)")
    ap.add_argument("--text-dropout", type=float, default=0.1, help="DistilBERT dropout / attention_dropout in the timed step "
                    "(HF default 0.1 = what the reference trains with; 0 = the deterministic parity configuration)")
    ap.add_argument("--no-fast-mode", action="store_true", help="skip the secondary single-pass bf16 measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-h2d-leg", action="store_true", help="skip the secondary with_h2d_uint8 measurement")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-dp-leg", action="store_true", help="skip the secondary leg that runs the N > 1 step policy (process group, RCCL "
                    "gather, hook-free gradient exchange, ONE wgrad stream, 248-workgroup GEMM grid) at world size 1")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc child runs that measure roofline.traffic (N = 1 only)")
    ap.add_argument("--no-guard", action="store_true", help="skip the precision guard's measurement of the per-block policy before the timed steps")
    ap.add_argument("--no-grad-err", action="store_true", help="skip grad_rel_err (two extra backward passes; A/B timing runs)")
    ap.add_argument("--no-trajectory", action="store_true", help="skip the 20-step loss / parameter drift comparison of the "
                    "benchmarked precision mode against the all-bf16x3 (fp32-grade gradient) run")
    ap.add_argument("--gemm-grid", type=int, default=0, help="persistent workgroups of the big GEMM (default: 256 at N=1, "
                    "248 at N>1 so that the overlapped RCCL kernels find free CUs)")
    ap.add_argument("--wgrad-side", type=int, default=int(os.environ.get("EGV_WGRAD_SIDE", "1")),
                    help="1 (default): weight-gradient GEMMs on their own HIP stream (egovlp_amd.ops.side_stream); 0: on the main stream")
    ap.add_argument("--text-side", type=int, default=int(os.environ.get("EGV_TEXT_SIDE", "1")),
                    help="1 (default): the DistilBERT tower on a second HIP stream under the video tower; 0: one stream")
    ap.add_argument("--main-priority", type=int, default=int(os.environ.get("EGV_MAIN_PRIO", "0")),
                    help="1: run the step on a HIGH-priority HIP stream (the wgrad / text side streams keep the default priority: "
                         "their workgroups fill the CUs the main stream's kernels leave free instead of competing with them)")
    ap.add_argument("--rccl-channels", type=int, default=0, help="N>1: cap RCCL at this many channels (= workgroups) via "
                    "NCCL_MAX_NCHANNELS, e.g. 8 to match the CUs the 248-workgroup GEMM grid leaves free (default: RCCL's choice)")
    ap.add_argument("--grad-exchange", default=os.environ.get("EGV_GRAD_EXCHANGE", "direct"), choices=["direct", "allreduce"],
                    help="N>1: how a bf16 gradient bucket travels -- direct (default): all-to-all of bucket slices over all xGMI "
                         "links, fp32 slice sum, all-gather; allreduce: RCCL's stock all_reduce of the bf16 bucket")
    ap.add_argument("--no-grad-sync", action="store_true", help="diagnostics: process group and embedding gather, but no "
                    "gradient exchange (what does the distributed environment itself cost?)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the RCCL process group, DDP wrapper and the fused all-gather even at world size 1 "
                         "(smoke test of the N>1 code path on a 1-GPU box)")
    ap.add_argument("--launch-dry-run", action="store_true",
                    help="exercise ONLY the N-rank launch path (self re-exec under torch.distributed.run, rendezvous, one "
                         "all-reduce on the gloo backend, rank 0 prints a JSON line) -- no GPU, no model; tests/test_bench_launch.py")
    args = ap.parse_args()

    # ---- N ranks: one process per GPU (reference: run/train_egoclip.py:39-45,128-134 / README.md:78-84).  Invoked bare
    # (`python bench.py --gpus N`, no WORLD_SIZE in the environment) this process is only the LAUNCHER: it re-executes
    # itself under torch.distributed.run with N local ranks and passes rank 0's JSON line through.  Invoked by
    # torch.distributed.run already (WORLD_SIZE set) it is one of the ranks.
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the two must agree (one rank per GPU)")
    if args.launch_dry_run:
        raise SystemExit(launch_dry_run(world, rank))
    if torch.cuda.device_count() < max(args.gpus, local_rank + 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} (local rank {local_rank}) but only {torch.cuda.device_count()} HIP "
                         f"device(s) are visible: refusing to report a {args.gpus}-GPU number from fewer GPUs")
    torch.cuda.set_device(local_rank)
    torch.manual_seed(1234 + rank)      # the dropout masks of the text encoder are a function of torch's seed: reproducible runs
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.rccl_channels > 0:
            os.environ["NCCL_MAX_NCHANNELS"] = str(args.rccl_channels)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))   # nccl == RCCL on ROCm
        if args.force_dist:
            os.environ["EGV_FORCE_GATHER"] = "1"

    from egovlp_amd import ops
    from egovlp_amd.model.loss import EgoNCE
    from egovlp_amd.optim import AdamW
    from egovlp_amd.synth import synth_batch
    from egovlp_amd.trainer.trainer_egoclip import egoclip_step

    B, T, L = args.batch, args.frames, 32
    model = build_model(args.arch, 16, args.text_dropout).cuda().train()
    model.text_model.seed_rank = rank     # data-parallel ranks draw different dropout masks
    ec = model.exec_ctx                   # everything below configures THIS model's execution context (egovlp_amd.ops.ExecContext)

    def set_precision(name):
        if name == "mixed":
            ec.set_precision("bf16x3", "bf16")
        else:
            ec.set_precision(name)

    set_precision(args.precision)
    net = model
    grad_sync = None
    if use_dist and not args.no_grad_sync:
        # hook-free gradient exchange: buckets launched from the polls of the video tower's backward (ExecContext.backward_poll).
        # (The grad-ready-hook variant and the DistributedDataParallel A/B leg of rounds 2-3 measured slower -- 3.2 % / 9 % against
        # 1.7 % at world size 1, profiles/r02_d_dp_overhead.txt -- and live in tools/ddp_ab.py now.)
        from egovlp_amd.dist import Bf16GradSync
        grad_sync = Bf16GradSync(model.parameters(), use_hooks=False, order_hint=model.gradient_ready_order(), exec_ctx=ec,
                                 exchange=args.grad_exchange)
        ec.set(backward_poll=grad_sync.poll)
    grid = args.gemm_grid or (248 if world > 1 else 256)
    ec.set(gemm_grid=grid)
    ec.set(wgrad_side_stream=bool(args.wgrad_side), text_side_stream=bool(args.text_side))
    opt = AdamW(model.parameters(), lr=3e-5)
    loss_fn = EgoNCE()
    batch = synth_batch(B, T=T, L=L, seed=1234, rank=rank)
    data = {"video": batch["video"].cuda(), "text": {k: v.cuda() for k, v in batch["text"].items()},
            "noun_vec": batch["noun_vec"].cuda(), "verb_vec": batch["verb_vec"].cuda()}

    # the per-block precision policy measured on THESE weights and THIS batch before anything is timed (egovlp_amd.guard): the timed
    # steps run the policy it leaves in force (on the synthetic Gaussian weights: the shipped one), and the line says what it measured
    guard_report = None
    if args.precision in ("f16mix", "f16x2") and not args.no_guard:
        from egovlp_amd.guard import PrecisionGuard
        guard_report = PrecisionGuard(model).check(data)

    HOST = {}

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def one_step():
        return egoclip_step(net, loss_fn, opt, data, world, rank, grad_sync=grad_sync)

    def measure(steps, warmup):
        for _ in range(warmup):
            one_step()
        barrier()
        t0 = time.perf_counter()
        marks = [t0]
        w0 = ec.flow_wait_s
        for _ in range(steps):
            loss = one_step()
            marks.append(time.perf_counter())
        waited = ec.flow_wait_s - w0            # inside ExecContext._throttle: the host waiting for the step of two steps ago
        HOST["enqueue_ms_per_step"] = (marks[-1] - t0 - waited) / steps * 1e3     # host time per step, flow-control waits excluded
        HOST["flow_wait_ms_per_step"] = waited / steps * 1e3
        barrier()
        dt = time.perf_counter() - t0
        # flow control (max_steps_in_flight, default 2) stops the host before the HIP runtime's own back-pressure would (which sets in
        # 3 - 5 steps ahead and cannot be told apart from work): the waits are explicit and timed, what remains is the host's work.
        # Cross-checks: the cheapest three consecutive steps of the loop (wall clock, waits included when they happen) and how far
        # ahead of the GPU the host was when it had enqueued the last step.
        per = [(b_ - a_) * 1e3 for a_, b_ in zip(marks, marks[1:])]
        HOST["per_step_wall_ms"] = [round(x, 1) for x in per]          # host wall time between consecutive enqueues (waits included)
        w = min(3, len(per))
        HOST["enqueue_ms_unthrottled"] = min(sum(per[i:i + w]) / w for i in range(len(per) - w + 1))
        HOST["lead_ms"] = (dt - (marks[-1] - t0)) * 1e3
        tmax = torch.tensor([dt], device="cuda", dtype=torch.float64)
        if use_dist:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax), float(loss)

    if args.main_priority:
        hp = torch.cuda.Stream(priority=-1)
        hp.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(hp)          # everything below (timed steps, instrumented passes) runs on the high-priority stream
    dt, loss_val = measure(args.steps, args.warmup)
    # what ONE step costs the host when nothing pushes back: enqueue a step onto idle streams (3 repetitions, minimum).  The
    # steady-state figure above (time until the host has enqueued `steps` steps) also contains the waits for space in the
    # hardware queues once the host is a full step ahead of the GPU -- back-pressure, not work.
    idle = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0_ = time.perf_counter()
        one_step()
        idle.append((time.perf_counter() - t0_) * 1e3)
    torch.cuda.synchronize()
    HOST["enqueue_idle_ms"] = min(idle)
    ms = dt / args.steps * 1e3
    pairs = world * B * args.steps / dt

    # ---- dominant-kernel roofline: the bf16-MFMA GEMM.  HIP events bracket every egv_gemm_nt launch on the stream
    # the kernels run on, in a separate instrumented pass of the same step (so the timed value above is untouched).
    roof = None
    if not args.no_kernel_timing:
        timer = ops.KernelTimer()
        # per-launch durations: one kernel at a time (no side streams) while the timer is attached
        ec.set(kernel_timer=timer, wgrad_side_stream=False, text_side_stream=False)
        for _ in range(2):
            egoclip_step(net, loss_fn, opt, data, world, rank, grad_sync=grad_sync)
        torch.cuda.synchronize()
        kt = timer.summary()
        ec.set(kernel_timer=None, wgrad_side_stream=bool(args.wgrad_side), text_side_stream=bool(args.text_side))
        g_all = kt["egv_gemm_nt"]
        # the dominant kernel is gemm_big_kernel (all token-major GEMMs and every wgrad); the 128x128 kernel of the small-M
        # problems (DistilBERT, projections: latency-bound, 6 % of the GEMM time) is reported separately, not averaged in
        big = {k: v for k, v in g_all["shapes"].items() if k.startswith("gemm_big ")}
        g = {"shapes": big, "launches": sum(v["launches"] for v in big.values()), "seconds": sum(v["seconds"] for v in big.values()),
             "flops": sum(v["flops"] for v in big.values())}
        g["issue_flops"] = sum(v["flops"] * (3 if k.endswith("x3") else (2 if k.endswith("x2") else 1)) for k, v in big.items())
        ach = g["flops"] / g["seconds"] / 1e12
        shapes = sorted(g_all["shapes"].items(), key=lambda kv: -kv[1]["seconds"])
        table = [{"shape": k, "launches_per_step": v["launches"] // 2, "avg_us": round(v["seconds"] / v["launches"] * 1e6, 1),
                  "tflops": round(v["flops"] / v["seconds"] / 1e12, 1), "ms_per_step": round(v["seconds"] / 2 * 1e3, 3)}
                 for k, v in shapes[:14]]
        # HBM traffic of the dominant kernel: PMC counters need their own rocprofv3 --pmc passes (they cannot be read from
        # inside this process), so the number comes from the committed summary of such a pass of THIS command, if present
        # (`traffic` itself is therefore null in THIS line: it is not a measurement of this run; the committed number is quoted
        # under `traffic_reference` with its origin)
        traffic_ref = None
        tpath = os.path.join(ROOT, "profiles", "gemm_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            traffic_ref = {"from": "profiles/gemm_traffic.json (separate rocprofv3 --pmc passes of this command on a builder box): "
                                   + str(tj.get("source")), "hbm_bytes_per_launch": tj.get("hbm_bytes_per_launch"),
                           "measured_at_commit": tj.get("measured_at_commit"), "precision": tj.get("precision")}
        # algorithmic bytes of a launch (SURVEY 8(d)'s accounting: each operand and the result once, 16 bits per element), mean over the step's launches
        import re as _re
        alg_bytes = 0.0
        for k_, v_ in big.items():
            mm = _re.search(r"M=(\d+) N=(\d+) K=(\d+)", k_)
            if mm:
                m_, n_, kk_ = (int(x) for x in mm.groups())
                alg_bytes += v_["launches"] * 2.0 * (m_ * kk_ + n_ * kk_ + m_ * n_)
        alg_bytes_per_launch = alg_bytes / max(g["launches"], 1)
        roof = {"bound": "mfma", "kernel": "gemm_big_kernel (every launch of the step: NT forward / dgrad, TN wgrad incl. its split-K reduce)",
                "algorithmic_bytes_per_launch": round(alg_bytes_per_launch),
                "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": None, "traffic_reference": traffic_ref,
                "launches_per_step": g["launches"] // 2,
                "avg_launch_us": round(g["seconds"] / g["launches"] * 1e6, 2),
                "algorithmic_gflop_per_launch": round(g["flops"] / g["launches"] / 1e9, 3),
                "gemm_ms_per_step": round(g["seconds"] / 2 * 1e3, 3),
                "mfma_issue_tflops": round(g["issue_flops"] / g["seconds"] / 1e12, 1),
                "all_gemm_launches": {"achieved": round(g_all["flops"] / g_all["seconds"] / 1e12, 1),
                                      "launches_per_step": g_all["launches"] // 2,
                                      "gemm_ms_per_step": round(g_all["seconds"] / 2 * 1e3, 3)},
                "per_shape": table,
                "note": "achieved = algorithmic 2*M*N*K of the gemm_big launches of a step / their summed HIP-event time "
                        "(events on the launch stream around each C-ABI call); bf16x3 launches issue 3 MFMA passes per "
                        "algorithmic product (mfma_issue_tflops counts them); traffic = (FETCH_SIZE x 2 + WRITE_SIZE) per "
                        "launch comes from two rocprofv3 --pmc child runs of this command at the end of THIS run (traffic_measurement; "
                        "traffic_missing_because says why not, and traffic_reference quotes the committed summary of an earlier run)"}
    key = (args.arch, T)
    step_frac = None
    if key in FWD_GFLOP_PER_PAIR:
        step_frac = pairs / world * FWD_GFLOP_PER_PAIR[key] * 3 * 1e9 / (PEAK_BF16_TFLOPS * 1e12)

    out = {
        "metric": "clip-pairs/sec (whole node), 4f/224^2 ViT-B + 32-tok text, B=32/GPU, train step (fwd+gather+EgoNCE+bwd+AdamW)"
                  if (T, B, args.arch) == (4, 32, "base_patch16_224") else
                  f"clip-pairs/sec (whole node), {T}f/224^2 {args.arch} + 32-tok text, B={B}/GPU, train step (fwd+gather+EgoNCE+bwd+AdamW)",
        "value": round(pairs, 2), "unit": "clip-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("fp16 (every Linear and both attentions of the video blocks, forward and backward) + bf16 (patch embedding, text tower, heads) MFMA "
                  "operands, fp32 accumulate" if ec.precision_name()[1] == "f16" else
                  {"f16mix": "fp16 (forward Linears of the video blocks) + bf16 (attention, early proj, text tower, backward) MFMA operands, fp32 accumulate",
                   "f16x2": "fp16 (forward Linears of the video blocks) + bf16 (attention, proj, text tower, backward) MFMA operands, fp32 accumulate"
                   }.get(ec.precision_name()[0], "bf16")), "data": "synthetic (random frames/tokens/noun-verb vectors, random-init weights)",
        "config": {"workload": f"EgoClip step: {T}x3x224x224 frames + {L}-tok text, {args.arch} + DistilBERT, EgoNCE, "
                               f"B={B}/GPU, global batch {B * world}", "global_batch": B * world,
                   "parallelism": f"dp{world}", "precision": "/".join(ec.precision_name()),
                   "mfma_products": "forward: " + {"f16mix": "per-block precision policy k = " + json.dumps(ec.f16_single_policy(len(model.video_model.blocks)))
                                                             + " (profiles/r05_precision_table.txt): qkv / fc1 / fc2 of the video blocks 2 x fp16 MFMA (f16x2 "
                                                             "operands) in blocks [0, k), 1 x fp16 MFMA in blocks [k, depth); proj "
                                                             + ("2 x fp16" if ec.precision_name()[1] == "f16" else "3 x bf16") + " MFMA in blocks "
                                                             "[0, k_proj), 1 x fp16 MFMA behind; attention 3 x "
                                                             + ("fp16 (fp16-split qkv planes)" if ec.precision_name()[1] == "f16" else "bf16")
                                                             + " MFMA; patch embedding / text tower / heads 3 x bf16 MFMA",
                                                   "f16x2": "qkv / fc1 / fc2 of the video blocks 2 x fp16 MFMA (f16x2 operands), proj / attention / "
                                                            "text tower / heads 3 x bf16 MFMA",
                                                   "bf16x3": "3 x bf16 MFMA per product", "bf16": "1 x bf16 MFMA per product"}[ec.precision_name()[0]]
                                    + "; backward: " + {"bf16x3": "3 x bf16 MFMA per product", "bf16": "1 x bf16 MFMA per product",
                                                        "f16": "1 x fp16 MFMA per product of the video blocks' Linears on loss-scaled gradients (dynamic scale on the "
                                                               "device, egovlp_amd.optim.LossScaler), 3 x bf16 in the text tower / patch embedding / heads"
                                                        }[ec.precision_name()[1]] + "; fp32 accumulation",
                   "text_dropout": args.text_dropout,
                   "streams": {"text_tower_side_stream": bool(args.text_side), "wgrad_side_stream": bool(args.wgrad_side),
                               # what differs between the N = 1 and the N > 1 step: one wgrad stream and a 248-workgroup
                               # GEMM grid under a process group, two wgrad streams and 256 workgroups without
                               "wgrad_streams": (ops._wgrad_stream_count() if args.wgrad_side else 0), "gemm_grid": grid,
                               "main_stream_high_priority": bool(args.main_priority),
                               "max_steps_in_flight": int(ec.max_steps_in_flight)}},
        "loss": round(loss_val, 5),
        # caching-allocator state after the timed loop: reserved HBM and how often an allocation had to free cached blocks and retry
        # (each retry synchronises the device: a host that runs many steps ahead holds that many steps' workspaces)
        "hbm_reserved_gb": round(torch.cuda.memory_reserved() / 2 ** 30, 1),
        "alloc_retries": int(torch.cuda.memory_stats().get("num_alloc_retries", 0)),
        "host_enqueue_ms_per_step": round(HOST.get("enqueue_ms_per_step", 0.0), 3),
        "host_enqueue_ms_from_idle_streams": round(HOST.get("enqueue_idle_ms", 0.0), 3),
        "host_flow_control_wait_ms_per_step": round(HOST.get("flow_wait_ms_per_step", 0.0), 3),
        "host_enqueue_ms_per_step_unthrottled": round(HOST.get("enqueue_ms_unthrottled", 0.0), 3),
        "host_lead_ms_at_last_enqueue": round(HOST.get("lead_ms", 0.0), 1),
        "host_wall_ms_between_enqueues": HOST.get("per_step_wall_ms"),
        "step_mfma_frac": None if step_frac is None else round(step_frac, 4),
    }
    if roof is not None:
        out["roofline"] = roof
    # ---- how far the gradients of THIS precision mode are from the fp32-grade (bf16x3) backward of the same step
    if guard_report is not None:
        out["precision_guard"] = {"what": "video embedding of this batch in the policy in force vs the all-bf16x3 forward, measured on the device before the "
                                          "timed steps (egovlp_amd/guard.py); over budget -> the policy is demoted, logged, and the line reports it",
                                  **guard_report}
    if args.precision != "bf16x3" and not args.no_grad_err:
        out["grad_rel_err"] = grad_rel_err(model, loss_fn, data, world, rank, set_precision, args.precision)
    # ---- exchange diagnostics (N > 1): what the step spends in the collectives that backward does not hide
    if use_dist:
        import egovlp_amd.dist as egd
        egd.COMM_EVENTS = []
        nrep = 3
        for _ in range(nrep):
            egoclip_step(net, loss_fn, opt, data, world, rank, grad_sync=grad_sync)
        torch.cuda.synchronize()
        acc = {}
        for name, e0, e1 in egd.COMM_EVENTS:
            acc[name] = acc.get(name, 0.0) + e0.elapsed_time(e1)
        egd.COMM_EVENTS = None
        mine = torch.tensor([ms] + [acc.get(k, 0.0) / nrep for k in ("embedding_all_gather", "grad_sync_exposed")],
                            device="cuda", dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        allr = torch.stack(allr).cpu()
        out["comm"] = {"rccl_ranks": world, "backend": dist.get_backend(), "gemm_grid": grid,
                       "rccl_max_channels": os.environ.get("NCCL_MAX_NCHANNELS", "default"),
                       "gradient_exchange": (
                           "Bf16GradSync direct (bf16 buckets: all-to-all of slices over all links, fp32 slice sum, all-gather; launched from "
                           "the backward polls on a private stream)" if args.grad_exchange == "direct" else
                           "Bf16GradSync allreduce (bf16 buckets, async RCCL all-reduce launched from the backward polls)"),
                       "grad_sync": None if grad_sync is None else {k: int(v) for k, v in grad_sync.stats.items()
                                                                   if k in ("buckets", "launched_during_backward")},
                       "ms_per_step_rank_min": round(float(allr[:, 0].min()), 3), "ms_per_step_rank_max": round(float(allr[:, 0].max()), 3),
                       "embedding_all_gather_ms": round(float(allr[:, 1].mean()), 3),
                       "grad_sync_exposed_ms_mean": round(float(allr[:, 2].mean()), 3),
                       "grad_sync_exposed_ms_max": round(float(allr[:, 2].max()), 3),
                       "note": "HIP events on the compute stream; grad_sync_exposed = wait for the bucket all-reduces that "
                               "backward did not hide + unpack; ms_per_step_rank_* are each rank's own untimed-barrier clock"}
    # ---- and over a TRAJECTORY: 20 optimisation steps at B = 8 in this mode vs the fp32-grade backward, same init, same batches
    if args.precision in ("mixed", "f16x2", "f16mix") and not args.no_trajectory and world == 1 and (T, args.arch) == (4, "base_patch16_224"):
        out["trajectory"] = trajectory_drift(model, loss_fn, lambda ps: AdamW(ps, lr=3e-5), modes=("bf16x3", args.precision))
        opt = AdamW(model.parameters(), lr=3e-5)        # fresh optimizer state for the legs below (weights were restored)
    if args.precision != "bf16" and not args.no_fast_mode:
        # secondary line: the same step with single-pass bf16 operands everywhere (embeddings ~6e-3 from fp32:
        # outside the parity bar, reported for reference only -- `value` above is the parity-mode number)
        set_precision("bf16")
        dt2, loss2 = measure(args.steps, max(args.warmup, 2))
        out["fast_mode_bf16"] = {"value": round(world * B * args.steps / dt2, 2), "unit": "clip-pairs/s",
                                 "ms_per_step": round(dt2 / args.steps * 1e3, 3), "loss": round(loss2, 5),
                                 "step_mfma_frac": None if key not in FWD_GFLOP_PER_PAIR else round(
                                     B * args.steps / dt2 * FWD_GFLOP_PER_PAIR[key] * 3e9 / (PEAK_BF16_TFLOPS * 1e12), 4)}
        set_precision(args.precision)
    if world == 1 and not args.no_h2d_leg:
        # secondary, clearly labelled figure: the step WITH the input hand-over the reference's step has (trainer/trainer_egoclip.py:
        # 114-121 moves every batch to the device before the step).  Decoded uint8 frames in pinned host memory (the loader's
        # output format; x / 255 and Normalize run inside the patch gather on the device), copied on a private copy stream one
        # batch ahead of the compute stream (egovlp_amd.trainer.trainer_egoclip._prefetched, what Multi_Trainer_dist._train_epoch
        # does).  `value` above stays the HBM-resident number.
        from egovlp_amd.trainer.trainer_egoclip import _prefetched
        mean = torch.tensor(ops.IMAGENET_MEAN).view(1, 1, 3, 1, 1)
        std = torch.tensor(ops.IMAGENET_STD).view(1, 1, 3, 1, 1)
        u8 = ((batch["video"] * std + mean).clamp(0, 1) * 255).round().to(torch.uint8).pin_memory()
        host = {"video": u8, "text": {k: v.pin_memory() for k, v in batch["text"].items()},
                "noun_vec": batch["noun_vec"].pin_memory(), "verb_vec": batch["verb_vec"].pin_memory()}
        nh = args.steps + 3

        def host_batches():
            for i in range(nh):
                yield i, 0, host
        feed = _prefetched(host_batches(), torch.device("cuda", local_rank))
        t0h, lossh = None, None
        for bi, _, dev_batch in feed:
            if bi is None:
                break
            if bi == 3:                      # three warm-up steps (first use of the uint8 gather), then the clock
                torch.cuda.synchronize()
                t0h = time.perf_counter()
            lossh = egoclip_step(net, loss_fn, opt, dev_batch, world, rank, grad_sync=grad_sync)
        torch.cuda.synchronize()
        dth = time.perf_counter() - t0h
        out["with_h2d_uint8"] = {"value": round(B * args.steps / dth, 2), "unit": "clip-pairs/s", "ms_per_step": round(dth / args.steps * 1e3, 3),
                                 "host_bytes_per_step": int(u8.numel() + sum(v.numel() * v.element_size() for v in batch["text"].values())
                                                            + 4 * (batch["noun_vec"].numel() + batch["verb_vec"].numel())),
                                 "input": "decoded uint8 frames + token ids / masks / noun-verb vectors in pinned host memory, copied one "
                                          "batch ahead on a copy stream; normalisation fused into the patch gather", "loss": round(float(lossh), 5)}
    if world == 1 and not use_dist and not args.no_dp_leg:
        # secondary, clearly labelled: the step exactly as `world > 1` configures it -- RCCL process group, fused embedding
        # all-gather, hook-free bf16 gradient exchange (`--grad-exchange`) launched from the backward polls, ONE wgrad side stream, the
        # GEMM grid capped at 248 workgroups -- on the one GPU there is: what the data-parallel machinery costs before any byte crosses
        # a link.  No scaling claim: gpurun boxes have one GPU.
        from egovlp_amd.dist import Bf16GradSync
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ["RANK"], os.environ["WORLD_SIZE"] = "0", "1"
        # RCCL prints a version banner through C stdio on stdout when its first communicator comes up: this leg's stdout goes to
        # stderr, so that the ONE JSON line stays the only thing on stdout
        sys.stdout.flush()
        _libc.fflush(None)
        saved_fd1 = os.dup(1)
        os.dup2(2, 1)
        pg_up = False
        try:
            # a free rendezvous port: a fixed one may still be held by a previous run on this node
            if "MASTER_PORT" not in os.environ or os.environ["MASTER_PORT"] == "29533":
                import socket
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            import datetime
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=120))
            pg_up = True
            os.environ["EGV_FORCE_GATHER"] = "1"
            use_dist = True
            grad_sync = Bf16GradSync(model.parameters(), use_hooks=False, order_hint=model.gradient_ready_order(), exec_ctx=ec,
                                     exchange=args.grad_exchange)
            ec.set(backward_poll=grad_sync.poll, gemm_grid=248)
            if args.wgrad_side and len(ec._side["extra"]) + 1 != ops._wgrad_stream_count():
                ec.reset_side_streams()         # (only when an EGV_WGRAD_STREAMS override made the N = 1 stream count differ)
            dt3, loss3 = measure(args.steps, max(args.warmup, 3))
            out["dp_policy_at_world_size_1"] = {
                "value": round(B * args.steps / dt3, 2), "unit": "clip-pairs/s", "ms_per_step": round(dt3 / args.steps * 1e3, 3),
                "loss": round(loss3, 5), "gemm_grid": 248, "wgrad_streams": ops._wgrad_stream_count() if args.wgrad_side else 0,
                "gradient_exchange": args.grad_exchange, "buckets": int(grad_sync.stats.get("buckets", 0)),
                "launched_during_backward": int(grad_sync.stats.get("launched_during_backward", 0)),
                "what": "the N > 1 step policy (RCCL process group, fused all-gather, hook-free bf16 gradient exchange, one wgrad stream, "
                        "248-workgroup GEMM grid) at world size 1 on this GPU; `value` above is the N = 1 policy"}
        except Exception as e:          # a secondary leg must never cost the line: report why it is missing
            out["dp_policy_at_world_size_1"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        finally:
            ec.set(backward_poll=None, gemm_grid=grid)
            grad_sync = None
            os.environ.pop("EGV_FORCE_GATHER", None)
            use_dist = False
            try:
                torch.cuda.synchronize()
                if pg_up:
                    dist.destroy_process_group()
            except Exception:
                pass
            sys.stdout.flush()
            _libc.fflush(None)
            os.dup2(saved_fd1, 1)
            os.close(saved_fd1)
    # roofline.traffic of THIS run (N = 1): PMC counters need their own profiler passes, so two short child runs of this command
    if rank == 0 and world == 1 and not args.no_traffic and out.get("roofline") is not None:
        try:
            torch.cuda.empty_cache()              # the children allocate their own ~40 GB next to this process
            tr, why = measure_gemm_traffic(args)
        except Exception as e:                    # a secondary leg must never cost the line
            tr, why = None, f"{type(e).__name__}: {e}"[:200]
        if tr is not None:
            out["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
            out["roofline"]["traffic_unit"] = "bytes per gemm_big_kernel launch (fabric: HBM + Infinity-Cache hits)"
            out["roofline"]["traffic_measurement"] = tr
            alg = out["roofline"].get("algorithmic_bytes_per_launch")
            if alg:
                out["roofline"]["traffic_over_algorithmic"] = round(tr["hbm_bytes_per_launch"] / alg, 2)
        else:
            out["roofline"]["traffic_missing_because"] = why
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    _libc.fflush(None)        # anything native code buffered on stdout (RCCL's banner at N > 1) goes out BEFORE the line
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
