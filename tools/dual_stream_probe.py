"""EXPERIMENT: forward + backward of the EgoClip step as two half-batches on two HIP streams (GEMM grids capped at 128
workgroups each) vs the whole batch on one stream.  The two halves run the same kernel sequence phase-shifted, so the HBM-bound
kernels and epilogue bursts of one half can hide under the matrix-core-bound main loops of the other.
usage: python tools/dual_stream_probe.py [mixed|bf16|bf16x3] [B]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from egovlp_amd import ops, _lib
from egovlp_amd.model.loss import EgoNCE
from egovlp_amd.synth import synth_batch

prec = sys.argv[1] if len(sys.argv) > 1 else "mixed"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ops.Precision.set("bf16x3", "bf16") if prec == "mixed" else ops.Precision.set(prec)
model = bench.build_model("base_patch16_224", 16).cuda().train()
lossf = EgoNCE()
b = synth_batch(B, T=4, L=32, seed=1234)
data = {"video": b["video"].cuda(), "text": {k: v.cuda() for k, v in b["text"].items()}}
nv, vv = b["noun_vec"].cuda(), b["verb_vec"].cuda()


def part(lo, hi):
    return {"video": data["video"][lo:hi].contiguous(), "text": {k: v[lo:hi].contiguous() for k, v in data["text"].items()}}


def head(te, ve):
    loss = lossf.fused(te, ve, nv, vv)
    loss.backward()
    return loss


def single():
    for p in model.parameters():
        p.grad = None
    te, ve = model(data)
    return head(te, ve)


streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def dual(nsplit=2, offset_fn=None):
    for p in model.parameters():
        p.grad = None
    main = torch.cuda.current_stream()
    outs = []
    h = B // nsplit
    for i in range(nsplit):
        s = streams[i % 2]
        s.wait_stream(main)
        with torch.cuda.stream(s):
            outs.append(model(part(i * h, (i + 1) * h)))
    for s in streams:
        main.wait_stream(s)
    te = torch.cat([o[0] for o in outs])
    ve = torch.cat([o[1] for o in outs])
    return head(te, ve)


def timeit(fn, n=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        l = fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return 1e3 * (t2 - t0) / n, 1e3 * (t1 - t0) / n, float(l)


lib = _lib.lib()
lib.egv_gemm_set_grid(256)
t, h, l = timeit(single)
print(f"{prec} B={B} one stream, grid 256: {t:7.2f} ms fwd+bwd  (host enqueue {h:6.2f} ms)  loss {l:.5f}")
for grid in (128, 256):
    lib.egv_gemm_set_grid(grid)
    t, h, l = timeit(dual)
    print(f"{prec} B={B} two streams x B/2, grid {grid}: {t:7.2f} ms fwd+bwd  (host enqueue {h:6.2f} ms)  loss {l:.5f}")
lib.egv_gemm_set_grid(128)
t, h, l = timeit(lambda: dual(4))
print(f"{prec} B={B} two streams x 4 quarter batches, grid 128: {t:7.2f} ms fwd+bwd  (host enqueue {h:6.2f} ms)  loss {l:.5f}")
lib.egv_gemm_set_grid(256)
