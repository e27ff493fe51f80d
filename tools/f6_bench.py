"""Isolated timing of the video blocks' forward GEMMs in the f16f6 form (passes = 2) next to the fused bf16x3 form (passes = 3),
with the epilogues the step gives them.  usage: [EGOVLP_HIP_LIB=...] python tools/f6_bench.py [tokens]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egovlp_amd import ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32 * 785
D, H3, HD = (int(os.environ.get("F6_D", 768)),) * 1 + (0, 0)
H3, HD = 3 * D, 4 * D
dev = "cuda"
torch.manual_seed(0)


def operand(r, c, P):
    x = torch.rand(r, c, device=dev) * 2 - 1
    return ops.f16f6_encode(x) if P == 2 else ops.split_f32(x, P)[0]


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tot = {}
for P in (3, 2):
    res = torch.rand(M, D, device=dev)
    o32 = torch.empty(M, D, device=dev)
    bias = {n: torch.zeros(n, device=dev) for n in (D, H3, HD)}
    qkv = ops.empty_planes(M, H3, 3, dev)
    h = ops.empty_planes_f16f6(M, HD, dev, want_bf=True) if P == 2 else ops.empty_planes(M, HD, 3, dev)
    z = torch.empty(M, HD, device=dev, dtype=torch.bfloat16)
    xs = {k: operand(M, k, P) for k in (D, HD)}
    ws = {(n, k): operand(n, k, P) for n, k in ((H3, D), (D, D), (HD, D), (D, HD))}
    cases = [
        ("qkv  fwd", H3, D, lambda: ops.gemm_nt(xs[D], ws[(H3, D)], passes=P, bias=bias[H3], out_planes=qkv)),
        ("proj fwd", D, D, lambda: ops.gemm_nt(xs[D], ws[(D, D)], passes=P, bias=bias[D], residual=res, out_f32=o32)),
        ("fc1  fwd", HD, D, lambda: ops.gemm_nt(xs[D], ws[(HD, D)], passes=P, bias=bias[HD], act=ops.ACT_GELU, aux_out=z, out_planes=h,
                                                aux_is_grad=True)),
        ("fc2  fwd", D, HD, lambda: ops.gemm_nt(xs[HD], ws[(D, HD)], passes=P, bias=bias[D], residual=res, out_f32=o32)),
    ]
    t_all = 0.0
    for name, n, k, fn in cases:
        us = timed(fn)
        mult = 2 if name.startswith("qkv") or name.startswith("proj") else 1
        t_all += us * mult
        print("passes %d  %s  M=%d N=%4d K=%4d  %7.1f us  %6.1f TF algorithmic" % (P, name, M, n, k, us, 2.0 * M * n * k / us * 1e-6))
    tot[P] = t_all
    print("passes %d  forward Linears of one block (2 qkv + 2 proj + fc1 + fc2): %.1f us" % (P, t_all))
print("lib %s: f16f6 block / bf16x3 block = %.3f" % (os.environ.get("EGOVLP_HIP_LIB", "default"), tot[2] / tot[3]))
