"""Two-product f16x2 GEMMs of the step: the fused two-product loop (EGV_X2_SEG=0) vs two k-segments of the plain fp16 loop (=2: every
shape; =1: K, N <= 768 only = the proj Linears).  Forward epilogues as in the step.  python tools/x2seg_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egovlp_amd import ops  # noqa: E402

M, D, H3, HD = 32 * 785, 768, 2304, 3072
dev = "cuda"


def enc(r, c, role):
    return ops.f16x2_encode(torch.rand(r, c, device=dev) * 2 - 1, role)


res, o32 = torch.rand(M, D, device=dev), torch.empty(M, D, device=dev)
cases = []
a, b = enc(M, D, 0), enc(D, D, 1)
bias = torch.zeros(D, device=dev)
cases.append(("proj fwd  N=768  K=768 ", M, D, D, lambda a=a, b=b: ops.gemm_nt(a, b, passes=2, bias=bias, residual=res, out_f32=o32)))
a2, b2 = enc(M, D, 0), enc(H3, D, 1)
qkv = ops.empty_planes_f16x2(M, H3, dev, split=True)
bq = torch.zeros(H3, device=dev)
cases.append(("qkv fwd   N=2304 K=768 ", M, H3, D, lambda: ops.gemm_nt(a2, b2, passes=2, bias=bq, out_planes=qkv)))
b3 = enc(HD, D, 1)
h = ops.empty_planes_f16x2(M, HD, dev)
z = torch.empty(M, HD, dtype=torch.float16, device=dev)
b1 = torch.zeros(HD, device=dev)
cases.append(("fc1 fwd   N=3072 K=768 ", M, HD, D, lambda: ops.gemm_nt(a2, b3, passes=2, bias=b1, act=ops.ACT_GELU, aux_out=z, out_planes=h, aux_is_grad=True)))
a4, b4 = enc(M, HD, 0), enc(D, HD, 1)
cases.append(("fc2 fwd   N=768  K=3072", M, D, HD, lambda: ops.gemm_nt(a4, b4, passes=2, bias=bias, residual=res, out_f32=o32)))
for name, m, n, k, run in cases:
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("EGV_X2_SEG=%s %s: %7.1f us  %6.1f TF algorithmic" % (os.environ.get("EGV_X2_SEG", "(default 1)"), name, us, 2.0 * m * n * k / us / 1e6))
