"""What bounds the fc1 forward's epilogue?  (round-5 verdict: "0.31 MFMA busy: its epilogue writes three planes").  The one-product fc1
GEMM of the fp16 backward mode (h as one fp16 plane + gelu' as fp16: TWO planes now) timed (a) as shipped, (b) with every global store
of the epilogue sunk into registers -- loads, LDS traffic, bias, erf-GELU, gelu', conversions all still run -- (EGV_GEMM_DBG=90), (c)
with the epilogue skipped altogether (EGV_GEMM_DBG=100: main loop only), next to the qkv forward (plain bias epilogue, two fp16 planes).
Diagnostics library: make -C egovlp_amd/csrc diag; EGOVLP_HIP_LIB=egovlp_amd/libegovlp_hip_diag.so EGV_GEMM_DBG=<n> python tools/fc1_epilogue_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egovlp_amd import ops  # noqa: E402

M, D, H3, HD = 32 * 785, 768, 2304, 3072
dev = "cuda"
a = ops.Planes((torch.rand(M, D, device=dev) * 2 - 1).to(torch.float16), None, M, D, "f16")
w1, wq = ops.f16x2_encode(torch.rand(HD, D, device=dev) * 0.1 - 0.05, 1), ops.f16x2_encode(torch.rand(H3, D, device=dev) * 0.1 - 0.05, 1)
h = ops.empty_planes_f16x2(M, HD, dev, single=True)
z = torch.empty(M, HD, dtype=torch.float16, device=dev)
qkv = ops.empty_planes_f16x2(M, H3, dev, split=True)
b1, bq = torch.zeros(HD, device=dev), torch.zeros(H3, device=dev)
cases = [("fc1 fwd (GELU, h + gelu' as fp16)", M, HD, D, lambda: ops.gemm_nt(a, w1, passes=4, bias=b1, act=ops.ACT_GELU, aux_out=z, out_planes=h, aux_is_grad=True)),
         ("qkv fwd (bias, fp16 split planes)", M, H3, D, lambda: ops.gemm_nt(a, wq, passes=4, bias=bq, out_planes=qkv))]
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
    print("EGV_GEMM_DBG=%-4s %-36s %7.1f us  %6.1f TF" % (os.environ.get("EGV_GEMM_DBG", "0"), name, us, 2.0 * m * n * k / us / 1e6))
