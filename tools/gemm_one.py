"""Run one GEMM shape N times (for rocprofv3 PMC passes): python tools/gemm_one.py passes M N K [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egovlp_amd import ops
passes, m, n, k = [int(x) for x in sys.argv[1:5]]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
a = ops.split_f32(torch.rand(m, k, device="cuda") * 2 - 1, passes)[0]
b = ops.split_f32(torch.rand(n, k, device="cuda") * 2 - 1, passes)[0]
out = torch.empty(m, n, device="cuda")
for _ in range(iters):
    ops.gemm_nt(a, b, passes=passes, out_f32=out)
torch.cuda.synchronize()
