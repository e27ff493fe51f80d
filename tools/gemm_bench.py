"""Micro-benchmark of egv_gemm_nt on the hot-path shapes (uniform random operands, HIP-event timed).
usage: [EGV_GEMM_KERNEL=1|2|3|4|5|14] python tools/gemm_bench.py [passes]     (wgrad rows use the TN kernel)"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egovlp_amd import ops  # noqa: E402

passes_list = [int(sys.argv[1])] if len(sys.argv) > 1 else [3, 1]
M = 32 * 785
# (name, M, N, K, kind): kind 1 = NT with bias epilogue, 0 = wgrad (TN, contraction over the M token rows)
shapes = [("qkv   fwd", M, 2304, 768, 1), ("proj  fwd", M, 768, 768, 1), ("fc1   fwd", M, 3072, 768, 1),
          ("fc2   fwd", M, 768, 3072, 1), ("qkv dgrad", M, 768, 2304, 1), ("qkv wgrad", 2304, 768, M, 0),
          ("proj wgrad", 768, 768, M, 0), ("fc1 wgrad", 3072, 768, M, 0), ("fc2 wgrad", 768, 3072, M, 0),
          ("text  lin", 1024, 768, 768, 1), ("text wgrad", 768, 768, 1024, 0)]
for passes in passes_list:
    tot_t = tot_f = 0.0
    for name, m, n, k, kind in shapes:
        if kind:
            a = ops.split_f32(torch.rand(m, k, device="cuda") * 2 - 1, passes)[0]
            b = ops.split_f32(torch.rand(n, k, device="cuda") * 2 - 1, passes)[0]
            out = torch.empty(m, n, device="cuda")
            bias = torch.zeros(n, device="cuda")
            if os.environ.get("BENCH_OUT") == "bf16":
                outp = ops.empty_planes(m, n, 1, "cuda")
                run = lambda: ops.gemm_nt(a, b, passes=passes, out_planes=outp, bias=bias)
            else:
                run = lambda: ops.gemm_nt(a, b, passes=passes, out_f32=out, bias=bias)
        else:
            a = ops.split_f32(torch.rand(k, m, device="cuda") * 2 - 1, passes)[0]
            b = ops.split_f32(torch.rand(k, n, device="cuda") * 2 - 1, passes)[0]
            out = torch.empty(m, n, device="cuda")
            run = lambda: ops.gemm_tn(a, b, passes=passes, out_f32=out, want_colsum=True)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        tf = 2.0 * m * n * k / us / 1e6
        tot_t += us; tot_f += 2.0 * m * n * k
        print(f"passes={passes} {name:10s} M={m:6d} N={n:5d} K={k:6d}: {us:8.1f} us  {tf:7.1f} TF algorithmic  ({tf*passes:7.1f} TF MFMA issue)")
    print(f"passes={passes} TOTAL {tot_f/tot_t/1e6:.1f} TF algorithmic, variant={os.environ.get('EGV_GEMM_KERNEL','auto')}")
