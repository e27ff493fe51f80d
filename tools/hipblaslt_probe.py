"""What does the vendor library (torch.matmul -> hipBLASLt / rocBLAS) reach on the step's GEMM shapes, single-pass bf16?
A yardstick for gemm_big only -- nothing in the product calls it.  usage: python tools/hipblaslt_probe.py"""
import torch
shapes = [("qkv fwd", 25120, 2304, 768, "nt"), ("proj fwd", 25120, 768, 768, "nt"), ("fc1 fwd", 25120, 3072, 768, "nt"),
          ("fc2 fwd", 25120, 768, 3072, "nt"), ("qkv dgrad", 25120, 768, 2304, "nt"),
          ("qkv wgrad", 2304, 768, 25120, "tn"), ("proj wgrad", 768, 768, 25120, "tn"), ("fc1 wgrad", 3072, 768, 25120, "tn")]
for name, M, N, K, kind in shapes:
    for out_dtype in (torch.bfloat16,):
        if kind == "nt":
            a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
            b = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
            f = lambda: torch.matmul(a, b.t())
        else:
            a = torch.randn(K, M, device="cuda", dtype=torch.bfloat16)
            b = torch.randn(K, N, device="cuda", dtype=torch.bfloat16)
            f = lambda: torch.matmul(a.t(), b)
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print("%-11s M=%6d N=%5d K=%6d %s  bf16 out: %8.1f us  %7.1f TF" % (name, M, N, K, kind, us, 2.0 * M * N * K / us / 1e6))
