import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egovlp_amd import ops
B, T, n, H = (int(os.environ.get(k, d)) for k, d in (("ATT_B", 32), ("ATT_T", 4), ("ATT_N", 196), ("ATT_H", 12)))
print("B=%d T=%d n=%d H=%d" % (B, T, n, H))
S = 1 + T * n
for passes in (3, 1):
    qkv = ops.split_f32(torch.randn(B * S, 3 * H * 64, device="cuda"), passes)[0]
    for mode, name in ((0, "space"), (1, "time")):
        f = lambda: ops.divided_attn_fwd(qkv, B, T, n, H, mode, passes)
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        print("passes=%d %s attention fwd: %.1f us" % (passes, name, e0.elapsed_time(e1) * 100))

# backward (single pass): dQ + dK/dV kernels
for mode, name in ((0, "space"), (1, "time")):
    qkv = ops.split_f32(torch.randn(B * S, 3 * H * 64, device="cuda"), 1)[0]
    out, lse = ops.divided_attn_fwd(qkv, B, T, n, H, mode, 1)
    d_out = ops.split_f32(torch.randn(B * S, H * 64, device="cuda"), 1)[0]
    f = lambda: ops.divided_attn_bwd(qkv, out, d_out, lse, B, T, n, H, mode, 1)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print("passes=1 %s attention bwd: %.1f us" % (name, e0.elapsed_time(e1) * 100))

# the fp16 attention of the fp16 backward mode (round 6): qkv planes = an fp16 split, three-product forward, single-product backward on fp16
x = torch.randn(B * S, 3 * H * 64, device="cuda")
hi = x.to(torch.float16)
qkv16 = ops.Planes(hi, (x - hi.float()).to(torch.float16), B * S, 3 * H * 64, "f16s")
for mode, name in ((0, "space"), (1, "time")):
    for fmt in ("f16x2", "f16"):
        f = lambda: ops.divided_attn_fwd(qkv16, B, T, n, H, mode, 3, out_fmt=fmt)
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        print("fp16 x3 %s attention fwd (output %s): %.1f us" % (name, fmt, e0.elapsed_time(e1) * 100))
    out, lse = ops.divided_attn_fwd(qkv16, B, T, n, H, mode, 3, out_fmt="f16")
    d_out = ops.f16_cast(torch.randn(B * S, H * 64, device="cuda"))
    f = lambda: ops.divided_attn_bwd(qkv16, out, d_out, lse, B, T, n, H, mode, 1, grad_f16=True)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print("fp16 x1 %s attention bwd: %.1f us" % (name, e0.elapsed_time(e1) * 100))
