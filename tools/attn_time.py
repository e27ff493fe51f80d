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
