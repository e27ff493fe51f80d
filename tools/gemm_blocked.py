"""EXPERIMENT: K-blocked operand layout vs row-major for the v2 GEMM.  EGV_GEMM_KERNEL=2 EGV_BLOCKED=0|1"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egovlp_amd import ops
blocked = int(os.environ.get("EGV_BLOCKED", "0"))
M = 32 * 785
for passes in (3, 1):
    for name, m, n, k in (("qkv", M, 2304, 768), ("proj", M, 768, 768), ("fc1", M, 3072, 768), ("fc2", M, 768, 3072)):
        af = torch.rand(m, k, device="cuda") * 2 - 1
        bf = torch.rand(n, k, device="cuda") * 2 - 1
        a = ops.split_f32(af, passes)[0]
        b = ops.split_f32(bf, passes)[0]
        if blocked:
            def blk(pl):
                f = lambda t: None if t is None else t.view(pl.rows, pl.cols // 32, 32).permute(1, 0, 2).contiguous().view(pl.cols // 32 * pl.rows, 32)
                hi, lo = f(pl.hi), f(pl.lo)
                q = ops.Planes(hi, lo, pl.rows, pl.cols)
                return q
            class BP(ops.Planes):
                @property
                def ld(self):          # the blocked loader reads lda/ldb as the row count
                    return self.rows
            a2, b2 = blk(a), blk(b)
            a2 = BP(a2.hi, a2.lo, a2.rows, a2.cols); b2 = BP(b2.hi, b2.lo, b2.rows, b2.cols)
            a, b = a2, b2
        out = torch.empty(m, n, device="cuda")
        run = lambda: ops.gemm_nt(a, b, passes=passes, out_f32=out, K=k)
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        err = float((out.double() - af.double() @ bf.double().t()).norm() / (af.double() @ bf.double().t()).norm())
        print(f"blocked={blocked} passes={passes} {name}: {us:8.1f} us {2.0*m*n*k/us/1e6*passes:7.1f} TF issue  rel err {err:.1e}")
