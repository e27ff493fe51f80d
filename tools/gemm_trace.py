"""Per-block timeline of gemm_big (diagnostic): EGV_GEMM_DBG=200 python tools/gemm_trace.py M N K"""
import os, sys, torch
os.environ["EGV_GEMM_DBG"] = "200"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egovlp_amd import ops
m, n, k = [int(x) for x in sys.argv[1:4]]
a = ops.split_f32(torch.rand(m, k, device="cuda") * 2 - 1, 1)[0]
b = ops.split_f32(torch.rand(n, k, device="cuda") * 2 - 1, 1)[0]
out = torch.empty(m, n, device="cuda")
nb = 4096
ts = torch.zeros(nb * 4, dtype=torch.int64, device="cuda")
for it in range(3):
    ts.zero_()
    torch.cuda.synchronize()
    ops.gemm_nt(a, b, passes=1, out_f32=out, aux_out=ts.view(torch.float32))
    torch.cuda.synchronize()
t = ts.view(-1, 4).cpu()
t = t[t[:, 0] > 0].double() / 100.0   # us
t0 = t[:, 0].min()
t = t - t0
order = torch.argsort(t[:, 0])
t = t[order]
nblk = t.shape[0]
print(f"M={m} N={n} K={k}: {nblk} blocks, kernel span {t[:,3].max():.1f} us")
print("start times (us): p0 %.1f p25 %.1f p50 %.1f p75 %.1f p100 %.1f" % tuple(t[:, 0].quantile(torch.tensor([0, .25, .5, .75, 1.0], dtype=torch.float64)).tolist()))
print("prologue  (us): mean %.2f max %.2f" % ((t[:, 1] - t[:, 0]).mean(), (t[:, 1] - t[:, 0]).max()))
print("main loop (us): mean %.2f min %.2f max %.2f" % ((t[:, 2] - t[:, 1]).mean(), (t[:, 2] - t[:, 1]).min(), (t[:, 2] - t[:, 1]).max()))
print("epilogue  (us): mean %.2f max %.2f" % ((t[:, 3] - t[:, 2]).mean(), (t[:, 3] - t[:, 2]).max()))
for i in list(range(0, min(nblk, 12))) + list(range(max(0, nblk - 6), nblk)):
    print("  blk#%4d start %7.2f  pro %6.2f  loop %7.2f  epi %6.2f  end %7.2f" % (i, t[i, 0], t[i, 1] - t[i, 0], t[i, 2] - t[i, 1], t[i, 3] - t[i, 2], t[i, 3]))
