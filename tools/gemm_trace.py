"""Per-output-tile timeline of gemm_big (diagnostic): python tools/gemm_trace.py M N K [tn [ksplit]]
Each persistent workgroup stamps s_memrealtime (100 MHz) at tile start / first MFMA / end of main loop / end of epilogue
(EGV_GEMM_DBG=200: results are not stored)."""
import ctypes as C
import os
import sys

import torch

os.environ["EGV_GEMM_DBG"] = str(200 + int(os.environ.get("TRACE_DIAG", "0"), 0))
# the timeline switches only exist in the diagnostics build of the library (`make -C egovlp_amd/csrc diag`)
os.environ.setdefault("EGOVLP_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                     "egovlp_amd", "libegovlp_hip_diag.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egovlp_amd import _lib, ops  # noqa: E402

m, n, k = [int(x) for x in sys.argv[1:4]]
tn = len(sys.argv) > 4 and sys.argv[4] == "tn"
ksplit = int(sys.argv[5]) if len(sys.argv) > 5 else 1
def gen(r, c):   # TRACE_DATA=ones|zeros: operand bit patterns that barely toggle the multipliers (power / clock experiment)
    kind = os.environ.get("TRACE_DATA", "rand")
    if kind == "ones":
        return torch.ones(r, c, device="cuda")
    if kind == "zeros":
        return torch.zeros(r, c, device="cuda")
    return torch.rand(r, c, device="cuda") * 2 - 1


if tn:
    a = ops.split_f32(gen(k, m), 1)[0]
    b = ops.split_f32(gen(k, n), 1)[0]
else:
    a = ops.split_f32(gen(m, k), 1)[0]
    b = ops.split_f32(gen(n, k), 1)[0]
out = torch.empty(m, n, device="cuda")
partial = torch.empty(ksplit * (m * n + m), device="cuda") if ksplit > 1 else None
ts = torch.zeros(8192 * 4, dtype=torch.int64, device="cuda")
d = _lib.GemmDesc()
d.a_hi, d.lda, d.b_hi, d.ldb = a.hi.data_ptr(), a.ld, b.hi.data_ptr(), b.ld
d.M, d.N, d.K, d.passes, d.alpha, d.act = m, n, k, 1, 1.0, 0
d.out_f32, d.ldo = out.data_ptr(), n
d.ksplit, d.partial = ksplit, (partial.data_ptr() if partial is not None else None)
d.trans = int(tn)
d.aux_out = ts.data_ptr()
for it in range(3):
    ts.zero_()
    torch.cuda.synchronize()
    _lib.check(_lib.lib().egv_gemm_nt(C.byref(d), torch.cuda.current_stream().cuda_stream), "egv_gemm_nt")
    torch.cuda.synchronize()
t = ts.view(-1, 4).cpu()
t = t[t[:, 0] > 0].double() / 100.0   # us
t0 = t[:, 0].min()
t = t - t0
t = t[torch.argsort(t[:, 0])]
nblk = t.shape[0]
print(f"M={m} N={n} K={k} tn={tn} ksplit={ksplit}: {nblk} tiles, kernel span {t[:,3].max():.1f} us")
q = torch.tensor([0, .25, .5, .75, 1.0], dtype=torch.float64)
print("tile start (us): p0 %.1f p25 %.1f p50 %.1f p75 %.1f p100 %.1f" % tuple(t[:, 0].quantile(q).tolist()))
print("prologue  (us): mean %.2f max %.2f" % ((t[:, 1] - t[:, 0]).mean(), (t[:, 1] - t[:, 0]).max()))
print("main loop (us): mean %.2f min %.2f max %.2f" % ((t[:, 2] - t[:, 1]).mean(), (t[:, 2] - t[:, 1]).min(), (t[:, 2] - t[:, 1]).max()))
print("epilogue  (us): mean %.2f max %.2f" % ((t[:, 3] - t[:, 2]).mean(), (t[:, 3] - t[:, 2]).max()))
for i in list(range(0, min(nblk, 6))) + list(range(max(0, nblk - 4), nblk)):
    print("  tile#%4d start %7.2f  pro %6.2f  loop %7.2f  epi %6.2f  end %7.2f" % (i, t[i, 0], t[i, 1] - t[i, 0], t[i, 2] - t[i, 1], t[i, 3] - t[i, 2], t[i, 3]))

if int(os.environ.get("TRACE_DIAG", "0"), 0) & 0x4000:
    # hand-over stamps of workgroup 0's first tile: [wave][k-tile][arrive, waited, barrier passed, DMA issued] (s_memtime)
    st = ts[16384:16384 + 8 * 256].view(8, 64, 4).cpu().double()
    nkt = min(64, (m if tn else k) // 64 // max(ksplit, 1)) - 1
    base = st[:, 0, 0].min()
    print("hand-over stamps, workgroup 0, tile 0 (s_memtime ticks relative to first arrival at k-tile 0):")
    print("  kt | arrive: min max (who last) | wait vm/lgkm: max | barrier exit: min max | dma issue: max(loaders) | period")
    prev = None
    for t in range(nkt):
        a, w, b, d = st[:, t, 0] - base, st[:, t, 1] - base, st[:, t, 2] - base, st[:, t, 3] - base
        per = "" if prev is None else "%6.0f" % (b.min() - prev)
        prev = b.min()
        if t < 12 or t >= nkt - 3:
            print("  %2d | %7.0f %7.0f (w%d) | %5.0f | %7.0f %7.0f | %5.0f | %s" % (
                t, a.min(), a.max(), int(a.argmax()), (w - a).max(), b.min(), b.max(), (d - b).max(), per))
    for wv in range(8):
        print("   wave %d arrive-rel-to-first per k-tile 4..11:" % wv, " ".join("%5.0f" % (st[wv, t, 0] - st[:, t, 0].min()) for t in range(4, 12)),
              "| exit->next arrive:", " ".join("%5.0f" % (st[wv, t + 1, 0] - st[wv, t, 2]) for t in range(4, 12)))
