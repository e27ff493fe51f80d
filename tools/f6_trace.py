"""Where a k-tile of the f16f6 main loop spends its time (diagnostic; needs `make -C egovlp_amd/csrc diag`):
per-wave s_memtime stamps of workgroup 0's first output tile + the per-tile timeline.   python tools/f6_trace.py M N K
stamps per k-tile: 0 top | 1 phase 0 starts (B_h(0), A_h landed) | 2 phase 1 (slots landed) | 3 phase 4 | 4 phases done | 5 own reads
returned | 6 own DMA landed | 7 barrier passed"""
import ctypes as C
import os
import sys

import torch

os.environ["EGV_GEMM_DBG"] = str(0x4000 + 200)
os.environ.setdefault("EGOVLP_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                     "egovlp_amd", "libegovlp_hip_diag.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egovlp_amd import _lib, ops  # noqa: E402

m, n, k = [int(x) for x in sys.argv[1:4]]
a = ops.f16f6_encode(torch.rand(m, k, device="cuda") * 2 - 1)
b = ops.f16f6_encode(torch.rand(n, k, device="cuda") * 2 - 1)
out = torch.empty(m, n, device="cuda")
ts = torch.zeros(8192 * 4, dtype=torch.int64, device="cuda")
d = _lib.GemmDesc()
d.a_hi, d.a_lo, d.lda, d.b_hi, d.b_lo, d.ldb = a.hi.data_ptr(), a.lo.data_ptr(), a.ld, b.hi.data_ptr(), b.lo.data_ptr(), b.ld
d.M, d.N, d.K, d.passes, d.alpha, d.act = m, n, k, 2, 1.0, 0
d.out_f32, d.ldo = out.data_ptr(), n
d.ksplit = 1
d.aux_out = ts.data_ptr()
for it in range(3):
    ts.zero_()
    torch.cuda.synchronize()
    _lib.check(_lib.lib().egv_gemm_nt(C.byref(d), torch.cuda.current_stream().cuda_stream), "egv_gemm_nt")
    torch.cuda.synchronize()
t = ts.view(-1, 4)[:4096].cpu()
t = t[t[:, 0] > 0].double() / 100.0
t = t - t[:, 0].min()
t = t[torch.argsort(t[:, 0])]
print(f"M={m} N={n} K={k} f16f6: {t.shape[0]} tiles, kernel span {t[:,3].max():.1f} us; main loop mean {(t[:,2]-t[:,1]).mean():.2f} us "
      f"({(t[:,2]-t[:,1]).mean() / (k // 32):.3f} us per k-tile), epilogue mean {(t[:,3]-t[:,2]).mean():.2f} us, prologue {(t[:,1]-t[:,0]).mean():.2f}")
st = ts[16384:16384 + 8 * 256].view(8, 32, 8).cpu().double()
nkt = min(32, k // 32) - 1
names = ["top", "ph0", "ph1", "ph4", "done", "rd_ret", "dma_in", "barrier"]
print("per wave, mean over k-tiles 2..%d of the interval ENDING at each stamp (s_memtime ticks), and the k-tile period:" % (nkt - 1))
print("  wave | " + " ".join("%8s" % x for x in names[1:]) + " |  next top | period")
for wv in range(8):
    iv = []
    for i in range(1, 8):
        iv.append(float((st[wv, 2:nkt, i] - st[wv, 2:nkt, i - 1]).mean()))
    nxt = float((st[wv, 3:nkt + 1, 0] - st[wv, 2:nkt, 7]).mean())
    per = float((st[wv, 3:nkt + 1, 0] - st[wv, 2:nkt, 0]).mean())
    print("  %4d | " % wv + " ".join("%8.0f" % x for x in iv) + " | %9.0f | %6.0f" % (nxt, per))
