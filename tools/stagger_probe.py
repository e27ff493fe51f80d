"""Probe: does de-phasing the epilogue store bursts of the persistent GEMM help?  The qkv-forward GEMM (M x 2304 x 768,
plane outputs) is run (a) as one launch on 256 workgroups and (b) as two launches of half the rows each on two streams, 128
persistent workgroups each, the second delayed by a fraction of a tile period (torch.cuda._sleep on its stream), so that
the two halves of the chip store at different times.  usage: python tools/stagger_probe.py [passes]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egovlp_amd import _lib, ops  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 3
M, N, K = 32 * 785, 2304, 768
dev = "cuda"
a = ops.split_f32(torch.rand(M, K, device=dev) * 2 - 1, P)[0]
b = ops.split_f32(torch.rand(N, K, device=dev) * 2 - 1, P)[0]
bias = torch.zeros(N, device=dev)
out = ops.empty_planes(M, N, P, dev)
h = M // 2 // 320 * 320                      # split on a tile boundary
parts = []
for r0, r1 in ((0, h), (h, M)):
    pa = ops.Planes(a.hi[r0:r1], None if a.lo is None else a.lo[r0:r1], r1 - r0, K)
    po = ops.Planes(out.hi[r0:r1], None if out.lo is None else out.lo[r0:r1], r1 - r0, N)
    parts.append((pa, po))
lib = _lib.lib()


def full():
    ops.gemm_nt(a, b, passes=P, bias=bias, out_planes=out)


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def halves(delay_cycles):
    cur = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(cur)
    for i, (st, (pa, po)) in enumerate(zip((s1, s2), parts)):
        st.wait_event(ev)
        with torch.cuda.stream(st):
            if i == 1 and delay_cycles:
                torch.cuda._sleep(delay_cycles)
            ops.gemm_nt(pa, b, passes=P, bias=bias, out_planes=po)
    for st in (s1, s2):
        e = torch.cuda.Event()
        e.record(st)
        cur.wait_event(e)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


lib.egv_gemm_set_grid(256)
print(f"passes={P} full launch, 256 workgroups: {timeit(full):8.1f} us")
lib.egv_gemm_set_grid(128)
print(f"passes={P} full launch, 128 workgroups: {timeit(full):8.1f} us")
for d in (0, 20000, 40000, 80000, 120000):
    print(f"passes={P} two half launches on two streams, 128 workgroups each, second delayed {d:6d} cycles: {timeit(lambda: halves(d)):8.1f} us")
lib.egv_gemm_set_grid(256)
