"""Sustained MFMA rate of the chip (no memory traffic): python tools/mfma_peak.py
256 workgroups (one per CU) x `waves` waves x iters x 40 MFMA 16x16x32 bf16; reports TFLOP/s for 4 and 8 waves per CU."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egovlp_amd import _lib  # noqa: E402

out = torch.zeros(512, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for waves in (4, 8):
    for iters in (2000, 20000):
        _lib.check(_lib.lib().egv_diag_mfma_peak(200, waves, out.data_ptr(), st), "warm")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(_lib.lib().egv_diag_mfma_peak(iters, waves, out.data_ptr(), st), "egv_diag_mfma_peak")
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        flop = 256.0 * waves * iters * 40 * 2 * 16 * 16 * 32
        cyc16 = iters * 40 * 16 * (waves / 4.0)   # cycles per SIMD if one MFMA issues every 16 cycles
        print(f"waves/CU={waves} iters={iters}: {ms*1e3:9.1f} us  {flop/ms/1e9:8.1f} TFLOP/s  "
              f"(implied clock if 16 cycles/MFMA: {cyc16/ms/1e6:.2f} GHz)")

for mode, name in ((100, "register-operand pattern (5 A x 2 B fragments)"), (200, "same + LDS fragment re-reads (13 ds_read_b128 / 40 MFMA)")):
    for waves in (4, 8):
        iters = 2000
        _lib.check(_lib.lib().egv_diag_mfma_peak(50, mode + waves, out.data_ptr(), st), "warm")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(_lib.lib().egv_diag_mfma_peak(iters, mode + waves, out.data_ptr(), st), "egv_diag_mfma_peak")
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        flop = 256.0 * waves * iters * 120 * 2 * 16 * 16 * 32
        print(f"{name}: waves/CU={waves}: {ms*1e3:9.1f} us  {flop/ms/1e9:8.1f} TFLOP/s")

for waves in (4, 8):
    iters = 20000
    _lib.check(_lib.lib().egv_diag_mfma_peak(-200, waves, out.data_ptr(), st), "warm")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(_lib.lib().egv_diag_mfma_peak(-iters, waves, out.data_ptr(), st), "egv_diag_mfma_peak")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    flop = 256.0 * waves * iters * 40 * 2 * 16 * 16 * 32
    print(f"random operands, waves/CU={waves}: {ms*1e3:9.1f} us  {flop/ms/1e9:8.1f} TFLOP/s")
