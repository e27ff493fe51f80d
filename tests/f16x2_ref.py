"""Host restatement of the f16x2 operand format (egovlp_amd/csrc/f16x2.h, include/egovlp_hip.h)  --  TEST INFRASTRUCTURE ONLY.

`encode` follows the device arithmetic step by step (the same fp32 operations and the same two roundings per value), so its planes
are what the HIP encoders must produce BIT FOR BIT; `product` is the value the f16x2 GEMM computes from encoded operands (exact
products, fp64 accumulation).  The format has no counterpart in the reference (fp32 there): what is pinned against the reference is
the END of the pipeline (embeddings / loss at 1e-3, tests/test_gpu_model.py); these helpers pin the pieces.
"""
import torch

E = 2.0 ** -6


def _h(x):
    return x.to(torch.float16)


def encode(x: torch.Tensor, role: int):
    """fp32 [rows, cols] -> (plane 1, plane 2) float16, bf bfloat16.
    role 0 (first operand):  p1 = fp16((1 - e) x),  p2 = fp16(x - p1)
    role 1 (second operand): p1 = fp16(x),          p2 = fp16(p1 + (x - p1) / e)            e = 2^-6"""
    raw = x.float().contiguous()
    x = raw.clamp(-65504.0, 65504.0)                 # the fp16 planes saturate; the bf16 copy keeps the value
    if role == 0:
        p1 = _h(x - x * torch.tensor(E, dtype=torch.float32))
        p2 = _h(x - p1.float())
    else:
        p1 = _h(x)
        p2 = _h(p1.float() + (x - p1.float()) * torch.tensor(1.0 / E, dtype=torch.float32))
    return p1, p2, raw.to(torch.bfloat16)


def product(a1, a2, b1, b2) -> torch.Tensor:
    """What egv_gemm_nt(passes = 2) computes from encoded operands: A1 B1^T + A2 B2^T, in fp64."""
    return a1.double() @ b1.double().t() + a2.double() @ b2.double().t()
