"""Host restatement of the f16f6 operand format (egovlp_amd/csrc/f6.h, include/egovlp_hip.h)  --  TEST INFRASTRUCTURE ONLY.

`encode` follows the device arithmetic step by step (same fp32 operations in the same order), so its bytes are what the HIP
encoders must produce BIT FOR BIT; `decode` turns planes back into numbers; `product` is the value the f16f6 GEMM computes from
encoded operands (exact products, fp64 accumulation).  The format has no counterpart in the reference (fp32 there): what is pinned
against the reference is the END of the pipeline (embeddings / loss at 1e-3, tests/test_gpu_model.py); these helpers pin the pieces.
"""
import numpy as np
import torch

E2M3 = np.array([(m / 8.0 if e == 0 else (1.0 + m / 8.0) * 2.0 ** (e - 1)) * (-1.0 if s else 1.0)
                 for s in (0, 1) for e in range(4) for m in range(8)], dtype=np.float64)          # value of code s e e m m m


def _scale_byte(amax: torch.Tensor) -> torch.Tensor:
    """E8M0 exponent byte of a block: (bits(amax * fl(1 / 7.5)) + 0x7fffff) >> 23, clamped to 247 (f6_scale_byte)."""
    t = (amax.float() * torch.tensor(0.13333334, dtype=torch.float32)).contiguous()
    eb = (t.view(torch.int32).to(torch.int64) + 0x7FFFFF) >> 23
    return eb.clamp(max=247)


def _codes(y: torch.Tensor) -> torch.Tensor:
    """fp32 (already multiplied by 2^-(scale - 127) 2^-6) -> 6-bit E2M3 codes through the E4M3 grid (f6_codes8)."""
    b = y.to(torch.float8_e4m3fn).view(torch.uint8).to(torch.int64)
    return (b & 0x1F) | ((b >> 2) & 0x20)


def _pack(codes: torch.Tensor) -> torch.Tensor:
    """[..., 32] codes -> [..., 24] bytes, element i at bits [6 i, 6 i + 6) little-endian."""
    c = codes.to(torch.int64)
    out = torch.zeros(c.shape[:-1] + (24,), dtype=torch.int64)
    for i in range(32):
        bit = 6 * i
        byte, off = bit // 8, bit % 8
        out[..., byte] |= (c[..., i] << off) & 0xFF
        if off > 2:
            out[..., byte + 1] |= c[..., i] >> (8 - off)
    return out.to(torch.uint8)


def encode(x: torch.Tensor):
    """fp32 [rows, cols] (cols % 32 == 0) -> (h16 float16 [rows, cols], bf bfloat16 [rows, cols],
    slots uint8 [rows, cols // 32, 2, 32]: [.., 0, :] the c6 slot, [.., 1, :] the l6 slot; bytes 0..23 codes, byte 24 scale, rest 0)."""
    x = x.float().contiguous()
    rows, cols = x.shape
    h16 = x.clamp(-65504.0, 65504.0).to(torch.float16)
    bf = x.to(torch.bfloat16)
    r = x - h16.float()
    slots = torch.zeros(rows, cols // 32, 2, 32, dtype=torch.uint8)
    for k, v in enumerate((x, r)):
        blk = v.view(rows, cols // 32, 32)
        eb = _scale_byte(blk.abs().amax(dim=-1))
        pre = ((248 - eb) << 23).to(torch.int32).view(torch.float32)
        slots[:, :, k, :24] = _pack(_codes(blk * pre.unsqueeze(-1)))
        slots[:, :, k, 24] = eb.to(torch.uint8)
    return h16, bf, slots


def unpack_slots(raw: torch.Tensor, rows: int, cols: int) -> torch.Tensor:
    """The slot plane as the device stores it (int16 / uint8 tensor of rows * cols * 2 bytes; per block four 16-byte chunks:
    c6 bytes 0-15 | l6 bytes 0-15 | c6 bytes 16-23, scale, pad | l6 bytes 16-23, scale, pad) -> uint8 [rows, cols // 32, 2, 32]
    with [.., 0, :] the c6 slot and [.., 1, :] the l6 slot (24 code bytes, scale byte, pad)."""
    ch = raw.contiguous().view(torch.uint8).reshape(rows, cols // 32, 4, 16)
    return torch.stack([torch.cat([ch[:, :, 0], ch[:, :, 2]], dim=-1), torch.cat([ch[:, :, 1], ch[:, :, 3]], dim=-1)], dim=2)


def decode_slots(slots: torch.Tensor):
    """uint8 [rows, nblk, 2, 32] -> (c6 values, l6 values) float64 [rows, nblk * 32]."""
    s = slots.to(torch.int64)
    rows, nblk = s.shape[:2]
    out = []
    for k in range(2):
        data = s[:, :, k, :24]
        codes = torch.zeros(rows, nblk, 32, dtype=torch.int64)
        for i in range(32):
            bit = 6 * i
            byte, off = bit // 8, bit % 8
            v = data[..., byte] >> off
            if off > 2:
                v = v | (data[..., byte + 1] << (8 - off))
            codes[..., i] = v & 0x3F
        vals = torch.from_numpy(E2M3)[codes] * torch.pow(2.0, (s[:, :, k, 24] - 127).double()).unsqueeze(-1)
        out.append(vals.reshape(rows, nblk * 32))
    return out[0], out[1]


def product(a_h16, a_slots, b_h16, b_slots) -> torch.Tensor:
    """What egv_gemm_nt(passes = 2) computes from encoded operands: A_h B_h^T + c6(A) l6(B)^T + l6(A) c6(B)^T, in fp64."""
    ac, al = decode_slots(a_slots)
    bc, bl = decode_slots(b_slots)
    ah, bh = a_h16.double(), b_h16.double()
    return ah @ bh.t() + ac @ bl.t() + al @ bc.t()
