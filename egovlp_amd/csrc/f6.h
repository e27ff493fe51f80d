// "f16f6" operand format of the forward GEMMs (gfx950 only): x ~= h + l with
//   h  = fp16(x)                       (11 significant bits, saturating at +-65504)           -> MFMA 16x16x32 f16
//   c6 = MXFP6-E2M3(x),  l6 = MXFP6-E2M3(x - h)   one E8M0 scale per 32 consecutive k-elements -> v_mfma_scale_f32_16x16x128_f8f6f4
// and the product  A . B^T ~= A_h B_h + c6(A) l6(B) + l6(A) c6(B)   (error ~2^-12.3 x 2^-5.3 per operand instead of bf16x3's 2^-17 --
// measured on the CPU oracle: embeddings 1.3e-4 of the 1e-3 bar on ViT-L, tests/precision_table.py, profiles/r04_precision_table.txt).
//
// HBM / LDS layout.  An operand [rows, K] (K % 32 == 0) is TWO planes of 2 bytes per element each -- the same sizes and leading
// dimensions as the split-bf16 (hi, lo) pair it replaces, so the LDS-DMA staging of gemm_big is unchanged:
//   plane 0 ("hi")   : fp16 [rows, ld]
//   plane 1 ("slots"): per row and per 32-element block b, 64 bytes at byte offset (row * ld + 32 b) * 2, four 16-byte chunks:
//        chunk 0: c6 codes bytes 0-15 | chunk 1: l6 codes bytes 0-15 | chunk 2: c6 codes bytes 16-23, c6 scale byte, 7 B pad |
//        chunk 3: l6 codes bytes 16-23, l6 scale byte, 7 B pad
//     codes: element i of the block at bits [6 i, 6 i + 6) little-endian of the 24 bytes (the register image the MFMA reads);
//     scale byte e (E8M0): element value = code value x 2^(e - 127), e = ceil(log2(amax / 7.5)) + 127 (no saturation).
// A lane of the MFMA (row = lane & 15, k-group g = lane >> 4) fetches ONE slot (c6 or l6) with two ds_read_b128 -- chunks (0, 2) or
// (1, 3): k-group g reading chunk g first is the access pattern the 64-byte-row LDS image of gemm_big is conflict-free for --
// dwords 0-5 of the 8 are the operand, byte 0 of dword 6 its scale (op_sel 0).  Per 32-deep k-tile only groups 0 and 1 carry data
// (c6.l6 and l6.c6); lanes 32-63 are switched off through their scale.
//
// E2M3 codes come from the hardware fp32 -> E4M3 converter: the four lowest binades of OCP E4M3 (subnormals and exponents 1-3:
// spacings 2^-9, 2^-9, 2^-8, 2^-7 from 0 to 7.5 x 2^-6) are E2M3's grid (spacings 1/8, 1/8, 1/4, 1/2 from 0 to 7.5) scaled by 2^-6, so
// code6 = (e4m3(y / 64) & 0x1f) | sign << 5 with the converter's own round-to-nearest-even (checked exhaustively on hardware:
// tools/mx_probe.hip, profiles/r04_mx_probe.txt).
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(3))) unsigned int u32x3_t;

constexpr int EGV_DPP_QUAD_SWAP1 = 0xB1;   // quad_perm [1, 0, 3, 2]
constexpr int EGV_DPP_QUAD_SWAP2 = 0x4E;   // quad_perm [2, 3, 0, 1]

__device__ __forceinline__ float f6_quad_max(float v) {
  v = fmaxf(v, __uint_as_float(__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), EGV_DPP_QUAD_SWAP1, 0xf, 0xf, true)));
  v = fmaxf(v, __uint_as_float(__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), EGV_DPP_QUAD_SWAP2, 0xf, 0xf, true)));
  return v;
}

// E8M0 exponent byte of a block whose largest magnitude is amax: the smallest power of two s with amax <= 7.5 s.
// (amax * fl(1 / 7.5) in fp32: at an exact boundary the product may round up one ulp and the block gets the next larger scale --
// one bit of its codes unused, never a saturated code; the host reference in tests/f16f6_ref.py follows the same arithmetic.)
__device__ __forceinline__ unsigned f6_scale_byte(float amax) {
  const unsigned eb = (__float_as_uint(amax * 0.13333334f) + 0x7fffffu) >> 23;
  return eb > 247u ? 247u : eb;
}
// 2^-(eb - 127) * 2^-6: what a value is multiplied by before the E4M3 converter
__device__ __forceinline__ float f6_prescale(unsigned eb) { return __uint_as_float((248u - eb) << 23); }

// 8 values -> 48 bits of E2M3 codes (element 0 in the low bits): d0 = bits 0..31, d1 = bits 32..47
__device__ __forceinline__ void f6_codes8(const float (&y)[8], float pre, unsigned& d0, unsigned& d1) {
  int w0 = 0, w1 = 0;
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(y[0] * pre, y[1] * pre, w0, false);
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(y[2] * pre, y[3] * pre, w0, true);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(y[4] * pre, y[5] * pre, w1, false);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(y[6] * pre, y[7] * pre, w1, true);
  auto pack4 = [](unsigned u) {                                    // four E4M3 bytes -> 24 bits of codes
    u = (u & 0x1f1f1f1fu) | ((u >> 2) & 0x20202020u);              // s 0 0 e e m m m -> s e e m m m
    const unsigned t = (u & 0x003f003fu) | ((u >> 2) & 0x0fc00fc0u);
    return (t & 0xfffu) | ((t >> 4) & 0xfff000u);
  };
  const unsigned r0 = pack4((unsigned)w0), r1 = pack4((unsigned)w1);
  d0 = r0 | (r1 << 24);
  d1 = r1 >> 8;
}

// What one lane contributes to the f16f6 image of 8 consecutive values v[0..7] whose 32-element block is shared with the other
// lanes of its quad (lane & 3 = position of the 8 values inside the block; ALL four lanes must call this together).
//   h16  : 8 x fp16 (16 B at element offset of v[0] in plane 0)
//   bf   : 8 x bf16 of v (the single-pass operand of the backward GEMMs), 16 B
//   piece: this lane's share of the 64-byte slot pair -- even lanes hold c6 codes, odd lanes l6 codes, of THEIR lane pair
//          (16 elements = 96 bits = dwords 0..2); dword 3 = the scale byte.  f6_store_piece puts it where the layout above wants it:
//          lanes 0, 1 of a quad own code bytes 0-11 (chunk 0 / 1), lanes 2, 3 code bytes 12-15 (end of chunk 0 / 1) and 16-23 +
//          the scale (start of chunk 2 / 3).
struct F6Lane {
  u32x4_t h16, bf, piece;
};

__device__ __forceinline__ F6Lane f6_encode8(const float (&v)[8], int lane) {
  F6Lane o;
  float r[8];
  float am_c = 0.f, am_l = 0.f;
  unsigned hw[4], bw[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float a = __builtin_amdgcn_fmed3f(v[2 * e], -65504.f, 65504.f), b = __builtin_amdgcn_fmed3f(v[2 * e + 1], -65504.f, 65504.f);
    const f16x2_t h = __builtin_convertvector((f32x2_t){a, b}, f16x2_t);
    hw[e] = __builtin_bit_cast(unsigned, h);
    bw[e] = f32x2_to_bf16x2(v[2 * e], v[2 * e + 1]);
    r[2 * e] = v[2 * e] - (float)h[0];
    r[2 * e + 1] = v[2 * e + 1] - (float)h[1];
    am_c = fmaxf(am_c, fmaxf(fabsf(v[2 * e]), fabsf(v[2 * e + 1])));
    am_l = fmaxf(am_l, fmaxf(fabsf(r[2 * e]), fabsf(r[2 * e + 1])));
  }
  o.h16 = (u32x4_t){hw[0], hw[1], hw[2], hw[3]};
  o.bf = (u32x4_t){bw[0], bw[1], bw[2], bw[3]};
  const unsigned eb_c = f6_scale_byte(f6_quad_max(am_c)), eb_l = f6_scale_byte(f6_quad_max(am_l));
  unsigned c0, c1, l0, l1;
  f6_codes8(v, f6_prescale(eb_c), c0, c1);
  f6_codes8(r, f6_prescale(eb_l), l0, l1);
  // lane pairs trade halves: the even lane keeps both c6 halves of the pair, the odd lane both l6 halves
  const bool odd = lane & 1;
  const unsigned g0 = odd ? c0 : l0, g1 = odd ? c1 : l1;          // given away
  const unsigned k0 = odd ? l0 : c0, k1 = odd ? l1 : c1;          // kept
  const unsigned p0 = (unsigned)__builtin_amdgcn_mov_dpp((int)g0, EGV_DPP_QUAD_SWAP1, 0xf, 0xf, true);
  const unsigned p1 = (unsigned)__builtin_amdgcn_mov_dpp((int)g1, EGV_DPP_QUAD_SWAP1, 0xf, 0xf, true);
  const unsigned f0 = odd ? p0 : k0, f1 = odd ? p1 : k1;          // first 48 bits: the even lane's elements
  const unsigned s0 = odd ? k0 : p0, s1 = odd ? k1 : p1;          // second 48 bits: the odd lane's
  o.piece = (u32x4_t){f0, (f1 & 0xffffu) | (s0 << 16), (s0 >> 16) | (s1 << 16), odd ? eb_l : eb_c};
  return o;
}

template <int SITE>
__device__ __forceinline__ void f6_store_piece(char* slot_pair, int lane, const u32x4_t& piece) {
  char* d = slot_pair + ((lane & 1) << 4);
  if (lane & 2) {
    egv_store<SITE>(d + 12, piece[0]);
    egv_store<SITE>(d + 32, (u32x3_t){piece[1], piece[2], piece[3]});
  } else {
    egv_store<SITE>(d, (u32x3_t){piece[0], piece[1], piece[2]});
  }
}
