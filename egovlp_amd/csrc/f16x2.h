// The f16x2 operand format: a product A B^T in fp32-grade accuracy from TWO fp16 MFMA products instead of three bf16 ones.
//
//   A (activations, first operand):   a1 = fp16((1 - e) a)      a2 = fp16(a - a1)            e = 2^-6
//   B (weights, second operand):      b1 = fp16(b)              b2 = fp16(b1 + (b - b1) / e)
//   a b  ~=  a1 b1 + a2 b2
//
// a2 = e a - rho (rho = the rounding error of a1), so a2 b2 = (e a - rho)(b1 + (b - b1) / e) = e a b1 + a (b - b1) - rho b1 - ...: the
// second product hands back the e a b1 that a1 left out, carries the residual of b, and cancels a1's rounding error; what is NOT
// compensated -- the roundings of a2 and b2 and the cross term rho (b - b1) / e -- is attenuated by e or by 2^-12 / e: ~2^-17 relative
// per product, the class of the split-bf16 three-product scheme (measured on random operands: 5.3e-6 against 4.4e-6; embeddings of
// the EgoClip model vs the fp32 oracle 3.2e-5 against 2.7e-5, tests/quant_emul.py scheme "fp16x2").  Both products accumulate into the
// same fp32 accumulator, nothing is rescaled, and an operand is two fp16 planes -- the same bytes and the same [rows, ld] geometry as
// the split-bf16 (hi, lo) planes, so the GEMM's stage layout, DMA and fragment reads are the three-product kernel's.
// Range: fp16 (6e-5 .. 65504 normal); values below ~4e-3 lose relative (not absolute) precision in a2.  The format is used for
// FORWARD operands (activations of O(1), weights); gradients keep bf16's exponent range.
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;

constexpr float F16X2_E = 0.015625f;          // e = 2^-6
constexpr float F16X2_INV_E = 64.0f;

__device__ __forceinline__ float f16x2_clamp(float x) { return fminf(fmaxf(x, -65504.0f), 65504.0f); }

// first-operand (activation) encoding
__device__ __forceinline__ void f16x2_a(float x, _Float16& a1, _Float16& a2) {
  x = f16x2_clamp(x);
  a1 = (_Float16)(x - x * F16X2_E);            // (1 - e) x is exact in fp32 for e = 2^-6 up to one rounding of the subtraction
  a2 = (_Float16)(x - (float)a1);
}
// second-operand (weight) encoding
__device__ __forceinline__ void f16x2_b(float x, _Float16& b1, _Float16& b2) {
  x = f16x2_clamp(x);
  b1 = (_Float16)x;
  b2 = (_Float16)((float)b1 + (x - (float)b1) * F16X2_INV_E);
}

__device__ __forceinline__ uint32_t f16x2_pack(_Float16 lo, _Float16 hi) {
  return (uint32_t)__builtin_bit_cast(unsigned short, lo) | ((uint32_t)__builtin_bit_cast(unsigned short, hi) << 16);
}
// two fp32 -> one word of two fp16 (round to nearest even, overflow -> inf: what two scalar casts give, bit for bit) as ONE instruction:
// a <2 x float> -> <2 x half> truncation selects gfx950's v_cvt_pk_f16_f32, where two scalar casts and a pack cost v_cvt_f16_f32 +
// v_cvt_f16_f32_sdwa + v_or_b32.  Every fp16 plane writer goes through here (GEMM / LayerNorm / attention epilogues are VALU-bound or
// close to it).
typedef __attribute__((ext_vector_type(2))) float f16pk_f2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16pk_h2_t;
__device__ __forceinline__ uint32_t f16_pk(float a, float b) {
  const f16pk_f2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16pk_h2_t));
}

// eight consecutive values -> the two 16-byte plane pieces (ROLE 0: first operand, 1: second operand) [+ the bf16 piece]
template <int ROLE>
__device__ __forceinline__ void f16x2_encode8(const float (&v)[8], u32x4_t& p1, u32x4_t& p2) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {      // f16x2_a / f16x2_b on a pair of values, packed conversions
    const float x0 = f16x2_clamp(v[2 * e]), x1 = f16x2_clamp(v[2 * e + 1]);
    const f16pk_f2_t t = ROLE == 0 ? (f16pk_f2_t){x0 - x0 * F16X2_E, x1 - x1 * F16X2_E} : (f16pk_f2_t){x0, x1};
    const f16pk_h2_t h = __builtin_convertvector(t, f16pk_h2_t);
    p1[e] = __builtin_bit_cast(uint32_t, h);
    const float r0 = x0 - (float)h[0], r1 = x1 - (float)h[1];
    p2[e] = ROLE == 0 ? f16_pk(r0, r1) : f16_pk((float)h[0] + r0 * F16X2_INV_E, (float)h[1] + r1 * F16X2_INV_E);
  }
}
// eight consecutive values -> ONE 16-byte piece of plain fp16(value) (saturating): the operand of a single-fp16-product GEMM
// (egv_gemm_nt passes == 4) -- the Linears whose share of the 1e-3 parity budget allows one product (the per-op table)
__device__ __forceinline__ u32x4_t f16_piece8(const float (&v)[8]) {
  u32x4_t p;
#pragma unroll
  for (int e = 0; e < 4; ++e) p[e] = f16_pk(f16x2_clamp(v[2 * e]), f16x2_clamp(v[2 * e + 1]));
  return p;
}
// The same WITHOUT saturation: gradient planes of the fp16 backward.  A scaled gradient beyond fp16's range must become inf (and then
// NaN / inf in every gradient behind it), because that is what the dynamic loss scale detects and answers with a skipped step and a
// halved scale (egv_loss_scale_*); a clamp would silently train on a clipped gradient instead.
__device__ __forceinline__ u32x4_t f16_grad_piece8(const float (&v)[8]) {
  u32x4_t p;
#pragma unroll
  for (int e = 0; e < 4; ++e) p[e] = f16_pk(v[2 * e], v[2 * e + 1]);
  return p;
}
// eight consecutive values -> an fp16 SPLIT (hi = fp16(v), lo = fp16(v - hi); saturating): the qkv planes of the fp16 attention -- the
// three-product forward multiplies (hi, lo) pairs like the split-bf16 one (fp32-grade), the backward reads hi alone (2^-11)
__device__ __forceinline__ void f16_split8(const float (&v)[8], u32x4_t& p1, u32x4_t& p2) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float a = f16x2_clamp(v[2 * e]), b = f16x2_clamp(v[2 * e + 1]);
    const f16pk_h2_t h = __builtin_convertvector(((f16pk_f2_t){a, b}), f16pk_h2_t);
    p1[e] = __builtin_bit_cast(uint32_t, h);
    p2[e] = f16_pk(a - (float)h[0], b - (float)h[1]);
  }
}
__device__ __forceinline__ uint32_t f16_grad_pack2(float a, float b) { return f16_pk(a, b); }
// two fp16 values of one 32-bit word -> fp32
__device__ __forceinline__ void f16x2_unpack(uint32_t w, float& a, float& b) {
  a = (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu));
  b = (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16));
}
__device__ __forceinline__ u32x4_t bf16_piece8(const float (&v)[8]) {
  return (u32x4_t){f32x2_to_bf16x2(v[0], v[1]), f32x2_to_bf16x2(v[2], v[3]), f32x2_to_bf16x2(v[4], v[5]), f32x2_to_bf16x2(v[6], v[7])};
}
