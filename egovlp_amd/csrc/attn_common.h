// Shared pieces of the MFMA attention kernels (space attention of the SpaceTimeTransformer and the
// DistilBERT masked MHA).  Head dim is fixed at 64 (ViT-B/16, ViT-L/14 and DistilBERT all use 64).
//
// LDS image of a [rows][64] operand: bf16 planes with 128-B rows, 16-B chunks XOR-swizzled by (row & 7):
//   byte(row, col) = row*128 + (((col >> 3) ^ (row & 7)) << 4) + (col & 7)*2
// This one image serves BOTH MFMA read patterns conflict-free:
//   * contraction along columns (d):  ds_read_b128, lane -> row (l&15), chunk (l>>4) + 4*ks
//   * contraction along rows (keys / queries): ds_read_b64_tr_b16 (CDNA4 transpose read): each 16-lane
//     group reads a [4 rows][16 cols] block and lane i receives column i, so no transposed copy of
//     K / V / Q / dO ever exists in LDS or HBM.
// k-index convention for row-contraction fragments (A and B operands use the same one, which is all an
// MFMA needs): element j of lane-group g covers row  r0 + 4g + j  (j < 4)  and  r0 + 16 + 4g + (j-4).
// It is chosen so that the C-layout of a preceding 16x16 MFMA pair (rows 4g..4g+3 of two fragments) IS
// the B operand of the next MFMA without any cross-lane movement.
#pragma once
#include "common.h"
#include "f16x2.h"

#define ATT_D 64
#define ATT_ROW_BYTES 128

typedef __attribute__((ext_vector_type(4))) short s16x4_t;

__device__ __forceinline__ int att_off(int row, int col) {
  return row * ATT_ROW_BYTES + ((((col >> 3) ^ (row & 7)) << 4) | ((col & 7) << 1));
}

// Stage `nrows` rows (zero-filled up to `prows`) of a [*, 64] fp32 operand into swizzled split planes.
// row_ptr(r) returns the global pointer of row r (64 contiguous floats).  256 threads.
template <typename RowPtr>
__device__ __forceinline__ void att_stage(char* hi, char* lo, int nrows, int prows, float scale, RowPtr row_ptr) {
  for (int t = threadIdx.x; t < prows * 8; t += blockDim.x) {
    const int row = t >> 3, chunk = t & 7;
    f32x4_t a = {0.f, 0.f, 0.f, 0.f}, b = a;
    if (row < nrows) {
      const float* p = row_ptr(row) + chunk * 8;
      a = *(const f32x4_t*)p;
      b = *(const f32x4_t*)(p + 4);
    }
    bf16_t h[8], l[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      split_bf16(a[e] * scale, h[e], l[e]);
      split_bf16(b[e] * scale, h[4 + e], l[4 + e]);
    }
    const int off = row * ATT_ROW_BYTES + ((chunk ^ (row & 7)) << 4);
    *(u32x4_t*)(hi + off) = (u32x4_t){pack2(h[0], h[1]), pack2(h[2], h[3]), pack2(h[4], h[5]), pack2(h[6], h[7])};
    if (lo) *(u32x4_t*)(lo + off) = (u32x4_t){pack2(l[0], l[1]), pack2(l[2], l[3]), pack2(l[4], l[5]), pack2(l[6], l[7])};
  }
}

// column-contraction fragment: lane (row r0 + (l&15), cols 8*((l>>4) + 4*ks) .. +7)
__device__ __forceinline__ bf16x8_t att_frag_cols(const char* plane, int r0, int ks, int lane) {
  const int row = r0 + (lane & 15);
  const int chunk = (lane >> 4) + 4 * ks;
  return *(const bf16x8_t*)(plane + row * ATT_ROW_BYTES + ((chunk ^ (row & 7)) << 4));
}

// row-contraction fragment via the hardware transpose read: lane (g = l>>4, c = l&15) receives column
// col0 + c at rows r0+4g..r0+4g+3 (elements 0..3) and r0+16+4g..+3 (elements 4..7).
__device__ __forceinline__ bf16x8_t att_frag_rows(const char* plane, int r0, int col0, int lane) {
  const int g = lane >> 4, p = lane & 15;
  const int col = col0 + ((p & 3) << 2);
  const int ra = r0 + 4 * g + (p >> 2);
  const int rb = ra + 16;
#if defined(EGV_NO_TR_READ)
  // reference path (slow, 8 scalar LDS reads) used to validate the transpose-read semantics on hardware
  bf16x8_t out;
  const int c = col0 + p;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned short a = *(const unsigned short*)(plane + att_off(r0 + 4 * g + j, c));
    const unsigned short b = *(const unsigned short*)(plane + att_off(r0 + 16 + 4 * g + j, c));
    out[j] = __builtin_bit_cast(__bf16, a);
    out[4 + j] = __builtin_bit_cast(__bf16, b);
  }
  return out;
#else
  const s16x4_t x = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4_t*)(plane + att_off(ra, col)));
  const s16x4_t y = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4_t*)(plane + att_off(rb, col)));
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  const s16x8_t z = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
  return __builtin_bit_cast(bf16x8_t, z);
#endif
}

// The same with the lane's address already resolved (p = plane + att_off(r0 + 4g + (p >> 2), col0 + 4 (p & 3))): the kernels whose chunk
// loops are VALU-issue-bound resolve the lane part once per tile and add the chunk's rows as one offset.
#if !defined(EGV_NO_TR_READ)
__device__ __forceinline__ bf16x8_t att_frag_rows_at(const char* p) {
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  const s16x4_t x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
  const s16x4_t y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p + 16 * ATT_ROW_BYTES));
  const s16x8_t z = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
  return __builtin_bit_cast(bf16x8_t, z);
}
#endif

// 8 fp32 -> split bf16x8 pair (four v_cvt_pk_bf16_f32 per plane); F16: the same split in fp16 (hi = fp16(v), lo = fp16(v - hi), NOT
// saturating: the values are probabilities in [0, 1] or scaled gradients, whose overflow must surface as inf) in the same registers
template <bool F16 = false>
__device__ __forceinline__ void att_split8(const float* v, bf16x8_t& hi, bf16x8_t& lo) {
  u32x4_t h, l;
  if constexpr (F16) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const f16pk_h2_t hp = __builtin_convertvector(((f16pk_f2_t){v[2 * e], v[2 * e + 1]}), f16pk_h2_t);
      h[e] = __builtin_bit_cast(uint32_t, hp);
      l[e] = f16_pk(v[2 * e] - (float)hp[0], v[2 * e + 1] - (float)hp[1]);
    }
    hi = __builtin_bit_cast(bf16x8_t, h);
    lo = __builtin_bit_cast(bf16x8_t, l);
    return;
  }
  uint32_t a, b;
  split_bf16x2(v[0], v[1], a, b); h[0] = a; l[0] = b;
  split_bf16x2(v[2], v[3], a, b); h[1] = a; l[1] = b;
  split_bf16x2(v[4], v[5], a, b); h[2] = a; l[2] = b;
  split_bf16x2(v[6], v[7], a, b); h[3] = a; l[3] = b;
  hi = __builtin_bit_cast(bf16x8_t, h);
  lo = __builtin_bit_cast(bf16x8_t, l);
}

// B-operand fragment straight from global: lane -> row (l&15) given by `p` (already row-resolved),
// cols 8*((l>>4) + 4*ks) .. +7, scaled, split.
__device__ __forceinline__ void att_gfrag(const float* rowp, int ks, int lane, float scale, bf16x8_t& hi, bf16x8_t& lo) {
  const float* p = rowp + ((lane >> 4) + 4 * ks) * 8;
  const f32x4_t a = *(const f32x4_t*)p;
  const f32x4_t b = *(const f32x4_t*)(p + 4);
  float v[8] = {a[0] * scale, a[1] * scale, a[2] * scale, a[3] * scale,
                b[0] * scale, b[1] * scale, b[2] * scale, b[3] * scale};
  att_split8(v, hi, lo);
}

// ---- the same two loaders for operands that already ARE split-bf16 planes in HBM (the fused qkv buffer written by
// the qkv GEMM epilogue, the attention-output gradient written by the proj dgrad epilogue): a straight 16-B copy.
// row_off(r) returns the ELEMENT offset of row r (64 contiguous bf16) inside the hi / lo planes.  256 threads.
template <typename RowOff>
__device__ __forceinline__ void att_stage_planes(char* hi, char* lo, const bf16_t* ph, const bf16_t* pl, int nrows,
                                                 int prows, RowOff row_off) {
  for (int t = threadIdx.x; t < prows * 8; t += blockDim.x) {
    const int row = t >> 3, chunk = t & 7;
    u32x4_t a = {0u, 0u, 0u, 0u}, b = a;
    if (row < nrows) {
      const long off = row_off(row) + chunk * 8;
      a = *(const u32x4_t*)(ph + off);
      if (lo) b = *(const u32x4_t*)(pl + off);
    }
    const int o = row * ATT_ROW_BYTES + ((chunk ^ (row & 7)) << 4);
    *(u32x4_t*)(hi + o) = a;
    if (lo) *(u32x4_t*)(lo + o) = b;
  }
}

// B-operand fragment straight from global planes: `off` = element offset of this lane's row (l&15 resolved by the
// caller); columns 8*((l>>4) + 4*ks) .. +7.
__device__ __forceinline__ void att_gfrag_planes(const bf16_t* ph, const bf16_t* pl, long off, int ks, int lane,
                                                 bf16x8_t& hi, bf16x8_t& lo) {
  const long o = off + ((lane >> 4) + 4 * ks) * 8;
  hi = *(const bf16x8_t*)(ph + o);
  lo = hi;
  if (pl) lo = *(const bf16x8_t*)(pl + o);
}

// F16: the operand registers hold fp16 (the fp16 attention of the fp16 backward mode: q / k / v / dO planes, P and dS are fp16 -- 2^-11
// per operand where bf16 has 2^-8; the three-product form multiplies (hi, lo) fp16 splits, fp32-grade like the bf16 one)
template <int PASSES, bool F16 = false>
__device__ __forceinline__ f32x4_t att_mma(bf16x8_t ah, bf16x8_t al, bf16x8_t bh, bf16x8_t bl, f32x4_t c) {
  if constexpr (F16) {
    const f16x8_t a0 = __builtin_bit_cast(f16x8_t, ah), a1 = __builtin_bit_cast(f16x8_t, al);
    const f16x8_t b0 = __builtin_bit_cast(f16x8_t, bh), b1 = __builtin_bit_cast(f16x8_t, bl);
    if (PASSES == 3) {
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, c, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, c, 0, 0, 0);
  } else {
    if (PASSES == 3) {
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
  }
}

// Geometry of one attention "group" (the rows that attend to each other).
//  MODE_SPACE: qkv [B,S,3,H,64]; group = (b, f, h); nq = n queries (tokens 1+f*n+i), nk = n+1 keys
//              (key 0 = CLS token 0, key j = token f*n + j)            model/video_transformer.py:114-124
//  MODE_TEXT : separate q,k,v [B,L,H*64]; group = (b, h); nq = nk = L; key j masked if mask[b,j] == 0
enum { MODE_SPACE = 0, MODE_TEXT = 2 };

// one output pair in the plane formats of the attention forward (egv_divided_attn_fwd, mode >> 1):
//   0  hi = bf16 pair, lo = the bf16 residual pair (split planes: a three-product proj)
//   1  hi = bf16 pair (what a bf16 backward reads), lo = the fp16 pair of the VALUES (the proj Linear runs ONE fp16 product)
//   2  the f16x2 operand format, first-operand role (csrc/f16x2.h): hi = a1 = fp16((1 - e) v), lo = a2 = fp16(v - a1) -- a two-product proj
//      whose backward is fp16 as well (the weight gradient reads a1, the attention backward takes O = a1 / (1 - e))
//   3  hi = fp16(value), nothing else (the second plane does not exist): one-product proj, fp16 backward
enum { ATT_OUT_SPLIT = 0, ATT_OUT_BF16_F16 = 1, ATT_OUT_F16X2 = 2, ATT_OUT_F16 = 3, ATT_GRAD_F16 = 4 };
__device__ __forceinline__ void att_out2(float a, float b, int fmt, uint32_t& hi, uint32_t& lo) {
  if (fmt == ATT_OUT_F16X2) {       // f16x2_a on the pair, packed conversions
    const float x0 = f16x2_clamp(a), x1 = f16x2_clamp(b);
    const f16pk_h2_t h = __builtin_convertvector(((f16pk_f2_t){x0 - x0 * F16X2_E, x1 - x1 * F16X2_E}), f16pk_h2_t);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = f16_pk(x0 - (float)h[0], x1 - (float)h[1]);
    return;
  }
  if (fmt == ATT_OUT_F16) {
    hi = lo = f16_pk(f16x2_clamp(a), f16x2_clamp(b));
    return;
  }
  if (fmt == ATT_GRAD_F16) {       // a scaled gradient of the fp16 backward: ONE plane, NOT saturating (overflow -> inf -> skipped step)
    hi = lo = f16_grad_pack2(a, b);
    return;
  }
  split_bf16x2(a, b, hi, lo);
  if (fmt) lo = f16_pk(f16x2_clamp(a), f16x2_clamp(b));
}
// the forward output O as the backward reads it back (delta = rowsum(dO o O)): one 32-bit word of the first plane (and of the second,
// fmt 0 only) -> two fp32 values
__device__ __forceinline__ void att_o_unpack(uint32_t w_hi, uint32_t w_lo, bool has_lo, int fmt, float& x, float& y) {
  if (fmt == ATT_OUT_F16X2 || fmt == ATT_OUT_F16) {
    f16x2_unpack(w_hi, x, y);
    if (fmt == ATT_OUT_F16X2) { x *= (1.0f / (1.0f - F16X2_E)); y *= (1.0f / (1.0f - F16X2_E)); }
    return;
  }
  x = __uint_as_float(w_hi << 16);
  y = __uint_as_float(w_hi & 0xffff0000u);
  if (has_lo && fmt == ATT_OUT_SPLIT) { x += __uint_as_float(w_lo << 16); y += __uint_as_float(w_lo & 0xffff0000u); }
}

struct AttGeom {
  const float* q;    // MODE_TEXT: fp32 sources
  const float* k;
  const float* v;
  const bf16_t* ph;  // MODE_SPACE: the fused qkv buffer [B, S, 3, H, 64] as split-bf16 planes (pl == nullptr: hi only)
  const bf16_t* pl;
  long tok_stride;   // elements between consecutive tokens in q/k/v (or in the planes)
  int B, T, n, H, S; // S = tokens per batch item (1+T*n or L)
  int nq, nk;
  const long long* mask;  // MODE_TEXT only
  EgvDrop drop;           // MODE_TEXT only: attention-probability dropout (thresh == 0: none); element index
                          // ((b * H + h) * S + query) * S + key
  int out_fmt;            // format of the output planes (ATT_OUT_*, see att_out2)
  int f16;                // MODE_SPACE: the qkv planes (and, backward, the dO planes) hold fp16 -- (hi, lo) = (fp16(x), fp16(x - hi)) -- and
                          // every product runs on the fp16 MFMA
};

template <int MODE>
struct AttGroup {
  int b, f, h;
  long tok0;  // first token of batch item b
  __device__ __forceinline__ AttGroup(const AttGeom& g, int gid) {
    h = gid % g.H;
    int r = gid / g.H;
    if (MODE == MODE_SPACE) {
      f = r % g.T;
      b = r / g.T;
    } else {
      f = 0;
      b = r;
    }
    tok0 = (long)b * g.S;
  }
  __device__ __forceinline__ long q_tok(const AttGeom& g, int i) const {
    return MODE == MODE_SPACE ? tok0 + 1 + (long)f * g.n + i : tok0 + i;
  }
  __device__ __forceinline__ long k_tok(const AttGeom& g, int j) const {
    return MODE == MODE_SPACE ? (j == 0 ? tok0 : tok0 + (long)f * g.n + j) : tok0 + j;
  }
};
