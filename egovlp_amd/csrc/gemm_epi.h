// Shared by gemm_big.hip and gemm_duo.hip: the epilogue flavours and the per-row-piece epilogue.
#pragma once
#include "common.h"
#include "egovlp_hip.h"

// epilogue flavours (template parameter: each kernel instance carries only its own epilogue code)
enum { EPI_RAW = 0,      // store the accumulators (split-K partial slab, or plain fp32 output)
       EPI_LINEAR = 1,   // + bias, + residual -> fp32 and/or planes
       EPI_GELU = 2,     // + bias, pre-activation -> aux_out, gelu -> fp32 and/or planes
       EPI_GELU_BWD = 3, // * gelu'(aux_in) -> fp32 and/or planes
       EPI_GENERIC = 4 };// everything at run time (alpha, ReLU', ...)

// One 4-column piece of one output row: alpha, + bias, activation, + residual, stores (include/egovlp_hip.h order).
template <int EPI>
__device__ __forceinline__ void epilogue4(const egv_gemm_desc& p, f32x4_t v, int m, int n, int z, int ksplit) {
  if (EPI == EPI_RAW) {
    if (ksplit > 1) *(f32x4_t*)(p.partial + ((long)z * p.M + m) * p.N + n) = v;
    else *(f32x4_t*)(p.out_f32 + (long)m * p.ldo + n) = v;
    return;
  }
  if (EPI == EPI_GENERIC && p.alpha != 1.0f) v *= p.alpha;
  if (EPI != EPI_GELU_BWD && p.bias) v += *(const f32x4_t*)(p.bias + n);
  if (EPI == EPI_GELU || (EPI == EPI_GENERIC && p.act == EGV_ACT_GELU)) {
    if (p.aux_out) {
      if (p.aux_bf16)
        *(u32x2_t*)((bf16_t*)p.aux_out + (long)m * p.ldaux + n) =
            (u32x2_t){pack2(f32_to_bf16(v[0]), f32_to_bf16(v[1])), pack2(f32_to_bf16(v[2]), f32_to_bf16(v[3]))};
      else
        *(f32x4_t*)(p.aux_out + (long)m * p.ldaux + n) = v;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
  } else if (EPI == EPI_GELU_BWD || (EPI == EPI_GENERIC && p.act == EGV_ACT_GELU_BWD)) {
    f32x4_t zv;
    if (p.aux_bf16) {
      const u32x2_t zb = *(const u32x2_t*)((const bf16_t*)p.aux_in + (long)m * p.ldaux + n);
      zv = (f32x4_t){__uint_as_float(zb[0] << 16), __uint_as_float(zb[0] & 0xffff0000u), __uint_as_float(zb[1] << 16),
                     __uint_as_float(zb[1] & 0xffff0000u)};
    } else {
      zv = *(const f32x4_t*)(p.aux_in + (long)m * p.ldaux + n);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= gelu_grad_f(zv[e]);
  } else if (EPI == EPI_GENERIC && p.act == EGV_ACT_RELU_BWD) {
    const f32x4_t zv = *(const f32x4_t*)(p.aux_in + (long)m * p.ldaux + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = zv[e] > 0.f ? v[e] : 0.f;
  }
  if (p.residual) v += *(const f32x4_t*)(p.residual + (long)m * p.ldr + n);
  if (p.out_f32) *(f32x4_t*)(p.out_f32 + (long)m * p.ldo + n) = v;
  if (p.out_hi) {
    bf16_t h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split_bf16(v[e], h[e], l[e]);
    *(u32x2_t*)(p.out_hi + (long)m * p.ldoh + n) = (u32x2_t){pack2(h[0], h[1]), pack2(h[2], h[3])};
    if (p.out_lo) *(u32x2_t*)(p.out_lo + (long)m * p.ldoh + n) = (u32x2_t){pack2(l[0], l[1]), pack2(l[2], l[3])};
  }
}

