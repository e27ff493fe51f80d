// Shared GEMM epilogue: one 16x16 accumulator fragment of the "swapped" MFMA (weights as the A operand), i.e. this
// lane holds C[m][n .. n+3].  Order of operations (include/egovlp_hip.h): alpha, + bias[n], activation, + residual.
#pragma once
#include "common.h"
#include "egovlp_hip.h"

__device__ __forceinline__ void egv_gemm_store4(const egv_gemm_desc& p, f32x4_t v, int m, int n, int z, int ksplit) {
  if (m >= p.M || n >= p.N) return;  // N is a multiple of 4
  if (ksplit > 1) {
    *(f32x4_t*)(p.partial + ((long)z * p.M + m) * p.N + n) = v;
    return;
  }
  if (p.alpha != 1.0f) v *= p.alpha;
  if (p.bias) v += *(const f32x4_t*)(p.bias + n);
  if (p.act == EGV_ACT_GELU) {
    if (p.aux_out) *(f32x4_t*)(p.aux_out + (long)m * p.ldaux + n) = v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
  } else if (p.act == EGV_ACT_GELU_BWD) {
    const f32x4_t zv = *(const f32x4_t*)(p.aux_in + (long)m * p.ldaux + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= gelu_grad_f(zv[e]);
  } else if (p.act == EGV_ACT_RELU_BWD) {
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
