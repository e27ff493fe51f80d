// Diagnostics (not on the product path): the sustained MFMA issue rate of this chip under its power limit.
// egv_diag_mfma_peak runs `iters` rounds of 40 independent v_mfma_f32_16x16x32_bf16 per wave (the accumulator footprint of
// a gemm_big wave, 160 registers), no memory traffic in the loop, `waves` waves per workgroup (8 = two per SIMD, the
// gemm_big occupancy), one workgroup per CU.  tools/mfma_peak.py turns the time into TFLOP/s: the number to read the
// GEMM main-loop rate against when the nominal 2.5 PFLOP/s assumes the peak engine clock.
#include "common.h"
#include "egovlp_hip.h"

namespace {

__global__ __launch_bounds__(512, 2) void mfma_peak_kernel(int iters, float* out) {
  f32x4_t acc[40];
#pragma unroll
  for (int i = 0; i < 40; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 av = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, (short)(0x3F80 + (threadIdx.x & 1))};
  bf16x8_t a = __builtin_bit_cast(bf16x8_t, av), b = a;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 40; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  f32x4_t s = acc[0];
#pragma unroll
  for (int i = 1; i < 40; ++i) s += acc[i];
  if (s[0] + s[1] + s[2] + s[3] == -1.0f) out[threadIdx.x] = s[0];   // never true: keeps the loop alive
}

}  // namespace

extern "C" int egv_diag_mfma_peak(int32_t iters, int32_t waves, float* out, void* stream) {
  if (iters <= 0 || waves < 1 || waves > 8 || !out) return EGV_ERR_ARG;
  EGV_LAUNCH(mfma_peak_kernel, dim3(256), dim3(64 * waves), 0, (hipStream_t)stream, iters, out);
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}
