// Shared device helpers for the EgoVLP gfx950 (MI355X / CDNA4) kernels.
// Everything here assumes wave64 and gfx950; there is no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned short us4_t;
typedef __attribute__((ext_vector_type(8))) unsigned short us8_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

#define EGV_OK 0
#define EGV_ERR_ARG 1
#define EGV_ERR_LAUNCH 2

// hipGetLastError() is sticky per thread: an unrelated earlier runtime call of the host application (e.g. a failed
// capability probe inside PyTorch) must not be reported as OUR launch failing, so clear it right before launching.
#define EGV_LAUNCH(...)          \
  do {                           \
    (void)hipGetLastError();     \
    hipLaunchKernelGGL(__VA_ARGS__); \
  } while (0)

#define EGV_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return EGV_ERR_LAUNCH + (int)e__; \
  } while (0)

// Global store / load, optionally non-temporal (streaming: `global_store ... nt`).  Which streams use it is a per-site
// compile-time choice (EGV_NT_MASK bits; Makefile `nt` target builds A/B libraries):
//   1 what fc1 saves for backward (read a whole forward later)   2 time-attention forward output planes
//   4 plane outputs of the plain GEMM epilogue (qkv, dgrad dX)  8 fp32 GEMM outputs   8192 h planes of the GELU epilogue
// 16384 dZ planes of the GELU' epilogue
//  16 LayerNorm outputs (fwd planes / fp32, bwd dx + planes)      32 AdamW stores (p, m, v)      64 AdamW loads (g, m, v, p)
// 128 reduced weight gradients (split-K reduce output)           256 space-attention outputs (fwd / bwd, 8-byte pieces)
// 512 time-attention backward outputs (8-byte-lane kernel only)   1024 LOADS of the GEMM epilogue (residual / saved gelu')   2048 LOADS of time attention fwd
// 4096 LOADS of the split-K reduce
#ifndef EGV_NT_MASK
#define EGV_NT_MASK (519 + 8192 + 16384)
#endif
enum { EGV_NT_SAVED = 1, EGV_NT_ATTN_OUT = 2, EGV_NT_GEMM_PLANES = 4, EGV_NT_GEMM_F32 = 8, EGV_NT_LN = 16, EGV_NT_ADAMW_ST = 32,
       EGV_NT_ADAMW_LD = 64, EGV_NT_WGRAD = 128, EGV_NT_SPACE_ATTN = 256, EGV_NT_TIME_BWD = 512,
       EGV_NT_EPI_LD = 1024, EGV_NT_TIME_LD = 2048, EGV_NT_REDUCE_LD = 4096,
       EGV_NT_GELU_PLANES = 8192, EGV_NT_GELUBWD_PLANES = 16384 };
template <int SITE, typename V>
__device__ __forceinline__ void egv_store(void* p, V v) {
  if constexpr ((EGV_NT_MASK & SITE) != 0) __builtin_nontemporal_store(v, (V*)p);
  else *(V*)p = v;
}
template <int SITE, typename V>
__device__ __forceinline__ void egv_store16(void* p, V v) { egv_store<SITE, V>(p, v); }
template <int SITE, typename V>
__device__ __forceinline__ V egv_load(const void* p) {
  if constexpr ((EGV_NT_MASK & SITE) != 0) return __builtin_nontemporal_load((const V*)p);
  else return *(const V*)p;
}

// ---- bf16 <-> f32 -----------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// fp32 -> bf16, round-to-nearest-even, on the CDNA4 conversion instruction v_cvt_pk_bf16_f32 (two values per issue; the
// integer-add emulation of older targets costs ~4 VALU ops per value and showed up in every plane-writing epilogue).
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t f32x2_to_bf16x2(float a, float b) {       // a -> bits 0..15, b -> bits 16..31
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){a, b}, bf16x2_t));
}
__device__ __forceinline__ bf16_t f32_to_bf16(float x) { return (bf16_t)(f32x2_to_bf16x2(x, 0.f) & 0xffffu); }

// "split-bf16": x ~= hi + lo with hi = bf16(x), lo = bf16(x - hi); |x - hi - lo| <= 2^-17 |x|.
// Three bf16 MFMA passes (hi*hi + hi*lo + lo*hi) on such pairs reproduce an fp32 product to ~2^-16
// relative, which is what lets bf16 matrix cores meet the 1e-3 parity bar against the fp32 reference.
__device__ __forceinline__ void split_bf16(float x, bf16_t& hi, bf16_t& lo) {
  hi = f32_to_bf16(x);
  lo = f32_to_bf16(x - bf16_to_f32(hi));
}
// the same for a pair, packed (a in the low half): 2 conversions + 2 subtractions + 2 unpack ops for two values
__device__ __forceinline__ void split_bf16x2(float a, float b, uint32_t& hi, uint32_t& lo) {
  hi = f32x2_to_bf16x2(a, b);
  lo = f32x2_to_bf16x2(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}

__device__ __forceinline__ uint32_t pack2(bf16_t a, bf16_t b) { return (uint32_t)a | ((uint32_t)b << 16); }

// ---- wave64 reductions ------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// exact (erf) GELU and its derivative -- nn.GELU default, model/video_transformer.py:37.
// erf by Abramowitz & Stegun 7.1.26 (|abs error| <= 1.5e-7, i.e. fp32 round-off class): one v_rcp, one v_exp and five
// FMAs instead of ocml's ~25-instruction branchy erff.  The GELU epilogues of fc1 / fc2-dgrad apply this to 77 M elements
// per GEMM with the matrix pipe idle, and the SAME exponential e^{-x^2/2} serves the Gaussian pdf of the derivative.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& pdf) {
  const float ax = fabsf(x);
  // raw v_exp_f32 / v_rcp_f32 (1 ulp): __expf adds a denormal-range rescue (compare + two selects) and __frcp_rn expands to
  // the IEEE division sequence (~10 VALU ops) -- together a third of the GELU epilogue's instructions, for accuracy the
  // 1.5e-7 approximation cannot use
  const float e = __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x);      // e^{-x^2/2} = e^{-u^2}, u = x / sqrt(2)
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, ax, 1.0f));
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float half = 0.5f * poly * e;                         // 0.5 * erfc(|u|): no cancellation in the negative tail
  cdf = (x < 0.f) ? half : 1.0f - half;
  pdf = 0.3989422804014327f * e;
}
__device__ __forceinline__ float gelu_f(float x) {
  float cdf, pdf;
  gelu_parts(x, cdf, pdf);
  return x * cdf;
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  float cdf, pdf;
  gelu_parts(x, cdf, pdf);
  return cdf + x * pdf;
}

// ---- counter-based dropout mask ------------------------------------------------------------------------
// keep(seed, idx) is a pure function of a 64-bit seed (one per dropout site and step, chosen by the host) and the element
// index, so the backward kernels regenerate the mask of the forward instead of storing it (HF DistilBERT, modeling_distilbert.py:
// embedding / attention-probability / FFN dropout).  Two rounds of a 32-bit avalanche hash over the index words; this is NOT
// PyTorch's Philox stream -- dropout masks are not part of the parity contract, only their statistics (keep rate 1 - p,
// survivors scaled by 1 / (1 - p)) and the forward / backward consistency are (tests/test_gpu_dropout.py).
__device__ __forceinline__ uint32_t egv_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
struct EgvDrop {
  uint32_t s0, s1, thresh;   // seed words; keep <=> hash >= thresh (thresh = p * 2^32)
  float scale;               // 1 / (1 - p); 0 threshold and scale 1 when p == 0
  const uint32_t* dev;       // optional DEVICE seed words {lo, hi}, XOR-ed into (s0, s1) by the kernel: a step captured into a
                             // HIP graph keeps its host-side seed as a launch argument forever, so what changes from replay to
                             // replay has to live in memory the graph reads (a caller that captures a training step; none in this package)
};
__device__ __forceinline__ EgvDrop egv_drop_resolve(EgvDrop d) {   // once per kernel, before the first egv_drop_scale
  if (d.dev) {
    d.s0 ^= d.dev[0];
    d.s1 ^= d.dev[1];
  }
  return d;
}
__device__ __forceinline__ float egv_drop_scale(const EgvDrop& d, uint64_t idx) {
  const uint32_t h = egv_mix32(egv_mix32((uint32_t)idx ^ d.s0) ^ (uint32_t)(idx >> 32) ^ d.s1);
  return h >= d.thresh ? d.scale : 0.0f;
}
static inline EgvDrop egv_make_drop(float p, uint64_t seed, const uint64_t* seed_dev = nullptr) {
  EgvDrop d;
  d.s0 = (uint32_t)seed; d.s1 = (uint32_t)(seed >> 32);
  d.dev = (const uint32_t*)seed_dev;
  if (!(p > 0.f)) { d.thresh = 0u; d.scale = 1.0f; return d; }
  const double t = (double)p * 4294967296.0;
  d.thresh = t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
  d.scale = 1.0f / (1.0f - p);
  return d;
}

// XCD-aware, bijective remap of a 1-D block id: blocks that the dispatcher places on one XCD
// (bid % 8) receive a CONTIGUOUS range of work ids, so neighbouring tiles share that XCD's L2.
// Speed only -- correctness never depends on placement.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (bid >> 3);
}
