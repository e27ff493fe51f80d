// The CLS-row helpers of divided space-time attention (model/video_transformer.py:109-112: the clip's CLS query attends to all S
// keys per (b, head)).  The CLS query has no attention kernel of its own: each group of the space kernel (attn_mfma_*.hip) and
// each unit of the time kernel (attn_time_mfma.hip) carries it as one more query against ITS keys -- forward as an un-normalised
// softmax partial (o[64], m, l) merged here by egv_attn_cls_combine, backward with the global log-sum-exp and delta
// (egv_attn_cls_delta), so every dK / dV row leaves those kernels complete (patch queries + CLS query) and is written ONCE, as bf16
// planes.  Only the CLS token's own gradients (shared by all groups of a clip) go through fp32 atomics + egv_attn_cls_finish.
// (Rounds 1 - 3 also had the vector-ALU time-attention kernels in this file; they lost to the matrix-core kernels at every T --
// T = 4 in-step: 835 -> 846 pairs/s, profiles/r04n_* -- and were removed in round 4.)
#include "common.h"
#include "f16x2.h"
#include "egovlp_hip.h"

namespace {

constexpr int D = 64;

// ------------------------------------------------------------------------------------------- CLS row helpers
// forward: merge the G softmax partials (o[64], m, l) of one (clip, head) -> output planes of token 0 + its lse.
// 256 threads: thread = (sub = t >> 4, 4 channels); the 16 subs walk the groups interleaved with an online merge, then
// one LDS round merges the subs.
__global__ __launch_bounds__(256) void attn_cls_combine_kernel(const float* __restrict__ ws, int G, int S, int H,
                                                              bf16_t* __restrict__ out_hi, bf16_t* __restrict__ out_lo,
                                                              float* __restrict__ lse, int out_fmt) {
  __shared__ float sm[16], sl[16], so[16][64];
  const int t = threadIdx.x;
  const int sub = t >> 4, c4 = (t & 15) * 4;
  const int h = blockIdx.x % H, b = blockIdx.x / H;
  const float* w = ws + (long)blockIdx.x * G * 68;
  float m = -3e38f, l = 0.f;
  f32x4_t o = {0.f, 0.f, 0.f, 0.f};
  for (int g = sub; g < G; g += 16) {
    const float mg = w[(long)g * 68 + 64], lg = w[(long)g * 68 + 65];
    const f32x4_t og = *(const f32x4_t*)(w + (long)g * 68 + c4);
    const float mn = fmaxf(m, mg);
    const float ea = __expf(m - mn), eb = __expf(mg - mn);
    l = l * ea + lg * eb;
    o = o * ea + og * eb;
    m = mn;
  }
  if ((t & 15) == 0) {
    sm[sub] = m;
    sl[sub] = l;
  }
  *(f32x4_t*)&so[sub][c4] = o;
  __syncthreads();
  if (t < 64) {
    float mm = -3e38f;
#pragma unroll
    for (int i = 0; i < 16; ++i) mm = fmaxf(mm, sm[i]);
    float ll = 0.f, oo = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float e = __expf(sm[i] - mm);   // subs that saw no group: m = -3e38 -> e = 0
      ll += sl[i] * e;
      oo += so[i][t] * e;
    }
    oo /= ll;
    bf16_t hh, lo2;
    split_bf16(oo, hh, lo2);
    if (out_fmt == 1) lo2 = __builtin_bit_cast(unsigned short, (_Float16)f16x2_clamp(oo));      // second plane = fp16(value)
    if (out_fmt == 2) {          // f16x2, first-operand role
      _Float16 a1, a2;
      f16x2_a(oo, a1, a2);
      hh = __builtin_bit_cast(unsigned short, a1);
      lo2 = __builtin_bit_cast(unsigned short, a2);
    }
    if (out_fmt == 3) hh = __builtin_bit_cast(unsigned short, (_Float16)f16x2_clamp(oo));         // the only plane = fp16(value)
    const long off = (long)b * S * H * D + (long)h * D + t;
    out_hi[off] = hh;
    if (out_lo) out_lo[off] = lo2;
    if (t == 0) lse[((long)b * H + h) * S] = mm + __logf(ll);
  }
}

// backward prologue: delta of the CLS query row = sum_d dO[d] * O[d] (== sum_j P_j dP_j over ALL keys)
__global__ __launch_bounds__(64) void attn_cls_delta_kernel(const bf16_t* __restrict__ oh, const bf16_t* __restrict__ ol,
                                                            const bf16_t* __restrict__ doh,
                                                            const bf16_t* __restrict__ dol, int S, int H,
                                                            float* __restrict__ delta, float* __restrict__ dcls, int o_fmt, int f16) {
  const int lane = threadIdx.x;
  const int h = blockIdx.x % H, b = blockIdx.x / H;
  if (dcls) {   // this (clip, head)'s raw dq / dk / dv accumulators start at zero: saves a memset node per attention backward
    float* a = dcls + (long)blockIdx.x * 192;
    a[lane] = 0.f; a[64 + lane] = 0.f; a[128 + lane] = 0.f;
  }
  const long off = (long)b * S * H * D + (long)h * D + lane;
  float o, g = f16 ? (float)__builtin_bit_cast(_Float16, doh[off]) : bf16_to_f32(doh[off]);      // dO: an fp16 plane in the fp16 attention backward
  if (o_fmt == 2 || o_fmt == 3) {       // the forward wrote fp16 planes (attn_common.h ATT_OUT_F16X2 / ATT_OUT_F16): O from the first one
    o = (float)__builtin_bit_cast(_Float16, oh[off]);
    if (o_fmt == 2) o *= 1.0f / (1.0f - F16X2_E);
  } else {
    o = bf16_to_f32(oh[off]);
    if (ol && o_fmt == 0) o += bf16_to_f32(ol[off]);
  }
  if (dol) g += bf16_to_f32(dol[off]);
  const float d = wave_sum(o * g);
  if (lane == 0) delta[((long)b * H + h) * S] = d;
}

// backward epilogue: the CLS token's accumulated raw dq / dk / dv -> gradient planes of token 0 (q and k carry 64^-0.5)
__global__ __launch_bounds__(64) void attn_cls_finish_kernel(const float* __restrict__ dcls, int S, int H,
                                                             bf16_t* __restrict__ gh, bf16_t* __restrict__ gl, int gfmt) {
  const int lane = threadIdx.x;
  const int h = blockIdx.x % H, b = blockIdx.x / H;
  const float* a = dcls + (long)blockIdx.x * 192;
  const long HD = (long)H * D;
  const long off = (long)b * S * 3 * HD + (long)h * D + lane;
  const float v[3] = {a[lane] * 0.125f, a[64 + lane] * 0.125f, a[128 + lane]};
#pragma unroll
  for (int part = 0; part < 3; ++part) {
    bf16_t hh, ll;
    split_bf16(v[part], hh, ll);
    if (gfmt == 4) hh = __builtin_bit_cast(unsigned short, (_Float16)v[part]);      // un-clamped fp16 (the fp16 backward)
    gh[off + part * HD] = hh;
    if (gl) gl[off + part * HD] = ll;
  }
}

}  // namespace

int egv_attn_time_mfma_fwd_impl(const bf16_t* qh, const bf16_t* ql, int B, int T, int n, int H, bf16_t* oh, bf16_t* ol, float* lse,
                                float* ws, int out_fmt, int f16, hipStream_t s);
int egv_attn_time_mfma_bwd_impl(const bf16_t* qh, const bf16_t* ql, const bf16_t* doh, const bf16_t* dol, const float* lse,
                                const float* delta, int B, int T, int n, int H, bf16_t* gh, bf16_t* gl, float* dcls, int gfmt, int f16, hipStream_t s);

int egv_attn_time_fwd_impl(const bf16_t* qh, const bf16_t* ql, int B, int T, int n, int H, bf16_t* oh, bf16_t* ol,
                           float* lse, float* ws, int out_fmt, int f16, hipStream_t s) {
  if (T > 16) return EGV_ERR_ARG;
  return egv_attn_time_mfma_fwd_impl(qh, ql, B, T, n, H, oh, ol, lse, ws, out_fmt, f16, s);
}

int egv_attn_time_bwd_impl(const bf16_t* qh, const bf16_t* ql, const bf16_t* doh, const bf16_t* dol, const float* lse,
                           const float* delta, int B, int T, int n, int H, bf16_t* gh, bf16_t* gl, float* dcls, int gfmt, int f16,
                           hipStream_t s) {
  if (T > 16) return EGV_ERR_ARG;
  return egv_attn_time_mfma_bwd_impl(qh, ql, doh, dol, lse, delta, B, T, n, H, gh, gl, dcls, gfmt, f16, s);
}

int egv_attn_cls_combine_impl(const float* ws, int B, int G, int S, int H, bf16_t* oh, bf16_t* ol, float* lse, int out_fmt,
                              hipStream_t s) {
  EGV_LAUNCH(attn_cls_combine_kernel, dim3(B * H), dim3(256), 0, s, ws, G, S, H, oh, ol, lse, out_fmt);
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}

int egv_attn_cls_delta_impl(const bf16_t* oh, const bf16_t* ol, const bf16_t* doh, const bf16_t* dol, int B, int S, int H,
                            float* delta, float* dcls, int o_fmt, int f16, hipStream_t s) {
  EGV_LAUNCH(attn_cls_delta_kernel, dim3(B * H), dim3(64), 0, s, oh, ol, doh, dol, S, H, delta, dcls, o_fmt, f16);
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}

int egv_attn_cls_finish_impl(const float* dcls, int B, int S, int H, bf16_t* gh, bf16_t* gl, int gfmt, hipStream_t s) {
  EGV_LAUNCH(attn_cls_finish_kernel, dim3(B * H), dim3(64), 0, s, dcls, S, H, gh, gl, gfmt);
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}
