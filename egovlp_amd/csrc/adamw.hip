// Fused multi-tensor AdamW, transformers==4.2.1 semantics (the optimizer configs/pt/egoclip.json:49-54
// + run/train_egoclip.py:73 instantiate):  m,v EMA -> denom = sqrt(v) + eps (eps OUTSIDE the bias
// correction) -> p -= lr*sqrt(1-b2^t)/(1-b1^t) * m/denom -> decoupled weight decay p -= lr*wd*p.
// The reference loops over 327 tensors with ~8 ATen launches each; here the 180.9 M parameters are one
// pass of ~10 launches (pointer tables travel in the kernel arguments), 16 B/lane accesses: the step is
// pure HBM traffic, 16 B read + 12 B written per parameter (+4 B when the bf16 planes are refreshed).
#include "common.h"
#include "egovlp_hip.h"

namespace {

constexpr int MAX_T = 48;        // tensors per launch
constexpr int CHUNK = 16384;     // elements per block

struct AdamTable {
  float* p[MAX_T];
  const float* g[MAX_T];
  float* m[MAX_T];
  float* v[MAX_T];
  bf16_t* wh[MAX_T];
  bf16_t* wl[MAX_T];
  long numel[MAX_T];
  int blk_start[MAX_T + 1];  // first block of each tensor (prefix sum of chunk counts)
  int count;
};

__global__ __launch_bounds__(256) void adamw_kernel(const AdamTable t, float lr, float b1, float b2, float eps, float wd,
                                                    float step_size, float grad_scale, const float* __restrict__ hyper) {
  // hyper (optional, device): {lr, step_size, grad_scale, skip} of THIS step, written by egv_loss_scale_update on the same stream:
  // whether the step happens at all (an overflow of the fp16 backward skips it), the 1 / S that un-scales the gradients and the bias
  // correction at the number of steps actually APPLIED are decided on the device -- the host never waits for a found-inf flag
  if (hyper) {
    if (hyper[3] != 0.f) return;       // skipped step: parameters and moments stay as they are
    lr = hyper[0];
    step_size = hyper[1];
    grad_scale = hyper[2];
  }
  int ti = 0;
  while (ti + 1 < t.count && (int)blockIdx.x >= t.blk_start[ti + 1]) ++ti;   // <= 47 scalar compares
  const long base = (long)((int)blockIdx.x - t.blk_start[ti]) * CHUNK;
  const long n = t.numel[ti];
  float* __restrict__ p = t.p[ti];
  const float* __restrict__ g = t.g[ti];
  float* __restrict__ m = t.m[ti];
  float* __restrict__ v = t.v[ti];
  bf16_t* wh = t.wh[ti];
  bf16_t* wl = t.wl[ti];
  const long end = min(n, base + CHUNK);
  // 16-byte accesses only where all four streams allow them (bucket views / odd offsets fall back to the scalar loop)
  const bool vec = ((n & 3) == 0) && (((((size_t)p) | ((size_t)g) | ((size_t)m) | ((size_t)v)) & 15) == 0) &&
                   (!wh || (((size_t)wh) & 7) == 0) && (!wl || (((size_t)wl) & 7) == 0);
  if (vec) {
    for (long i = base + threadIdx.x * 4; i < end; i += 256 * 4) {
      f32x4_t pv = egv_load<EGV_NT_ADAMW_LD, f32x4_t>(p + i), gv = egv_load<EGV_NT_ADAMW_LD, f32x4_t>(g + i),
               mv = egv_load<EGV_NT_ADAMW_LD, f32x4_t>(m + i), vv = egv_load<EGV_NT_ADAMW_LD, f32x4_t>(v + i);
      bf16_t h[4], l[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float ge = gv[e] * grad_scale;
        mv[e] = mv[e] * b1 + (1.0f - b1) * ge;
        vv[e] = vv[e] * b2 + (1.0f - b2) * ge * ge;
        const float denom = sqrtf(vv[e]) + eps;
        pv[e] = pv[e] - step_size * (mv[e] / denom);
        if (wd > 0.f) pv[e] = pv[e] - lr * wd * pv[e];
        split_bf16(pv[e], h[e], l[e]);
      }
      egv_store<EGV_NT_ADAMW_ST>(p + i, pv);
      egv_store<EGV_NT_ADAMW_ST>(m + i, mv);
      egv_store<EGV_NT_ADAMW_ST>(v + i, vv);
      if (wh) *(u32x2_t*)(wh + i) = (u32x2_t){pack2(h[0], h[1]), pack2(h[2], h[3])};
      if (wl) *(u32x2_t*)(wl + i) = (u32x2_t){pack2(l[0], l[1]), pack2(l[2], l[3])};
    }
  } else {
    for (long i = base + threadIdx.x; i < end; i += 256) {
      const float ge = g[i] * grad_scale;
      const float me = m[i] * b1 + (1.0f - b1) * ge;
      const float ve = v[i] * b2 + (1.0f - b2) * ge * ge;
      float pe = p[i] - step_size * (me / (sqrtf(ve) + eps));
      if (wd > 0.f) pe = pe - lr * wd * pe;
      p[i] = pe;
      m[i] = me;
      v[i] = ve;
      bf16_t h, l;
      split_bf16(pe, h, l);
      if (wh) wh[i] = h;
      if (wl) wl[i] = l;
    }
  }
}

// ---- dynamic loss scale of the fp16 backward ---------------------------------------------------------------------------------------
// (1) any non-finite value in any gradient -> state.found_inf
constexpr int NF_MAX_T = 96;
struct NonfiniteTable {
  const float* g[NF_MAX_T];
  long numel[NF_MAX_T];
  int blk_start[NF_MAX_T + 1];
  int count;
};
__global__ __launch_bounds__(256) void grad_nonfinite_kernel(const NonfiniteTable t, int* __restrict__ state) {
  int ti = 0;
  while (ti + 1 < t.count && (int)blockIdx.x >= t.blk_start[ti + 1]) ++ti;
  const long base = (long)((int)blockIdx.x - t.blk_start[ti]) * (4 * CHUNK);
  const long n = t.numel[ti];
  const float* __restrict__ g = t.g[ti];
  const long end = min(n, base + 4 * CHUNK);
  // |x| as an integer: finite <=> exponent field < 255 <=> (bits & 0x7fffffff) < 0x7f800000; the OR of the "bad" bits of all elements
  unsigned bad = 0u;
  if (((n & 3) == 0) && ((((size_t)g) & 15) == 0)) {
    // four independent 16-byte loads per lane in flight (a pure read stream: 724 MB per step, HBM-bound)
    long i = base + threadIdx.x * 4;
    for (; i + 3 * 1024 < end; i += 4 * 1024) {
      u32x4_t w[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) w[u] = __builtin_bit_cast(u32x4_t, egv_load<EGV_NT_ADAMW_LD, f32x4_t>(g + i + u * 1024));
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) bad |= ((w[u][e] & 0x7fffffffu) >= 0x7f800000u) ? 1u : 0u;
    }
    for (; i < end; i += 1024) {
      const u32x4_t w = __builtin_bit_cast(u32x4_t, egv_load<EGV_NT_ADAMW_LD, f32x4_t>(g + i));
#pragma unroll
      for (int e = 0; e < 4; ++e) bad |= ((w[e] & 0x7fffffffu) >= 0x7f800000u) ? 1u : 0u;
    }
  } else {
    for (long i = base + threadIdx.x; i < end; i += 256) bad |= ((__float_as_uint(g[i]) & 0x7fffffffu) >= 0x7f800000u) ? 1u : 0u;
  }
  if (__any(bad != 0u) && (threadIdx.x & 63) == 0) atomicOr(state + 2, 1);      // one atomic per wave that saw one; normally none
}

// (2) the per-step decision, one thread.  state (8 x 32 bits, DEVICE): [0] float scale S; [1] int good steps since the last change of
// S; [2] int found_inf (set by (1), cleared here); [3] int steps skipped so far; [4..7] float {lr, step_size, 1 / S, skip} = the hyper
// block egv_adamw_multi reads (hyper_out may also point to a caller's own 4 floats: one block per parameter group).
__global__ void loss_scale_update_kernel(int* __restrict__ state, float* __restrict__ hyper_out, float lr, float beta1, float beta2,
                                         int step, int correct_bias, float growth, float backoff, int interval, float max_scale, int advance) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float* fs = (float*)state;
  float skip = fs[7], inv = fs[6];
  if (advance) {
    const float S = fs[0];
    inv = 1.0f / S;                              // the scale the gradients of THIS step carry
    if (state[2] != 0) {
      skip = 1.0f;
      state[3] += 1;
      state[1] = 0;
      fs[0] = fmaxf(S * backoff, 1.0f);
    } else {
      skip = 0.0f;
      const int good = state[1] + 1;
      if (good >= interval) {
        fs[0] = fminf(S * growth, max_scale);
        state[1] = 0;
      } else {
        state[1] = good;
      }
    }
    state[2] = 0;
    fs[6] = inv;
    fs[7] = skip;
  }
  // bias correction at the number of steps actually applied (a skipped step does not count): t = step - skipped
  const int t = max(step - state[3], 1);
  float step_size = lr;
  if (correct_bias) {
    const double bc1 = 1.0 - pow((double)beta1, (double)t);
    const double bc2 = 1.0 - pow((double)beta2, (double)t);
    step_size = (float)((double)lr * sqrt(bc2) / bc1);
  }
  hyper_out[0] = lr;
  hyper_out[1] = step_size;
  hyper_out[2] = inv;
  hyper_out[3] = skip;
}

}  // namespace

extern "C" int egv_grad_nonfinite_multi(int32_t count, const float* const* g, const int64_t* numel, int32_t* state, void* stream) {
  if (count < 0 || !g || !numel || !state) return EGV_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  NonfiniteTable t;
  int nt = 0, nb = 0;
  auto flush = [&]() -> int {
    if (nt == 0) return EGV_OK;
    t.blk_start[nt] = nb;
    t.count = nt;
    EGV_LAUNCH(grad_nonfinite_kernel, dim3(nb), dim3(256), 0, s, t, state);
    EGV_CHECK_LAUNCH();
    nt = 0;
    nb = 0;
    return EGV_OK;
  };
  for (int i = 0; i < count; ++i) {
    if (numel[i] <= 0) continue;
    if (!g[i]) return EGV_ERR_ARG;
    if (nt == NF_MAX_T) {
      const int rc = flush();
      if (rc) return rc;
    }
    t.g[nt] = g[i];
    t.numel[nt] = numel[i];
    t.blk_start[nt] = nb;
    nb += (int)((numel[i] + 4 * CHUNK - 1) / (4 * CHUNK));
    ++nt;
  }
  return flush();
}

extern "C" int egv_loss_scale_update(int32_t* state, float* hyper_out, float lr, float beta1, float beta2, int32_t step,
                                     int32_t correct_bias, float growth_factor, float backoff_factor, int32_t growth_interval,
                                     float max_scale, int32_t advance, void* stream) {
  if (!state || step < 1 || !(growth_factor >= 1.0f) || !(backoff_factor > 0.f && backoff_factor <= 1.0f) || growth_interval < 1 ||
      !(max_scale >= 1.0f))
    return EGV_ERR_ARG;
  EGV_LAUNCH(loss_scale_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, hyper_out ? hyper_out : (float*)state + 4, lr,
             beta1, beta2, step, correct_bias, growth_factor, backoff_factor, growth_interval, max_scale, advance);
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}

extern "C" int egv_adamw_multi(int32_t count, float* const* p, const float* const* g, float* const* m, float* const* v,
                               egv_bf16* const* w_hi, egv_bf16* const* w_lo, const int64_t* numel, float lr,
                               float beta1, float beta2, float eps, float weight_decay, int32_t step,
                               int32_t correct_bias, float grad_scale, const float* hyper_dev, void* stream) {
  if (count < 0 || !p || !g || !m || !v || !numel || step < 1) return EGV_ERR_ARG;
  float step_size = lr;
  if (correct_bias) {
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    step_size = (float)((double)lr * sqrt(bc2) / bc1);
  }
  hipStream_t s = (hipStream_t)stream;
  AdamTable t;
  int nt = 0, nb = 0;
  auto flush = [&]() -> int {
    if (nt == 0) return EGV_OK;
    t.blk_start[nt] = nb;
    t.count = nt;
    EGV_LAUNCH(adamw_kernel, dim3(nb), dim3(256), 0, s, t, lr, beta1, beta2, eps, weight_decay, step_size,
                       grad_scale, hyper_dev);
    EGV_CHECK_LAUNCH();
    nt = 0;
    nb = 0;
    return EGV_OK;
  };
  for (int i = 0; i < count; ++i) {
    if (numel[i] <= 0) continue;
    if (!p[i] || !g[i] || !m[i] || !v[i]) return EGV_ERR_ARG;
    if (nt == MAX_T) {
      const int rc = flush();
      if (rc) return rc;
    }
    t.p[nt] = p[i]; t.g[nt] = g[i]; t.m[nt] = m[i]; t.v[nt] = v[i];
    t.wh[nt] = w_hi ? w_hi[i] : nullptr;
    t.wl[nt] = w_lo ? w_lo[i] : nullptr;
    t.numel[nt] = numel[i];
    t.blk_start[nt] = nb;
    nb += (int)((numel[i] + CHUNK - 1) / CHUNK);
    ++nt;
  }
  return flush();
}
