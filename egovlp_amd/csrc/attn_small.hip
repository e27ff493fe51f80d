// The two "thin" attention shapes of divided space-time attention, on the vector ALU in exact fp32:
//   * time attention (model/video_transformer.py:114-124, '(b n) f d'): per (b, location, head) only
//     T queries x (CLS + T) keys (4 x 5 at T=4, 16 x 17 at T=16) -- far below an MFMA tile; the kernel is
//     HBM-bound (it streams the whole qkv buffer once), one wave64 per group with lane = head channel d.
//   * the CLS query row (:109-112): 1 query x all S keys per (b, head): a GEMV-shaped, HBM-bound pass
//     over all K and V rows, 16 lanes per key (float4 each) so every row is one coalesced 256-B read.
// Both read q/k/v straight out of the fused qkv buffer [B,S,3,H,64] and write split-bf16 planes.
#include "common.h"
#include "egovlp_hip.h"

namespace {

constexpr int D = 64;

__device__ __forceinline__ float sum16(float v) {  // reduce over the 16 lanes of a key group
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}

// ------------------------------------------------------------------------------------------- time fwd
template <int TMAX>
__global__ __launch_bounds__(256) void attn_time_fwd_kernel(const float* __restrict__ qkv, int B, int T, int n, int H,
                                                            bf16_t* __restrict__ out_hi, bf16_t* __restrict__ out_lo,
                                                            float* __restrict__ lse) {
  const int lane = threadIdx.x & 63;
  const long gid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long ngroups = (long)B * n * H;
  if (gid >= ngroups) return;
  const int h = (int)(gid % H);
  const long r = gid / H;
  const int i = (int)(r % n);
  const int b = (int)(r / n);
  const long S = 1 + (long)T * n;
  const long HD = (long)H * D;
  const long ts = 3 * HD;
  const float* base = qkv + (long)b * S * ts + (long)h * D + lane;
  const float kc = base[HD], vc = base[2 * HD];
  float q[TMAX], k[TMAX], v[TMAX];
#pragma unroll
  for (int f = 0; f < TMAX; ++f) {
    q[f] = k[f] = v[f] = 0.f;
    if (f < T) {
      const float* p = base + (1 + (long)f * n + i) * ts;
      q[f] = p[0] * 0.125f;
      k[f] = p[HD];
      v[f] = p[2 * HD];
    }
  }
#pragma unroll
  for (int f = 0; f < TMAX; ++f) {
    if (f < T) {
      float s[TMAX + 1];
      s[0] = wave_sum(q[f] * kc);
      float m = s[0];
#pragma unroll
      for (int j = 0; j < TMAX; ++j) {
        s[j + 1] = -3e38f;
        if (j < T) {
          s[j + 1] = wave_sum(q[f] * k[j]);
          m = fmaxf(m, s[j + 1]);
        }
      }
      float p0 = __expf(s[0] - m);
      float l = p0;
      float o = p0 * vc;
#pragma unroll
      for (int j = 0; j < TMAX; ++j) {
        if (j < T) {
          const float pj = __expf(s[j + 1] - m);
          l += pj;
          o += pj * v[j];
        }
      }
      o /= l;
      const long tok = (long)b * S + 1 + (long)f * n + i;
      bf16_t hh, ll;
      split_bf16(o, hh, ll);
      out_hi[tok * HD + (long)h * D + lane] = hh;
      if (out_lo) out_lo[tok * HD + (long)h * D + lane] = ll;
      if (lane == 0 && lse) lse[((long)b * H + h) * S + 1 + (long)f * n + i] = m + __logf(l);
    }
  }
}

// ------------------------------------------------------------------------------------------- time bwd
// one workgroup = (b, h, 16 consecutive locations); each wave walks 4 locations and keeps the CLS-key
// gradient in registers, so the CLS rows receive one atomicAdd per workgroup and channel.
template <int TMAX>
__global__ __launch_bounds__(256) void attn_time_bwd_kernel(const float* __restrict__ qkv,
                                                            const float* __restrict__ d_out,
                                                            const float* __restrict__ lse, int B, int T, int n, int H,
                                                            float* __restrict__ dqkv) {
  __shared__ float red[2][4][D];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int chunks = (n + 15) / 16;
  const int ic = blockIdx.x % chunks;
  const int bh = blockIdx.x / chunks;
  const int h = bh % H, b = bh / H;
  const long S = 1 + (long)T * n;
  const long HD = (long)H * D;
  const long ts = 3 * HD;
  const float* base = qkv + (long)b * S * ts + (long)h * D + lane;
  float* dbase = dqkv + (long)b * S * ts + (long)h * D + lane;
  const float kc = base[HD], vc = base[2 * HD];
  float dkc = 0.f, dvc = 0.f;
  for (int ii = 0; ii < 4; ++ii) {
    const int i = ic * 16 + wave * 4 + ii;
    if (i >= n) break;
    float q[TMAX], k[TMAX], v[TMAX], go[TMAX], dk[TMAX], dv[TMAX];
#pragma unroll
    for (int f = 0; f < TMAX; ++f) {
      q[f] = k[f] = v[f] = go[f] = dk[f] = dv[f] = 0.f;
      if (f < T) {
        const long tok = 1 + (long)f * n + i;
        const float* p = base + tok * ts;
        q[f] = p[0] * 0.125f;
        k[f] = p[HD];
        v[f] = p[2 * HD];
        go[f] = d_out[((long)b * S + tok) * HD + (long)h * D + lane];
      }
    }
#pragma unroll
    for (int f = 0; f < TMAX; ++f) {
      if (f < T) {
        const float L = lse[((long)b * H + h) * S + 1 + (long)f * n + i];
        float p[TMAX + 1], dp[TMAX + 1];
        p[0] = __expf(wave_sum(q[f] * kc) - L);
        dp[0] = wave_sum(go[f] * vc);
        float delta = p[0] * dp[0];
#pragma unroll
        for (int j = 0; j < TMAX; ++j) {
          p[j + 1] = dp[j + 1] = 0.f;
          if (j < T) {
            p[j + 1] = __expf(wave_sum(q[f] * k[j]) - L);
            dp[j + 1] = wave_sum(go[f] * v[j]);
            delta += p[j + 1] * dp[j + 1];
          }
        }
        const float ds0 = p[0] * (dp[0] - delta);
        float dq = ds0 * kc;
        dkc += ds0 * q[f];
        dvc += p[0] * go[f];
#pragma unroll
        for (int j = 0; j < TMAX; ++j) {
          if (j < T) {
            const float ds = p[j + 1] * (dp[j + 1] - delta);
            dq += ds * k[j];
            dk[j] += ds * q[f];
            dv[j] += p[j + 1] * go[f];
          }
        }
        dbase[(1 + (long)f * n + i) * ts] = dq * 0.125f;
      }
    }
#pragma unroll
    for (int f = 0; f < TMAX; ++f) {
      if (f < T) {
        float* p = dbase + (1 + (long)f * n + i) * ts;
        p[HD] = dk[f];
        p[2 * HD] = dv[f];
      }
    }
  }
  red[0][wave][lane] = dkc;
  red[1][wave][lane] = dvc;
  __syncthreads();
  if (wave == 0) {
    atomicAdd(dbase + HD, red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane]);
    atomicAdd(dbase + 2 * HD, red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane]);
  }
}

// ------------------------------------------------------------------------------------------- CLS fwd
// one workgroup per (b, h): 16 key-groups (4 per wave) x 16 lanes (float4 of the 64-d row each)
__global__ __launch_bounds__(256) void attn_cls_fwd_kernel(const float* __restrict__ qkv, int B, int S, int H,
                                                           bf16_t* __restrict__ out_hi, bf16_t* __restrict__ out_lo,
                                                           float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sc = (float*)smem_raw;        // [S] scores -> probabilities
  float* red = sc + ((S + 3) & ~3);    // [16][64] partial outputs, then scalars
  const int tid = threadIdx.x;
  const int kg = tid >> 4;             // key group 0..15
  const int l16 = tid & 15;
  const int h = blockIdx.x % H, b = blockIdx.x / H;
  const long HD = (long)H * D;
  const long ts = 3 * HD;
  const float* base = qkv + (long)b * S * ts + (long)h * D + l16 * 4;
  f32x4_t q = *(const f32x4_t*)base;
  q *= 0.125f;
  float mloc = -3e38f;
  for (int j = kg; j < S; j += 16) {
    const f32x4_t kv = *(const f32x4_t*)(base + (long)j * ts + HD);
    const float s = sum16(q[0] * kv[0] + q[1] * kv[1] + q[2] * kv[2] + q[3] * kv[3]);
    if (l16 == 0) sc[j] = s;
    mloc = fmaxf(mloc, s);
  }
  // block max
  mloc = wave_max(mloc);
  __shared__ float wred[8];
  if ((tid & 63) == 0) wred[tid >> 6] = mloc;
  __syncthreads();
  const float m = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
  float lloc = 0.f;
  for (int j = tid; j < S; j += 256) {
    const float p = __expf(sc[j] - m);
    sc[j] = p;
    lloc += p;
  }
  lloc = wave_sum(lloc);
  if ((tid & 63) == 0) wred[4 + (tid >> 6)] = lloc;
  __syncthreads();
  const float l = wred[4] + wred[5] + wred[6] + wred[7];
  f32x4_t o = {0.f, 0.f, 0.f, 0.f};
  for (int j = kg; j < S; j += 16) {
    const f32x4_t vv = *(const f32x4_t*)(base + (long)j * ts + 2 * HD);
    o += sc[j] * vv;
  }
  *(f32x4_t*)(red + kg * 64 + l16 * 4) = o;
  __syncthreads();
  if (tid < 64) {
    float acc = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) acc += red[g * 64 + tid];
    acc /= l;
    bf16_t hh, ll;
    split_bf16(acc, hh, ll);
    const long o_off = (long)b * S * HD + (long)h * D + tid;  // token 0
    out_hi[o_off] = hh;
    if (out_lo) out_lo[o_off] = ll;
    if (tid == 0 && lse) lse[((long)b * H + h) * S] = m + __logf(l);
  }
}

// ------------------------------------------------------------------------------------------- CLS bwd
// dq_cls is stored; dk_j / dv_j are ADDED (plain read-modify-write: every (b,h,j) element has exactly one
// owner here and the patch kernels that wrote / atomically accumulated the same rows ran earlier on the
// same stream).
__global__ __launch_bounds__(256) void attn_cls_bwd_kernel(const float* __restrict__ qkv,
                                                           const float* __restrict__ d_out,
                                                           const float* __restrict__ lse, int B, int S, int H,
                                                           float* __restrict__ dqkv) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* pj = (float*)smem_raw;            // [S]
  float* dpj = pj + ((S + 3) & ~3);        // [S]
  float* red = dpj + ((S + 3) & ~3);       // [16][64]
  __shared__ float wred[4];
  const int tid = threadIdx.x;
  const int kg = tid >> 4, l16 = tid & 15;
  const int h = blockIdx.x % H, b = blockIdx.x / H;
  const long HD = (long)H * D;
  const long ts = 3 * HD;
  const float* base = qkv + (long)b * S * ts + (long)h * D + l16 * 4;
  float* dbase = dqkv + (long)b * S * ts + (long)h * D + l16 * 4;
  f32x4_t q = *(const f32x4_t*)base;
  q *= 0.125f;
  const f32x4_t go = *(const f32x4_t*)(d_out + (long)b * S * HD + (long)h * D + l16 * 4);
  const float L = lse[((long)b * H + h) * S];
  float dloc = 0.f;
  for (int j = kg; j < S; j += 16) {
    const f32x4_t kv = *(const f32x4_t*)(base + (long)j * ts + HD);
    const f32x4_t vv = *(const f32x4_t*)(base + (long)j * ts + 2 * HD);
    const float s = sum16(q[0] * kv[0] + q[1] * kv[1] + q[2] * kv[2] + q[3] * kv[3]);
    const float dp = sum16(go[0] * vv[0] + go[1] * vv[1] + go[2] * vv[2] + go[3] * vv[3]);
    const float p = __expf(s - L);
    if (l16 == 0) {
      pj[j] = p;
      dpj[j] = dp;
      dloc += p * dp;
    }
  }
  dloc = wave_sum(dloc);
  if ((tid & 63) == 0) wred[tid >> 6] = dloc;
  __syncthreads();
  const float delta = wred[0] + wred[1] + wred[2] + wred[3];
  f32x4_t dq = {0.f, 0.f, 0.f, 0.f};
  for (int j = kg; j < S; j += 16) {
    const float p = pj[j];
    const float ds = p * (dpj[j] - delta);
    const f32x4_t kv = *(const f32x4_t*)(base + (long)j * ts + HD);
    dq += ds * kv;
    float* dk = dbase + (long)j * ts + HD;
    float* dv = dbase + (long)j * ts + 2 * HD;
    *(f32x4_t*)dk = *(const f32x4_t*)dk + ds * q;
    *(f32x4_t*)dv = *(const f32x4_t*)dv + p * go;
  }
  *(f32x4_t*)(red + kg * 64 + l16 * 4) = dq;
  __syncthreads();
  if (tid < 64) {
    float acc = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) acc += red[g * 64 + tid];
    dqkv[(long)b * S * ts + (long)h * D + tid] = acc * 0.125f;
  }
}

}  // namespace

int egv_attn_time_fwd_impl(const float* qkv, int B, int T, int n, int H, bf16_t* oh, bf16_t* ol, float* lse,
                           hipStream_t s) {
  const long ngroups = (long)B * n * H;
  const dim3 grid((unsigned)((ngroups + 3) / 4)), block(256);
  if (T <= 4)
    EGV_LAUNCH(attn_time_fwd_kernel<4>, grid, block, 0, s, qkv, B, T, n, H, oh, ol, lse);
  else if (T <= 8)
    EGV_LAUNCH(attn_time_fwd_kernel<8>, grid, block, 0, s, qkv, B, T, n, H, oh, ol, lse);
  else if (T <= 16)
    EGV_LAUNCH(attn_time_fwd_kernel<16>, grid, block, 0, s, qkv, B, T, n, H, oh, ol, lse);
  else
    return EGV_ERR_ARG;
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}

int egv_attn_time_bwd_impl(const float* qkv, const float* d_out, const float* lse, int B, int T, int n, int H,
                           float* dqkv, hipStream_t s) {
  const dim3 grid((unsigned)(B * H * ((n + 15) / 16))), block(256);
  if (T <= 4)
    EGV_LAUNCH(attn_time_bwd_kernel<4>, grid, block, 0, s, qkv, d_out, lse, B, T, n, H, dqkv);
  else if (T <= 8)
    EGV_LAUNCH(attn_time_bwd_kernel<8>, grid, block, 0, s, qkv, d_out, lse, B, T, n, H, dqkv);
  else if (T <= 16)
    EGV_LAUNCH(attn_time_bwd_kernel<16>, grid, block, 0, s, qkv, d_out, lse, B, T, n, H, dqkv);
  else
    return EGV_ERR_ARG;
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}

int egv_attn_cls_fwd_impl(const float* qkv, int B, int S, int H, bf16_t* oh, bf16_t* ol, float* lse, hipStream_t s) {
  const size_t lds = (size_t)(((S + 3) & ~3) + 16 * 64) * sizeof(float);
  EGV_LAUNCH(attn_cls_fwd_kernel, dim3(B * H), dim3(256), lds, s, qkv, B, S, H, oh, ol, lse);
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}

int egv_attn_cls_bwd_impl(const float* qkv, const float* d_out, const float* lse, int B, int S, int H, float* dqkv,
                          hipStream_t s) {
  const size_t lds = (size_t)(2 * ((S + 3) & ~3) + 16 * 64) * sizeof(float);
  EGV_LAUNCH(attn_cls_bwd_kernel, dim3(B * H), dim3(256), lds, s, qkv, d_out, lse, B, S, H, dqkv);
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}
