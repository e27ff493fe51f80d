// LayerNorm forward / backward for rows of <= 1024 channels (ViT-B 768, ViT-L 1024, DistilBERT 768).
// HBM-bound: one wave64 owns a row, the row lives in registers (<= 4 float4 per lane), statistics by
// wave shuffles (two-pass mean / centred variance, matching the fp32 reference's numerics), outputs
// written once as split-bf16 planes (the next GEMM's operand format) and/or fp32.
#include "common.h"
#include "f16x2.h"
#include "egovlp_hip.h"

namespace {

constexpr int MAXV = 4;  // float4 per lane -> cols <= 1024

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ xadd, long ldx, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, int rows, int cols, float* __restrict__ sum_out,
    bf16_t* __restrict__ yhi, bf16_t* __restrict__ ylo, float* __restrict__ yf, long ldy, float* __restrict__ mean_out,
    float* __restrict__ rstd_out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv = cols / 4;
  f32x4_t v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c4 = lane + i * 64;
    v[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (c4 < nv) {
      v[i] = *(const f32x4_t*)(x + (long)row * ldx + c4 * 4);
      if (xadd) v[i] += *(const f32x4_t*)(xadd + (long)row * ldx + c4 * 4);
      if (sum_out) *(f32x4_t*)(sum_out + (long)row * ldx + c4 * 4) = v[i];
      s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
  }
  const float mean = wave_sum(s) / (float)cols;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c4 = lane + i * 64;
    if (c4 < nv) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[i][e] - mean;
        q += d * d;
      }
    }
  }
  const float var = wave_sum(q) / (float)cols;
  const float rstd = 1.0f / sqrtf(var + eps);
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c4 = lane + i * 64;
    if (c4 < nv) {
      const f32x4_t g = *(const f32x4_t*)(gamma + c4 * 4);
      const f32x4_t b = *(const f32x4_t*)(beta + c4 * 4);
      f32x4_t y;
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
      if (yf) egv_store<EGV_NT_LN>(yf + (long)row * ldy + c4 * 4, y);
      if (yhi) {
        bf16_t h[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split_bf16(y[e], h[e], l[e]);
        egv_store<EGV_NT_LN>(yhi + (long)row * ldy + c4 * 4, (u32x2_t){pack2(h[0], h[1]), pack2(h[2], h[3])});
        if (ylo) egv_store<EGV_NT_LN>(ylo + (long)row * ldy + c4 * 4, (u32x2_t){pack2(l[0], l[1]), pack2(l[2], l[3])});
      }
    }
  }
}

// backward: each wave walks rows wave_id, wave_id + nwaves, ...; per-lane partial dgamma/dbeta stay in
// registers; one LDS reduction per block at the end -> work[block][2][cols]; a second kernel sums blocks.
// HBM-bound (dy + x [+ add1 + add2] in, dx + planes out: 270 .. 424 MB per launch at M = 25 120, D = 768), so what matters is
// bytes in flight: EVERY input stream of a row (the residual-gradient addends too, which the first version fetched only after
// the two wave reductions) is requested at the top of the row, the kernel is instantiated per row width (NV float4 per
// lane: 3 for D = 768, so no register is spent on a fourth that is never used) and stays under 128 VGPRs, and the grid is four
// waves per SIMD (1024 blocks): 16 waves x 12 KiB requested per CU where the old shape had 8 x 6 KiB.
template <int NV>
__global__ __launch_bounds__(256, NV <= 3 ? 4 : 2) void layernorm_bwd_kernel(
    const float* __restrict__ dy, const bf16_t* __restrict__ dyh, const bf16_t* __restrict__ dyl, long lddy,
    const float* __restrict__ x, long ldx, const float* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ rstd, int rows, int cols, const float* __restrict__ add1,
    const float* __restrict__ add2, float* __restrict__ dx, long lddx, bf16_t* __restrict__ dxh,
    bf16_t* __restrict__ dxl, float* __restrict__ work, float* __restrict__ dgamma, float* __restrict__ dbeta, int dx_fmt) {
  __shared__ float red[2][4][NV * 256];   // [dgamma/dbeta][wave][col]
  if (blockIdx.x == 0) {   // the reduce kernel (next launch on the stream) accumulates into these with atomics
    for (int c = threadIdx.x; c < cols; c += 256) {
      if (dgamma) dgamma[c] = 0.f;
      if (dbeta) dbeta[c] = 0.f;
    }
  }
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nv = cols / 4;
  f32x4_t dg[NV], db[NV], g[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    dg[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    db[i] = dg[i];
    g[i] = dg[i];
    const int c4 = lane + i * 64;
    if (c4 < nv) g[i] = *(const f32x4_t*)(gamma + c4 * 4);
  }
  const float inv_cols = 1.0f / (float)cols;
  const bool full = (nv == NV * 64);      // D = 768 with NV = 3, D = 1024 with NV = 4: no lane is ever out of range
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    f32x4_t gy[NV], xh[NV], ad[NV];
    // ---- every load of the row, back to back
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c4 = lane + i * 64;
      gy[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      xh[i] = gy[i];
      ad[i] = gy[i];
      if (full || c4 < nv) {
        if (dyh && (dx_fmt & 2)) {   // dy as ONE plane of un-clamped fp16 (the dgrad GEMMs of the fp16 backward write it so): 2 B per element
          const u32x2_t a = *(const u32x2_t*)(dyh + (long)row * lddy + c4 * 4);
          float a0, a1, a2, a3;
          f16x2_unpack(a[0], a0, a1);
          f16x2_unpack(a[1], a2, a3);
          gy[i] = (f32x4_t){a0, a1, a2, a3};
        } else if (dyh) {   // dy as split-bf16 planes (written by the dgrad GEMM epilogue): half the bytes of fp32
          const u32x2_t a = *(const u32x2_t*)(dyh + (long)row * lddy + c4 * 4);
          gy[i] = (f32x4_t){__uint_as_float(a[0] << 16), __uint_as_float(a[0] & 0xffff0000u), __uint_as_float(a[1] << 16),
                            __uint_as_float(a[1] & 0xffff0000u)};
          if (dyl) {
            const u32x2_t b = *(const u32x2_t*)(dyl + (long)row * lddy + c4 * 4);
            gy[i] += (f32x4_t){__uint_as_float(b[0] << 16), __uint_as_float(b[0] & 0xffff0000u), __uint_as_float(b[1] << 16),
                               __uint_as_float(b[1] & 0xffff0000u)};
          }
        } else {
          gy[i] = *(const f32x4_t*)(dy + (long)row * lddy + c4 * 4);
        }
        xh[i] = *(const f32x4_t*)(x + (long)row * ldx + c4 * 4);
        if (add1) ad[i] = *(const f32x4_t*)(add1 + (long)row * lddx + c4 * 4);
        if (add2) ad[i] += *(const f32x4_t*)(add2 + (long)row * lddx + c4 * 4);
      }
    }
    const float mu = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c4 = lane + i * 64;
      if (full || c4 < nv) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = gy[i][e];
          xh[i][e] = (xh[i][e] - mu) * rs;
          gy[i][e] = d * g[i][e];
          dg[i][e] += d * xh[i][e];
          db[i][e] += d;
          s1 += gy[i][e];
          s2 += gy[i][e] * xh[i][e];
        }
      }
    }
    const float c1 = wave_sum(s1) * inv_cols;
    const float c2 = wave_sum(s2) * inv_cols;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c4 = lane + i * 64;
      if (full || c4 < nv) {
        f32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rs * (gy[i][e] - c1 - xh[i][e] * c2) + ad[i][e];
        egv_store<EGV_NT_LN>(dx + (long)row * lddx + c4 * 4, o);
        if (dxh && (dx_fmt & 1)) {   // ... as ONE plane of un-clamped fp16 (the fp16 backward: a scaled gradient; overflow -> inf -> skipped step)
          egv_store<EGV_NT_LN>(dxh + (long)row * cols + c4 * 4, (u32x2_t){f16_grad_pack2(o[0], o[1]), f16_grad_pack2(o[2], o[3])});
        } else if (dxh) {   // the same gradient as the next GEMM's operand (row-major split-bf16 planes, ld = cols)
          uint32_t h0, h1, l0, l1;
          split_bf16x2(o[0], o[1], h0, l0);
          split_bf16x2(o[2], o[3], h1, l1);
          egv_store<EGV_NT_LN>(dxh + (long)row * cols + c4 * 4, (u32x2_t){h0, h1});
          if (dxl) egv_store<EGV_NT_LN>(dxl + (long)row * cols + c4 * 4, (u32x2_t){l0, l1});
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = lane + i * 64;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[0][wave][c4 * 4 + e] = dg[i][e];
      red[1][wave][c4 * 4 + e] = db[i][e];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < cols; c += 256) {
    work[((long)blockIdx.x * 2 + 0) * cols + c] = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
    work[((long)blockIdx.x * 2 + 1) * cols + c] = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
  }
}

// second stage: work[parts][2][cols] -> dgamma/dbeta (zeroed by the launcher).  grid (cols/64, ceil(parts/64)): a block
// sums 64 parts (16 per wave, all loads in flight at once), one atomicAdd per column per block.
struct LnReduceTable {          // up to four LayerNorm backwards reduced by ONE launch (blockIdx.z picks the entry)
  const float* work[4];
  float* dgamma[4];
  float* dbeta[4];
};
__global__ __launch_bounds__(256) void layernorm_bwd_reduce_kernel(const LnReduceTable t, int parts, int cols) {
  const float* __restrict__ work = t.work[blockIdx.z];
  float* __restrict__ dgamma = t.dgamma[blockIdx.z];
  float* __restrict__ dbeta = t.dbeta[blockIdx.z];
  __shared__ float red[2][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int p0 = blockIdx.y * 64 + wave * 16;
  float a = 0.f, b = 0.f;
  if (c < cols) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int p = p0 + i;
      if (p < parts) {
        a += work[((long)p * 2 + 0) * cols + c];
        b += work[((long)p * 2 + 1) * cols + c];
      }
    }
  }
  red[0][wave][lane] = a;
  red[1][wave][lane] = b;
  __syncthreads();
  if (wave == 0 && c < cols) {
    if (dgamma) atomicAdd(dgamma + c, red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane]);
    if (dbeta) atomicAdd(dbeta + c, red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane]);
  }
}

}  // namespace

extern "C" int egv_layernorm_fwd(const float* x, const float* x_add, int64_t ldx, const float* gamma,
                                 const float* beta, float eps, int32_t rows, int32_t cols, float* sum_out,
                                 egv_bf16* y_hi, egv_bf16* y_lo, float* y_f32, int64_t ldy, float* mean, float* rstd,
                                 void* stream) {
  if (!x || !gamma || !beta || rows <= 0 || cols <= 0 || cols % 4 != 0 || cols > MAXV * 256) return EGV_ERR_ARG;
  if (!y_hi && !y_f32) return EGV_ERR_ARG;
  EGV_LAUNCH(layernorm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, x_add, ldx,
                     gamma, beta, eps, rows, cols, sum_out, y_hi, y_lo, y_f32, ldy, mean, rstd);
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}

extern "C" int egv_layernorm_bwd_parts(int32_t rows) {
  const int b = (rows + 3) / 4;
  return b < 1024 ? b : 1024;       // 1024 blocks of 4 waves = 16 waves per CU (four per SIMD)
}

extern "C" int egv_layernorm_bwd(const float* dy, const egv_bf16* dy_hi, const egv_bf16* dy_lo, int64_t lddy,
                                 const float* x, int64_t ldx, const float* gamma,
                                 const float* mean, const float* rstd, int32_t rows, int32_t cols, const float* add1,
                                 const float* add2, float* dx, int64_t lddx, egv_bf16* dx_hi, egv_bf16* dx_lo,
                                 float* dgamma, float* dbeta, float* work, void* stream) {
  return egv_layernorm_bwd_fmt(dy, dy_hi, dy_lo, lddy, x, ldx, gamma, mean, rstd, rows, cols, add1, add2, dx, lddx, dx_hi, dx_lo, 0,
                               dgamma, dbeta, work, stream);
}

extern "C" int egv_layernorm_bwd_fmt(const float* dy, const egv_bf16* dy_hi, const egv_bf16* dy_lo, int64_t lddy,
                                     const float* x, int64_t ldx, const float* gamma,
                                     const float* mean, const float* rstd, int32_t rows, int32_t cols, const float* add1,
                                     const float* add2, float* dx, int64_t lddx, egv_bf16* dx_hi, egv_bf16* dx_lo, int32_t dx_fmt,
                                     float* dgamma, float* dbeta, float* work, void* stream) {
  const int rc = egv_layernorm_bwd_partial(dy, dy_hi, dy_lo, lddy, x, ldx, gamma, mean, rstd, rows, cols, add1, add2, dx, lddx, dx_hi, dx_lo,
                                           dx_fmt, dgamma, dbeta, work, stream);
  if (rc) return rc;
  const float* w1[1] = {work};
  float* g1[1] = {dgamma};
  float* b1[1] = {dbeta};
  return egv_layernorm_bwd_reduce(1, w1, rows, cols, g1, b1, stream);
}

// second stage for up to four LayerNorm backwards of the same (rows, cols) in ONE launch: work[i] -> dgamma[i] / dbeta[i]
extern "C" int egv_layernorm_bwd_reduce(int32_t count, const float* const* work, int32_t rows, int32_t cols, float* const* dgamma,
                                        float* const* dbeta, void* stream) {
  if (count < 1 || count > 4 || !work || !dgamma || !dbeta || rows <= 0 || cols <= 0) return EGV_ERR_ARG;
  LnReduceTable t = {};
  for (int i = 0; i < count; ++i) {
    if (!work[i]) return EGV_ERR_ARG;
    t.work[i] = work[i]; t.dgamma[i] = dgamma[i]; t.dbeta[i] = dbeta[i];
  }
  const int parts = egv_layernorm_bwd_parts(rows);
  EGV_LAUNCH(layernorm_bwd_reduce_kernel, dim3((cols + 63) / 64, (parts + 63) / 64, count), dim3(256), 0, (hipStream_t)stream, t, parts, cols);
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}

// first stage only: dx (+ planes), per-block partial sums of dgamma / dbeta -> work; dgamma / dbeta are ZEROED (the reduce accumulates)
extern "C" int egv_layernorm_bwd_partial(const float* dy, const egv_bf16* dy_hi, const egv_bf16* dy_lo, int64_t lddy,
                                         const float* x, int64_t ldx, const float* gamma,
                                         const float* mean, const float* rstd, int32_t rows, int32_t cols, const float* add1,
                                         const float* add2, float* dx, int64_t lddx, egv_bf16* dx_hi, egv_bf16* dx_lo, int32_t dx_fmt,
                                         float* dgamma, float* dbeta, float* work, void* stream) {
  if ((!dy && !dy_hi) || !x || !gamma || !mean || !rstd || !dx || !work) return EGV_ERR_ARG;
  if (dx_fmt < 0 || dx_fmt > 3 || ((dx_fmt & 1) && dx_lo)) return EGV_ERR_ARG;
  if ((dx_fmt & 2) && (!dy_hi || dy_lo || dy)) return EGV_ERR_ARG;          // an fp16 dy is ONE plane, given in dy_hi
  if (rows <= 0 || cols <= 0 || cols % 4 != 0 || cols > MAXV * 256) return EGV_ERR_ARG;
  const int parts = egv_layernorm_bwd_parts(rows);
  hipStream_t s = (hipStream_t)stream;
  const int nv64 = (cols / 4 + 63) / 64;      // float4 per lane
#define EGV_LN_BWD(NV)                                                                                                      \
  EGV_LAUNCH(layernorm_bwd_kernel<NV>, dim3(parts), dim3(256), 0, s, dy, dy_hi, dy_lo, lddy, x, ldx, gamma, mean, rstd, rows, \
             cols, add1, add2, dx, lddx, dx_hi, dx_lo, work, dgamma, dbeta, dx_fmt)
  if (nv64 <= 1) EGV_LN_BWD(1);
  else if (nv64 == 2) EGV_LN_BWD(2);
  else if (nv64 == 3) EGV_LN_BWD(3);
  else EGV_LN_BWD(4);
#undef EGV_LN_BWD
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}
