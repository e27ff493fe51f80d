// Producers of the f16x2 operand format (csrc/f16x2.h): fp32 -> two fp16 planes [+ a bf16 plane] for weights and test operands, and
// the LayerNorm forward that writes its output directly in that format (model/video_transformer.py:146,156,159 feed the qkv / fc1
// Linears of :103 and :47).  HBM-bound; every lane owns 8 consecutive elements.
#include "f16x2.h"
#include "egovlp_hip.h"

namespace {

// ---- fp32 [rows, cols] -> f16x2 planes (cols % 8 == 0) ----------------------------------------------------------------------------
template <int ROLE>
__device__ __forceinline__ void encode_piece(const float* __restrict__ x, long ldx, int cols, unsigned short* __restrict__ p1,
                                             unsigned short* __restrict__ p2, unsigned short* __restrict__ bf, long ldo, long piece) {
  const int ppr = cols >> 3;                       // 8-element pieces per row
  const int r = (int)(piece / ppr), c = (int)(piece - (long)r * ppr) * 8;
  const f32x4_t a = *(const f32x4_t*)(x + (long)r * ldx + c), b = *(const f32x4_t*)(x + (long)r * ldx + c + 4);
  const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  u32x4_t o1, o2;
  f16x2_encode8<ROLE>(v, o1, o2);
  *(u32x4_t*)(p1 + (long)r * ldo + c) = o1;
  *(u32x4_t*)(p2 + (long)r * ldo + c) = o2;
  if (bf) *(u32x4_t*)(bf + (long)r * ldo + c) = bf16_piece8(v);
}

// fp32 [rows, cols] -> ONE plane of plain fp16, NOT saturating (role 2: a scaled gradient handed to the fp16 backward as fp32 --
// the loss gradient arriving at the last block, or a block input gradient whose plane hand-over was dropped)
__global__ __launch_bounds__(256) void f16_cast_kernel(const float* __restrict__ x, long ldx, int rows, int cols,
                                                       unsigned short* __restrict__ p1, long ldo) {
  const long piece = (long)blockIdx.x * 256 + threadIdx.x;
  if (piece >= (long)rows * (cols >> 3)) return;
  const int ppr = cols >> 3;
  const int r = (int)(piece / ppr), c = (int)(piece - (long)r * ppr) * 8;
  const f32x4_t a = *(const f32x4_t*)(x + (long)r * ldx + c), b = *(const f32x4_t*)(x + (long)r * ldx + c + 4);
  const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  *(u32x4_t*)(p1 + (long)r * ldo + c) = f16_grad_piece8(v);
}

template <int ROLE>
__global__ __launch_bounds__(256) void f16x2_encode_kernel(const float* __restrict__ x, long ldx, int rows, int cols,
                                                           unsigned short* __restrict__ p1, unsigned short* __restrict__ p2,
                                                           unsigned short* __restrict__ bf, long ldo) {
  const long piece = (long)blockIdx.x * 256 + threadIdx.x;
  if (piece >= (long)rows * (cols >> 3)) return;
  encode_piece<ROLE>(x, ldx, cols, p1, p2, bf, ldo, piece);
}

constexpr int ENC_MAX_T = 48;
struct EncodeTable {
  const float* x[ENC_MAX_T];
  unsigned short* p1[ENC_MAX_T];
  unsigned short* p2[ENC_MAX_T];
  int ldx[ENC_MAX_T], ldo[ENC_MAX_T], rows[ENC_MAX_T], cols[ENC_MAX_T];
  int blk_start[ENC_MAX_T + 1];
  int count;
};
template <int ROLE>
__global__ __launch_bounds__(256) void f16x2_encode_multi_kernel(const EncodeTable t) {
  int ti = 0;
  while (ti + 1 < t.count && (int)blockIdx.x >= t.blk_start[ti + 1]) ++ti;
  const long piece = (long)((int)blockIdx.x - t.blk_start[ti]) * 256 + threadIdx.x;
  if (piece >= (long)t.rows[ti] * (t.cols[ti] >> 3)) return;
  encode_piece<ROLE>(t.x[ti], t.ldx[ti], t.cols[ti], t.p1[ti], t.p2[ti], nullptr, t.ldo[ti], piece);
}

// ---- LayerNorm forward -> f16x2 (first-operand role) ------------------------------------------------------------------------------
// One wave per row as in layernorm_fwd_kernel; a lane owns 8 CONSECUTIVE channels per round (piece p = lane + 64 i covers channels
// [8 p, 8 p + 8)): 16-byte stores on all three planes.  cols % 8 == 0, cols <= 1024 (NP rounds of 512 channels).
template <int NP>
__global__ __launch_bounds__(256) void layernorm_fwd_f16x2_kernel(
    const float* __restrict__ x, long ldx, const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int rows,
    int cols, unsigned short* __restrict__ y1, unsigned short* __restrict__ y2, unsigned short* __restrict__ ybf, long ldy,
    float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int np = cols >> 3;
  f32x4_t v[NP][2];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int p = lane + i * 64;
    v[i][0] = v[i][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (p < np) {
      v[i][0] = *(const f32x4_t*)(x + (long)row * ldx + p * 8);
      v[i][1] = *(const f32x4_t*)(x + (long)row * ldx + p * 8 + 4);
      s += (v[i][0][0] + v[i][0][1] + v[i][0][2] + v[i][0][3]) + (v[i][1][0] + v[i][1][1] + v[i][1][2] + v[i][1][3]);
    }
  }
  const float mean = wave_sum(s) / (float)cols;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    if (lane + i * 64 < np) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = v[i][h][e] - mean;
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
  for (int i = 0; i < NP; ++i) {
    const int p = lane + i * 64;
    if (p < np) {
      const f32x4_t g0 = *(const f32x4_t*)(gamma + p * 8), g1 = *(const f32x4_t*)(gamma + p * 8 + 4);
      const f32x4_t b0 = *(const f32x4_t*)(beta + p * 8), b1 = *(const f32x4_t*)(beta + p * 8 + 4);
      float y[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        y[e] = (v[i][0][e] - mean) * rstd * g0[e] + b0[e];
        y[4 + e] = (v[i][1][e] - mean) * rstd * g1[e] + b1[e];
      }
      if (y2) {
        u32x4_t o1, o2;
        f16x2_encode8<0>(y, o1, o2);
        egv_store<EGV_NT_LN>(y1 + (long)row * ldy + p * 8, o1);
        egv_store<EGV_NT_LN>(y2 + (long)row * ldy + p * 8, o2);
      } else {                          // ONE plane of plain fp16: the consumer runs a single fp16 product
        egv_store<EGV_NT_LN>(y1 + (long)row * ldy + p * 8, f16_piece8(y));
      }
      if (ybf) egv_store<EGV_NT_LN>(ybf + (long)row * ldy + p * 8, bf16_piece8(y));
    }
  }
}

}  // namespace

extern "C" int egv_f16x2_encode(const float* x, int64_t ldx, int32_t rows, int32_t cols, uint16_t* p1, uint16_t* p2, egv_bf16* bf,
                                int64_t ldo, int32_t role, void* stream) {
  if (!x || !p1 || (!p2 && role != 2) || rows <= 0 || cols <= 0 || cols % 8 != 0 || ldo % 8 != 0 || ldx % 4 != 0 || role < 0 || role > 2)
    return EGV_ERR_ARG;
  const long pieces = (long)rows * (cols >> 3);
  const dim3 grid((unsigned)((pieces + 255) / 256));
  if (role == 2)
    EGV_LAUNCH(f16_cast_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, (long)ldx, rows, cols, p1, (long)ldo);
  else if (role == 0)
    EGV_LAUNCH(f16x2_encode_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, x, (long)ldx, rows, cols, p1, p2, bf, (long)ldo);
  else
    EGV_LAUNCH(f16x2_encode_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, x, (long)ldx, rows, cols, p1, p2, bf, (long)ldo);
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}

extern "C" int egv_f16x2_encode_multi(int32_t count, const float* const* x, const int64_t* ldx, const int32_t* rows,
                                      const int32_t* cols, uint16_t* const* p1, uint16_t* const* p2, const int64_t* ldo, int32_t role,
                                      void* stream) {
  if (count < 0 || !x || !ldx || !rows || !cols || !p1 || !p2 || !ldo || (role != 0 && role != 1)) return EGV_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  EncodeTable t;
  int nt = 0, nb = 0;
  auto flush = [&]() -> int {
    if (nt == 0) return EGV_OK;
    t.blk_start[nt] = nb;
    t.count = nt;
    if (role == 0) EGV_LAUNCH(f16x2_encode_multi_kernel<0>, dim3(nb), dim3(256), 0, s, t);
    else EGV_LAUNCH(f16x2_encode_multi_kernel<1>, dim3(nb), dim3(256), 0, s, t);
    EGV_CHECK_LAUNCH();
    nt = 0;
    nb = 0;
    return EGV_OK;
  };
  for (int i = 0; i < count; ++i) {
    if (!x[i] || !p1[i] || !p2[i] || rows[i] <= 0 || cols[i] <= 0 || cols[i] % 8 != 0 || ldo[i] % 8 != 0 || ldx[i] % 4 != 0 ||
        ldx[i] > 0x7fffffff || ldo[i] > 0x7fffffff)
      return EGV_ERR_ARG;
    if (nt == ENC_MAX_T) {
      const int rc = flush();
      if (rc) return rc;
    }
    const long pieces = (long)rows[i] * (cols[i] >> 3);
    t.x[nt] = x[i]; t.p1[nt] = p1[i]; t.p2[nt] = p2[i];
    t.ldx[nt] = (int)ldx[i]; t.ldo[nt] = (int)ldo[i]; t.rows[nt] = rows[i]; t.cols[nt] = cols[i];
    t.blk_start[nt] = nb;
    nb += (int)((pieces + 255) / 256);
    ++nt;
  }
  return flush();
}

extern "C" int egv_layernorm_fwd_f16x2(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, int32_t rows,
                                       int32_t cols, uint16_t* y1, uint16_t* y2, egv_bf16* ybf, int64_t ldy, float* mean,
                                       float* rstd, void* stream) {
  if (!x || !gamma || !beta || !y1 || rows <= 0 || cols <= 0 || cols % 8 != 0 || cols > 1024 || ldx % 4 != 0 || ldy % 8 != 0)
    return EGV_ERR_ARG;
  const dim3 grid((rows + 3) / 4), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (cols <= 512)
    EGV_LAUNCH(layernorm_fwd_f16x2_kernel<1>, grid, block, 0, s, x, (long)ldx, gamma, beta, eps, rows, cols, y1, y2, ybf, (long)ldy,
               mean, rstd);
  else
    EGV_LAUNCH(layernorm_fwd_f16x2_kernel<2>, grid, block, 0, s, x, (long)ldx, gamma, beta, eps, rows, cols, y1, y2, ybf, (long)ldy,
               mean, rstd);
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}
