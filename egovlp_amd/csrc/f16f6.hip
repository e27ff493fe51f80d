// Producers of the f16f6 operand format (csrc/f6.h): fp32 -> (fp16 plane, MXFP6 slot plane[, bf16 plane]) for weights and test
// operands, and the LayerNorm forward that writes its output directly in that format (model/video_transformer.py:146,156,159 feed
// the qkv / fc1 Linears of :103 and :47).  HBM-bound; every lane owns 8 consecutive elements, the four lanes of a quad one MX block.
#include "f6.h"
#include "egovlp_hip.h"

namespace {

// ---- fp32 [rows, cols] -> f16f6 planes (cols % 32 == 0) -----------------------------------------------------------------------
__device__ __forceinline__ void encode_piece(const float* __restrict__ x, long ldx, int cols, unsigned short* __restrict__ h16,
                                             unsigned short* __restrict__ slots, unsigned short* __restrict__ bf, long ldo,
                                             long piece, int lane) {
  const int ppr = cols >> 3;                       // 8-element pieces per row
  const int r = (int)(piece / ppr), c = (int)(piece - (long)r * ppr) * 8;
  const f32x4_t a = *(const f32x4_t*)(x + (long)r * ldx + c), b = *(const f32x4_t*)(x + (long)r * ldx + c + 4);
  const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  const F6Lane o = f6_encode8(v, lane);
  *(u32x4_t*)(h16 + (long)r * ldo + c) = o.h16;
  if (bf) *(u32x4_t*)(bf + (long)r * ldo + c) = o.bf;
  f6_store_piece<0>((char*)(slots + (long)r * ldo + (c & ~31)), lane, o.piece);
}

__global__ __launch_bounds__(256) void f16f6_encode_kernel(const float* __restrict__ x, long ldx, int rows, int cols,
                                                           unsigned short* __restrict__ h16, unsigned short* __restrict__ slots,
                                                           unsigned short* __restrict__ bf, long ldo) {
  const long piece = (long)blockIdx.x * 256 + threadIdx.x;
  // a quad never straddles the end: rows * cols / 8 is a multiple of 4 (cols % 32 == 0)
  if (piece >= (long)rows * (cols >> 3)) return;
  encode_piece(x, ldx, cols, h16, slots, bf, ldo, piece, threadIdx.x & 63);
}

constexpr int ENC_MAX_T = 48;
struct EncodeTable {
  const float* x[ENC_MAX_T];
  unsigned short* h16[ENC_MAX_T];
  unsigned short* slots[ENC_MAX_T];
  int ldx[ENC_MAX_T], ldo[ENC_MAX_T], rows[ENC_MAX_T], cols[ENC_MAX_T];
  int blk_start[ENC_MAX_T + 1];
  int count;
};
__global__ __launch_bounds__(256) void f16f6_encode_multi_kernel(const EncodeTable t) {
  int ti = 0;
  while (ti + 1 < t.count && (int)blockIdx.x >= t.blk_start[ti + 1]) ++ti;
  const long piece = (long)((int)blockIdx.x - t.blk_start[ti]) * 256 + threadIdx.x;
  if (piece >= (long)t.rows[ti] * (t.cols[ti] >> 3)) return;
  encode_piece(t.x[ti], t.ldx[ti], t.cols[ti], t.h16[ti], t.slots[ti], nullptr, t.ldo[ti], piece, threadIdx.x & 63);
}

// ---- LayerNorm forward -> f16f6 -------------------------------------------------------------------------------------------------
// One wave per row as in layernorm_fwd_kernel, but a lane owns 8 CONSECUTIVE channels per round (piece p = lane + 64 i covers
// channels [8 p, 8 p + 8)): the four lanes of a quad then hold one 32-channel MX block and the format's cross-lane work is two
// quad permutes.  cols % 32 == 0, cols <= 1024 (NP rounds: 2 covers ViT-B's 768 = 64 + 32 pieces and ViT-L's 1024 = 2 x 64).
template <int NP>
__global__ __launch_bounds__(256) void layernorm_fwd_f16f6_kernel(
    const float* __restrict__ x, long ldx, const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int rows,
    int cols, unsigned short* __restrict__ y16, unsigned short* __restrict__ yslots, unsigned short* __restrict__ ybf, long ldy,
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
    if (p < np) {            // np % 4 == 0: a quad is in range or out of range as a whole
      const f32x4_t g0 = *(const f32x4_t*)(gamma + p * 8), g1 = *(const f32x4_t*)(gamma + p * 8 + 4);
      const f32x4_t b0 = *(const f32x4_t*)(beta + p * 8), b1 = *(const f32x4_t*)(beta + p * 8 + 4);
      float y[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        y[e] = (v[i][0][e] - mean) * rstd * g0[e] + b0[e];
        y[4 + e] = (v[i][1][e] - mean) * rstd * g1[e] + b1[e];
      }
      const F6Lane o = f6_encode8(y, lane);
      egv_store<EGV_NT_LN>(y16 + (long)row * ldy + p * 8, o.h16);
      if (ybf) egv_store<EGV_NT_LN>(ybf + (long)row * ldy + p * 8, o.bf);
      f6_store_piece<EGV_NT_LN>((char*)(yslots + (long)row * ldy + ((p * 8) & ~31)), lane, o.piece);
    }
  }
}

}  // namespace

extern "C" int egv_f16f6_encode(const float* x, int64_t ldx, int32_t rows, int32_t cols, uint16_t* h16, uint16_t* slots,
                                egv_bf16* bf, int64_t ldo, void* stream) {
  if (!x || !h16 || !slots || rows <= 0 || cols <= 0 || cols % 32 != 0 || ldo % 32 != 0 || ldx % 4 != 0) return EGV_ERR_ARG;
  const long pieces = (long)rows * (cols >> 3);
  EGV_LAUNCH(f16f6_encode_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, rows,
             cols, h16, slots, bf, (long)ldo);
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}

extern "C" int egv_f16f6_encode_multi(int32_t count, const float* const* x, const int64_t* ldx, const int32_t* rows,
                                      const int32_t* cols, uint16_t* const* h16, uint16_t* const* slots, const int64_t* ldo,
                                      void* stream) {
  if (count < 0 || !x || !ldx || !rows || !cols || !h16 || !slots || !ldo) return EGV_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  EncodeTable t;
  int nt = 0, nb = 0;
  auto flush = [&]() -> int {
    if (nt == 0) return EGV_OK;
    t.blk_start[nt] = nb;
    t.count = nt;
    EGV_LAUNCH(f16f6_encode_multi_kernel, dim3(nb), dim3(256), 0, s, t);
    EGV_CHECK_LAUNCH();
    nt = 0;
    nb = 0;
    return EGV_OK;
  };
  for (int i = 0; i < count; ++i) {
    if (!x[i] || !h16[i] || !slots[i] || rows[i] <= 0 || cols[i] <= 0 || cols[i] % 32 != 0 || ldo[i] % 32 != 0 || ldx[i] % 4 != 0 ||
        ldx[i] > 0x7fffffff || ldo[i] > 0x7fffffff)
      return EGV_ERR_ARG;
    if (nt == ENC_MAX_T) {
      const int rc = flush();
      if (rc) return rc;
    }
    const long pieces = (long)rows[i] * (cols[i] >> 3);
    t.x[nt] = x[i]; t.h16[nt] = h16[i]; t.slots[nt] = slots[i];
    t.ldx[nt] = (int)ldx[i]; t.ldo[nt] = (int)ldo[i]; t.rows[nt] = rows[i]; t.cols[nt] = cols[i];
    t.blk_start[nt] = nb;
    nb += (int)((pieces + 255) / 256);
    ++nt;
  }
  return flush();
}

extern "C" int egv_layernorm_fwd_f16f6(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, int32_t rows,
                                       int32_t cols, uint16_t* y16, uint16_t* yslots, egv_bf16* ybf, int64_t ldy, float* mean,
                                       float* rstd, void* stream) {
  if (!x || !gamma || !beta || !y16 || !yslots || rows <= 0 || cols <= 0 || cols % 32 != 0 || cols > 1024 || ldx % 4 != 0 ||
      ldy % 32 != 0)
    return EGV_ERR_ARG;
  const dim3 grid((rows + 3) / 4), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (cols <= 512)
    EGV_LAUNCH(layernorm_fwd_f16f6_kernel<1>, grid, block, 0, s, x, (long)ldx, gamma, beta, eps, rows, cols, y16, yslots, ybf, (long)ldy,
               mean, rstd);
  else
    EGV_LAUNCH(layernorm_fwd_f16f6_kernel<2>, grid, block, 0, s, x, (long)ldx, gamma, beta, eps, rows, cols, y16, yslots, ybf, (long)ldy,
               mean, rstd);
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}
