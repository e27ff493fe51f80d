// C-ABI entry points of divided space-time attention (VarAttention core, model/video_transformer.py:104-133):
// patch queries by the space (MFMA) or time (VALU) kernel + the CLS query row by the CLS kernel.
#include "common.h"
#include "egovlp_hip.h"

int egv_attn_space_fwd_impl(const float* qkv, int B, int T, int n, int H, int passes, bf16_t* out_hi, bf16_t* out_lo,
                            float* lse, hipStream_t s);
int egv_attn_space_bwd_impl(const float* qkv, const float* d_out, const float* lse, float* delta, int B, int T, int n,
                            int H, int passes, float* dqkv, hipStream_t s);
int egv_attn_time_fwd_impl(const float* qkv, int B, int T, int n, int H, bf16_t* oh, bf16_t* ol, float* lse,
                           hipStream_t s);
int egv_attn_time_bwd_impl(const float* qkv, const float* d_out, const float* lse, int B, int T, int n, int H,
                           float* dqkv, hipStream_t s);
int egv_attn_cls_fwd_impl(const float* qkv, int B, int S, int H, bf16_t* oh, bf16_t* ol, float* lse, hipStream_t s);
int egv_attn_cls_bwd_impl(const float* qkv, const float* d_out, const float* lse, int B, int S, int H, float* dqkv,
                          hipStream_t s);

extern "C" int egv_divided_attn_fwd(const float* qkv, int32_t B, int32_t T, int32_t n, int32_t H, int32_t mode,
                                    int32_t passes, egv_bf16* out_hi, egv_bf16* out_lo, float* lse, void* stream) {
  if (!qkv || !out_hi || !lse || B <= 0 || T <= 0 || n <= 0 || H <= 0) return EGV_ERR_ARG;
  if (passes != 1 && passes != 3) return EGV_ERR_ARG;
  if (passes == 3 && !out_lo) return EGV_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (mode == 0)
    rc = egv_attn_space_fwd_impl(qkv, B, T, n, H, passes, out_hi, out_lo, lse, s);
  else if (mode == 1)
    rc = egv_attn_time_fwd_impl(qkv, B, T, n, H, out_hi, out_lo, lse, s);
  else
    return EGV_ERR_ARG;
  if (rc) return rc;
  return egv_attn_cls_fwd_impl(qkv, B, 1 + T * n, H, out_hi, out_lo, lse, s);
}

extern "C" int egv_divided_attn_bwd(const float* qkv, const float* d_out, const float* lse, int32_t B, int32_t T,
                                    int32_t n, int32_t H, int32_t mode, int32_t passes, float* dqkv, float* work,
                                    void* stream) {
  if (!qkv || !d_out || !lse || !dqkv || B <= 0 || T <= 0 || n <= 0 || H <= 0) return EGV_ERR_ARG;
  if (passes != 1 && passes != 3) return EGV_ERR_ARG;
  if (mode == 0 && !work) return EGV_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const long S = 1 + (long)T * n;
  const long row = 3L * H * 64;
  // the CLS token's k/v rows are accumulated atomically by the patch kernels: zero token 0 of every clip
  if (hipMemset2DAsync(dqkv, S * row * sizeof(float), 0, row * sizeof(float), B, s) != hipSuccess)
    return EGV_ERR_LAUNCH;
  int rc;
  if (mode == 0)
    rc = egv_attn_space_bwd_impl(qkv, d_out, lse, work, B, T, n, H, passes, dqkv, s);
  else if (mode == 1)
    rc = egv_attn_time_bwd_impl(qkv, d_out, lse, B, T, n, H, dqkv, s);
  else
    return EGV_ERR_ARG;
  if (rc) return rc;
  return egv_attn_cls_bwd_impl(qkv, d_out, lse, B, (int)S, H, dqkv, s);
}
