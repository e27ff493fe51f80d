// C-ABI entry points of divided space-time attention (VarAttention core, model/video_transformer.py:104-133).
// Patch queries run in the space kernels (attn_mfma_fwd / _bwd.hip) or the time kernels (attn_time_mfma.hip) -- both on the matrix
// cores since round 4 --; the clip's CLS query row (:109-112) rides along in every
// group of those kernels (see attn_small.hip) and is finished by tiny combine / delta / finish kernels.
#include "common.h"
#include "egovlp_hip.h"

int egv_attn_space_fwd_impl(const bf16_t* qkv_hi, const bf16_t* qkv_lo, int B, int T, int n, int H, int passes,
                            bf16_t* out_hi, bf16_t* out_lo, float* lse, float* cls_ws, int out_fmt, int f16, hipStream_t s);
int egv_attn_space_bwd_impl(const bf16_t* qkv_hi, const bf16_t* qkv_lo, const bf16_t* out_hi, const bf16_t* out_lo,
                            const bf16_t* do_hi, const bf16_t* do_lo,
                            const float* lse, float* delta, float* dcls, int B, int T, int n, int H, int passes,
                            bf16_t* dqkv_hi, bf16_t* dqkv_lo, int o_fmt, int g_fmt, int f16, hipStream_t s);
int egv_attn_time_fwd_impl(const bf16_t* qh, const bf16_t* ql, int B, int T, int n, int H, bf16_t* oh, bf16_t* ol,
                           float* lse, float* ws, int out_fmt, int f16, hipStream_t s);
int egv_attn_time_bwd_impl(const bf16_t* qh, const bf16_t* ql, const bf16_t* doh, const bf16_t* dol, const float* lse,
                           const float* delta, int B, int T, int n, int H, bf16_t* gh, bf16_t* gl, float* dcls, int gfmt, int f16,
                           hipStream_t s);
int egv_attn_cls_combine_impl(const float* ws, int B, int G, int S, int H, bf16_t* oh, bf16_t* ol, float* lse, int out_fmt,
                              hipStream_t s);
int egv_attn_cls_delta_impl(const bf16_t* oh, const bf16_t* ol, const bf16_t* doh, const bf16_t* dol, int B, int S, int H,
                            float* delta, float* dcls, int o_fmt, int f16, hipStream_t s);
int egv_attn_cls_finish_impl(const float* dcls, int B, int S, int H, bf16_t* gh, bf16_t* gl, int gfmt, hipStream_t s);

extern "C" int64_t egv_divided_attn_fwd_work_floats(int32_t B, int32_t T, int32_t n, int32_t H, int32_t mode) {
  return (int64_t)B * H * (mode == 0 ? T : n) * 68;
}

extern "C" int64_t egv_divided_attn_bwd_work_floats(int32_t B, int32_t T, int32_t n, int32_t H) {
  return (int64_t)B * H * (1 + (int64_t)T * n) + (int64_t)B * H * 192;
}

extern "C" int egv_divided_attn_fwd(const egv_bf16* qkv_hi, const egv_bf16* qkv_lo, int32_t B, int32_t T, int32_t n,
                                    int32_t H, int32_t mode, int32_t passes, egv_bf16* out_hi, egv_bf16* out_lo,
                                    float* lse, float* work, void* stream) {
  if (!qkv_hi || !out_hi || !lse || !work || B <= 0 || T <= 0 || n <= 0 || H <= 0) return EGV_ERR_ARG;
  if (passes != 1 && passes != 3) return EGV_ERR_ARG;
  if (mode < 0 || mode > 15) return EGV_ERR_ARG;
  const int out_fmt = (mode >> 1) & 3;           // mode bits 1-2: the format of the output planes (attn_common.h ATT_OUT_*: 0 split-bf16,
                                                 // 1 bf16 + fp16(value), 2 f16x2 first-operand planes, 3 ONE plane of fp16(value))
  const int f16 = (mode >> 3) & 1;               // mode bit 3: the qkv planes are an fp16 split (fp16(x), fp16(x - hi)): fp16 MFMA products
  mode &= 1;
  if (f16 && passes != 3) return EGV_ERR_ARG;
  if (out_fmt && passes != 3) return EGV_ERR_ARG;
  if (out_fmt == 3) out_lo = nullptr;
  if (passes == 3 && ((!out_lo && out_fmt != 3) || !qkv_lo)) return EGV_ERR_ARG;
  if (passes == 1) { qkv_lo = nullptr; out_lo = nullptr; }
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (mode == 0)
    rc = egv_attn_space_fwd_impl(qkv_hi, qkv_lo, B, T, n, H, passes, out_hi, out_lo, lse, work, out_fmt, f16, s);
  else
    rc = egv_attn_time_fwd_impl(qkv_hi, qkv_lo, B, T, n, H, out_hi, out_lo, lse, work, out_fmt, f16, s);
  if (rc) return rc;
  return egv_attn_cls_combine_impl(work, B, mode == 0 ? T : n, 1 + T * n, H, out_hi, out_lo, lse, out_fmt, s);
}

extern "C" int egv_divided_attn_bwd(const egv_bf16* qkv_hi, const egv_bf16* qkv_lo, const egv_bf16* out_hi,
                                    const egv_bf16* out_lo, const egv_bf16* dout_hi, const egv_bf16* dout_lo,
                                    const float* lse, int32_t B, int32_t T, int32_t n, int32_t H, int32_t mode,
                                    int32_t passes, egv_bf16* dqkv_hi, egv_bf16* dqkv_lo, float* work, void* stream) {
  if (!qkv_hi || !out_hi || !dout_hi || !lse || !dqkv_hi || !work || B <= 0 || T <= 0 || n <= 0 || H <= 0)
    return EGV_ERR_ARG;
  if (passes != 1 && passes != 3) return EGV_ERR_ARG;
  // mode: bit 0 = time (else space); bits 1-2 = the format the forward wrote its output planes in (egv_divided_attn_fwd: mode >> 1);
  // bit 3 = dqkv as ONE plane of un-clamped fp16 instead of split-bf16 (the fp16 backward: the qkv dgrad / wgrad multiply fp16)
  // bit 4 (with bit 3, passes == 1) = q / k / v (the hi plane of an fp16-split qkv) and dO are fp16 planes: fp16 MFMA products throughout
  if (mode < 0 || mode > 31) return EGV_ERR_ARG;
  const int o_fmt = (mode >> 1) & 3, g_fmt = (mode & 8) ? 4 : 0, f16 = (mode >> 4) & 1;
  mode &= 1;
  if (f16 && (passes != 1 || !g_fmt)) return EGV_ERR_ARG;
  if ((o_fmt >= 2 || g_fmt) && passes != 1) return EGV_ERR_ARG;
  if (o_fmt) out_lo = nullptr;                   // only the split-bf16 format has a residual plane
  if (passes == 3 && (!qkv_lo || (!out_lo && o_fmt == 0) || !dout_lo || !dqkv_lo)) return EGV_ERR_ARG;
  // single-pass: hi planes only -- except the forward's output, whose lo plane (if the caller has one: the benchmarked mode
  // runs a three-pass forward) makes delta = rowsum(dO o O) exact in O at no cost
  if (passes == 1) { qkv_lo = nullptr; dout_lo = nullptr; dqkv_lo = nullptr; }
  hipStream_t s = (hipStream_t)stream;
  const int S = 1 + T * n;
  float* delta = work;                          // [B, H, S]
  float* dcls = work + (long)B * H * S;         // [B, H, 3, 64] raw fp32 accumulators of the CLS token
  int rc = egv_attn_cls_delta_impl(out_hi, out_lo, dout_hi, dout_lo, B, S, H, delta, dcls, o_fmt, f16, s);   // also zeroes dcls
  if (rc) return rc;
  if (mode == 0)
    rc = egv_attn_space_bwd_impl(qkv_hi, qkv_lo, out_hi, out_lo, dout_hi, dout_lo, lse, delta, dcls, B, T, n, H, passes,
                                 dqkv_hi, dqkv_lo, o_fmt, g_fmt, f16, s);
  else if (mode == 1)
    rc = egv_attn_time_bwd_impl(qkv_hi, qkv_lo, dout_hi, dout_lo, lse, delta, B, T, n, H, dqkv_hi, dqkv_lo, dcls, g_fmt, f16, s);
  else
    return EGV_ERR_ARG;
  if (rc) return rc;
  return egv_attn_cls_finish_impl(dcls, B, S, H, dqkv_hi, dqkv_lo, g_fmt, s);
}
