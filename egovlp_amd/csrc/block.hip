// One C-ABI call per SpaceTimeBlock forward and one per backward (model/video_transformer.py:163-177 and its autograd
// transpose): the host enqueues the block's ~11 / ~25 kernels from C with pointers into ONE workspace arena per direction, instead
// of ~50 Python-level tensor allocations and ~30 ctypes calls.  Nothing new is computed here: every launch below is one of the
// library's own entry points (egv_layernorm_*, egv_gemm_nt, egv_divided_attn_*), with the arguments the per-kernel Python path
// (egovlp_amd/model/video_transformer.py::_SpaceTimeBlockFn, kept as the reference) gives them, in the same order -- results are
// bit-identical (tests/test_gpu_block.py).  The layout of the two arenas is a pure function of the geometry (egv_block_layout).
#include <hip/hip_runtime.h>

#include "common.h"
#include "egovlp_hip.h"

namespace {

constexpr int64_t ALIGN = 256;
inline int64_t up(int64_t b) { return (b + ALIGN - 1) / ALIGN * ALIGN; }

struct Bump {
  int64_t off = 0;
  int64_t take(int64_t bytes) {
    const int64_t o = off;
    off += up(bytes);
    return o;
  }
};

// forward arena: what the block's kernels hand to each other and what the backward needs again
struct FwdLayout {
  int64_t n3_hi, n3_lo, mean3, rstd3, qkvt_hi, qkvt_lo, at_hi, at_lo, lse_t, work_t;
  int64_t tr, n1_hi, n1_lo, mean1, rstd1, qkvs_hi, qkvs_lo, as_hi, as_lo, lse_s, work_s;
  int64_t sr, n2_hi, n2_lo, mean2, rstd2, h_hi, h_lo, z;
  int64_t n3_bf, n1_bf, n2_bf, h_bf;     // f16x2 forward (P == 2) of a training step: bf16 copies for the single-pass backward
  int64_t total;
};

struct Geo {
  int64_t M, S, D, Hd;
  int P, Pb;
};

Geo geo_of(const egv_block_geom& g) {
  Geo o;
  o.S = 1 + (int64_t)g.T * g.n;
  o.M = (int64_t)g.B * o.S;
  o.D = g.D;
  o.Hd = g.Hd;
  o.P = g.fwd_passes;
  o.Pb = g.bwd_passes;
  return o;
}

bool geom_ok(const egv_block_geom& g) {
  if (g.B <= 0 || g.T <= 0 || g.n <= 0 || g.H <= 0 || g.D != g.H * 64 || g.Hd <= 0 || g.D % 32 || g.Hd % 32) return false;
  // bwd_passes: 3 = split-bf16 three-product, 1 = one bf16 product, 4 = one FP16 product on scaled gradients (the fp16 backward: needs
  // the f16x2 / f16mix forward, whose saved operands ARE fp16 planes -- no bf16 copies are written then)
  if (g.fwd_passes < 1 || g.fwd_passes > 3 || (g.bwd_passes != 1 && g.bwd_passes != 3 && g.bwd_passes != 4)) return false;
  if (g.bwd_passes == 3 && g.fwd_passes != 3) return false;
  if (g.bwd_passes == 4 && g.fwd_passes != 2) return false;
  if (g.fwd_passes == 2 && (g.bwd_passes == 3 || (g.train && !g.z_bf16))) return false;   // f16x2 forward: single-product backward, 16-bit saved gelu'
  if (g.f16_single < 0 || g.f16_single > 15 || (g.f16_single && g.fwd_passes != 2)) return false;
  return true;
}

FwdLayout fwd_layout(const egv_block_geom& g) {
  const Geo o = geo_of(g);
  const bool lo = o.P != 1;              // split-bf16 (P == 3) and f16x2 (P == 2) operands are two planes
  const bool h16 = o.Pb == 4;            // fp16 backward: it reads the forward's own fp16 planes
  const bool bf = o.P == 2 && g.train && !h16;
  FwdLayout L;
  Bump b;
  auto plane = [&](int64_t cols) { return b.take(o.M * cols * 2); };
  auto plane_lo = [&](int64_t cols) { return lo ? b.take(o.M * cols * 2) : (int64_t)-1; };
  auto plane_bf = [&](int64_t cols) { return bf ? b.take(o.M * cols * 2) : (int64_t)-1; };
  // f16x2 forward with single-product Linears (g.f16_single): their first operand is ONE plain fp16 plane
  const bool q1 = (g.f16_single & 4) != 0;
  L.n3_hi = plane(o.D); L.n3_lo = q1 ? (int64_t)-1 : plane_lo(o.D);
  L.mean3 = b.take(o.M * 4); L.rstd3 = b.take(o.M * 4);
  L.qkvt_hi = plane(3 * o.D); L.qkvt_lo = plane_lo(3 * o.D);
  const bool a1p = h16 && (g.f16_single & 8);   // attention output as ONE fp16 plane (one-product proj, fp16 backward)
  L.at_hi = plane(o.D); L.at_lo = a1p ? (int64_t)-1 : plane_lo(o.D);
  L.lse_t = b.take((int64_t)g.B * g.H * o.S * 4);
  L.work_t = b.take(egv_divided_attn_fwd_work_floats(g.B, g.T, g.n, g.H, 1) * 4);
  L.tr = b.take(o.M * o.D * 4);
  L.n1_hi = plane(o.D); L.n1_lo = q1 ? (int64_t)-1 : plane_lo(o.D);
  L.mean1 = b.take(o.M * 4); L.rstd1 = b.take(o.M * 4);
  L.qkvs_hi = plane(3 * o.D); L.qkvs_lo = plane_lo(3 * o.D);
  L.as_hi = plane(o.D); L.as_lo = a1p ? (int64_t)-1 : plane_lo(o.D);
  L.lse_s = b.take((int64_t)g.B * g.H * o.S * 4);
  L.work_s = b.take(egv_divided_attn_fwd_work_floats(g.B, g.T, g.n, g.H, 0) * 4);
  L.sr = b.take(o.M * o.D * 4);
  L.n2_hi = plane(o.D); L.n2_lo = (g.f16_single & 1) ? (int64_t)-1 : plane_lo(o.D);
  L.mean2 = b.take(o.M * 4); L.rstd2 = b.take(o.M * 4);
  L.h_hi = plane(o.Hd); L.h_lo = (g.f16_single & 2) ? (int64_t)-1 : plane_lo(o.Hd);
  L.z = g.train ? b.take(o.M * o.Hd * (g.z_bf16 ? 2 : 4)) : (int64_t)-1;
  L.n3_bf = plane_bf(o.D); L.n1_bf = plane_bf(o.D); L.n2_bf = plane_bf(o.D); L.h_bf = plane_bf(o.Hd);
  L.total = b.off;
  return L;
}

// weights of a block, in the order of egv_block_params: tqkv, tproj, sqkv, sproj, fc1, fc2  ->  (N, K) of W[N,K]
void wshape(const Geo& o, int i, int64_t& N, int64_t& K) {
  switch (i) {
    case 0: case 2: N = 3 * o.D; K = o.D; break;
    case 1: case 3: N = o.D; K = o.D; break;
    case 4: N = o.Hd; K = o.D; break;
    default: N = o.D; K = o.Hd; break;
  }
}

// gradient buffer: [dW x 6][db x 6][dgamma3, dbeta3, dgamma1, dbeta1, dgamma2, dbeta2] back to back, offsets in floats
void grad_layout(const egv_block_geom& g, int64_t off[18], int64_t& total) {
  const Geo o = geo_of(g);
  int64_t p = 0;
  auto take = [&](int64_t n) { const int64_t q = p; p += n; return q; };   // back to back: every size is a multiple of 4 floats (D % 32 == 0)
  for (int i = 0; i < 6; ++i) { int64_t N, K; wshape(o, i, N, K); off[i] = take(N * K); }
  for (int i = 0; i < 6; ++i) { int64_t N, K; wshape(o, i, N, K); off[6 + i] = take(N); }
  for (int i = 0; i < 6; ++i) off[12 + i] = take(o.D);
  total = p;
}

struct BwdLayout {
  int64_t g_hi, g_lo, dz_hi, dz_lo, d_n2, d_sr, dsr_hi, dsr_lo, das_hi, das_lo, dqkvs_hi, dqkvs_lo, d_n1, d_tr, dtr_hi, dtr_lo, dat_hi,
      dat_lo, dqkvt_hi, dqkvt_lo, d_n3, ln_work[3], attn_work, partial[6];
  int64_t total;
};

BwdLayout bwd_layout(const egv_block_geom& g, const int32_t* ksplit) {
  const Geo o = geo_of(g);
  const bool lo = o.Pb == 3;
  BwdLayout L;
  Bump b;
  auto plane = [&](int64_t cols) { return b.take(o.M * cols * 2); };
  auto plane_lo = [&](int64_t cols) { return lo ? b.take(o.M * cols * 2) : (int64_t)-1; };
  L.g_hi = plane(o.D); L.g_lo = plane_lo(o.D);
  L.dz_hi = plane(o.Hd); L.dz_lo = plane_lo(o.Hd);
  L.d_n2 = b.take(o.M * o.D * 4);
  L.d_sr = b.take(o.M * o.D * 4);
  L.dsr_hi = plane(o.D); L.dsr_lo = plane_lo(o.D);
  L.das_hi = plane(o.D); L.das_lo = plane_lo(o.D);
  L.dqkvs_hi = plane(3 * o.D); L.dqkvs_lo = plane_lo(3 * o.D);
  L.d_n1 = b.take(o.M * o.D * 4);
  L.d_tr = b.take(o.M * o.D * 4);
  L.dtr_hi = plane(o.D); L.dtr_lo = plane_lo(o.D);
  L.dat_hi = plane(o.D); L.dat_lo = plane_lo(o.D);
  L.dqkvt_hi = plane(3 * o.D); L.dqkvt_lo = plane_lo(3 * o.D);
  L.d_n3 = b.take(o.M * o.D * 4);
  for (int i = 0; i < 3; ++i) L.ln_work[i] = b.take(2 * o.D * (int64_t)egv_layernorm_bwd_parts((int32_t)o.M) * 4);   // one per LayerNorm: reduced together at the end
  L.attn_work = b.take(egv_divided_attn_bwd_work_floats(g.B, g.T, g.n, g.H) * 4);
  for (int i = 0; i < 6; ++i) {
    int64_t N, K;
    wshape(o, i, N, K);
    const int ks = ksplit ? ksplit[i] : 1;
    L.partial[i] = ks > 1 ? b.take((int64_t)ks * (N * K + N) * 4) : (int64_t)-1;
  }
  L.total = b.off;
  return L;
}

template <class T>
T* at(void* base, int64_t off) { return off < 0 ? nullptr : (T*)((char*)base + off); }
template <class T>
const T* at(const void* base, int64_t off) { return off < 0 ? nullptr : (const T*)((const char*)base + off); }

egv_gemm_desc nt_desc(const egv_bf16* a_hi, const egv_bf16* a_lo, int64_t lda, const egv_bf16* b_hi, const egv_bf16* b_lo, int64_t ldb,
                      int64_t M, int64_t N, int64_t K, int passes, int grid_cap) {
  egv_gemm_desc d = {};
  d.a_hi = a_hi; d.a_lo = a_lo; d.lda = lda;
  d.b_hi = b_hi; d.b_lo = b_lo; d.ldb = ldb;
  d.M = (int32_t)M; d.N = (int32_t)N; d.K = (int32_t)K; d.passes = passes;
  d.alpha = 1.0f;
  d.ksplit = 1;
  d.grid_cap = grid_cap;
  return d;
}

#define EGV_TRY(call)            \
  do {                           \
    const int rc__ = (call);     \
    if (rc__ != EGV_OK) return rc__; \
  } while (0)

}  // namespace

extern "C" int64_t egv_block_fwd_arena_bytes(const egv_block_geom* g) { return (g && geom_ok(*g)) ? fwd_layout(*g).total : -1; }

extern "C" int64_t egv_block_bwd_arena_bytes(const egv_block_geom* g, const int32_t* wgrad_ksplit) {
  return (g && geom_ok(*g)) ? bwd_layout(*g, wgrad_ksplit).total : -1;
}

extern "C" int egv_block_grad_layout(const egv_block_geom* g, int64_t* offsets18, int64_t* total_floats) {
  if (!g || !geom_ok(*g) || !offsets18 || !total_floats) return EGV_ERR_ARG;
  grad_layout(*g, offsets18, *total_floats);
  return EGV_OK;
}

// Byte offsets (into the forward arena) of what Python keeps handles to: the planes the attached gradient hand-over and the
// tests look at.  order: n3_hi, at_hi, n1_hi, as_hi, n2_hi, h_hi, qkvt_hi, qkvs_hi, tr, sr, z  (-1: absent)
extern "C" int egv_block_fwd_offsets(const egv_block_geom* g, int64_t* off11) {
  if (!g || !geom_ok(*g) || !off11) return EGV_ERR_ARG;
  const FwdLayout L = fwd_layout(*g);
  const int64_t v[11] = {L.n3_hi, L.at_hi, L.n1_hi, L.as_hi, L.n2_hi, L.h_hi, L.qkvt_hi, L.qkvs_hi, L.tr, L.sr, L.z};
  for (int i = 0; i < 11; ++i) off11[i] = v[i];
  return EGV_OK;
}

extern "C" int egv_block_fwd(const egv_block_geom* gp, const egv_block_params* pp, const float* x, float* out, void* arena, void* stream) {
  if (!gp || !pp || !x || !out || !arena || !geom_ok(*gp)) return EGV_ERR_ARG;
  const egv_block_geom& g = *gp;
  const egv_block_params& p = *pp;
  const Geo o = geo_of(g);
  const FwdLayout L = fwd_layout(g);
  const int P = o.P;
  const int Pa = P == 2 ? 3 : P;         // attention and the proj Linears: split-bf16 three-product operands in the f16x2 mode
  const int32_t M = (int32_t)o.M, D = (int32_t)o.D, Hd = (int32_t)o.Hd;
  for (int i = 0; i < 6; ++i)
    if (!p.w_hi[i] || (P != 1 && !p.w_lo[i])) return EGV_ERR_ARG;
  const int P_fc1 = (g.f16_single & 1) ? 4 : P, P_fc2 = (g.f16_single & 2) ? 4 : P, P_qkv = (g.f16_single & 4) ? 4 : P;   // 4: ONE fp16 product
  const bool proj1 = (g.f16_single & 8) != 0;      // proj Linears: ONE fp16 product on the attention's fp16(value) plane (its second output plane)
  const bool h16 = o.Pb == 4;                      // fp16 backward: no bf16 copies; the attention output is fp16 planes only (format 3: ONE
                                                   // plane of fp16(value), one-product proj; format 2: f16x2 planes, TWO-product proj)
  const int afmt = h16 ? (proj1 ? 3 : 2) : (proj1 ? 1 : 0);
  const int amode = (afmt << 1) | (h16 ? 8 : 0);     // fp16 backward: the attention itself runs on fp16 operands (qkv planes = an fp16 split)
  // the proj GEMM of one attention branch: operand planes and product count by format
  auto proj_desc = [&](const egv_bf16* a_hi, const egv_bf16* a_lo, int wi) {
    if (afmt == 3) return nt_desc(a_hi, nullptr, D, p.w_hi[wi], p.w_lo[wi], p.ldw[wi], M, D, D, 4, g.grid_cap);
    if (afmt == 2) return nt_desc(a_hi, a_lo, D, p.w_hi[wi], p.w_lo[wi], p.ldw[wi], M, D, D, 2, g.grid_cap);
    if (afmt == 1) return nt_desc(a_lo, nullptr, D, p.w_hi[wi], p.w_lo[wi], p.ldw[wi], M, D, D, 4, g.grid_cap);
    return nt_desc(a_hi, a_lo, D, p.w_hi[wi], p.w_lo[wi], p.ldw[wi], M, D, D, Pa, g.grid_cap);
  };
  char* A = (char*)arena;
  // LayerNorm -> operand planes of the qkv / fc1 Linears: split-bf16, or f16x2 (first-operand role, + the bf16 copy the backward reads)
  auto ln = [&](const float* in, const float* gw, const float* gb, egv_bf16* y_hi, egv_bf16* y_lo, int64_t bf_off, float* mean, float* rstd) -> int {
    if (P == 2)
      return egv_layernorm_fwd_f16x2(in, D, gw, gb, g.eps, M, D, (uint16_t*)y_hi, (uint16_t*)y_lo, at<egv_bf16>(A, bf_off), D, mean, rstd, stream);
    return egv_layernorm_fwd(in, nullptr, D, gw, gb, g.eps, M, D, nullptr, y_hi, y_lo, nullptr, D, mean, rstd, stream);
  };
  egv_bf16 *n3_hi = at<egv_bf16>(A, L.n3_hi), *n3_lo = at<egv_bf16>(A, L.n3_lo);
  egv_bf16 *qt_hi = at<egv_bf16>(A, L.qkvt_hi), *qt_lo = at<egv_bf16>(A, L.qkvt_lo);
  egv_bf16 *at_hi = at<egv_bf16>(A, L.at_hi), *at_lo = at<egv_bf16>(A, L.at_lo);
  egv_bf16 *n1_hi = at<egv_bf16>(A, L.n1_hi), *n1_lo = at<egv_bf16>(A, L.n1_lo);
  egv_bf16 *qs_hi = at<egv_bf16>(A, L.qkvs_hi), *qs_lo = at<egv_bf16>(A, L.qkvs_lo);
  egv_bf16 *as_hi = at<egv_bf16>(A, L.as_hi), *as_lo = at<egv_bf16>(A, L.as_lo);
  egv_bf16 *n2_hi = at<egv_bf16>(A, L.n2_hi), *n2_lo = at<egv_bf16>(A, L.n2_lo);
  egv_bf16 *h_hi = at<egv_bf16>(A, L.h_hi), *h_lo = at<egv_bf16>(A, L.h_lo);
  float *tr = at<float>(A, L.tr), *sr = at<float>(A, L.sr);

  // ---- temporal attention branch (:166-167)
  EGV_TRY(ln(x, p.n3w, p.n3b, n3_hi, n3_lo, L.n3_bf, at<float>(A, L.mean3), at<float>(A, L.rstd3)));
  {
    egv_gemm_desc d = nt_desc(n3_hi, n3_lo, D, p.w_hi[0], p.w_lo[0], p.ldw[0], M, 3 * D, D, P_qkv, g.grid_cap);
    d.bias = p.bias[0]; d.out_hi = qt_hi; d.out_lo = qt_lo; d.ldoh = 3 * D;
    if (h16) d.out_fmt = 3;              // qkv as an fp16 split (fp16(x), fp16(x - hi)): the fp16 attention's operand planes
    EGV_TRY(egv_gemm_nt(&d, stream));
  }
  EGV_TRY(egv_divided_attn_fwd(qt_hi, qt_lo, g.B, g.T, g.n, g.H, 1 | amode, Pa, at_hi, at_lo, at<float>(A, L.lse_t), at<float>(A, L.work_t), stream));
  {
    egv_gemm_desc d = proj_desc(at_hi, at_lo, 1);
    d.bias = p.bias[1]; d.residual = x; d.ldr = D; d.out_f32 = tr; d.ldo = D;
    EGV_TRY(egv_gemm_nt(&d, stream));
  }
  // ---- spatial attention branch (:168-171; the residual is the block INPUT x, :171)
  EGV_TRY(ln(tr, p.n1w, p.n1b, n1_hi, n1_lo, L.n1_bf, at<float>(A, L.mean1), at<float>(A, L.rstd1)));
  {
    egv_gemm_desc d = nt_desc(n1_hi, n1_lo, D, p.w_hi[2], p.w_lo[2], p.ldw[2], M, 3 * D, D, P_qkv, g.grid_cap);
    d.bias = p.bias[2]; d.out_hi = qs_hi; d.out_lo = qs_lo; d.ldoh = 3 * D;
    if (h16) d.out_fmt = 3;
    EGV_TRY(egv_gemm_nt(&d, stream));
  }
  EGV_TRY(egv_divided_attn_fwd(qs_hi, qs_lo, g.B, g.T, g.n, g.H, 0 | amode, Pa, as_hi, as_lo, at<float>(A, L.lse_s), at<float>(A, L.work_s), stream));
  {
    egv_gemm_desc d = proj_desc(as_hi, as_lo, 3);
    d.bias = p.bias[3]; d.residual = x; d.ldr = D; d.out_f32 = sr; d.ldo = D;
    EGV_TRY(egv_gemm_nt(&d, stream));
  }
  // ---- MLP (:175, :46-52): exact-erf GELU in the fc1 epilogue
  EGV_TRY(ln(sr, p.n2w, p.n2b, n2_hi, n2_lo, L.n2_bf, at<float>(A, L.mean2), at<float>(A, L.rstd2)));
  {
    egv_gemm_desc d = nt_desc(n2_hi, n2_lo, D, p.w_hi[4], p.w_lo[4], p.ldw[4], M, Hd, D, P_fc1, g.grid_cap);
    d.bias = p.bias[4]; d.act = EGV_ACT_GELU; d.out_hi = h_hi; d.out_lo = h_lo; d.ldoh = Hd;
    // h as fp16 operand planes (+ bf16 copy when training): the f16x2 format, or one plain plane when fc2 runs a single product
    if (P == 2) { d.out_fmt = P_fc2 == 4 ? 2 : 1; d.out_bf = at<egv_bf16>(A, L.h_bf); }
    if (g.train) {
      d.aux_out = at<float>(A, L.z); d.ldaux = Hd;
      d.aux_bf16 = g.z_bf16 ? (h16 ? 3 : 2) : 0;      // 16 bits: gelu'(z) itself (bf16; fp16 for the fp16 backward), else the fp32 pre-activation
    }
    EGV_TRY(egv_gemm_nt(&d, stream));
  }
  {
    egv_gemm_desc d = nt_desc(h_hi, h_lo, Hd, p.w_hi[5], p.w_lo[5], p.ldw[5], M, D, Hd, P_fc2, g.grid_cap);
    d.bias = p.bias[5]; d.residual = sr; d.ldr = D; d.out_f32 = out; d.ldo = D;
    EGV_TRY(egv_gemm_nt(&d, stream));
  }
  return EGV_OK;
}

extern "C" int egv_block_bwd(const egv_block_geom* gp, const egv_block_params* pp, const egv_block_bwd_io* iop, void* stream) {
  if (!gp || !pp || !iop || !geom_ok(*gp) || !gp->train) return EGV_ERR_ARG;
  const egv_block_geom& g = *gp;
  const egv_block_params& p = *pp;
  const egv_block_bwd_io& io = *iop;
  if (!io.g_out || !io.x || !io.fwd_arena || !io.bwd_arena || !io.d_x || !io.dx_hi || !io.grads) return EGV_ERR_ARG;
  const Geo o = geo_of(g);
  const int Pb = o.Pb;
  const FwdLayout F = fwd_layout(g);
  const BwdLayout L = bwd_layout(g, io.wgrad_ksplit);
  int64_t goff[18], gtot;
  grad_layout(g, goff, gtot);
  const int32_t M = (int32_t)o.M, D = (int32_t)o.D, Hd = (int32_t)o.Hd;
  for (int i = 0; i < 6; ++i)
    if (!p.wt_hi[i] || (Pb == 3 && !p.wt_lo[i])) return EGV_ERR_ARG;
  if (Pb == 3 && (!io.dx_lo || (io.g_hi && !io.g_lo))) return EGV_ERR_ARG;
  // The fp16 backward (Pb == 4).  Every gradient of the pass carries the loss scale S (egv_loss_scale_*; linear all the way, so nothing
  // here knows S).  Operand planes: dY = ONE plane of un-clamped fp16 (LayerNorm-backward, the GELU' epilogue, the attention backward
  // write it so; g_hi / dx_hi of the io struct are such planes too); X = plane 1 of the forward's own fp16 operand (plain fp16(x), or
  // a1 = fp16((1 - e) x) of an f16x2 encoding: the weight gradient is rescaled by 1 / (1 - e)); W^T = fp16 planes in p.wt_hi.  The
  // attention backward multiplies fp16 as well: q / k / v = the hi plane of the fp16-split qkv planes the forward wrote, dO = an fp16 plane
  // from the proj dgrad epilogue, P and dS rounded to fp16 (2^-11 per operand; with a bf16 attention backward the weight gradients of the
  // first blocks stayed at 1.8e-2 whatever the GEMMs did, profiles/r06_fp16_backward_bringup.txt).
  const bool h16 = Pb == 4;
  constexpr float A1_INV = 1.0f / (1.0f - 0.015625f);         // csrc/f16x2.h: a1 = fp16((1 - 2^-6) x)
  const char* FA = (const char*)io.fwd_arena;
  char* A = (char*)io.bwd_arena;
  hipStream_t main_s = (hipStream_t)stream;
  auto fpl = [&](int64_t hi, int64_t lo, const egv_bf16*& ph, const egv_bf16*& pl) {
    ph = at<egv_bf16>(FA, hi);
    pl = Pb == 3 ? at<egv_bf16>(FA, lo) : nullptr;
  };
  const bool x2 = o.P == 2 && !h16;      // f16x2 forward, bf16 backward: the activations' single-pass operands are their bf16 copies
  const egv_bf16 *n3_hi, *n3_lo, *at_hi, *at_lo, *n1_hi, *n1_lo, *as_hi, *as_lo, *n2_hi, *n2_lo, *h_hi, *h_lo, *qt_hi, *qt_lo, *qs_hi, *qs_lo;
  fpl(x2 ? F.n3_bf : F.n3_hi, F.n3_lo, n3_hi, n3_lo); fpl(F.at_hi, F.at_lo, at_hi, at_lo); fpl(x2 ? F.n1_bf : F.n1_hi, F.n1_lo, n1_hi, n1_lo);
  fpl(F.as_hi, F.as_lo, as_hi, as_lo); fpl(x2 ? F.n2_bf : F.n2_hi, F.n2_lo, n2_hi, n2_lo); fpl(x2 ? F.h_bf : F.h_hi, F.h_lo, h_hi, h_lo);
  fpl(F.qkvt_hi, F.qkvt_lo, qt_hi, qt_lo); fpl(F.qkvs_hi, F.qkvs_lo, qs_hi, qs_lo);
  // the attention backward takes the forward output's lo plane whenever the forward wrote one, also in a single-pass backward
  // (delta = rowsum(dO o O) exact in O: egv_divided_attn_bwd)
  // (not when that plane holds fp16(value) for a single-product proj, egv_block_geom.f16_single bit 3: delta then comes from the bf16 plane)
  const bool proj1 = (g.f16_single & 8) != 0;
  const egv_bf16 *as_lo_f = (proj1 || h16) ? nullptr : at<egv_bf16>(FA, F.as_lo), *at_lo_f = (proj1 || h16) ? nullptr : at<egv_bf16>(FA, F.at_lo);
  const int afmt = h16 ? (proj1 ? 3 : 2) : (proj1 ? 1 : 0);       // the format egv_block_fwd wrote the attention outputs in
  const int amode = (afmt << 1) | (h16 ? 8 | 16 : 0);             // + dqkv as fp16, + q / k / v / dO read as fp16 (fp16 products throughout)
  // weight-gradient rescale: X is a1 of an f16x2 encoding wherever the op does NOT run one product (egv_block_geom.f16_single)
  const float walpha[6] = {h16 && !(g.f16_single & 4) ? A1_INV : 1.0f, h16 && !proj1 ? A1_INV : 1.0f, h16 && !(g.f16_single & 4) ? A1_INV : 1.0f,
                           h16 && !proj1 ? A1_INV : 1.0f, h16 && !(g.f16_single & 1) ? A1_INV : 1.0f, h16 && !(g.f16_single & 2) ? A1_INV : 1.0f};
  const int Pg = Pb;                     // passes code of every GEMM of the pass (4: one fp16 product)
  const float *tr = at<float>(FA, F.tr), *sr = at<float>(FA, F.sr);
  float* grads = io.grads;

  // The split-K slabs of the six weight gradients are reduced by ONE launch at the end of the call (egv_splitk_reduce_multi) when all
  // six run on the same stream -- the single wgrad side stream, or the main stream --: 12 reduce launches per step instead of 72 on
  // the stream whose queue drains last.  Weight gradients dealt to different streams keep their own reduce.
  bool defer_reduce = true;
  for (int i = 1; i < 6; ++i) defer_reduce = defer_reduce && io.side_stream[i] == io.side_stream[0];
#ifdef EGV_NO_MULTI_REDUCE
  defer_reduce = false;
#endif
  // the weight gradient dW[N,K] = dY^T X (TN kernel, bias gradient from the same pass) of weight i, on its side stream if it has one
  auto wgrad = [&](int i, const egv_bf16* dy_hi, const egv_bf16* dy_lo, int64_t lddy, const egv_bf16* x_hi, const egv_bf16* x_lo,
                   int64_t ldx) -> int {
    int64_t N, K;
    wshape(o, i, N, K);
    hipStream_t s = main_s;
    if (io.side_stream[i]) {
      s = (hipStream_t)io.side_stream[i];
      if (hipEventRecord((hipEvent_t)io.side_event[i], main_s) != hipSuccess) return EGV_ERR_LAUNCH;
      if (hipStreamWaitEvent(s, (hipEvent_t)io.side_event[i], 0) != hipSuccess) return EGV_ERR_LAUNCH;
    }
    egv_gemm_desc d = {};
    d.a_hi = dy_hi; d.a_lo = dy_lo; d.lda = lddy;
    d.b_hi = x_hi; d.b_lo = x_lo; d.ldb = ldx;
    d.M = (int32_t)N; d.N = (int32_t)K; d.K = M; d.passes = Pg;
    d.alpha = walpha[i];
    d.out_f32 = grads + goff[i]; d.ldo = K;
    d.ksplit = io.wgrad_ksplit[i] > 1 ? io.wgrad_ksplit[i] : 1;
    d.partial = at<float>(A, L.partial[i]);
    d.trans = 1;
    d.colsum = grads + goff[6 + i];
    d.grid_cap = g.grid_cap;
    if (defer_reduce && d.ksplit > 1) d.accumulate = 2;        // slabs only; reduced below
    return egv_gemm_nt(&d, s);
  };

  // ---- G as planes (handed over by the next block's LayerNorm-backward, or split here)
  const egv_bf16 *g_hi = io.g_hi, *g_lo = Pb == 3 ? io.g_lo : nullptr;
  if (!g_hi) {
    egv_bf16 *gh = at<egv_bf16>(A, L.g_hi), *gl = at<egv_bf16>(A, L.g_lo);
    if (h16) EGV_TRY(egv_f16x2_encode(io.g_out, D, M, D, gh, nullptr, nullptr, D, 2, stream));
    else EGV_TRY(egv_split_f32(io.g_out, D, M, D, gh, gl, D, nullptr, nullptr, 0, nullptr, stream));
    g_hi = gh; g_lo = gl;
  }
  // ---- MLP backward: dZ = (G . W2) * gelu'(z) leaves the fc2-dgrad epilogue already split
  egv_bf16 *dz_hi = at<egv_bf16>(A, L.dz_hi), *dz_lo = at<egv_bf16>(A, L.dz_lo);
  {
    egv_gemm_desc d = nt_desc(g_hi, g_lo, D, p.wt_hi[5], p.wt_lo[5], p.ldwt[5], M, Hd, D, Pg, g.grid_cap);
    d.act = EGV_ACT_GELU_BWD; d.aux_in = at<float>(FA, F.z); d.ldaux = Hd; d.aux_bf16 = g.z_bf16 ? (h16 ? 3 : 2) : 0;
    d.out_hi = dz_hi; d.out_lo = dz_lo; d.ldoh = Hd;
    if (h16) d.out_fmt = 4;               // dZ as one plane of un-clamped fp16
    EGV_TRY(egv_gemm_nt(&d, stream));
  }
  EGV_TRY(wgrad(5, g_hi, g_lo, D, h_hi, h_lo, Hd));
  EGV_TRY(wgrad(4, dz_hi, dz_lo, Hd, n2_hi, n2_lo, D));
  // the three dgrads that feed a LayerNorm backward (d_n2, d_n1, d_n3): fp32; in the fp16 backward ONE plane of un-clamped fp16 in the
  // same buffer (half the bytes written here and read there: 77 -> 38.6 MB per launch at M = 25 120)
  auto ln_in = [&](egv_gemm_desc& d, float* buf) {
    if (h16) { d.out_hi = (egv_bf16*)buf; d.ldoh = D; d.out_fmt = 4; }
    else { d.out_f32 = buf; d.ldo = D; }
  };
  const int ln_fmt = h16 ? 3 : 0;        // egv_layernorm_bwd_partial: dx plane (bit 0) and dy plane (bit 1) as un-clamped fp16
  float* d_n2 = at<float>(A, L.d_n2);
  {
    egv_gemm_desc d = nt_desc(dz_hi, dz_lo, Hd, p.wt_hi[4], p.wt_lo[4], p.ldwt[4], M, D, Hd, Pg, g.grid_cap);
    ln_in(d, d_n2);
    EGV_TRY(egv_gemm_nt(&d, stream));
  }
  float* d_sr = at<float>(A, L.d_sr);
  egv_bf16 *dsr_hi = at<egv_bf16>(A, L.dsr_hi), *dsr_lo = at<egv_bf16>(A, L.dsr_lo);
  EGV_TRY(egv_layernorm_bwd_partial(h16 ? nullptr : d_n2, h16 ? (const egv_bf16*)d_n2 : nullptr, nullptr, D, sr, D, p.n2w, at<float>(FA, F.mean2),
                            at<float>(FA, F.rstd2), M, D, io.g_out, nullptr,
                            d_sr, D, dsr_hi, dsr_lo, ln_fmt, grads + goff[16], grads + goff[17], at<float>(A, L.ln_work[0]), stream));
  // ---- spatial attention backward
  EGV_TRY(wgrad(3, dsr_hi, dsr_lo, D, as_hi, as_lo, D));
  egv_bf16 *das_hi = at<egv_bf16>(A, L.das_hi), *das_lo = at<egv_bf16>(A, L.das_lo);
  {
    egv_gemm_desc d = nt_desc(dsr_hi, dsr_lo, D, p.wt_hi[3], p.wt_lo[3], p.ldwt[3], M, D, D, Pg, g.grid_cap);
    d.out_hi = das_hi; d.out_lo = das_lo; d.ldoh = D;
    if (h16) d.out_fmt = 4;              // dO as one plane of un-clamped fp16
    EGV_TRY(egv_gemm_nt(&d, stream));
  }
  egv_bf16 *dqs_hi = at<egv_bf16>(A, L.dqkvs_hi), *dqs_lo = at<egv_bf16>(A, L.dqkvs_lo);
  EGV_TRY(egv_divided_attn_bwd(qs_hi, qs_lo, as_hi, as_lo_f, das_hi, das_lo, at<float>(FA, F.lse_s), g.B, g.T, g.n, g.H, 0 | amode, h16 ? 1 : Pb, dqs_hi, dqs_lo,
                               at<float>(A, L.attn_work), stream));
  EGV_TRY(wgrad(2, dqs_hi, dqs_lo, 3 * D, n1_hi, n1_lo, D));
  float* d_n1 = at<float>(A, L.d_n1);
  {
    egv_gemm_desc d = nt_desc(dqs_hi, dqs_lo, 3 * D, p.wt_hi[2], p.wt_lo[2], p.ldwt[2], M, D, 3 * D, Pg, g.grid_cap);
    ln_in(d, d_n1);
    EGV_TRY(egv_gemm_nt(&d, stream));
  }
  float* d_tr = at<float>(A, L.d_tr);
  egv_bf16 *dtr_hi = at<egv_bf16>(A, L.dtr_hi), *dtr_lo = at<egv_bf16>(A, L.dtr_lo);
  EGV_TRY(egv_layernorm_bwd_partial(h16 ? nullptr : d_n1, h16 ? (const egv_bf16*)d_n1 : nullptr, nullptr, D, tr, D, p.n1w, at<float>(FA, F.mean1),
                            at<float>(FA, F.rstd1), M, D, nullptr, nullptr,
                            d_tr, D, dtr_hi, dtr_lo, ln_fmt, grads + goff[14], grads + goff[15], at<float>(A, L.ln_work[1]), stream));
  // ---- temporal attention backward
  EGV_TRY(wgrad(1, dtr_hi, dtr_lo, D, at_hi, at_lo, D));
  egv_bf16 *dat_hi = at<egv_bf16>(A, L.dat_hi), *dat_lo = at<egv_bf16>(A, L.dat_lo);
  {
    egv_gemm_desc d = nt_desc(dtr_hi, dtr_lo, D, p.wt_hi[1], p.wt_lo[1], p.ldwt[1], M, D, D, Pg, g.grid_cap);
    d.out_hi = dat_hi; d.out_lo = dat_lo; d.ldoh = D;
    if (h16) d.out_fmt = 4;
    EGV_TRY(egv_gemm_nt(&d, stream));
  }
  egv_bf16 *dqt_hi = at<egv_bf16>(A, L.dqkvt_hi), *dqt_lo = at<egv_bf16>(A, L.dqkvt_lo);
  EGV_TRY(egv_divided_attn_bwd(qt_hi, qt_lo, at_hi, at_lo_f, dat_hi, dat_lo, at<float>(FA, F.lse_t), g.B, g.T, g.n, g.H, 1 | amode, h16 ? 1 : Pb, dqt_hi, dqt_lo,
                               at<float>(A, L.attn_work), stream));
  EGV_TRY(wgrad(0, dqt_hi, dqt_lo, 3 * D, n3_hi, n3_lo, D));
  float* d_n3 = at<float>(A, L.d_n3);
  {
    egv_gemm_desc d = nt_desc(dqt_hi, dqt_lo, 3 * D, p.wt_hi[0], p.wt_lo[0], p.ldwt[0], M, D, 3 * D, Pg, g.grid_cap);
    ln_in(d, d_n3);
    EGV_TRY(egv_gemm_nt(&d, stream));
  }
  // x feeds norm3, the tr residual and the sr residual: dx = d_tr + d_sr + LN3'(d_n3)
  EGV_TRY(egv_layernorm_bwd_partial(h16 ? nullptr : d_n3, h16 ? (const egv_bf16*)d_n3 : nullptr, nullptr, D, io.x, D, p.n3w, at<float>(FA, F.mean3),
                            at<float>(FA, F.rstd3), M, D, d_tr, d_sr,
                            io.d_x, D, io.dx_hi, Pb == 3 ? io.dx_lo : nullptr, ln_fmt, grads + goff[12], grads + goff[13],
                            at<float>(A, L.ln_work[2]), stream));
  if (defer_reduce) {
    const float* part[6]; float* outp[6]; float* csp[6]; int64_t mn[6]; int32_t ksv[6], mv[6];
    int cnt = 0;
    for (int i = 0; i < 6; ++i) {
      if (io.wgrad_ksplit[i] <= 1) continue;
      int64_t N, K;
      wshape(o, i, N, K);
      part[cnt] = at<float>(A, L.partial[i]); outp[cnt] = grads + goff[i]; csp[cnt] = grads + goff[6 + i];
      mn[cnt] = N * K; ksv[cnt] = io.wgrad_ksplit[i]; mv[cnt] = (int32_t)N;
      ++cnt;
    }
    // on the stream the weight gradients ran on (stream order: behind the last of them)
    if (cnt) EGV_TRY(egv_splitk_reduce_multi(cnt, part, outp, mn, ksv, csp, mv, io.side_stream[0] ? io.side_stream[0] : stream));
  }
  // the affine gradients of the three LayerNorms: their per-block partial sums are reduced by ONE launch (three before)
  {
    const float* w3[3] = {at<float>(A, L.ln_work[0]), at<float>(A, L.ln_work[1]), at<float>(A, L.ln_work[2])};
    float* g3[3] = {grads + goff[16], grads + goff[14], grads + goff[12]};
    float* b3[3] = {grads + goff[17], grads + goff[15], grads + goff[13]};
    EGV_TRY(egv_layernorm_bwd_reduce(3, w3, M, D, g3, b3, stream));
  }
  return EGV_OK;
}
