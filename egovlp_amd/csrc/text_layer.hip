// One C-ABI call per DistilBERT TransformerBlock direction (HF modeling_distilbert.py TransformerBlock :227-259, post-LN:
// sa = LN(out_lin(MHA(x)) + x); out = LN(lin2(gelu(lin1(sa))) + sa)) -- the text tower's counterpart of block.hip.  Nothing new is
// computed: every launch below is one of the library's own entry points with the arguments the per-kernel Python path
// (egovlp_amd/model/text_transformer.py::_TextLayerFn, kept as the reference) gives them, in the same order, on ONE stream; the
// split-K factors of the latency-bound M = B*L GEMMs are the caller's (the same policy functions as the per-kernel path), so the
// results are bit-identical (tests/test_gpu_block.py).  Weight index: 0 = q/k/v fused [3D, D], 1 = out_lin, 2 = lin1, 3 = lin2.
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

template <class T>
T* at(void* base, int64_t off) { return off < 0 ? nullptr : (T*)((char*)base + off); }
template <class T>
const T* at(const void* base, int64_t off) { return off < 0 ? nullptr : (const T*)((const char*)base + off); }

#define EGV_TRY(call)                \
  do {                               \
    const int rc__ = (call);         \
    if (rc__ != EGV_OK) return rc__; \
  } while (0)

bool geom_ok(const egv_text_geom& g) {
  if (g.B <= 0 || g.L <= 0 || g.H <= 0 || g.D != g.H * 64 || g.Hd <= 0 || g.D % 32 || g.Hd % 32) return false;
  if ((g.fwd_passes != 1 && g.fwd_passes != 3) || (g.bwd_passes != 1 && g.bwd_passes != 3) || g.bwd_passes > g.fwd_passes) return false;
  if (!(g.attn_p >= 0.0f && g.attn_p < 1.0f) || !(g.ffn_p >= 0.0f && g.ffn_p < 1.0f)) return false;
  for (int i = 0; i < 4; ++i)
    if (g.nt_ksplit_fwd[i] < 1 || g.nt_ksplit_bwd[i] < 1 || g.wgrad_ksplit[i] < 1) return false;
  return true;
}

// (N, K) of weight i
void wshape(const egv_text_geom& g, int i, int64_t& N, int64_t& K) {
  switch (i) {
    case 0: N = 3 * (int64_t)g.D; K = g.D; break;
    case 1: N = g.D; K = g.D; break;
    case 2: N = g.Hd; K = g.D; break;
    default: N = g.D; K = g.Hd; break;
  }
}

struct FwdLayout {
  int64_t x_hi, x_lo, qkv, c_hi, c_lo, lse, s1, sa_hi, sa_lo, sa, mean1, rstd1, h_hi, h_lo, z, s2raw, s2, mean2, rstd2, partial;
  int64_t total;
};

FwdLayout fwd_layout(const egv_text_geom& g) {
  const int64_t M = (int64_t)g.B * g.L, D = g.D, Hd = g.Hd;
  const bool lo = g.fwd_passes == 3;
  FwdLayout L;
  Bump b;
  auto plane = [&](int64_t cols) { return b.take(M * cols * 2); };
  auto plane_lo = [&](int64_t cols) { return lo ? b.take(M * cols * 2) : (int64_t)-1; };
  L.x_hi = plane(D); L.x_lo = plane_lo(D);
  L.qkv = b.take(M * 3 * D * 4);
  L.c_hi = plane(D); L.c_lo = plane_lo(D);
  L.lse = b.take((int64_t)g.B * g.H * g.L * 4);
  L.s1 = b.take(M * D * 4);
  L.sa_hi = plane(D); L.sa_lo = plane_lo(D);
  L.sa = b.take(M * D * 4);
  L.mean1 = b.take(M * 4); L.rstd1 = b.take(M * 4);
  L.h_hi = plane(Hd); L.h_lo = plane_lo(Hd);
  L.z = g.train ? b.take(M * Hd * 4) : (int64_t)-1;
  L.s2raw = g.ffn_p > 0.0f ? b.take(M * D * 4) : (int64_t)-1;
  L.s2 = b.take(M * D * 4);
  L.mean2 = b.take(M * 4); L.rstd2 = b.take(M * 4);
  // split-K slabs of the four forward GEMMs: one region, the GEMMs follow each other on one stream
  int64_t pmax = 0;
  for (int i = 0; i < 4; ++i) {
    int64_t N, K;
    wshape(g, i, N, K);
    if (g.nt_ksplit_fwd[i] > 1) pmax = pmax > g.nt_ksplit_fwd[i] * M * N * 4 ? pmax : g.nt_ksplit_fwd[i] * M * N * 4;
  }
  L.partial = pmax ? b.take(pmax) : (int64_t)-1;
  L.total = b.off;
  return L;
}

struct BwdLayout {
  int64_t d_s2, g_in, g_hi, g_lo, dz_hi, dz_lo, d_sa, d_s1, ds1_hi, ds1_lo, d_ctx, dqkv, dqkv_hi, dqkv_lo, ln_work, attn_work, partial;
  int64_t total;
};

// the backward's NT GEMMs, in launch order: 0 dZ = g . W2^T^T [M, Hd] (k = D), 1 d_sa [M, D] (k = Hd), 2 d_ctx [M, D] (k = D),
// 3 d_x [M, D] (k = 3D)  ->  output columns
int64_t bwd_nt_cols(const egv_text_geom& g, int i) { return i == 0 ? g.Hd : g.D; }

BwdLayout bwd_layout(const egv_text_geom& g) {
  const int64_t M = (int64_t)g.B * g.L, D = g.D, Hd = g.Hd;
  const bool lo = g.bwd_passes == 3;
  BwdLayout L;
  Bump b;
  auto plane = [&](int64_t cols) { return b.take(M * cols * 2); };
  auto plane_lo = [&](int64_t cols) { return lo ? b.take(M * cols * 2) : (int64_t)-1; };
  L.d_s2 = b.take(M * D * 4);
  L.g_in = g.ffn_p > 0.0f ? b.take(M * D * 4) : (int64_t)-1;
  L.g_hi = plane(D); L.g_lo = plane_lo(D);
  L.dz_hi = plane(Hd); L.dz_lo = plane_lo(Hd);
  L.d_sa = b.take(M * D * 4);
  L.d_s1 = b.take(M * D * 4);
  L.ds1_hi = plane(D); L.ds1_lo = plane_lo(D);
  L.d_ctx = b.take(M * D * 4);
  L.dqkv = b.take(M * 3 * D * 4);
  L.dqkv_hi = plane(3 * D); L.dqkv_lo = plane_lo(3 * D);
  L.ln_work = b.take(2 * D * (int64_t)egv_layernorm_bwd_parts((int32_t)M) * 4);
  L.attn_work = b.take((int64_t)g.B * g.H * g.L * 4);
  int64_t pmax = 0;
  for (int i = 0; i < 4; ++i) {
    if (g.nt_ksplit_bwd[i] > 1) {
      const int64_t by = g.nt_ksplit_bwd[i] * M * bwd_nt_cols(g, i) * 4;
      pmax = pmax > by ? pmax : by;
    }
    int64_t N, K;
    wshape(g, i, N, K);
    if (g.wgrad_ksplit[i] > 1) {
      const int64_t by = (int64_t)g.wgrad_ksplit[i] * (N * K + N) * 4;
      pmax = pmax > by ? pmax : by;
    }
  }
  L.partial = pmax ? b.take(pmax) : (int64_t)-1;
  L.total = b.off;
  return L;
}

// gradient buffer (floats, back to back): dW x 4, db x 4, sa_layer_norm (dgamma, dbeta), output_layer_norm (dgamma, dbeta)
void grad_layout(const egv_text_geom& g, int64_t off[12], int64_t& total) {
  int64_t p = 0;
  auto take = [&](int64_t n) { const int64_t q = p; p += n; return q; };
  for (int i = 0; i < 4; ++i) { int64_t N, K; wshape(g, i, N, K); off[i] = take(N * K); }
  for (int i = 0; i < 4; ++i) { int64_t N, K; wshape(g, i, N, K); off[4 + i] = take(N); }
  for (int i = 0; i < 4; ++i) off[8 + i] = take(g.D);
  total = p;
}

egv_gemm_desc nt_desc(const egv_bf16* a_hi, const egv_bf16* a_lo, int64_t lda, const egv_bf16* b_hi, const egv_bf16* b_lo, int64_t ldb,
                      int64_t M, int64_t N, int64_t K, int passes, int grid_cap, int ksplit, float* partial) {
  egv_gemm_desc d = {};
  d.a_hi = a_hi; d.a_lo = a_lo; d.lda = lda;
  d.b_hi = b_hi; d.b_lo = b_lo; d.ldb = ldb;
  d.M = (int32_t)M; d.N = (int32_t)N; d.K = (int32_t)K; d.passes = passes;
  d.alpha = 1.0f;
  d.ksplit = ksplit;
  d.partial = ksplit > 1 ? partial : nullptr;
  d.grid_cap = grid_cap;
  return d;
}

}  // namespace

extern "C" int64_t egv_text_layer_fwd_arena_bytes(const egv_text_geom* g) { return (g && geom_ok(*g)) ? fwd_layout(*g).total : -1; }
extern "C" int64_t egv_text_layer_bwd_arena_bytes(const egv_text_geom* g) { return (g && geom_ok(*g)) ? bwd_layout(*g).total : -1; }

extern "C" int egv_text_layer_grad_layout(const egv_text_geom* g, int64_t* offsets12, int64_t* total_floats) {
  if (!g || !geom_ok(*g) || !offsets12 || !total_floats) return EGV_ERR_ARG;
  grad_layout(*g, offsets12, *total_floats);
  return EGV_OK;
}

extern "C" int egv_text_layer_fwd(const egv_text_geom* gp, const egv_text_params* pp, const float* x, const int64_t* mask, float* out,
                                  void* arena, void* stream) {
  if (!gp || !pp || !x || !mask || !out || !arena || !geom_ok(*gp)) return EGV_ERR_ARG;
  const egv_text_geom& g = *gp;
  const egv_text_params& p = *pp;
  const FwdLayout L = fwd_layout(g);
  const int P = g.fwd_passes;
  const int32_t M = g.B * g.L, D = g.D, Hd = g.Hd;
  for (int i = 0; i < 4; ++i)
    if (!p.w_hi[i] || (P == 3 && !p.w_lo[i])) return EGV_ERR_ARG;
  char* A = (char*)arena;
  egv_bf16 *x_hi = at<egv_bf16>(A, L.x_hi), *x_lo = at<egv_bf16>(A, L.x_lo);
  egv_bf16 *c_hi = at<egv_bf16>(A, L.c_hi), *c_lo = at<egv_bf16>(A, L.c_lo);
  egv_bf16 *sa_hi = at<egv_bf16>(A, L.sa_hi), *sa_lo = at<egv_bf16>(A, L.sa_lo);
  egv_bf16 *h_hi = at<egv_bf16>(A, L.h_hi), *h_lo = at<egv_bf16>(A, L.h_lo);
  float *qkv = at<float>(A, L.qkv), *s1 = at<float>(A, L.s1), *sa = at<float>(A, L.sa), *s2 = at<float>(A, L.s2);
  float* part = at<float>(A, L.partial);

  EGV_TRY(egv_split_f32(x, D, M, D, x_hi, x_lo, D, nullptr, nullptr, 0, nullptr, stream));
  {  // q_lin / k_lin / v_lin as one [M, D] x [3D, D]^T GEMM
    egv_gemm_desc d = nt_desc(x_hi, x_lo, D, p.w_hi[0], p.w_lo[0], p.ldw[0], M, 3 * D, D, P, g.grid_cap, g.nt_ksplit_fwd[0], part);
    d.bias = p.bias[0]; d.out_f32 = qkv; d.ldo = 3 * D;
    EGV_TRY(egv_gemm_nt(&d, stream));
  }
  EGV_TRY(egv_text_attn_fwd(qkv, qkv + D, qkv + 2 * D, 3 * D, mask, g.B, g.L, g.H, P, g.attn_p, g.attn_seed, g.seed_dev, c_hi, c_lo,
                            at<float>(A, L.lse), stream));
  {
    egv_gemm_desc d = nt_desc(c_hi, c_lo, D, p.w_hi[1], p.w_lo[1], p.ldw[1], M, D, D, P, g.grid_cap, g.nt_ksplit_fwd[1], part);
    d.bias = p.bias[1]; d.residual = x; d.ldr = D; d.out_f32 = s1; d.ldo = D;
    EGV_TRY(egv_gemm_nt(&d, stream));
  }
  EGV_TRY(egv_layernorm_fwd(s1, nullptr, D, p.ln1w, p.ln1b, g.eps, M, D, nullptr, sa_hi, sa_lo, sa, D, at<float>(A, L.mean1),
                            at<float>(A, L.rstd1), stream));
  {
    egv_gemm_desc d = nt_desc(sa_hi, sa_lo, D, p.w_hi[2], p.w_lo[2], p.ldw[2], M, Hd, D, P, g.grid_cap, g.nt_ksplit_fwd[2], part);
    d.bias = p.bias[2]; d.act = EGV_ACT_GELU; d.out_hi = h_hi; d.out_lo = h_lo; d.ldoh = Hd;
    if (g.train) { d.aux_out = at<float>(A, L.z); d.ldaux = Hd; }
    EGV_TRY(egv_gemm_nt(&d, stream));
  }
  {  // FFN.forward: dropout(lin2(gelu(lin1(x)))), then the block's residual: s2 = drop(y) + sa
    egv_gemm_desc d = nt_desc(h_hi, h_lo, Hd, p.w_hi[3], p.w_lo[3], p.ldw[3], M, D, Hd, P, g.grid_cap, g.nt_ksplit_fwd[3], part);
    d.bias = p.bias[3];
    if (g.ffn_p > 0.0f) {
      float* raw = at<float>(A, L.s2raw);
      d.out_f32 = raw; d.ldo = D;
      EGV_TRY(egv_gemm_nt(&d, stream));
      EGV_TRY(egv_dropout(raw, sa, s2, (int64_t)M * D, g.ffn_p, g.ffn_seed, g.seed_dev, stream));
    } else {
      d.residual = sa; d.ldr = D; d.out_f32 = s2; d.ldo = D;
      EGV_TRY(egv_gemm_nt(&d, stream));
    }
  }
  return egv_layernorm_fwd(s2, nullptr, D, p.ln2w, p.ln2b, g.eps, M, D, nullptr, nullptr, nullptr, out, D, at<float>(A, L.mean2),
                           at<float>(A, L.rstd2), stream);
}

extern "C" int egv_text_layer_bwd(const egv_text_geom* gp, const egv_text_params* pp, const float* g_out, const int64_t* mask,
                                  const void* fwd_arena, void* bwd_arena, float* d_x, float* grads, void* stream) {
  if (!gp || !pp || !g_out || !mask || !fwd_arena || !bwd_arena || !d_x || !grads || !geom_ok(*gp) || !gp->train) return EGV_ERR_ARG;
  const egv_text_geom& g = *gp;
  const egv_text_params& p = *pp;
  const FwdLayout F = fwd_layout(g);
  const BwdLayout L = bwd_layout(g);
  int64_t goff[12], gtot;
  grad_layout(g, goff, gtot);
  const int Pb = g.bwd_passes;
  const int32_t M = g.B * g.L, D = g.D, Hd = g.Hd;
  for (int i = 0; i < 4; ++i)
    if (!p.wt_hi[i] || (Pb == 3 && !p.wt_lo[i])) return EGV_ERR_ARG;
  const char* FA = (const char*)fwd_arena;
  char* A = (char*)bwd_arena;
  auto fpl = [&](int64_t hi, int64_t lo, const egv_bf16*& ph, const egv_bf16*& pl) {
    ph = at<egv_bf16>(FA, hi);
    pl = Pb == 3 ? at<egv_bf16>(FA, lo) : nullptr;
  };
  const egv_bf16 *x_hi, *x_lo, *c_hi, *c_lo, *sa_hi, *sa_lo, *h_hi, *h_lo;
  fpl(F.x_hi, F.x_lo, x_hi, x_lo); fpl(F.c_hi, F.c_lo, c_hi, c_lo); fpl(F.sa_hi, F.sa_lo, sa_hi, sa_lo); fpl(F.h_hi, F.h_lo, h_hi, h_lo);
  const float *qkv = at<float>(FA, F.qkv), *s1 = at<float>(FA, F.s1), *s2 = at<float>(FA, F.s2);
  float* part = at<float>(A, L.partial);
  float* ln_work = at<float>(A, L.ln_work);

  // dW[N,K] = dY^T X (TN kernel; the bias gradient from the same pass) of weight i
  auto wgrad = [&](int i, const egv_bf16* dy_hi, const egv_bf16* dy_lo, int64_t lddy, const egv_bf16* a_hi, const egv_bf16* a_lo,
                   int64_t lda) -> int {
    int64_t N, K;
    wshape(g, i, N, K);
    egv_gemm_desc d = {};
    d.a_hi = dy_hi; d.a_lo = dy_lo; d.lda = lddy;
    d.b_hi = a_hi; d.b_lo = a_lo; d.ldb = lda;
    d.M = (int32_t)N; d.N = (int32_t)K; d.K = M; d.passes = Pb;
    d.alpha = 1.0f;
    d.out_f32 = grads + goff[i]; d.ldo = K;
    d.ksplit = g.wgrad_ksplit[i];
    d.partial = g.wgrad_ksplit[i] > 1 ? part : nullptr;
    d.trans = 1;
    d.colsum = grads + goff[4 + i];
    d.grid_cap = g.grid_cap;
    return egv_gemm_nt(&d, stream);
  };

  // ---- output_layer_norm backward
  float* d_s2 = at<float>(A, L.d_s2);
  EGV_TRY(egv_layernorm_bwd(g_out, nullptr, nullptr, D, s2, D, p.ln2w, at<float>(FA, F.mean2), at<float>(FA, F.rstd2), M, D, nullptr, nullptr,
                            d_s2, D, nullptr, nullptr, grads + goff[10], grads + goff[11], ln_work, stream));
  // ---- FFN: d_s2 reaches lin2 through the dropout mask of the forward; the residual branch takes it as it is
  const float* g_in = d_s2;
  if (g.ffn_p > 0.0f) {
    float* gi = at<float>(A, L.g_in);
    EGV_TRY(egv_dropout(d_s2, nullptr, gi, (int64_t)M * D, g.ffn_p, g.ffn_seed, g.seed_dev, stream));
    g_in = gi;
  }
  egv_bf16 *g_hi = at<egv_bf16>(A, L.g_hi), *g_lo = at<egv_bf16>(A, L.g_lo);
  EGV_TRY(egv_split_f32(g_in, D, M, D, g_hi, g_lo, D, nullptr, nullptr, 0, nullptr, stream));
  egv_bf16 *dz_hi = at<egv_bf16>(A, L.dz_hi), *dz_lo = at<egv_bf16>(A, L.dz_lo);
  {
    egv_gemm_desc d = nt_desc(g_hi, g_lo, D, p.wt_hi[3], p.wt_lo[3], p.ldwt[3], M, Hd, D, Pb, g.grid_cap, g.nt_ksplit_bwd[0], part);
    d.act = EGV_ACT_GELU_BWD; d.aux_in = at<float>(FA, F.z); d.ldaux = Hd;
    d.out_hi = dz_hi; d.out_lo = dz_lo; d.ldoh = Hd;
    EGV_TRY(egv_gemm_nt(&d, stream));
  }
  EGV_TRY(wgrad(3, g_hi, g_lo, D, h_hi, h_lo, Hd));
  EGV_TRY(wgrad(2, dz_hi, dz_lo, Hd, sa_hi, sa_lo, D));
  float* d_sa = at<float>(A, L.d_sa);
  {  // d_sa = d_s2 + dZ . W1
    egv_gemm_desc d = nt_desc(dz_hi, dz_lo, Hd, p.wt_hi[2], p.wt_lo[2], p.ldwt[2], M, D, Hd, Pb, g.grid_cap, g.nt_ksplit_bwd[1], part);
    d.residual = d_s2; d.ldr = D; d.out_f32 = d_sa; d.ldo = D;
    EGV_TRY(egv_gemm_nt(&d, stream));
  }
  // ---- sa_layer_norm backward
  float* d_s1 = at<float>(A, L.d_s1);
  EGV_TRY(egv_layernorm_bwd(d_sa, nullptr, nullptr, D, s1, D, p.ln1w, at<float>(FA, F.mean1), at<float>(FA, F.rstd1), M, D, nullptr, nullptr,
                            d_s1, D, nullptr, nullptr, grads + goff[8], grads + goff[9], ln_work, stream));
  // ---- attention output projection
  egv_bf16 *ds1_hi = at<egv_bf16>(A, L.ds1_hi), *ds1_lo = at<egv_bf16>(A, L.ds1_lo);
  EGV_TRY(egv_split_f32(d_s1, D, M, D, ds1_hi, ds1_lo, D, nullptr, nullptr, 0, nullptr, stream));
  EGV_TRY(wgrad(1, ds1_hi, ds1_lo, D, c_hi, c_lo, D));
  float* d_ctx = at<float>(A, L.d_ctx);
  {
    egv_gemm_desc d = nt_desc(ds1_hi, ds1_lo, D, p.wt_hi[1], p.wt_lo[1], p.ldwt[1], M, D, D, Pb, g.grid_cap, g.nt_ksplit_bwd[2], part);
    d.out_f32 = d_ctx; d.ldo = D;
    EGV_TRY(egv_gemm_nt(&d, stream));
  }
  float* dqkv = at<float>(A, L.dqkv);
  EGV_TRY(egv_text_attn_bwd(qkv, qkv + D, qkv + 2 * D, 3 * D, mask, d_ctx, at<float>(FA, F.lse), g.B, g.L, g.H, Pb, g.attn_p, g.attn_seed,
                            g.seed_dev, dqkv, dqkv + D, dqkv + 2 * D, 3 * D, at<float>(A, L.attn_work), stream));
  // ---- fused q/k/v projection backward: one wgrad (dW [3D, D] + bias gradients) and one dgrad chained onto d_s1
  egv_bf16 *dq_hi = at<egv_bf16>(A, L.dqkv_hi), *dq_lo = at<egv_bf16>(A, L.dqkv_lo);
  EGV_TRY(egv_split_f32(dqkv, 3 * D, M, 3 * D, dq_hi, dq_lo, 3 * D, nullptr, nullptr, 0, nullptr, stream));
  EGV_TRY(wgrad(0, dq_hi, dq_lo, 3 * D, x_hi, x_lo, D));
  {
    egv_gemm_desc d = nt_desc(dq_hi, dq_lo, 3 * D, p.wt_hi[0], p.wt_lo[0], p.ldwt[0], M, D, 3 * D, Pb, g.grid_cap, g.nt_ksplit_bwd[3], part);
    d.residual = d_s1; d.ldr = D; d.out_f32 = d_x; d.ldo = D;
    EGV_TRY(egv_gemm_nt(&d, stream));
  }
  return EGV_OK;
}
