// Fused attention backward on bf16 MFMA (space attention / DistilBERT MHA), flash-style recomputation:
// probabilities are rebuilt from the saved log-sum-exp, nothing of size [q, k] ever reaches HBM.
//
// Two kernels per group because an MFMA contracts over the index that runs along lane-groups/elements:
//   dQ  kernel: wave owns 16-query tiles; S^T, dP^T with keys in registers -> contraction over KEYS
//               (dQ^T = K^T . dS^T, K^T fragments by transpose-read from the row-major K image);
//               also emits delta[q] = sum_k P dP for the second kernel.
//   dKV kernel: wave owns 16-key fragments; S, dP with queries in registers -> contraction over QUERIES
//               (dV^T = dO^T . P, dK^T = Q^T . dS, transposed fragments again by transpose-read of the
//               row-major Q / dO images in LDS).
// CLS-key gradients of the space mode are shared by the T frame-groups of a clip: atomicAdd.
#include <cstdlib>

#include "attn_common.h"
#include "egovlp_hip.h"

namespace {

struct AttGrad {
  float* dq;           // MODE_TEXT: fp32 outputs
  float* dk;
  float* dv;
  bf16_t* gh;          // MODE_SPACE: dqkv as split-bf16 planes [B, S, 3, H, 64] (gl == nullptr: hi only)
  bf16_t* gl;
  long tok_stride;     // elements between tokens in dq/dk/dv (or in the gradient planes)
  const float* d_out;  // MODE_TEXT: [B, S, H*64] fp32
  const bf16_t* doh;   // MODE_SPACE: d_out planes
  const bf16_t* dol;
  long do_stride;
  const bf16_t* oh;    // MODE_SPACE: the forward's attention output planes (delta = rowsum(dO o O) in the streaming dQ kernel)
  const bf16_t* ol;
  const float* lse;    // [B, H, S]
  float* delta;        // [B, H, S] workspace (written by dQ kernel, read by dKV kernel; slot 0 = the CLS row's delta,
                       //  precomputed by egv_attn_cls_delta in MODE_SPACE)
  float* dcls;         // MODE_SPACE: [B, H, 3, 64] fp32 accumulators of the CLS token's raw dq / dk / dv (zeroed first)
  int o_fmt;           // MODE_SPACE: format of the forward's output planes (attn_common.h ATT_OUT_*)
  int g_fmt;           // MODE_SPACE: format of the gradient planes: 0 = split-bf16 (hi[, lo]), ATT_GRAD_F16 = ONE plane of un-clamped fp16
};

__device__ __forceinline__ void store_planes4(bf16_t* hi, bf16_t* lo, long off, f32x4_t v, int fmt = 0) {
  uint32_t h0, h1, l0, l1;
  att_out2(v[0], v[1], fmt, h0, l0);
  att_out2(v[2], v[3], fmt, h1, l1);
  egv_store<EGV_NT_SPACE_ATTN>(hi + off, (u32x2_t){h0, h1});
  if (lo) egv_store<EGV_NT_SPACE_ATTN>(lo + off, (u32x2_t){l0, l1});
}

// ------------------------------------------------------------------------------------------------ dQ
// MODE_SPACE: operands are planes; the clip's CLS query rides as query row n (see attn_mfma_fwd.hip): its L and delta
// are the GLOBAL ones (lse[b,h,0], delta[b,h,0]); its dq partial over this frame's keys is accumulated atomically.
template <int MODE, int NKF, int PASSES, bool F16 = false>
__global__ __launch_bounds__(512) void attn_bwd_dq_kernel(const AttGeom g, const AttGrad gr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NKP = NKF * 16;
  constexpr int PLANE = NKP * ATT_ROW_BYTES;
  constexpr bool SP = (MODE == MODE_SPACE);
  char* k_hi = smem;
  char* v_hi = smem + PLANE;
  char* k_lo = (PASSES == 3) ? smem + 2 * PLANE : nullptr;
  char* v_lo = (PASSES == 3) ? smem + 3 * PLANE : nullptr;
  float* kbias = (float*)(smem + ((PASSES == 3) ? 4 : 2) * PLANE);

  const AttGroup<MODE> grp(g, blockIdx.x);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long hoff = (long)grp.h * ATT_D;
  const long HD = (long)g.H * ATT_D;

  if (SP) {
    att_stage_planes(k_hi, k_lo, g.ph, g.pl, g.nk, NKP, [&](int r) { return grp.k_tok(g, r) * g.tok_stride + HD + hoff; });
    att_stage_planes(v_hi, v_lo, g.ph, g.pl, g.nk, NKP, [&](int r) { return grp.k_tok(g, r) * g.tok_stride + 2 * HD + hoff; });
  } else {
    att_stage(k_hi, k_lo, g.nk, NKP, 1.0f, [&](int r) { return g.k + grp.k_tok(g, r) * g.tok_stride + hoff; });
    att_stage(v_hi, v_lo, g.nk, NKP, 1.0f, [&](int r) { return g.v + grp.k_tok(g, r) * g.tok_stride + hoff; });
  }
  for (int j = threadIdx.x; j < NKP; j += blockDim.x) {
    float bias = (j < g.nk) ? 0.f : -1e30f;
    if (MODE == MODE_TEXT && j < g.nk && g.mask[(long)grp.b * g.S + j] == 0) bias = -1e30f;
    kbias[j] = bias;
  }
  __syncthreads();

  const int gq = lane >> 4;
  const int nq_all = SP ? g.nq + 1 : g.nq;
  const int ntiles = (nq_all + 15) / 16;
  for (int qt = wave; qt < ntiles; qt += (int)(blockDim.x >> 6)) {
    asm volatile("" ::: "memory");  // K/V fragments are loop-invariant: stop LICM from hoisting ~900 VGPRs of them
    const int qi = qt * 16 + (lane & 15);
    const bool is_cls = SP && qi >= g.nq;
    const long tok = is_cls ? grp.tok0 : grp.q_tok(g, min(qi, g.nq - 1));
    bf16x8_t qh[2], ql[2], gh[2], gl[2];
    if (SP) {
      att_gfrag_planes(g.ph, g.pl, tok * g.tok_stride + hoff, 0, lane, qh[0], ql[0]);
      att_gfrag_planes(g.ph, g.pl, tok * g.tok_stride + hoff, 1, lane, qh[1], ql[1]);
      att_gfrag_planes(gr.doh, gr.dol, tok * gr.do_stride + hoff, 0, lane, gh[0], gl[0]);
      att_gfrag_planes(gr.doh, gr.dol, tok * gr.do_stride + hoff, 1, lane, gh[1], gl[1]);
    } else {
      const float* qrow = g.q + tok * g.tok_stride + hoff;
      const float* grow = gr.d_out + tok * gr.do_stride + hoff;
      att_gfrag(qrow, 0, lane, 1.0f, qh[0], ql[0]);
      att_gfrag(qrow, 1, lane, 1.0f, qh[1], ql[1]);
      att_gfrag(grow, 0, lane, 1.0f, gh[0], gl[0]);
      att_gfrag(grow, 1, lane, 1.0f, gh[1], gl[1]);
    }
    const long lrow = ((long)grp.b * g.H + grp.h) * g.S + (tok - grp.tok0);
    const float L = gr.lse[lrow];

    f32x4_t p[NKF], dp[NKF];
    float delta = 0.f;
#pragma unroll
    for (int kf = 0; kf < NKF; ++kf) {
      f32x4_t s = {0.f, 0.f, 0.f, 0.f};
      f32x4_t d = s;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8_t ah = att_frag_cols(k_hi, kf * 16, ks, lane);
        bf16x8_t al = ah;
        if (PASSES == 3) al = att_frag_cols(k_lo, kf * 16, ks, lane);
        s = att_mma<PASSES, F16>(ah, al, qh[ks], ql[ks], s);
        const bf16x8_t bh = att_frag_cols(v_hi, kf * 16, ks, lane);
        bf16x8_t bl = bh;
        if (PASSES == 3) bl = att_frag_cols(v_lo, kf * 16, ks, lane);
        d = att_mma<PASSES, F16>(bh, bl, gh[ks], gl[ks], d);
      }
      const f32x4_t kb = *(const f32x4_t*)(kbias + kf * 16 + 4 * gq);
      if (MODE == MODE_TEXT && g.drop.thresh != 0u) {
        // O = (P o M') V: d = dO . V is the gradient w.r.t. the DROPPED weights; dP = d o M' (the mask of the forward,
        // regenerated), and delta = sum_k P dP = rowsum(dO o O) as without dropout
        const uint64_t rowbase = (((uint64_t)grp.b * g.H + grp.h) * g.S + (uint64_t)min(qi, g.nq - 1)) * g.S;
        const EgvDrop dr = egv_drop_resolve(g.drop);
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r] *= egv_drop_scale(dr, rowbase + kf * 16 + 4 * gq + r);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p[kf][r] = __expf(s[r] * 0.125f + kb[r] - L);
        if (SP && kf == 0 && r == 0 && is_cls && grp.f > 0 && gq == 0) p[kf][r] = 0.f;   // CLS key x CLS query: group 0 only
        delta += p[kf][r] * d[r];
      }
      dp[kf] = d;
    }
    delta += __shfl_xor(delta, 16, 64);
    delta += __shfl_xor(delta, 32, 64);
    if (is_cls) delta = gr.delta[lrow];             // the CLS row's delta spans all frame groups: precomputed

    f32x4_t dq[4];
#pragma unroll
    for (int df = 0; df < 4; ++df) dq[df] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NKF / 2; ++c) {
      float dsv[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dsv[r] = p[2 * c][r] * (dp[2 * c][r] - delta);
        dsv[4 + r] = p[2 * c + 1][r] * (dp[2 * c + 1][r] - delta);
      }
      bf16x8_t sh, sl;
      att_split8<F16>(dsv, sh, sl);
#pragma unroll
      for (int df = 0; df < 4; ++df) {
        const bf16x8_t kh = att_frag_rows(k_hi, 32 * c, df * 16, lane);
        bf16x8_t kl = kh;
        if (PASSES == 3) kl = att_frag_rows(k_lo, 32 * c, df * 16, lane);
        dq[df] = att_mma<PASSES, F16>(kh, kl, sh, sl, dq[df]);
      }
    }
    if (SP && qi == g.nq) {
      float* a = gr.dcls + ((long)grp.b * g.H + grp.h) * 192;
#pragma unroll
      for (int df = 0; df < 4; ++df)
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(a + df * 16 + 4 * gq + r, dq[df][r]);
    } else if (qi < g.nq) {
      if (SP) {
#pragma unroll
        for (int df = 0; df < 4; ++df)
          store_planes4(gr.gh, gr.gl, tok * gr.tok_stride + hoff + df * 16 + 4 * gq, dq[df] * 0.125f, gr.g_fmt);
      } else {
        float* out = gr.dq + tok * gr.tok_stride + hoff;
#pragma unroll
        for (int df = 0; df < 4; ++df) *(f32x4_t*)(out + df * 16 + 4 * gq) = dq[df] * 0.125f;
      }
      if (gq == 0) gr.delta[lrow] = delta;
    }
  }
}

// ---- streaming dQ for the space mode ------------------------------------------------------------------------------------------
// The kernel above needs delta = sum_k P dP before it can form dS, so it keeps P and dP of ALL keys in registers (112 fp32,
// 171 VGPRs single-pass: two waves per SIMD, one workgroup per CU).  delta is also rowsum(dO o O) -- one dot product of the
// query's dO row with the forward's output row -- so here it is computed up front from the O planes and the keys are walked
// in 32-key chunks with nothing but the dQ accumulators live: under 128 VGPRs, i.e. two 8-wave workgroups per CU in
// single-pass mode (the staging of one hides under the tiles of the other) and 16 waves per workgroup in three-pass mode.
// a = dO (split-bf16), b = O in the format the forward wrote it (b_fmt: ATT_OUT_*)
template <bool F16 = false>
__device__ __forceinline__ float frag_dot8(bf16x8_t ah, bf16x8_t al, bool a_lo, bf16x8_t bh, bf16x8_t bl, bool b_lo, int b_fmt) {
  const u32x4_t a0 = __builtin_bit_cast(u32x4_t, ah), a1 = __builtin_bit_cast(u32x4_t, al);
  const u32x4_t b0 = __builtin_bit_cast(u32x4_t, bh), b1 = __builtin_bit_cast(u32x4_t, bl);
  float acc = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float ax, ay, bx, by;
    if constexpr (F16) {
      f16x2_unpack(a0[e], ax, ay);           // dO as an fp16 plane (the fp16 attention backward)
    } else {
      ax = __uint_as_float(a0[e] << 16), ay = __uint_as_float(a0[e] & 0xffff0000u);
      if (a_lo) { ax += __uint_as_float(a1[e] << 16); ay += __uint_as_float(a1[e] & 0xffff0000u); }
    }
    att_o_unpack(b0[e], b1[e], b_lo, b_fmt, bx, by);
    acc += ax * bx + ay * by;
  }
  return acc;
}

template <int NKF, int PASSES, bool F16 = false>
__global__ __launch_bounds__(PASSES == 3 ? 1024 : 512) void attn_bwd_dq_stream_kernel(const AttGeom g, const AttGrad gr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NKP = NKF * 16;
  constexpr int PLANE = NKP * ATT_ROW_BYTES;
  char* k_hi = smem;
  char* v_hi = smem + PLANE;
  char* k_lo = (PASSES == 3) ? smem + 2 * PLANE : nullptr;
  char* v_lo = (PASSES == 3) ? smem + 3 * PLANE : nullptr;
  float* kbias = (float*)(smem + ((PASSES == 3) ? 4 : 2) * PLANE);

  const AttGroup<MODE_SPACE> grp(g, blockIdx.x);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long hoff = (long)grp.h * ATT_D;
  const long HD = (long)g.H * ATT_D;
  att_stage_planes(k_hi, k_lo, g.ph, g.pl, g.nk, NKP, [&](int r) { return grp.k_tok(g, r) * g.tok_stride + HD + hoff; });
  att_stage_planes(v_hi, v_lo, g.ph, g.pl, g.nk, NKP, [&](int r) { return grp.k_tok(g, r) * g.tok_stride + 2 * HD + hoff; });
  for (int j = threadIdx.x; j < NKP; j += blockDim.x) kbias[j] = (j < g.nk) ? 0.f : -1e30f;
  __syncthreads();

  const int gq = lane >> 4;
  const int nq_all = g.nq + 1;
  const int ntiles = (nq_all + 15) / 16;
  for (int qt = wave; qt < ntiles; qt += (int)(blockDim.x >> 6)) {
    const int qi = qt * 16 + (lane & 15);
    const bool is_cls = qi >= g.nq;
    const long tok = is_cls ? grp.tok0 : grp.q_tok(g, min(qi, g.nq - 1));
    bf16x8_t qh[2], ql[2], gh[2], gl[2], oh[2], ol[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      att_gfrag_planes(g.ph, g.pl, tok * g.tok_stride + hoff, ks, lane, qh[ks], ql[ks]);
      att_gfrag_planes(gr.doh, gr.dol, tok * gr.do_stride + hoff, ks, lane, gh[ks], gl[ks]);
      att_gfrag_planes(gr.oh, gr.ol, tok * gr.do_stride + hoff, ks, lane, oh[ks], ol[ks]);
    }
    const long lrow = ((long)grp.b * g.H + grp.h) * g.S + (tok - grp.tok0);
    constexpr float LOG2E = 1.4426950408889634f;
    const float L2 = gr.lse[lrow] * LOG2E;          // the probabilities are rebuilt in the exp2 domain: P = 2^(s 64^-0.5 log2 e + bias - L log2 e)
    float delta = frag_dot8<F16>(gh[0], gl[0], gr.dol != nullptr, oh[0], ol[0], gr.ol != nullptr, gr.o_fmt) +
                  frag_dot8<F16>(gh[1], gl[1], gr.dol != nullptr, oh[1], ol[1], gr.ol != nullptr, gr.o_fmt);
    delta += __shfl_xor(delta, 16, 64);
    delta += __shfl_xor(delta, 32, 64);
    if (is_cls) delta = gr.delta[lrow];             // the CLS row's delta spans all frame groups: precomputed

    // As in the forward (attn_mfma_fwd.hip, attn_fwd_stream3_kernel): the lane part of every LDS fragment address is resolved once per
    // tile -- a chunk adds its 4 KiB, the rest (second fragment, k-step, V / lo plane) are immediates -- and every exponential is one
    // FMA + v_exp_f32: 8 % fewer instructions in the chunk loop.  (No measurable effect on the launch, profiles/r06ab_*: with two
    // workgroups per CU this kernel waits for its staging and its per-tile q / dO / O fetches, not for the VALU.)
    constexpr int VOFF = PLANE;                      // v_hi - k_hi
    constexpr int LOFF = 2 * PLANE;                  // k_lo - k_hi == v_lo - v_hi (PASSES == 3)
    const int r15 = lane & 15;
    const unsigned kc[2] = {(unsigned)(r15 * ATT_ROW_BYTES + ((((lane >> 4)) ^ (r15 & 7)) << 4)),
                            (unsigned)(r15 * ATT_ROW_BYTES + ((((lane >> 4) + 4) ^ (r15 & 7)) << 4))};
    unsigned ko[4];
#pragma unroll
    for (int df = 0; df < 4; ++df) ko[df] = (unsigned)att_off(4 * gq + (r15 >> 2), df * 16 + ((r15 & 3) << 2));

    f32x4_t dq[4];
#pragma unroll
    for (int df = 0; df < 4; ++df) dq[df] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int c = 0; c < NKF / 2; ++c) {
      float dsv[8];
      const char* kp[2] = {k_hi + (c * 4096 + kc[0]), k_hi + (c * 4096 + kc[1])};
      const float* kbp = kbias + (c * 32 + 4 * gq);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x4_t sc = {0.f, 0.f, 0.f, 0.f};
        f32x4_t d = sc;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8_t ah = *(const bf16x8_t*)(kp[ks] + h * 2048);
          bf16x8_t al = ah;
          if (PASSES == 3) al = *(const bf16x8_t*)(kp[ks] + h * 2048 + LOFF);
          sc = att_mma<PASSES, F16>(ah, al, qh[ks], ql[ks], sc);
          const bf16x8_t bh = *(const bf16x8_t*)(kp[ks] + h * 2048 + VOFF);
          bf16x8_t bl = bh;
          if (PASSES == 3) bl = *(const bf16x8_t*)(kp[ks] + h * 2048 + VOFF + LOFF);
          d = att_mma<PASSES, F16>(bh, bl, gh[ks], gl[ks], d);
        }
        const f32x4_t kb = *(const f32x4_t*)(kbp + h * 16);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float pr = __builtin_amdgcn_exp2f(sc[r] * (0.125f * LOG2E) + (kb[r] - L2));
          if (c == 0 && h == 0 && r == 0 && is_cls && grp.f > 0 && gq == 0) pr = 0.f;   // CLS key x CLS query: group 0 only
          dsv[4 * h + r] = pr * (d[r] - delta);
        }
      }
      bf16x8_t sh, sl;
      att_split8<F16>(dsv, sh, sl);
      const char* kr = k_hi + c * 4096;
#pragma unroll
      for (int df = 0; df < 4; ++df) {
#if defined(EGV_NO_TR_READ)
        const bf16x8_t kh = att_frag_rows(k_hi, 32 * c, df * 16, lane);
        bf16x8_t kl = kh;
        if (PASSES == 3) kl = att_frag_rows(k_lo, 32 * c, df * 16, lane);
#else
        const bf16x8_t kh = att_frag_rows_at(kr + ko[df]);
        bf16x8_t kl = kh;
        if (PASSES == 3) kl = att_frag_rows_at(kr + ko[df] + LOFF);
#endif
        dq[df] = att_mma<PASSES, F16>(kh, kl, sh, sl, dq[df]);
      }
    }
    if (qi == g.nq) {
      float* a = gr.dcls + ((long)grp.b * g.H + grp.h) * 192;
#pragma unroll
      for (int df = 0; df < 4; ++df)
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(a + df * 16 + 4 * gq + r, dq[df][r]);
    } else if (qi < g.nq) {
#pragma unroll
      for (int df = 0; df < 4; ++df)
        store_planes4(gr.gh, gr.gl, tok * gr.tok_stride + hoff + df * 16 + 4 * gq, dq[df] * 0.125f, gr.g_fmt);
      if (gq == 0) gr.delta[lrow] = delta;
    }
  }
}

// ---- streaming dQ for DistilBERT captions of 65 .. 288 tokens -------------------------------------------------------------------
// The register-resident dQ kernel keeps P and dP of ALL keys of a 16-query tile in registers: 112 / 144 fp32 at 14 / 18 key fragments
// -- the three-product instances spilled 33 / 113 VGPRs (round-5 verdict, weak #7: never hit by the 32-token benchmark, but the
// reference's tokenizer pads to the longest caption of a batch, trainer/trainer_egoclip.py:115-117).  The text attention has no saved
// output planes to take delta = rowsum(dO o O) from, so this kernel walks the keys TWICE: pass 1 accumulates delta = sum_k P dP, pass 2
// recomputes P and dP chunk by chunk and contracts dS = P o (dP - delta) with K.  Twice the score MFMAs of a kernel nobody waits for,
// and nothing but the dQ accumulators live across a chunk.
template <int NKF, int PASSES>
__global__ __launch_bounds__(512) void attn_bwd_dq_text_stream_kernel(const AttGeom g, const AttGrad gr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NKP = NKF * 16;
  constexpr int PLANE = NKP * ATT_ROW_BYTES;
  char* k_hi = smem;
  char* v_hi = smem + PLANE;
  char* k_lo = (PASSES == 3) ? smem + 2 * PLANE : nullptr;
  char* v_lo = (PASSES == 3) ? smem + 3 * PLANE : nullptr;
  float* kbias = (float*)(smem + ((PASSES == 3) ? 4 : 2) * PLANE);

  const AttGroup<MODE_TEXT> grp(g, blockIdx.x);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long hoff = (long)grp.h * ATT_D;
  att_stage(k_hi, k_lo, g.nk, NKP, 1.0f, [&](int r) { return g.k + grp.k_tok(g, r) * g.tok_stride + hoff; });
  att_stage(v_hi, v_lo, g.nk, NKP, 1.0f, [&](int r) { return g.v + grp.k_tok(g, r) * g.tok_stride + hoff; });
  for (int j = threadIdx.x; j < NKP; j += blockDim.x)
    kbias[j] = (j < g.nk && g.mask[(long)grp.b * g.S + j] != 0) ? 0.f : -1e30f;
  __syncthreads();

  const int gq = lane >> 4;
  const int ntiles = (g.nq + 15) / 16;
  const EgvDrop dr = egv_drop_resolve(g.drop);
  for (int qt = wave; qt < ntiles; qt += (int)(blockDim.x >> 6)) {
    const int qi = qt * 16 + (lane & 15);
    const long tok = grp.q_tok(g, min(qi, g.nq - 1));
    bf16x8_t qh[2], ql[2], gh[2], gl[2];
    const float* qrow = g.q + tok * g.tok_stride + hoff;
    const float* grow = gr.d_out + tok * gr.do_stride + hoff;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      att_gfrag(qrow, ks, lane, 1.0f, qh[ks], ql[ks]);
      att_gfrag(grow, ks, lane, 1.0f, gh[ks], gl[ks]);
    }
    const long lrow = ((long)grp.b * g.H + grp.h) * g.S + (tok - grp.tok0);
    const float L = gr.lse[lrow];
    const uint64_t rowbase = (((uint64_t)grp.b * g.H + grp.h) * g.S + (uint64_t)min(qi, g.nq - 1)) * g.S;
    // P and dP of key fragment kf for this lane's four keys (rows 4 gq .. + 3 of the fragment), dropout mask applied to dP
    auto p_dp = [&](int kf, f32x4_t& pr, f32x4_t& d) {
      f32x4_t sc = {0.f, 0.f, 0.f, 0.f};
      d = sc;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8_t ah = att_frag_cols(k_hi, kf * 16, ks, lane);
        bf16x8_t al = ah;
        if (PASSES == 3) al = att_frag_cols(k_lo, kf * 16, ks, lane);
        sc = att_mma<PASSES>(ah, al, qh[ks], ql[ks], sc);
        const bf16x8_t bh = att_frag_cols(v_hi, kf * 16, ks, lane);
        bf16x8_t bl = bh;
        if (PASSES == 3) bl = att_frag_cols(v_lo, kf * 16, ks, lane);
        d = att_mma<PASSES>(bh, bl, gh[ks], gl[ks], d);
      }
      const f32x4_t kb = *(const f32x4_t*)(kbias + kf * 16 + 4 * gq);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pr[r] = __expf(sc[r] * 0.125f + kb[r] - L);
        if (g.drop.thresh != 0u) d[r] *= egv_drop_scale(dr, rowbase + kf * 16 + 4 * gq + r);      // dP = (dO . V) o M'
      }
    };
    float delta = 0.f;
#pragma unroll 1
    for (int kf = 0; kf < NKF; ++kf) {
      f32x4_t pr, d;
      p_dp(kf, pr, d);
      delta += pr[0] * d[0] + pr[1] * d[1] + pr[2] * d[2] + pr[3] * d[3];
    }
    delta += __shfl_xor(delta, 16, 64);
    delta += __shfl_xor(delta, 32, 64);

    f32x4_t dq[4];
#pragma unroll
    for (int df = 0; df < 4; ++df) dq[df] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int c = 0; c < NKF / 2; ++c) {
      float dsv[8];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x4_t pr, d;
        p_dp(2 * c + h, pr, d);
#pragma unroll
        for (int r = 0; r < 4; ++r) dsv[4 * h + r] = pr[r] * (d[r] - delta);
      }
      bf16x8_t sh, sl;
      att_split8(dsv, sh, sl);
#pragma unroll
      for (int df = 0; df < 4; ++df) {
        const bf16x8_t kh = att_frag_rows(k_hi, 32 * c, df * 16, lane);
        bf16x8_t kl = kh;
        if (PASSES == 3) kl = att_frag_rows(k_lo, 32 * c, df * 16, lane);
        dq[df] = att_mma<PASSES>(kh, kl, sh, sl, dq[df]);
      }
    }
    if (qi < g.nq) {
      float* out = gr.dq + tok * gr.tok_stride + hoff;
#pragma unroll
      for (int df = 0; df < 4; ++df) *(f32x4_t*)(out + df * 16 + 4 * gq) = dq[df] * 0.125f;
      if (gq == 0) gr.delta[lrow] = delta;
    }
  }
}

// ----------------------------------------------------------------------------------------------- dKV
template <int MODE, int NQF, int PASSES, bool F16 = false>
__global__ __launch_bounds__(512) void attn_bwd_dkv_kernel(const AttGeom g, const AttGrad gr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NQP = NQF * 16;
  constexpr int PLANE = NQP * ATT_ROW_BYTES;
  constexpr bool SP = (MODE == MODE_SPACE);
  char* q_hi = smem;
  char* o_hi = smem + PLANE;
  char* q_lo = (PASSES == 3) ? smem + 2 * PLANE : nullptr;
  char* o_lo = (PASSES == 3) ? smem + 3 * PLANE : nullptr;
  float* lse_s = (float*)(smem + ((PASSES == 3) ? 4 : 2) * PLANE);
  float* del_s = lse_s + NQP;

  const AttGroup<MODE> grp(g, blockIdx.x);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long hoff = (long)grp.h * ATT_D;
  const long HD = (long)g.H * ATT_D;
  const int nq_all = SP ? g.nq + 1 : g.nq;          // query row n = the clip's CLS query (MODE_SPACE)
  auto qrow_tok = [&](int r) { return (SP && r >= g.nq) ? grp.tok0 : grp.q_tok(g, r); };

  if (SP) {
    att_stage_planes(q_hi, q_lo, g.ph, g.pl, nq_all, NQP, [&](int r) { return qrow_tok(r) * g.tok_stride + hoff; });
    att_stage_planes(o_hi, o_lo, gr.doh, gr.dol, nq_all, NQP, [&](int r) { return qrow_tok(r) * gr.do_stride + hoff; });
  } else {
    att_stage(q_hi, q_lo, g.nq, NQP, 1.0f, [&](int r) { return g.q + grp.q_tok(g, r) * g.tok_stride + hoff; });
    att_stage(o_hi, o_lo, g.nq, NQP, 1.0f, [&](int r) { return gr.d_out + grp.q_tok(g, r) * gr.do_stride + hoff; });
  }
  for (int i = threadIdx.x; i < NQP; i += blockDim.x) {
    float L = 1e30f, dl = 0.f;  // padded query rows: P = exp(s - 1e30) = 0
    if (i < nq_all) {
      const long lrow = ((long)grp.b * g.H + grp.h) * g.S + (qrow_tok(i) - grp.tok0);
      L = gr.lse[lrow];
      dl = gr.delta[lrow];
    }
    lse_s[i] = L * 1.4426950408889634f;        // exp2 domain (below)
    del_s[i] = dl;
  }
  __syncthreads();

  const int gq = lane >> 4;
  const int nkfrags = (g.nk + 15) / 16;
  // lane-resolved LDS fragment bases, as in the dQ kernel: a chunk adds its 4 KiB (32 query rows); second fragment, k-step, dO / lo
  // plane are immediates
  constexpr float LOG2E = 1.4426950408889634f;
  constexpr int OOFF = PLANE;                        // o_hi - q_hi
  constexpr int LOFF = 2 * PLANE;                    // q_lo - q_hi == o_lo - o_hi (PASSES == 3)
  const int r15 = lane & 15;
  const unsigned qc[2] = {(unsigned)(r15 * ATT_ROW_BYTES + ((((lane >> 4)) ^ (r15 & 7)) << 4)),
                          (unsigned)(r15 * ATT_ROW_BYTES + ((((lane >> 4) + 4) ^ (r15 & 7)) << 4))};
  unsigned qo[4];
#pragma unroll
  for (int df = 0; df < 4; ++df) qo[df] = (unsigned)att_off(4 * gq + (r15 >> 2), df * 16 + ((r15 & 3) << 2));
  for (int kf = wave; kf < nkfrags; kf += (int)(blockDim.x >> 6)) {
    const int kj = kf * 16 + (lane & 15);
    const int kc = min(kj, g.nk - 1);
    const long ktok = grp.k_tok(g, kc);
    bf16x8_t kh[2], kl[2], vh[2], vl[2];
    if (SP) {
      att_gfrag_planes(g.ph, g.pl, ktok * g.tok_stride + HD + hoff, 0, lane, kh[0], kl[0]);
      att_gfrag_planes(g.ph, g.pl, ktok * g.tok_stride + HD + hoff, 1, lane, kh[1], kl[1]);
      att_gfrag_planes(g.ph, g.pl, ktok * g.tok_stride + 2 * HD + hoff, 0, lane, vh[0], vl[0]);
      att_gfrag_planes(g.ph, g.pl, ktok * g.tok_stride + 2 * HD + hoff, 1, lane, vh[1], vl[1]);
    } else {
      const float* krow = g.k + ktok * g.tok_stride + hoff;
      const float* vrow = g.v + ktok * g.tok_stride + hoff;
      att_gfrag(krow, 0, lane, 1.0f, kh[0], kl[0]);
      att_gfrag(krow, 1, lane, 1.0f, kh[1], kl[1]);
      att_gfrag(vrow, 0, lane, 1.0f, vh[0], vl[0]);
      att_gfrag(vrow, 1, lane, 1.0f, vh[1], vl[1]);
    }
    float kb = (kj < g.nk) ? 0.f : -1e30f;      // (0 or -1e30: the same in the exp2 domain)
    if (MODE == MODE_TEXT && kj < g.nk && g.mask[(long)grp.b * g.S + kj] == 0) kb = -1e30f;
    const bool excl_cls = SP && grp.f > 0 && kj == 0;   // CLS key x CLS query is counted in frame-group 0 only

    f32x4_t dk[4], dv[4];
#pragma unroll
    for (int df = 0; df < 4; ++df) {
      dk[df] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      dv[df] = dk[df];
    }
#pragma unroll 1
    for (int c = 0; c < NQF / 2; ++c) {
      float pv[8], dsv[8];
      const char* qp[2] = {q_hi + (c * 4096 + qc[0]), q_hi + (c * 4096 + qc[1])};
      const float* lp = lse_s + (c * 32 + 4 * gq);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int r0 = 32 * c + 16 * t;
        f32x4_t s = {0.f, 0.f, 0.f, 0.f};
        f32x4_t d = s;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8_t ah = *(const bf16x8_t*)(qp[ks] + t * 2048);
          bf16x8_t al = ah;
          if (PASSES == 3) al = *(const bf16x8_t*)(qp[ks] + t * 2048 + LOFF);
          s = att_mma<PASSES, F16>(ah, al, kh[ks], kl[ks], s);
          const bf16x8_t bh = *(const bf16x8_t*)(qp[ks] + t * 2048 + OOFF);
          bf16x8_t bl = bh;
          if (PASSES == 3) bl = *(const bf16x8_t*)(qp[ks] + t * 2048 + OOFF + LOFF);
          d = att_mma<PASSES, F16>(bh, bl, vh[ks], vl[ks], d);
        }
        const f32x4_t L4 = *(const f32x4_t*)(lp + t * 16);              // L log2(e) (staged so)
        const f32x4_t D4 = *(const f32x4_t*)(lp + t * 16 + NQP);        // del_s = lse_s + NQP
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float pr = __builtin_amdgcn_exp2f(s[r] * (0.125f * LOG2E) + (kb - L4[r]));
          if (excl_cls && r0 + 4 * gq + r == g.nq) pr = 0.f;
          float mk = 1.0f;      // dropout mask of the forward for (query r0 + 4 gq + r, key kj), DistilBERT only
          if (MODE == MODE_TEXT && g.drop.thresh != 0u)
            mk = egv_drop_scale(egv_drop_resolve(g.drop), (((uint64_t)grp.b * g.H + grp.h) * g.S + (uint64_t)min(r0 + 4 * gq + r, g.nq - 1)) * g.S + kc);
          pv[4 * t + r] = pr * mk;                      // dV = (P o M')^T dO
          dsv[4 * t + r] = pr * (d[r] * mk - D4[r]);    // dS = P o (dP - delta), dP = (dO . V) o M'
        }
      }
      bf16x8_t ph, pl, sh, sl;
      att_split8<F16>(pv, ph, pl);
      att_split8<F16>(dsv, sh, sl);
      const char* qr = q_hi + c * 4096;
#pragma unroll
      for (int df = 0; df < 4; ++df) {
#if defined(EGV_NO_TR_READ)
        const bf16x8_t gh = att_frag_rows(o_hi, 32 * c, df * 16, lane);
        bf16x8_t gl = gh;
        if (PASSES == 3) gl = att_frag_rows(o_lo, 32 * c, df * 16, lane);
        const bf16x8_t qh = att_frag_rows(q_hi, 32 * c, df * 16, lane);
        bf16x8_t ql = qh;
        if (PASSES == 3) ql = att_frag_rows(q_lo, 32 * c, df * 16, lane);
#else
        const bf16x8_t gh = att_frag_rows_at(qr + qo[df] + OOFF);
        bf16x8_t gl = gh;
        if (PASSES == 3) gl = att_frag_rows_at(qr + qo[df] + OOFF + LOFF);
        const bf16x8_t qh = att_frag_rows_at(qr + qo[df]);
        bf16x8_t ql = qh;
        if (PASSES == 3) ql = att_frag_rows_at(qr + qo[df] + LOFF);
#endif
        dv[df] = att_mma<PASSES, F16>(gh, gl, ph, pl, dv[df]);
        dk[df] = att_mma<PASSES, F16>(qh, ql, sh, sl, dk[df]);
      }
    }
    if (kj < g.nk) {
      if (SP && kj == 0) {
        // the CLS key / value are shared by the T frame-groups of a clip: raw fp32 accumulation (finish kernel scales)
        float* a = gr.dcls + ((long)grp.b * g.H + grp.h) * 192;
#pragma unroll
        for (int df = 0; df < 4; ++df)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            atomicAdd(a + 64 + df * 16 + 4 * gq + r, dk[df][r]);
            atomicAdd(a + 128 + df * 16 + 4 * gq + r, dv[df][r]);
          }
      } else if (SP) {
#pragma unroll
        for (int df = 0; df < 4; ++df) {
          const long o = ktok * gr.tok_stride + hoff + df * 16 + 4 * gq;
          store_planes4(gr.gh, gr.gl, o + HD, dk[df] * 0.125f, gr.g_fmt);
          store_planes4(gr.gh, gr.gl, o + 2 * HD, dv[df], gr.g_fmt);
        }
      } else {
        float* okp = gr.dk + ktok * gr.tok_stride + hoff;
        float* ovp = gr.dv + ktok * gr.tok_stride + hoff;
#pragma unroll
        for (int df = 0; df < 4; ++df) {
          const int d = df * 16 + 4 * gq;
          *(f32x4_t*)(okp + d) = dk[df] * 0.125f;
          *(f32x4_t*)(ovp + d) = dv[df];
        }
      }
    }
  }
}

template <int MODE, int NF>
int launch_bwd(const AttGeom& g, const AttGrad& gr, int ngroups, int passes, hipStream_t s) {
  const int planes = passes == 3 ? 4 : 2;
  const size_t lds = (size_t)planes * NF * 16 * ATT_ROW_BYTES + 2 * NF * 16 * sizeof(float);
#ifndef EGV_BWD_STREAM18
#define EGV_BWD_STREAM18 1      // ViT-L/14's 257-key groups on the streaming dQ kernel as well: the register-resident attn_bwd_dq_kernel<0,18,3>
#endif                          // spills 51 VGPRs (the all-bf16x3 parity mode of config 5 dispatched it); 0: A/B builds
  if constexpr (MODE == MODE_SPACE && (NF == 14 || (NF == 18 && EGV_BWD_STREAM18))) {     // streaming dQ kernel (measured on ViT-B/16), then the dK/dV kernel
    if (gr.oh == nullptr) return EGV_ERR_ARG;          // delta = rowsum(dO o O) is taken from the forward's output planes
    {
      const size_t lds1 = (size_t)planes * NF * 16 * ATT_ROW_BYTES + NF * 16 * sizeof(float);
      if (passes == 3) {
        auto k1 = attn_bwd_dq_stream_kernel<NF, 3>;
        auto k2 = attn_bwd_dkv_kernel<MODE, NF, 3>;
        (void)hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
        (void)hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        EGV_LAUNCH(k1, dim3(ngroups), dim3(1024), lds1, s, g, gr);
        EGV_CHECK_LAUNCH();
        EGV_LAUNCH(k2, dim3(ngroups), dim3(256), lds, s, g, gr);
      } else if (g.f16) {
        auto k1 = attn_bwd_dq_stream_kernel<NF, 1, true>;
        auto k2 = attn_bwd_dkv_kernel<MODE, NF, 1, true>;
        (void)hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
        (void)hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        EGV_LAUNCH(k1, dim3(ngroups), dim3(512), lds1, s, g, gr);
        EGV_CHECK_LAUNCH();
        EGV_LAUNCH(k2, dim3(ngroups), dim3(512), lds, s, g, gr);
      } else {
        auto k1 = attn_bwd_dq_stream_kernel<NF, 1>;
        auto k2 = attn_bwd_dkv_kernel<MODE, NF, 1>;
        (void)hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
        (void)hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        EGV_LAUNCH(k1, dim3(ngroups), dim3(512), lds1, s, g, gr);
        EGV_CHECK_LAUNCH();
        EGV_LAUNCH(k2, dim3(ngroups), dim3(512), lds, s, g, gr);
      }
      EGV_CHECK_LAUNCH();
      return EGV_OK;
    }
  } else if constexpr (MODE == MODE_TEXT && (NF == 14 || NF == 18)) {     // captions of 65 .. 288 tokens: two-pass streaming dQ, then dK / dV
    if (passes == 3) {
      auto k1 = attn_bwd_dq_text_stream_kernel<NF, 3>;
      auto k2 = attn_bwd_dkv_kernel<MODE, NF, 3>;
      (void)hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      EGV_LAUNCH(k1, dim3(ngroups), dim3(512), lds, s, g, gr);
      EGV_CHECK_LAUNCH();
      EGV_LAUNCH(k2, dim3(ngroups), dim3(256), lds, s, g, gr);
    } else {
      auto k1 = attn_bwd_dq_text_stream_kernel<NF, 1>;
      auto k2 = attn_bwd_dkv_kernel<MODE, NF, 1>;
      (void)hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      EGV_LAUNCH(k1, dim3(ngroups), dim3(512), lds, s, g, gr);
      EGV_CHECK_LAUNCH();
      EGV_LAUNCH(k2, dim3(ngroups), dim3(512), lds, s, g, gr);
    }
    EGV_CHECK_LAUNCH();
    return EGV_OK;
  } else {        // (else-branch of the if constexpr: the register-resident dQ kernel is not even instantiated for the streamed sizes)
  if (passes == 3) {
    auto k1 = attn_bwd_dq_kernel<MODE, NF, 3>;
    auto k2 = attn_bwd_dkv_kernel<MODE, NF, 3>;
    (void)hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    EGV_LAUNCH(k1, dim3(ngroups), dim3(256), lds, s, g, gr);
    EGV_CHECK_LAUNCH();
    EGV_LAUNCH(k2, dim3(ngroups), dim3(256), lds, s, g, gr);
  } else if (MODE == MODE_SPACE && g.f16) {
    if constexpr (MODE == MODE_SPACE) {
      auto k1 = attn_bwd_dq_kernel<MODE, NF, 1, true>;
      auto k2 = attn_bwd_dkv_kernel<MODE, NF, 1, true>;
      (void)hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      EGV_LAUNCH(k1, dim3(ngroups), dim3(512), lds, s, g, gr);
      EGV_CHECK_LAUNCH();
      EGV_LAUNCH(k2, dim3(ngroups), dim3(512), lds, s, g, gr);
    }
  } else {
    auto k1 = attn_bwd_dq_kernel<MODE, NF, 1>;
    auto k2 = attn_bwd_dkv_kernel<MODE, NF, 1>;
    (void)hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    constexpr int nthr = 512;   // 8 waves per group
    EGV_LAUNCH(k1, dim3(ngroups), dim3(nthr), lds, s, g, gr);
    EGV_CHECK_LAUNCH();
    EGV_LAUNCH(k2, dim3(ngroups), dim3(nthr), lds, s, g, gr);
  }
  EGV_CHECK_LAUNCH();
  return EGV_OK;
  }
}

template <int MODE>
int dispatch_bwd(const AttGeom& g, const AttGrad& gr, int ngroups, int passes, hipStream_t s) {
  const int nq_all = (MODE == MODE_SPACE) ? g.nq + 1 : g.nq;
  const int m = g.nk > nq_all ? g.nk : nq_all;  // one fragment count covers both the key and the query extent
  if (m <= 32) return launch_bwd<MODE, 2>(g, gr, ngroups, passes, s);
  if (m <= 64) return launch_bwd<MODE, 4>(g, gr, ngroups, passes, s);
  if (m <= 224) return launch_bwd<MODE, 14>(g, gr, ngroups, passes, s);
  if (m <= 288) return launch_bwd<MODE, 18>(g, gr, ngroups, passes, s);
  return EGV_ERR_ARG;
}

}  // namespace

int egv_attn_space_bwd_impl(const bf16_t* qkv_hi, const bf16_t* qkv_lo, const bf16_t* out_hi, const bf16_t* out_lo,
                            const bf16_t* do_hi, const bf16_t* do_lo,
                            const float* lse, float* delta, float* dcls, int B, int T, int n, int H, int passes,
                            bf16_t* dqkv_hi, bf16_t* dqkv_lo, int o_fmt, int g_fmt, int f16, hipStream_t s) {
  AttGeom g;
  const long HD = (long)H * ATT_D;
  g.q = g.k = g.v = nullptr;
  g.ph = qkv_hi;
  g.pl = (passes == 3) ? qkv_lo : nullptr;
  g.tok_stride = 3 * HD;
  g.B = B; g.T = T; g.n = n; g.H = H; g.S = 1 + T * n;
  g.nq = n; g.nk = n + 1;
  g.mask = nullptr;
  g.drop = egv_make_drop(0.f, 0);
  g.out_fmt = 0;
  g.f16 = f16;
  if (f16 && passes != 1) return EGV_ERR_ARG;
  AttGrad gr;
  gr.dq = gr.dk = gr.dv = nullptr;
  gr.gh = dqkv_hi;
  gr.gl = (passes == 3) ? dqkv_lo : nullptr;
  gr.tok_stride = 3 * HD;
  gr.d_out = nullptr;
  gr.doh = do_hi;
  gr.dol = (passes == 3) ? do_lo : nullptr;
  gr.do_stride = HD;
  gr.oh = out_hi; gr.ol = out_lo;                   // [B*S, H*64] planes, same row stride as d_out
  gr.lse = lse; gr.delta = delta; gr.dcls = dcls;
  gr.o_fmt = o_fmt; gr.g_fmt = g_fmt;
  return dispatch_bwd<MODE_SPACE>(g, gr, B * T * H, passes, s);
}

extern "C" int egv_text_attn_bwd(const float* q, const float* k, const float* v, int64_t ldqkv, const int64_t* mask,
                                 const float* d_out, const float* lse, int32_t B, int32_t L, int32_t H, int32_t passes,
                                 float dropout_p, uint64_t seed, const uint64_t* seed_dev, float* dq, float* dk, float* dv,
                                 int64_t lddqkv, float* delta_work, void* stream) {
  if (!q || !k || !v || !mask || !d_out || !lse || !dq || !dk || !dv || !delta_work) return EGV_ERR_ARG;
  if (passes != 1 && passes != 3) return EGV_ERR_ARG;
  AttGeom g;
  const long HD = (long)H * ATT_D;
  g.q = q; g.k = k; g.v = v;
  g.ph = g.pl = nullptr;
  g.tok_stride = ldqkv;
  g.B = B; g.T = 1; g.n = L; g.H = H; g.S = L;
  g.nq = L; g.nk = L;
  g.mask = (const long long*)mask;
  if (ldqkv < HD || lddqkv < HD || ldqkv % 4 != 0 || lddqkv % 4 != 0) return EGV_ERR_ARG;
  if (!(dropout_p >= 0.f && dropout_p < 1.f)) return EGV_ERR_ARG;
  g.drop = egv_make_drop(dropout_p, seed, seed_dev);      // the (p, seed, device seed words) of the matching egv_text_attn_fwd call
  g.out_fmt = 0;
  g.f16 = 0;
  AttGrad gr;
  gr.dq = dq; gr.dk = dk; gr.dv = dv;
  gr.gh = gr.gl = nullptr;
  gr.tok_stride = lddqkv;
  gr.d_out = d_out; gr.doh = gr.dol = nullptr; gr.do_stride = HD;
  gr.oh = gr.ol = nullptr;
  gr.lse = lse; gr.delta = delta_work; gr.dcls = nullptr;
  gr.o_fmt = 0; gr.g_fmt = 0;
  return dispatch_bwd<MODE_TEXT>(g, gr, B * H, passes, (hipStream_t)stream);
}
