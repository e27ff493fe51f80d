// Fused attention forward on bf16 MFMA for the two "dense" attention shapes of the hot path:
//   space attention of the SpaceTimeTransformer (model/video_transformer.py:114-133: per (b, frame, head)
//   196 queries x (CLS + 196) keys) and DistilBERT's masked MHA (L x L per (b, head)).
//
// One workgroup (4 waves; 8 in three-pass mode) per group.  The whole K and V of a group (<= 288 keys x 64) live in LDS as
// swizzled split-bf16 planes, so scores never touch HBM (the reference materialises a 237 MB score
// tensor per block) and no online-softmax rescaling is needed: a wave takes a 16-query
// tile, computes S^T = K.Q^T for ALL keys into registers, does the row softmax with two wave shuffles
// (the key axis is register-local + lane groups 16/32 apart), and feeds P^T straight back as the B
// operand of O^T = V^T.P^T -- V^T fragments come from the row-major V image through the CDNA4
// transpose read.  q/k/v are read directly out of the fused qkv buffer with index arithmetic; none
// of the reference's rearrange / repeat / cat copies exist.
#include <cstdlib>
#include <type_traits>

#include "attn_common.h"
#include "egovlp_hip.h"

namespace {

// MODE_SPACE reads q/k/v from the split-bf16 planes of the fused qkv buffer and additionally carries the clip's CLS
// query as one more query row (index n, it rides in the padding of the last 16-query tile): its softmax over THIS
// group's keys is written as an un-normalised partial (o[64], m, l) to cls_ws; egv_attn_cls_combine merges the T
// partials of a (clip, head).  The CLS key is counted for the CLS query in frame-group 0 only.
// q is NOT pre-scaled: scores are multiplied by 64^-0.5 after the MFMA (exact for the power of two).
template <int MODE, int NKF, int PASSES, bool F16 = false>
__global__ __launch_bounds__(512) void attn_fwd_kernel(const AttGeom g, bf16_t* __restrict__ out_hi,
                                                       bf16_t* __restrict__ out_lo, long out_stride,
                                                       float* __restrict__ lse, float* __restrict__ cls_ws) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NKP = NKF * 16;
  constexpr int PLANE = NKP * ATT_ROW_BYTES;
  constexpr bool SP = (MODE == MODE_SPACE);
  char* k_hi = smem;
  char* v_hi = smem + PLANE;
  char* k_lo = (PASSES == 3) ? smem + 2 * PLANE : nullptr;
  char* v_lo = (PASSES == 3) ? smem + 3 * PLANE : nullptr;
  float* kbias = (float*)(smem + ((PASSES == 3) ? 4 : 2) * PLANE);  // [NKP] additive key bias (0 / -1e30)

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

  const int gq = lane >> 4;  // lane group
  const int nq_all = SP ? g.nq + 1 : g.nq;          // + the CLS query row
  const int ntiles = (nq_all + 15) / 16;
  for (int qt = wave; qt < ntiles; qt += (int)(blockDim.x >> 6)) {
    asm volatile("" ::: "memory");  // K/V fragments are loop-invariant: stop LICM from hoisting ~900 VGPRs of them
    const int qi = qt * 16 + (lane & 15);
    const bool is_cls = SP && qi >= g.nq;           // rows past n all alias the CLS row; only qi == n is stored
    const long qtok = is_cls ? grp.tok0 : grp.q_tok(g, min(qi, g.nq - 1));
    bf16x8_t qh[2], ql[2];
    if (SP) {
      att_gfrag_planes(g.ph, g.pl, qtok * g.tok_stride + hoff, 0, lane, qh[0], ql[0]);
      att_gfrag_planes(g.ph, g.pl, qtok * g.tok_stride + hoff, 1, lane, qh[1], ql[1]);
    } else {
      const float* qrow = g.q + qtok * g.tok_stride + hoff;
      att_gfrag(qrow, 0, lane, 1.0f, qh[0], ql[0]);
      att_gfrag(qrow, 1, lane, 1.0f, qh[1], ql[1]);
    }

    f32x4_t s[NKF];
#pragma unroll
    for (int kf = 0; kf < NKF; ++kf) {
      s[kf] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8_t ah = att_frag_cols(k_hi, kf * 16, ks, lane);
        bf16x8_t al = ah;
        if (PASSES == 3) al = att_frag_cols(k_lo, kf * 16, ks, lane);
        s[kf] = att_mma<PASSES, F16>(ah, al, qh[ks], ql[ks], s[kf]);
      }
    }
    // softmax over keys: lane holds keys kf*16 + 4*gq + r of query qi
    float m = -3e38f;
#pragma unroll
    for (int kf = 0; kf < NKF; ++kf) {
      const f32x4_t kb = *(const f32x4_t*)(kbias + kf * 16 + 4 * gq);
      s[kf] = s[kf] * 0.125f + kb;                  // q *= 64^-0.5 (video_transformer.py:106), applied to the scores
      if (SP && kf == 0 && is_cls && grp.f > 0 && gq == 0) s[0][0] = -1e30f;   // CLS key x CLS query: group 0 only
      m = fmaxf(m, fmaxf(fmaxf(s[kf][0], s[kf][1]), fmaxf(s[kf][2], s[kf][3])));
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int kf = 0; kf < NKF; ++kf) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[kf][r] = __expf(s[kf][r] - m);
        l += s[kf][r];
      }
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (MODE == MODE_TEXT && g.drop.thresh != 0u) {
      // dropout on the attention weights (HF eager attention: softmax -> dropout -> . V): the survivors of P are scaled by
      // 1 / (1 - p); the normaliser l is the softmax's and does not change
      const uint64_t rowbase = (((uint64_t)grp.b * g.H + grp.h) * g.S + (uint64_t)min(qi, g.nq - 1)) * g.S;
      const EgvDrop dr = egv_drop_resolve(g.drop);
#pragma unroll
      for (int kf = 0; kf < NKF; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[kf][r] *= egv_drop_scale(dr, rowbase + kf * 16 + 4 * gq + r);
    }

    f32x4_t o[4];
#pragma unroll
    for (int df = 0; df < 4; ++df) o[df] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NKF / 2; ++c) {
      float pv[8] = {s[2 * c][0], s[2 * c][1], s[2 * c][2], s[2 * c][3],
                     s[2 * c + 1][0], s[2 * c + 1][1], s[2 * c + 1][2], s[2 * c + 1][3]};
      bf16x8_t ph, pl;
      att_split8<F16>(pv, ph, pl);
#pragma unroll
      for (int df = 0; df < 4; ++df) {
        const bf16x8_t vh = att_frag_rows(v_hi, 32 * c, df * 16, lane);
        bf16x8_t vl = vh;
        if (PASSES == 3) vl = att_frag_rows(v_lo, 32 * c, df * 16, lane);
        o[df] = att_mma<PASSES, F16>(vh, vl, ph, pl, o[df]);
      }
    }
    if (SP && qi == g.nq) {
      // CLS query x this frame's keys: un-normalised partial for egv_attn_cls_combine
      float* w = cls_ws + (((long)grp.b * g.H + grp.h) * g.T + grp.f) * 68;
#pragma unroll
      for (int df = 0; df < 4; ++df) *(f32x4_t*)(w + df * 16 + 4 * gq) = o[df];
      if (gq == 0) {
        w[64] = m;
        w[65] = l;
      }
    } else if (qi < g.nq) {
      const float inv = 1.0f / l;
      const long tok = qtok;
      bf16_t* oh = out_hi + tok * out_stride + hoff;
      bf16_t* ol = out_lo ? out_lo + tok * out_stride + hoff : nullptr;
#pragma unroll
      for (int df = 0; df < 4; ++df) {
        uint32_t h0, h1, l0, l1;
        att_out2(o[df][0] * inv, o[df][1] * inv, g.out_fmt, h0, l0);
        att_out2(o[df][2] * inv, o[df][3] * inv, g.out_fmt, h1, l1);
        const int d = df * 16 + 4 * gq;
        egv_store<EGV_NT_SPACE_ATTN>(oh + d, (u32x2_t){h0, h1});
        if (ol) egv_store<EGV_NT_SPACE_ATTN>(ol + d, (u32x2_t){l0, l1});
      }
      if (gq == 0 && lse) {
        const long srow = tok - grp.tok0;
        lse[((long)grp.b * g.H + grp.h) * g.S + srow] = m + __logf(l);
      }
    }
  }
}

// ---- streaming variant for the three-pass space mode ---------------------------------------------------------------------
// The kernel above keeps the scores of ALL keys of a query tile in registers (56 fp32 + their split bf16 images: 256 VGPRs
// in three-pass mode, i.e. two waves per SIMD), and a wave runs QK^T (matrix pipe) -> softmax (VALU) -> P.V (matrix pipe)
// strictly one after the other: PMC shows VALU 25 %, MFMA 16 %, LDS 20 % of the launch, added up rather than overlapped
// (profiles/r02_i_pmc_attention.txt).  Here a wave walks the keys in chunks of 32 with the running-max / running-sum
// recurrence (m, l, O rescaled by exp(m_old - m_new) per chunk), so only one chunk of scores is live: <= 128 VGPRs, SIXTEEN
// waves per workgroup (four per SIMD) whose phases interleave, and the 13 query tiles of a ViT-B group run in one round.
// Measured 137 -> 130 us (profiles/r02_z_attention_streaming.txt): staging a group's K / V (one workgroup per CU, 115 KiB) is still
// not overlapped with the previous group's tiles -- that needs the K / V chunks themselves streamed through a small LDS ring.
// PERSISTENT (round 6): one workgroup per CU walks the groups gid, gid + grid, ... and fetches the NEXT group's operands into registers
// while it works on the current group's query tiles -- the staging round trip (every CU pulling 115 KiB at the same moment, six times per
// launch, then every wave its q fragments) was most of a group's 23 us with the matrix pipe and the VALU idle (a group's MFMAs are ~5 us):
//   * the hi planes of K / V (this thread's pieces: 16 VGPRs) at the top of the tile;
//   * the lo planes (16) and the wave's NEXT q fragments (16) behind the last Q.K^T of the tile, when the q registers and the score
//     registers of the earlier chunks are dead -- all 48 live at once with the tile's own 90 registers do not fit the 128 a 16-wave
//     workgroup has (both planes up front spilled);
//   * between two barriers, when every wave is done, the pieces go to LDS and the next group starts with its q in registers.
// A wave owns query tile `wave` of every group (13 tiles on 16 waves for 197 keys; waves without a tile only fetch); further tiles
// (257-key groups: 17 tiles) are run behind it with their q fetched on the spot.  Groups of more than 256 keys (3 pieces per plane and
// thread) fetch everything behind the tile.
template <int B, int E, class F>
__device__ __forceinline__ void static_for_planes(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for_planes<B + 1, E>(f);
  }
}

template <int NKF, bool F16 = false>
__global__ __launch_bounds__(1024) void attn_fwd_stream3_kernel(const AttGeom g, bf16_t* __restrict__ out_hi,
                                                                bf16_t* __restrict__ out_lo, long out_stride,
                                                                float* __restrict__ lse, float* __restrict__ cls_ws, const int ngroups) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NKP = NKF * 16;
  constexpr int PLANE = NKP * ATT_ROW_BYTES;
  char* k_hi = smem;
  char* v_hi = smem + PLANE;
  char* k_lo = smem + 2 * PLANE;
  char* v_lo = smem + 3 * PLANE;
  float* kbias = (float*)(smem + 4 * PLANE);

  // `tidx` / `lane` are made opaque at the top of every group (below): everything derived from them -- piece offsets, LDS fragment
  // addresses -- is recomputed per group instead of being hoisted out of the group loop, where it is live across all of it and spills
  int tidx = threadIdx.x;
  int lane = tidx & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long HD = (long)g.H * ATT_D;

  // this thread's pieces of a group's K / V image: piece t = tid + 1024 i -> row t / 8, 16-B chunk t % 8 (att_stage_planes' layout)
  constexpr int NLD = (NKP * 8 + 1023) / 1024;
  constexpr int EARLY = 0;                           // planes fetched at the top of the tile: none (the pipelined chunk loop has no registers to spare)
  u32x4_t pre[NLD][4];                               // [i][k_hi, v_hi, k_lo, v_lo]
  auto fetch = [&](const AttGroup<MODE_SPACE>& gr, auto P0, auto P1) {   // planes [P0, P1); branch-free (clamped rows)
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int t = min(tidx + 1024 * i, NKP * 8 - 1);
      const int row = min(t >> 3, g.nk - 1), chunk = t & 7;
      const long off = gr.k_tok(g, row) * g.tok_stride + HD + (long)gr.h * ATT_D + chunk * 8;
      static_for_planes<decltype(P0)::value, decltype(P1)::value>([&](auto Pc) {
        constexpr int pp = decltype(Pc)::value;
        pre[i][pp] = *(const u32x4_t*)((pp >= 2 ? g.pl : g.ph) + off + ((pp & 1) ? HD : 0));
      });
    }
  };
  using C0 = std::integral_constant<int, 0>;
  using CE = std::integral_constant<int, EARLY>;
  using C4 = std::integral_constant<int, 4>;
  using CM = std::integral_constant<int, NLD <= 2 ? 2 : EARLY>;      // planes [EARLY, CM) (k_hi, v_hi) are fetched behind the last Q.K^T of the tile,
                                                                     // [CM, 4) (k_lo, v_lo) behind its last P.V: the registers are free by then
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int t = tidx + 1024 * i;
      if (t < NKP * 8) {
        const int row = t >> 3, chunk = t & 7;
        const bool live = row < g.nk;                // rows past the group's keys are zeros (their scores are masked by kbias)
        const u32x4_t zero = {0u, 0u, 0u, 0u};
        const int o = row * ATT_ROW_BYTES + ((chunk ^ (row & 7)) << 4);
        *(u32x4_t*)(k_hi + o) = live ? pre[i][0] : zero;
        *(u32x4_t*)(v_hi + o) = live ? pre[i][1] : zero;
        *(u32x4_t*)(k_lo + o) = live ? pre[i][2] : zero;
        *(u32x4_t*)(v_lo + o) = live ? pre[i][3] : zero;
      }
    }
  };

  const int nq_all = g.nq + 1;                       // + the CLS query row (see attn_fwd_kernel)
  const int ntiles = (nq_all + 15) / 16;
  const bool has_tile = wave < ntiles;               // wave-uniform
  // q fragments of query tile qt of a group: (hi, lo) x two 32-wide k-steps
  auto load_q = [&](const AttGroup<MODE_SPACE>& gr, int qt, bf16x8_t (&h)[2], bf16x8_t (&l)[2]) {
    const int qi = qt * 16 + (lane & 15);
    const long qtok = qi >= g.nq ? gr.tok0 : gr.q_tok(g, min(qi, g.nq - 1));
    const long off = qtok * g.tok_stride + (long)gr.h * ATT_D + (lane >> 4) * 8;
    // (both planes exist in this kernel: plain loads -- att_gfrag_planes' "lo = hi" fallback makes the compiler wait for the hi load)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      h[ks] = *(const bf16x8_t*)(g.ph + off + 32 * ks);
      l[ks] = *(const bf16x8_t*)(g.pl + off + 32 * ks);
    }
  };

  int gid = blockIdx.x;
  AttGroup<MODE_SPACE> grp(g, gid);
  bf16x8_t qh[2], ql[2];
  if constexpr (NKF <= 16) load_q(grp, has_tile ? wave : 0, qh, ql);
  fetch(grp, C0{}, C4{});
  commit();
  for (int j = threadIdx.x; j < NKP; j += blockDim.x) kbias[j] = (j < g.nk) ? 0.f : -1e30f;
  __syncthreads();

  // One query tile: NKF / 2 chunks of 32 keys with the running-max / running-sum recurrence; `mid` runs behind the LAST chunk's Q.K^T
  // (the q fragments are dead from there on).
  auto run_tile = [&](const AttGroup<MODE_SPACE>& gr, const int qt, const bf16x8_t (&tqh)[2], const bf16x8_t (&tql)[2], auto&& mid, auto&& post) {
    const long hoff = (long)gr.h * ATT_D;
    const int gq = lane >> 4;
    const int qi = qt * 16 + (lane & 15);
    const bool is_cls = qi >= g.nq;
    const long qtok = is_cls ? gr.tok0 : gr.q_tok(g, min(qi, g.nq - 1));
    float m = -3e38f, l = 0.f;                       // m: uniform over the four lane groups of a query; l: this lane group's part
    f32x4_t o[4];
#pragma unroll
    for (int df = 0; df < 4; ++df) o[df] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // The kernel is VALU-issue-bound (~130 VALU instructions per chunk against 24 MFMAs; rocprofv3: VALU active 49 %, MFMA busy 20 %
    // of the launch), so the chunk body is written for instruction count:
    //   * the lane part of every LDS fragment address is resolved ONCE per tile (two column-fragment bases, four row-fragment bases);
    //     a chunk adds its own 4 KiB (32 rows) and everything else -- second fragment, k-step, lo plane -- is an immediate offset
    //     (att_frag_cols / att_frag_rows re-derive row, chunk and swizzle per call: 24 adds + 12 shifts / ors per chunk);
    //   * the running maximum lives in the exp2 domain: scores are scaled by 64^-0.5 log2(e) in the one FMA that adds the key bias and
    //     every exponential is a bare v_exp_f32 (exp() costs a multiply in front of it); lse and the CLS partial convert back;
    //   * F16: the probabilities (in [0, 1]: nothing to overflow) are split with v_cvt_pkrtz_f16_f32, two values per instruction
    //     (hi truncated instead of rounded: the lo plane carries the difference either way).
    constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    constexpr int LO = 2 * PLANE;                    // k_lo - k_hi == v_lo - v_hi
    const int r15 = lane & 15;
    const unsigned kc[2] = {(unsigned)(r15 * ATT_ROW_BYTES + ((((lane >> 4)) ^ (r15 & 7)) << 4)),
                            (unsigned)(r15 * ATT_ROW_BYTES + ((((lane >> 4) + 4) ^ (r15 & 7)) << 4))};
    unsigned vo[4];
#pragma unroll
    for (int df = 0; df < 4; ++df) vo[df] = (unsigned)att_off(4 * gq + (r15 >> 2), df * 16 + ((r15 & 3) << 2));
    auto scores = [&](const int c, f32x4_t (&s)[2]) {
      const char* kp0 = k_hi + (c * 4096 + kc[0]);
      const char* kp1 = k_hi + (c * 4096 + kc[1]);
      const float* kbp = kbias + (c * 32 + 4 * gq);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        s[h] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        s[h] = att_mma<3, F16>(*(const bf16x8_t*)(kp0 + h * 2048), *(const bf16x8_t*)(kp0 + h * 2048 + LO), tqh[0], tql[0], s[h]);
        s[h] = att_mma<3, F16>(*(const bf16x8_t*)(kp1 + h * 2048), *(const bf16x8_t*)(kp1 + h * 2048 + LO), tqh[1], tql[1], s[h]);
        const f32x4_t kb = *(const f32x4_t*)(kbp + h * 16);
        s[h] = s[h] * (0.125f * LOG2E) + kb;         // q *= 64^-0.5 (video_transformer.py:106), applied to the scores; exp2 domain
      }
      if (c == 0 && is_cls && gr.f > 0 && gq == 0) s[0][0] = -1e30f;   // CLS key x CLS query: group 0 only
    };
    auto accumulate = [&](const int c, const f32x4_t (&s)[2]) {
      float cm = fmaxf(fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3])),
                       fmaxf(fmaxf(s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3])));
      cm = fmaxf(cm, __shfl_xor(cm, 16, 64));
      cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
      const float mn = fmaxf(m, cm);
      const float alpha = __builtin_amdgcn_exp2f(m - mn);      // first chunk: 2^(-3e38 - mn) = 0 and l, o are 0 anyway
      m = mn;
      float pv[8];
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pv[r] = __builtin_amdgcn_exp2f(s[0][r] - mn);
        pv[4 + r] = __builtin_amdgcn_exp2f(s[1][r] - mn);
        ps += pv[r] + pv[4 + r];
      }
      l = l * alpha + ps;
      bf16x8_t ph, pl;
      if constexpr (F16) {
        typedef __attribute__((ext_vector_type(2))) __fp16 h2_t;
        u32x4_t hh, ll;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const h2_t hp = __builtin_amdgcn_cvt_pkrtz(pv[2 * e], pv[2 * e + 1]);
          hh[e] = __builtin_bit_cast(uint32_t, hp);
          ll[e] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(pv[2 * e] - (float)hp[0], pv[2 * e + 1] - (float)hp[1]));
        }
        ph = __builtin_bit_cast(bf16x8_t, hh);
        pl = __builtin_bit_cast(bf16x8_t, ll);
      } else {
        att_split8<false>(pv, ph, pl);
      }
      const char* vbase = v_hi + c * 4096;
#pragma unroll
      for (int df = 0; df < 4; ++df) {
#if defined(EGV_NO_TR_READ)
        const bf16x8_t vh = att_frag_rows(v_hi, 32 * c, df * 16, lane);
        const bf16x8_t vl = att_frag_rows(v_lo, 32 * c, df * 16, lane);
#else
        const bf16x8_t vh = att_frag_rows_at(vbase + vo[df]);
        const bf16x8_t vl = att_frag_rows_at(vbase + vo[df] + LO);
#endif
        o[df] = att_mma<3, F16>(vh, vl, ph, pl, o[df] * alpha);
      }
    };
    // Software-pipelined over the chunks: the Q.K^T MFMAs of chunk c + 1 are issued BEFORE the softmax of chunk c, so that the long
    // VALU chain (max, exp, sum, fp16 split: ~130 instructions per chunk against 24 MFMAs -- the kernel is VALU-issue-bound, rocprofv3:
    // VALU active 49 %, MFMA busy 20 % of the launch) has matrix work of its own wave to hide behind instead of leaving that to the
    // other waves of the SIMD alone.  Two score sets, the chunk loop unrolled by two (the chunk counts here -- 7, 9 -- are odd).
    // (The 257-key instance -- two tile bodies, 48 registers of K / V pieces at the end -- keeps the plain loop: pipelined it spills.)
    constexpr int NC = NKF / 2;
    f32x4_t sa[2], sb[2];
    if constexpr (NKF <= 16) {
      static_assert(NKF > 16 || NC % 2 == 1, "the two-chunk software pipeline below ends on an odd chunk count");
      scores(0, sa);
#pragma unroll 1
      for (int c = 0; c + 2 < NC; c += 2) {
        scores(c + 1, sb);
        __builtin_amdgcn_sched_barrier(0);
        accumulate(c, sa);
        __builtin_amdgcn_sched_barrier(0);
        scores(c + 2, sa);
        __builtin_amdgcn_sched_barrier(0);
        accumulate(c + 1, sb);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll 1
      for (int c = 0; c < NC - 1; ++c) {
        scores(c, sb);
        accumulate(c, sb);
      }
      scores(NC - 1, sa);
    }
    asm volatile("" ::: "memory");
    mid();
    asm volatile("" ::: "memory");
    accumulate(NC - 1, sa);
    asm volatile("" ::: "memory");
    post();
    asm volatile("" ::: "memory");
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (qi == g.nq) {
      // CLS query x this frame's keys: un-normalised partial for egv_attn_cls_combine
      float* w = cls_ws + (((long)gr.b * g.H + gr.h) * g.T + gr.f) * 68;
#pragma unroll
      for (int df = 0; df < 4; ++df) *(f32x4_t*)(w + df * 16 + 4 * gq) = o[df];
      if (gq == 0) {
        w[64] = m * LN2;                             // egv_attn_cls_combine merges the partials in natural units
        w[65] = l;
      }
    } else if (qi < g.nq) {
      const float inv = 1.0f / l;
      bf16_t* oh = out_hi + qtok * out_stride + hoff;
      bf16_t* ol = out_lo ? out_lo + qtok * out_stride + hoff : nullptr;      // ATT_OUT_F16: no second plane
#pragma unroll
      for (int df = 0; df < 4; ++df) {
        uint32_t h0, h1, l0, l1;
        att_out2(o[df][0] * inv, o[df][1] * inv, g.out_fmt, h0, l0);
        att_out2(o[df][2] * inv, o[df][3] * inv, g.out_fmt, h1, l1);
        const int d = df * 16 + 4 * gq;
        egv_store<EGV_NT_SPACE_ATTN>(oh + d, (u32x2_t){h0, h1});
        if (ol) egv_store<EGV_NT_SPACE_ATTN>(ol + d, (u32x2_t){l0, l1});
      }
      if (gq == 0 && lse) lse[((long)gr.b * g.H + gr.h) * g.S + (qtok - gr.tok0)] = (m + __log2f(l)) * LN2;
    }
  };

  for (;;) {
    asm volatile("" : "+v"(tidx), "+v"(lane));
    const int gnext = gid + (int)gridDim.x;
    const bool more = gnext < ngroups;
    const AttGroup<MODE_SPACE> nxt(g, more ? gnext : gid);     // the last group re-reads its own operands and drops them
    bf16x8_t nqh[2], nql[2];
    if (has_tile) {
      if constexpr (NKF > 16) {                      // more than 16 query tiles (only groups of more than 256 keys): the further ones FIRST,
        for (int qt = wave + (int)(blockDim.x >> 6); qt < ntiles; qt += (int)(blockDim.x >> 6)) {     // nothing fetched ahead is live under them
          bf16x8_t xh[2], xl[2];
          load_q(grp, qt, xh, xl);
          run_tile(grp, qt, xh, xl, []() {}, []() {});
        }
      }
      if constexpr (NKF > 16) load_q(grp, wave, qh, ql);     // (these groups fetch nothing ahead but K / V behind the last tile: registers)
      fetch(nxt, C0{}, CE{});                        // behind this wave's q loads, which are a whole group old
      run_tile(grp, wave, qh, ql, [&]() {
        if constexpr (NKF <= 16) load_q(nxt, wave, nqh, nql);
        fetch(nxt, CE{}, CM{});
      }, [&]() { fetch(nxt, CM{}, C4{}); });
    } else {
      if constexpr (NKF <= 16) load_q(nxt, 0, nqh, nql);     // (unused: keeps the hand-over below branch-free)
      fetch(nxt, C0{}, C4{});
    }
    if (!more) break;
    __syncthreads();                                 // every wave is done with this group's K / V
    commit();
    __syncthreads();
    gid = gnext;
    grp = nxt;
    if constexpr (NKF <= 16) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        qh[ks] = nqh[ks];
        ql[ks] = nql[ks];
      }
    }
  }
}

template <int MODE, int NKF>
int launch_fwd(const AttGeom& g, int ngroups, int passes, bf16_t* oh, bf16_t* ol, long ostride, float* lse,
               float* cls_ws, hipStream_t s) {
  const int planes = passes == 3 ? 4 : 2;
  const size_t lds = (size_t)planes * NKF * 16 * ATT_ROW_BYTES + NKF * 16 * sizeof(float);
#ifndef EGV_STREAM18
#define EGV_STREAM18 1      // the streaming kernel for ViT-L/14's 257-key groups as well (17 query tiles on 16 waves: two rounds, and still
#endif                      // +0.25 % on config 5 against attn_fwd_kernel<0,18,3>: 234.6 vs 234.0 pairs/s, profiles/r05e_ab_config5_stream18.txt; 0: A/B builds
  if constexpr (MODE == MODE_SPACE && (NKF == 14 || (NKF == 18 && EGV_STREAM18))) {   // measured on ViT-B/16 (13 query tiles on 16 waves): 137 -> 130 us
    if (passes == 3 && (ol != nullptr || g.out_fmt == ATT_OUT_F16)) {
      // persistent: one 16-wave workgroup per CU (115 / 148 KiB of LDS); EGV_ATTN_PERSIST=0 restores one workgroup per group (A/B)
      static const int persist = getenv("EGV_ATTN_PERSIST") ? atoi(getenv("EGV_ATTN_PERSIST")) : 1;
      const int grid = persist ? (ngroups < 256 ? ngroups : 256) : ngroups;
      if (g.f16) {
        auto kern = attn_fwd_stream3_kernel<NKF, true>;
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        EGV_LAUNCH(kern, dim3(grid), dim3(1024), lds, s, g, oh, ol, ostride, lse, cls_ws, ngroups);
      } else {
        auto kern = attn_fwd_stream3_kernel<NKF>;
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        EGV_LAUNCH(kern, dim3(grid), dim3(1024), lds, s, g, oh, ol, ostride, lse, cls_ws, ngroups);
      }
      EGV_CHECK_LAUNCH();
      return EGV_OK;
    }
    // a three-pass space forward of these sizes ALWAYS runs the streaming kernel (a missing second plane is an argument error): the
    // register-resident attn_fwd_kernel<0, 14, 3> spills and is not instantiated for them
    if (passes == 3) return EGV_ERR_ARG;
    {
      auto kern = attn_fwd_kernel<MODE, NKF, 1>;
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      EGV_LAUNCH(kern, dim3(ngroups), dim3(512), lds, s, g, oh, nullptr, ostride, lse, cls_ws);
      EGV_CHECK_LAUNCH();
      return EGV_OK;
    }
  } else {
  if (MODE == MODE_SPACE && g.f16) {
    if (passes != 3) return EGV_ERR_ARG;              // the fp16 forward is the three-product one
    if constexpr (MODE == MODE_SPACE) {
      auto kern = attn_fwd_kernel<MODE, NKF, 3, true>;
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      EGV_LAUNCH(kern, dim3(ngroups), dim3(512), lds, s, g, oh, ol, ostride, lse, cls_ws);
    }
  } else if (passes == 3) {
    auto kern = attn_fwd_kernel<MODE, NKF, 3>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // three-pass mode keeps four K/V planes in LDS (115 KiB: one workgroup per CU), so it runs 8 waves per workgroup to have
    // two waves per SIMD; the single-pass kernel fits two 4-wave workgroups per CU
    EGV_LAUNCH(kern, dim3(ngroups), dim3(512), lds, s, g, oh, ol, ostride, lse, cls_ws);
  } else {
    auto kern = attn_fwd_kernel<MODE, NKF, 1>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    constexpr int nthr = 512;
    EGV_LAUNCH(kern, dim3(ngroups), dim3(nthr), lds, s, g, oh, nullptr, ostride, lse, cls_ws);
  }
  EGV_CHECK_LAUNCH();
  return EGV_OK;
  }
}

}  // namespace

// dispatch on the padded key count (NKF 16-key fragments, even): 2 (<=32 keys: DistilBERT L<=32 and the
// tiny test configs), 4, 14 (ViT-B/16: 197 keys), 18 (ViT-L/14: 257 keys)
template <int MODE>
static int dispatch_fwd(const AttGeom& g, int ngroups, int passes, bf16_t* oh, bf16_t* ol, long ostride, float* lse,
                        float* cls_ws, hipStream_t s) {
  if (g.nk <= 32) return launch_fwd<MODE, 2>(g, ngroups, passes, oh, ol, ostride, lse, cls_ws, s);
  if (g.nk <= 64) return launch_fwd<MODE, 4>(g, ngroups, passes, oh, ol, ostride, lse, cls_ws, s);
  if (g.nk <= 224) return launch_fwd<MODE, 14>(g, ngroups, passes, oh, ol, ostride, lse, cls_ws, s);
  if (g.nk <= 288) return launch_fwd<MODE, 18>(g, ngroups, passes, oh, ol, ostride, lse, cls_ws, s);
  return EGV_ERR_ARG;
}

int egv_attn_space_fwd_impl(const bf16_t* qkv_hi, const bf16_t* qkv_lo, int B, int T, int n, int H, int passes,
                            bf16_t* out_hi, bf16_t* out_lo, float* lse, float* cls_ws, int out_fmt, int f16, hipStream_t s) {
  AttGeom g;
  g.out_fmt = out_fmt;
  g.f16 = f16;
  const long HD = (long)H * ATT_D;
  g.q = g.k = g.v = nullptr;
  g.ph = qkv_hi;
  g.pl = (passes == 3) ? qkv_lo : nullptr;
  g.tok_stride = 3 * HD;
  g.B = B; g.T = T; g.n = n; g.H = H; g.S = 1 + T * n;
  g.nq = n; g.nk = n + 1;
  g.mask = nullptr;
  g.drop = egv_make_drop(0.f, 0);
  return dispatch_fwd<MODE_SPACE>(g, B * T * H, passes, out_hi, out_lo, HD, lse, cls_ws, s);
}

extern "C" int egv_text_attn_fwd(const float* q, const float* k, const float* v, int64_t ldqkv, const int64_t* mask,
                                 int32_t B, int32_t L, int32_t H, int32_t passes, float dropout_p, uint64_t seed,
                                 const uint64_t* seed_dev, egv_bf16* out_hi, egv_bf16* out_lo, float* lse, void* stream) {
  if (!q || !k || !v || !out_hi || B <= 0 || L <= 0 || H <= 0) return EGV_ERR_ARG;
  if (passes != 1 && passes != 3) return EGV_ERR_ARG;
  if (passes == 3 && !out_lo) return EGV_ERR_ARG;
  AttGeom g;
  g.out_fmt = 0;
  g.f16 = 0;
  const long HD = (long)H * ATT_D;
  g.q = q; g.k = k; g.v = v;
  g.ph = g.pl = nullptr;
  g.tok_stride = ldqkv;
  g.B = B; g.T = 1; g.n = L; g.H = H; g.S = L;
  g.nq = L; g.nk = L;
  g.mask = (const long long*)mask;
  if (!mask || ldqkv < HD || ldqkv % 4 != 0) return EGV_ERR_ARG;
  if (!(dropout_p >= 0.f && dropout_p < 1.f)) return EGV_ERR_ARG;
  g.drop = egv_make_drop(dropout_p, seed, seed_dev);
  return dispatch_fwd<MODE_TEXT>(g, B * H, passes, out_hi, out_lo, HD, lse, nullptr, (hipStream_t)stream);
}
