// Time attention for 4 < T <= 16 frames (model/video_transformer.py:114-124, '(b n) f d': per (clip b, location i, head h) the T
// frame queries attend to the CLS key + the T frame keys of that location) on the matrix cores.
//
// The vector-ALU kernels of attn_small.hip hold q / k / v of a location in registers and reduce every dot product with DPP
// steps; that is HBM-bound at T = 4 (5 keys) and instruction-bound at T = 16: 16 x 17 dot products + as many axpys per head cost
// ~25 000 VALU issues per wave (394 us forward / 920 us backward per block at B = 16, config 4) where the qkv planes stream in
// ~25 us.  Here ONE wave owns one (b, i, h) and every product is an MFMA 16x16x32 on 16-row tiles:
//   * scores are computed TRANSPOSED, S' = K Q^T (rows = keys, columns = queries): the accumulator layout of a 16x16 tile
//     (lane -> column l & 15, rows 4 (l >> 4) + j) is then exactly the B-operand layout of the next product with the keys as the
//     contraction index (attn_common.h's k-index convention), so P goes from the softmax into O^T = V^T P without leaving its lane;
//   * the CLS key (a 17th key) and the clip's CLS query (which rides along in every location group, see attn_small.hip) are
//     one-row tiles: the CLS row sits at row 16 of the 32-row LDS images, rows 17..31 are zero, and elements 4..7 of a
//     row-contraction fragment (rows 16 + 4g + j) carry it through the SAME MFMA as the 16 frame rows -- no separate code path;
//   * operands with the head dimension as contraction index (Q, K, V, dO for the score / dP products) are loaded straight from
//     the bf16 planes in fragment layout (16 B per lane); operands contracted over keys / queries (V in forward; K, Q, dO in
//     backward) are written from those same registers into a per-wave LDS image and fetched with the CDNA4 transpose read;
//   * the backward needs dS in both orientations (dQ contracts over keys, dK / dV over queries): the scores and dP are simply
//     computed twice with the operands swapped (8 more MFMAs) instead of being transposed through LDS.
// Everything is wave-private (no barrier) except the reduction of the CLS token's gradient partials over a workgroup's four
// locations.  Masks: rows / columns >= T and the pad rows of the one-row tiles are forced to probability 0.
#include "attn_common.h"
#include "egovlp_hip.h"

namespace {

constexpr int HD64 = 64;
constexpr int PLANE = 32 * ATT_ROW_BYTES;      // one [32 rows][64] bf16 image

// write this lane's two column-contraction fragments (row p = l & 15, chunks g and g + 4) of a frame tile into an LDS image,
// plus the tile's CLS row (row 16, from the broadcast fragments `c`) and the zero rows 17..31
__device__ __forceinline__ void put_rows(char* plane, int lane, const bf16x8_t (&f)[2], const bf16x8_t (&c)[2]) {
  const int g = lane >> 4, p = lane & 15;
  const bf16x8_t z = __builtin_bit_cast(bf16x8_t, (u32x4_t){0u, 0u, 0u, 0u});
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int chunk = g + 4 * ks;
    *(bf16x8_t*)(plane + p * ATT_ROW_BYTES + ((chunk ^ (p & 7)) << 4)) = f[ks];
    const int r = 16 + p;
    *(bf16x8_t*)(plane + r * ATT_ROW_BYTES + ((chunk ^ (r & 7)) << 4)) = (p == 0) ? c[ks] : z;
  }
}

__device__ __forceinline__ float allg_max(float v) {   // over the four lane groups that share a column
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float allg_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}
__device__ __forceinline__ float row16_sum(float v) {   // over the 16 lanes of a lane group (DPP row)
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
  return v;
}

// row-contraction B operand from a tile pair's accumulator-layout values: elements 0..3 = rows 4g + j of the 16-row tile,
// element 4 = row 16 (the one-row tile; only lane group 0 holds it), 5..7 = 0
__device__ __forceinline__ void pack_b(const float (&a)[4], float one, bf16x8_t& hi, bf16x8_t& lo) {
  const float v[8] = {a[0], a[1], a[2], a[3], one, 0.f, 0.f, 0.f};
  att_split8(v, hi, lo);
}

template <int PASSES>
__device__ __forceinline__ f32x4_t mma2(const bf16x8_t (&ah)[2], const bf16x8_t (&al)[2], const bf16x8_t (&bh)[2], const bf16x8_t (&bl)[2]) {
  f32x4_t c = {0.f, 0.f, 0.f, 0.f};
  c = att_mma<PASSES>(ah[0], al[0], bh[0], bl[0], c);
  return att_mma<PASSES>(ah[1], al[1], bh[1], bl[1], c);
}

__device__ __forceinline__ void zero_unless(bool keep, bf16x8_t (&h)[2], bf16x8_t (&l)[2]) {
  if (!keep) {
    const bf16x8_t z = __builtin_bit_cast(bf16x8_t, (u32x4_t){0u, 0u, 0u, 0u});
    h[0] = h[1] = l[0] = l[1] = z;
  }
}

__device__ __forceinline__ void store4(bf16_t* __restrict__ ph, bf16_t* __restrict__ pl, long off, const f32x4_t& v, float scale) {
  uint32_t h0, h1, l0, l1;
  split_bf16x2(v[0] * scale, v[1] * scale, h0, l0);
  split_bf16x2(v[2] * scale, v[3] * scale, h1, l1);
  *(u32x2_t*)(ph + off) = (u32x2_t){h0, h1};
  if (pl) *(u32x2_t*)(pl + off) = (u32x2_t){l0, l1};
}

// ------------------------------------------------------------------------------------------------------------ forward
template <int PASSES>
__global__ __launch_bounds__(256) void attn_time_mfma_fwd_kernel(const bf16_t* __restrict__ qh, const bf16_t* __restrict__ ql, int B, int T,
                                                                 int n, int H, bf16_t* __restrict__ out_hi, bf16_t* __restrict__ out_lo,
                                                                 float* __restrict__ lse, float* __restrict__ cls_ws) {
  __shared__ __attribute__((aligned(128))) char smem[4][2][PLANE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long gid = (long)blockIdx.x * 4 + wave;
  if (gid >= (long)B * n * H) return;
  const int h = (int)(gid % H);
  const long r = gid / H;
  const int i = (int)(r % n), b = (int)(r / n);
  const int g = lane >> 4, p = lane & 15;
  const long S = 1 + (long)T * n, HD = (long)H * HD64, ts = 3 * HD;
  const int f = p < T ? p : T - 1;                              // rows >= T repeat the last frame (masked below)
  const long cbase = (long)b * S * ts + (long)h * HD64;         // the clip's CLS token, q part
  const long fbase = cbase + (1 + (long)f * n + i) * ts;        // this lane's frame token
  char* vhi = smem[wave][0];
  char* vlo = smem[wave][1];

  bf16x8_t k0h[2], k0l[2], q0h[2], q0l[2], kch[2], kcl[2], qch[2], qcl[2], vh[2], vl[2], vch[2], vcl[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    att_gfrag_planes(qh, ql, fbase, ks, lane, q0h[ks], q0l[ks]);
    att_gfrag_planes(qh, ql, fbase + HD, ks, lane, k0h[ks], k0l[ks]);
    att_gfrag_planes(qh, ql, fbase + 2 * HD, ks, lane, vh[ks], vl[ks]);
  }
  // one-row tiles: the CLS row is row / column 0 (lanes p == 0, which alone fetch it: a quarter of the vector-memory cycles of
  // a full-wave load), the other 15 rows are zero
  zero_unless(false, qch, qcl);
  zero_unless(false, kch, kcl);
  zero_unless(false, vch, vcl);
  if (p == 0) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      att_gfrag_planes(qh, ql, cbase, ks, lane, qch[ks], qcl[ks]);
      att_gfrag_planes(qh, ql, cbase + HD, ks, lane, kch[ks], kcl[ks]);
      att_gfrag_planes(qh, ql, cbase + 2 * HD, ks, lane, vch[ks], vcl[ks]);
    }
  }
  put_rows(vhi, lane, vh, vch);
  if (PASSES == 3) put_rows(vlo, lane, vl, vcl);

  // S' = K Q^T: rows = keys (4g + j), columns = queries (p)
  const f32x4_t s00 = mma2<PASSES>(k0h, k0l, q0h, q0l);      // frame keys x frame queries
  const f32x4_t s10 = mma2<PASSES>(kch, kcl, q0h, q0l);      // CLS key (row 0: group 0, j = 0) x frame queries
  const f32x4_t s01 = mma2<PASSES>(k0h, k0l, qch, qcl);      // frame keys x CLS query (column 0)
  const f32x4_t s11 = mma2<PASSES>(kch, kcl, qch, qcl);      // CLS key x CLS query

  bf16x8_t b0h, b0l, b1h, b1l;
  float m0, l0, m1, l1;
  {   // frame queries: softmax over CLS key + T frame keys
    float a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = (4 * g + j < T) ? s00[j] * 0.125f : -3e38f;
    const float c = (g == 0) ? s10[0] * 0.125f : -3e38f;
    m0 = allg_max(fmaxf(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])), c));
    float e[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) e[j] = (4 * g + j < T) ? __expf(a[j] - m0) : 0.f;
    const float ec = (g == 0) ? __expf(c - m0) : 0.f;
    l0 = allg_sum(e[0] + e[1] + e[2] + e[3] + ec);
    pack_b(e, ec, b0h, b0l);
  }
  {   // the clip's CLS query against this location's keys (+ the CLS key, counted in location 0 only): un-normalised partial
    float a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = (4 * g + j < T) ? s01[j] * 0.125f : -3e38f;
    const bool own = (g == 0) && (i == 0);
    const float c = own ? s11[0] * 0.125f : -3e38f;
    m1 = allg_max(fmaxf(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])), (i == 0) ? c : -1e30f));
    float e[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) e[j] = (4 * g + j < T) ? __expf(a[j] - m1) : 0.f;
    const float ec = own ? __expf(c - m1) : 0.f;
    l1 = allg_sum(e[0] + e[1] + e[2] + e[3] + ec);
    pack_b(e, ec, b1h, b1l);
  }
  const float inv0 = 1.0f / l0;
  const long otok = (long)b * S + 1 + (long)p * n + i;
  float* w = cls_ws + (((long)b * H + h) * n + i) * 68;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    // O^T tile c: rows = channels 16c + 4g + j, columns = queries; contraction over the 17 keys (rows 0..15 + row 16 of the image)
    const bf16x8_t ah = att_frag_rows(vhi, 0, 16 * c, lane);
    const bf16x8_t al = PASSES == 3 ? att_frag_rows(vlo, 0, 16 * c, lane) : ah;
    f32x4_t o0 = {0.f, 0.f, 0.f, 0.f}, o1 = o0;
    o0 = att_mma<PASSES>(ah, al, b0h, b0l, o0);
    o1 = att_mma<PASSES>(ah, al, b1h, b1l, o1);
    if (p < T) store4(out_hi, out_lo, otok * HD + (long)h * HD64 + 16 * c + 4 * g, o0, inv0);
    if (p == 0) *(f32x4_t*)(w + 16 * c + 4 * g) = o1;
  }
  if (g == 0 && p < T && lse) lse[((long)b * H + h) * S + 1 + (long)p * n + i] = m0 + __logf(l0);
  if (lane == 0) {
    w[64] = m1;
    w[65] = l1;
  }
}

// ------------------------------------------------------------------------------------------------------------ backward
// A workgroup = WPB consecutive locations of one (clip, head) (four; two in the three-product mode, whose images are twice as
// big): the CLS token's raw dq / dk / dv partials of its waves are summed in LDS and leave as one round of 192 atomics.
template <int PASSES, int WPB>
__global__ __launch_bounds__(64 * WPB) void attn_time_mfma_bwd_kernel(const bf16_t* __restrict__ qh, const bf16_t* __restrict__ ql,
                                                                 const bf16_t* __restrict__ doh, const bf16_t* __restrict__ dol,
                                                                 const float* __restrict__ lse, const float* __restrict__ delta, int B, int T,
                                                                 int n, int H, bf16_t* __restrict__ gh, bf16_t* __restrict__ gl,
                                                                 float* __restrict__ dcls) {
  constexpr int NPL = PASSES == 3 ? 2 : 1;
  __shared__ __attribute__((aligned(128))) char smem[WPB][3][NPL][PLANE];      // 48 KiB either way
  __shared__ float red[WPB][192];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int chunks = (n + WPB - 1) / WPB;
  const int ic = blockIdx.x % chunks;
  const int bh = blockIdx.x / chunks;
  const int h = bh % H, b = bh / H;
  const int i = ic * WPB + wave;
  const int g = lane >> 4, p = lane & 15;
  if (i < n) {
    const long S = 1 + (long)T * n, HD = (long)H * HD64, ts = 3 * HD;
    const int f = p < T ? p : T - 1;
    const long cbase = (long)b * S * ts + (long)h * HD64;
    const long fbase = cbase + (1 + (long)f * n + i) * ts;
    const long cob = (long)b * S * HD + (long)h * HD64;            // dO of the CLS token
    const long fob = cob + (1 + (long)f * n + i) * HD;
    const float* lb = lse + ((long)b * H + h) * S;
    char* kim = smem[wave][0][0];
    char* qim = smem[wave][1][0];
    char* gim = smem[wave][2][0];

    bf16x8_t k0h[2], k0l[2], q0h[2], q0l[2], v0h[2], v0l[2], g0h[2], g0l[2];
    bf16x8_t kch[2], kcl[2], qch[2], qcl[2], vch[2], vcl[2], gch[2], gcl[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      att_gfrag_planes(qh, ql, fbase, ks, lane, q0h[ks], q0l[ks]);
      att_gfrag_planes(qh, ql, fbase + HD, ks, lane, k0h[ks], k0l[ks]);
      att_gfrag_planes(qh, ql, fbase + 2 * HD, ks, lane, v0h[ks], v0l[ks]);
      att_gfrag_planes(doh, dol, fob, ks, lane, g0h[ks], g0l[ks]);
    }
    zero_unless(false, qch, qcl);
    zero_unless(false, kch, kcl);
    zero_unless(false, vch, vcl);
    zero_unless(false, gch, gcl);
    if (p == 0) {       // the one-row tiles' CLS row: fetched by the lanes that hold it
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        att_gfrag_planes(qh, ql, cbase, ks, lane, qch[ks], qcl[ks]);
        att_gfrag_planes(qh, ql, cbase + HD, ks, lane, kch[ks], kcl[ks]);
        att_gfrag_planes(qh, ql, cbase + 2 * HD, ks, lane, vch[ks], vcl[ks]);
        att_gfrag_planes(doh, dol, cob, ks, lane, gch[ks], gcl[ks]);
      }
    }
    put_rows(kim, lane, k0h, kch);
    put_rows(qim, lane, q0h, qch);
    put_rows(gim, lane, g0h, gch);
    if (PASSES == 3) {
      put_rows(kim + PLANE, lane, k0l, kcl);
      put_rows(qim + PLANE, lane, q0l, qcl);
      put_rows(gim + PLANE, lane, g0l, gcl);
    }
    const float Lc = lb[0], dlc = delta[((long)b * H + h) * S];
    const bool own = (i == 0);                                    // the CLS query sees the CLS key in location 0 only

    // ---- orientation 1: rows = keys (4g + j), columns = queries (p)  ->  dQ (contraction over keys)
    bf16x8_t dq0h, dq0l, dq1h, dq1l;
    {
      const f32x4_t s00 = mma2<PASSES>(k0h, k0l, q0h, q0l), s10 = mma2<PASSES>(kch, kcl, q0h, q0l);
      const f32x4_t s01 = mma2<PASSES>(k0h, k0l, qch, qcl), s11 = mma2<PASSES>(kch, kcl, qch, qcl);
      const f32x4_t d00 = mma2<PASSES>(v0h, v0l, g0h, g0l), d10 = mma2<PASSES>(vch, vcl, g0h, g0l);
      const f32x4_t d01 = mma2<PASSES>(v0h, v0l, gch, gcl), d11 = mma2<PASSES>(vch, vcl, gch, gcl);
      const float Lq = lb[1 + (long)f * n + i];
      float pr[4], pc, ds[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) pr[j] = (4 * g + j < T && p < T) ? __expf(s00[j] * 0.125f - Lq) : 0.f;
      pc = (g == 0 && p < T) ? __expf(s10[0] * 0.125f - Lq) : 0.f;
      const float dl = allg_sum(pr[0] * d00[0] + pr[1] * d00[1] + pr[2] * d00[2] + pr[3] * d00[3] + pc * d10[0]);
#pragma unroll
      for (int j = 0; j < 4; ++j) ds[j] = pr[j] * (d00[j] - dl);
      pack_b(ds, pc * (d10[0] - dl), dq0h, dq0l);
      // the CLS query (column 0)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float pj = (4 * g + j < T && p == 0) ? __expf(s01[j] * 0.125f - Lc) : 0.f;
        ds[j] = pj * (d01[j] - dlc);
      }
      const float pcc = (g == 0 && p == 0 && own) ? __expf(s11[0] * 0.125f - Lc) : 0.f;
      pack_b(ds, pcc * (d11[0] - dlc), dq1h, dq1l);
    }
    // ---- orientation 2: rows = queries (4g + j), columns = keys (p)  ->  dK, dV (contraction over queries)
    bf16x8_t dk0h, dk0l, dk1h, dk1l, pv0h, pv0l, pv1h, pv1l;
    {
      const f32x4_t t00 = mma2<PASSES>(q0h, q0l, k0h, k0l), t10 = mma2<PASSES>(qch, qcl, k0h, k0l);
      const f32x4_t t01 = mma2<PASSES>(q0h, q0l, kch, kcl), t11 = mma2<PASSES>(qch, qcl, kch, kcl);
      const f32x4_t e00 = mma2<PASSES>(g0h, g0l, v0h, v0l), e10 = mma2<PASSES>(gch, gcl, v0h, v0l);
      const f32x4_t e01 = mma2<PASSES>(g0h, g0l, vch, vcl), e11 = mma2<PASSES>(gch, gcl, vch, vcl);
      float p00[4], p01[4], ds0[4], ds1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int qf = 4 * g + j;
        const float Lr = lb[1 + (long)(qf < T ? qf : T - 1) * n + i];
        p00[j] = (qf < T && p < T) ? __expf(t00[j] * 0.125f - Lr) : 0.f;       // frame query x frame key
        p01[j] = (qf < T && p == 0) ? __expf(t01[j] * 0.125f - Lr) : 0.f;      // frame query x CLS key (column 0)
        const float dl = row16_sum(p00[j] * e00[j] + p01[j] * e01[j]);
        ds0[j] = p00[j] * (e00[j] - dl);
        ds1[j] = p01[j] * (e01[j] - dl);
      }
      // the CLS query (row 16 of the query images = element 4): x frame keys, x CLS key
      const float p10 = (g == 0 && p < T) ? __expf(t10[0] * 0.125f - Lc) : 0.f;
      const float p11 = (g == 0 && p == 0 && own) ? __expf(t11[0] * 0.125f - Lc) : 0.f;
      pack_b(ds0, p10 * (e10[0] - dlc), dk0h, dk0l);
      pack_b(ds1, p11 * (e11[0] - dlc), dk1h, dk1l);
      pack_b(p00, p10, pv0h, pv0l);
      pack_b(p01, p11, pv1h, pv1l);
    }
    bf16_t* gb = gh;
    bf16_t* gbl = gl;
    const long gtok = ((long)b * S + 1 + (long)p * n + i) * ts + (long)h * HD64;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bf16x8_t kth = att_frag_rows(kim, 0, 16 * c, lane), ktl = PASSES == 3 ? att_frag_rows(kim + PLANE, 0, 16 * c, lane) : kth;
      const bf16x8_t qth = att_frag_rows(qim, 0, 16 * c, lane), qtl = PASSES == 3 ? att_frag_rows(qim + PLANE, 0, 16 * c, lane) : qth;
      const bf16x8_t gth = att_frag_rows(gim, 0, 16 * c, lane), gtl = PASSES == 3 ? att_frag_rows(gim + PLANE, 0, 16 * c, lane) : gth;
      const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
      const f32x4_t dq = att_mma<PASSES>(kth, ktl, dq0h, dq0l, z);      // rows = channels 16c + 4g + j, columns = frame queries
      const f32x4_t dqc = att_mma<PASSES>(kth, ktl, dq1h, dq1l, z);     // column 0: the CLS query's partial
      const f32x4_t dk = att_mma<PASSES>(qth, qtl, dk0h, dk0l, z);      // columns = frame keys
      const f32x4_t dkc = att_mma<PASSES>(qth, qtl, dk1h, dk1l, z);     // column 0: the CLS key's partial
      const f32x4_t dv = att_mma<PASSES>(gth, gtl, pv0h, pv0l, z);
      const f32x4_t dvc = att_mma<PASSES>(gth, gtl, pv1h, pv1l, z);
      if (p < T) {
        const long o = gtok + 16 * c + 4 * g;
        store4(gb, gbl, o, dq, 0.125f);
        store4(gb, gbl, o + HD, dk, 0.125f);
        store4(gb, gbl, o + 2 * HD, dv, 1.0f);
      }
      if (p == 0) {
        *(f32x4_t*)&red[wave][16 * c + 4 * g] = dqc;
        *(f32x4_t*)&red[wave][64 + 16 * c + 4 * g] = dkc;
        *(f32x4_t*)&red[wave][128 + 16 * c + 4 * g] = dvc;
      }
    }
  } else {
    for (int x = lane; x < 192; x += 64) red[wave][x] = 0.f;
  }
  __syncthreads();
  for (int x = threadIdx.x; x < 192; x += 64 * WPB) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < WPB; ++w) a += red[w][x];
    atomicAdd(dcls + ((long)b * H + h) * 192 + x, a);
  }
}

}  // namespace

int egv_attn_time_mfma_fwd_impl(const bf16_t* qh, const bf16_t* ql, int B, int T, int n, int H, bf16_t* oh, bf16_t* ol, float* lse,
                                float* ws, hipStream_t s) {
  const long waves = (long)B * n * H;
  const dim3 grid((unsigned)((waves + 3) / 4));
  if (ql)
    EGV_LAUNCH((attn_time_mfma_fwd_kernel<3>), grid, dim3(256), 0, s, qh, ql, B, T, n, H, oh, ol, lse, ws);
  else
    EGV_LAUNCH((attn_time_mfma_fwd_kernel<1>), grid, dim3(256), 0, s, qh, ql, B, T, n, H, oh, ol, lse, ws);
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}

int egv_attn_time_mfma_bwd_impl(const bf16_t* qh, const bf16_t* ql, const bf16_t* doh, const bf16_t* dol, const float* lse,
                                const float* delta, int B, int T, int n, int H, bf16_t* gh, bf16_t* gl, float* dcls, hipStream_t s) {
  if (ql && dol)
    EGV_LAUNCH((attn_time_mfma_bwd_kernel<3, 2>), dim3((unsigned)((long)B * H * ((n + 1) / 2))), dim3(128), 0, s, qh, ql, doh, dol, lse,
               delta, B, T, n, H, gh, gl, dcls);
  else
    EGV_LAUNCH((attn_time_mfma_bwd_kernel<1, 4>), dim3((unsigned)((long)B * H * ((n + 3) / 4))), dim3(256), 0, s, qh, nullptr, doh,
               nullptr, lse, delta, B, T, n, H, gh, gl, dcls);
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}
