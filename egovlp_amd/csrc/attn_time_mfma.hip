// Time attention (model/video_transformer.py:114-124, '(b n) f d': per (clip b, location i, head h) the T <= 16 frame queries attend to
// the CLS key + the T frame keys of that location) on the matrix cores.
//
// Rounds 1 - 3 ran this on the vector ALU (q / k / v of a location in registers, every dot product reduced with DPP steps): HBM-bound
// at T = 4 (5 keys), instruction-bound at T = 16 -- 16 x 17 dot products + as many axpys per head cost ~25 000 VALU issues per wave
// (396 us forward / 926 us backward per block at B = 16, config 4, where the 616 MB of planes need ~80 us).  Here ONE wave owns one
// (b, h) and LOCS = 16 / TP consecutive locations (TP = T rounded up to 4, 8 or 16: one location at T = 16, four at T = 4 -- the rows
// of a tile are (location, frame) pairs and probabilities across locations are masked to 0) and every product is an MFMA 16x16x32
// on 16-row tiles:
//   * scores are computed TRANSPOSED, S' = K Q^T (rows = keys, columns = queries): the accumulator layout of a 16x16 tile
//     (lane -> column l & 15, rows 4 (l >> 4) + j) is then exactly the B-operand layout of the next product with the keys as the
//     contraction index (attn_common.h's k-index convention), so P goes from the softmax into O^T = V^T P without leaving its lane;
//   * the CLS key (one more key for every query) and the clip's CLS query (which rides along in every unit: one copy per location
//     of the unit, see attn_small.hip) are extra rows of the LDS images: rows 16..19 hold the CLS row / its copies / zeros, row 20
//     is zero, and elements 4..7 of a row-contraction fragment (rows 16 + j in lane group 0, the zero row in the others) carry
//     them through the SAME MFMA as the 16 frame rows -- no separate code path;
//   * every operand tile (16 rows x 128 B per plane) is fetched with TWO coalesced 16-B-per-lane loads (lane -> row l >> 3,
//     chunk l & 7) into a per-wave swizzled LDS image (attn_common.h) and read back as fragments: ds_read_b128 for the head
//     dimension as contraction index (scores, dP), the CDNA4 transpose read for keys / queries as contraction index (P V, dS K,
//     dS^T Q, P^T dO);
//   * the backward needs dS in both orientations (dQ contracts over keys, dK / dV over queries): the scores and dP are simply
//     computed twice with the operands swapped (8 more MFMAs) instead of being transposed through LDS.
// Everything is wave-private (no barrier) except the reduction of the CLS token's gradient partials over a workgroup's units.
// Masks: rows / columns whose frame >= T or location >= n, pairs from different locations and the pad rows of the CLS tiles are
// forced to probability 0.  What bounds it: HBM -- at T = 16, B = 16 the three-pass forward reads 462 MB and writes 154 MB in 141 us
// (4.4 TB/s = 0.55 of peak; the loads alone, diagnostic build: 82 us = 5.6 TB/s), the single-pass backward moves 539 MB in 138 us;
// fetching the tiles cooperatively per workgroup (adjacent tokens per instruction) changed nothing (profiles/r04q_*).
#include "attn_common.h"
#include "egovlp_hip.h"

namespace {

constexpr int HD64 = 64;
constexpr int IMG17 = 18 * ATT_ROW_BYTES;      // image read by column fragments only: 16 frame rows + the CLS row (+ 1 pad)
constexpr int IMG20 = 21 * ATT_ROW_BYTES;      // image read by the transpose read as well: rows 16..19 = CLS copies / zeros, row 20 = zero

// ---- one operand tile (16 frame rows of one plane, 128 B each) as two coalesced loads: lane -> (row 8 it + (l >> 3), chunk l & 7)
struct Tile { u32x4_t r[2]; };
// tile row -> (location i0 + row / TP, frame row % TP); rows whose frame >= T or location >= n repeat a valid token (masked by
// the callers)
template <int TP>
__device__ __forceinline__ long row_token(int row, int T, int n, int i0) {
  const int f = row % TP, il = i0 + row / TP;
  return 1 + (long)(f < T ? f : T - 1) * n + (il < n ? il : n - 1);
}
template <int TP>
__device__ __forceinline__ bool row_valid(int row, int T, int n, int i0) { return (row % TP) < T && i0 + row / TP < n; }
template <int TP>
__device__ __forceinline__ Tile load_tile(const bf16_t* __restrict__ plane, long part_base, int T, int n, int i0, long ts, int lane) {
  Tile t;
#pragma unroll
  for (int it = 0; it < 2; ++it)
    t.r[it] = *(const u32x4_t*)(plane + part_base + row_token<TP>(8 * it + (lane >> 3), T, n, i0) * ts + (lane & 7) * 8);
  return t;
}
__device__ __forceinline__ void put_tile(char* img, const Tile& t, int lane) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int row = 8 * it + (lane >> 3), chunk = lane & 7;
    *(u32x4_t*)(img + row * ATT_ROW_BYTES + ((chunk ^ (row & 7)) << 4)) = t.r[it];
  }
}
// rows 16 .. 16 + copies - 1 = the CLS row (from the lanes [8 slot, 8 slot + 8) of `v`); `fill` (images the transpose read walks):
// the rest up to row 19 and row 20 zero.  Rows 16..20: row & 7 = 0..4.
__device__ __forceinline__ void put_cls_rows(char* img, const u32x4_t& v, int slot, int lane, int copies, bool fill) {
  u32x4_t mine;
#pragma unroll
  for (int e = 0; e < 4; ++e) mine[e] = __shfl(v[e], 8 * slot + (lane & 7), 64);      // the slot's chunk (lane & 7) in every lane
  const int row = 16 + (lane >> 3), chunk = lane & 7;
  if (lane < 40 && (fill || (lane >> 3) < copies))
    *(u32x4_t*)(img + row * ATT_ROW_BYTES + ((chunk ^ (row & 7)) << 4)) = (lane >> 3) < copies ? mine : (u32x4_t){0u, 0u, 0u, 0u};
}
// column-contraction fragment of the CLS tile: row 16 in the lanes p < copies, zero elsewhere
__device__ __forceinline__ bf16x8_t frag_cls(const char* img, int ks, int lane, int copies) {
  const bf16x8_t z = __builtin_bit_cast(bf16x8_t, (u32x4_t){0u, 0u, 0u, 0u});
  const bf16x8_t v = *(const bf16x8_t*)(img + 16 * ATT_ROW_BYTES + (((lane >> 4) + 4 * ks) << 4));
  return (lane & 15) < copies ? v : z;
}
// row-contraction fragment of a 21-row image: elements 0..3 = rows 4g + j, elements 4..7 = rows 16 + j for lane group 0 (the CLS
// copies / zero rows), the zero row 20 for the groups g >= 1
__device__ __forceinline__ bf16x8_t frag_rows20(const char* img, int col0, int lane) {
  const int g = lane >> 4, p = lane & 15;
  const int col = col0 + ((p & 3) << 2);
  const int ra = 4 * g + (p >> 2);
  const int rb = g == 0 ? 16 + (p >> 2) : 20;
  const s16x4_t x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(img + att_off(ra, col)));
  const s16x4_t y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(img + att_off(rb, col)));
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  const s16x8_t z = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
  return __builtin_bit_cast(bf16x8_t, z);
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
// elements 4..7 = rows 16..19 (the CLS tile: its rows live in lane group 0 only -- the callers pass zeros elsewhere)
template <bool F16 = false>
__device__ __forceinline__ void pack_b(const float (&a)[4], const float (&x)[4], bf16x8_t& hi, bf16x8_t& lo) {
  const float v[8] = {a[0], a[1], a[2], a[3], x[0], x[1], x[2], x[3]};
  att_split8<F16>(v, hi, lo);
}
template <bool F16 = false>
__device__ __forceinline__ void pack_b(const float (&a)[4], float one, bf16x8_t& hi, bf16x8_t& lo) {
  const float x[4] = {one, 0.f, 0.f, 0.f};
  pack_b<F16>(a, x, hi, lo);
}

template <int PASSES, bool F16 = false>
__device__ __forceinline__ f32x4_t mma2(const bf16x8_t (&ah)[2], const bf16x8_t (&al)[2], const bf16x8_t (&bh)[2], const bf16x8_t (&bl)[2]) {
  f32x4_t c = {0.f, 0.f, 0.f, 0.f};
  c = att_mma<PASSES, F16>(ah[0], al[0], bh[0], bl[0], c);
  return att_mma<PASSES, F16>(ah[1], al[1], bh[1], bl[1], c);
}

__device__ __forceinline__ void store4(bf16_t* __restrict__ ph, bf16_t* __restrict__ pl, long off, const f32x4_t& v, float scale, int fmt = 0) {
  uint32_t h0, h1, l0, l1;
  att_out2(v[0] * scale, v[1] * scale, fmt, h0, l0);
  att_out2(v[2] * scale, v[3] * scale, fmt, h1, l1);
#if defined(EGV_TMF_DBG) && EGV_TMF_DBG >= 1      // diagnostics build: everything but the plane stores
  asm volatile("" ::"v"(h0), "v"(h1), "v"(l0), "v"(l1));
#elif defined(EGV_TMF_NT)
  __builtin_nontemporal_store((u32x2_t){h0, h1}, (u32x2_t*)(ph + off));
  if (pl) __builtin_nontemporal_store((u32x2_t){l0, l1}, (u32x2_t*)(pl + off));
#else
  *(u32x2_t*)(ph + off) = (u32x2_t){h0, h1};
  if (pl) *(u32x2_t*)(pl + off) = (u32x2_t){l0, l1};
#endif
}

// ---- output rows through LDS (round 5).  A lane of an output tile owns 4 channels of ONE token row (8 bytes of each plane): stored
// directly, a wave instruction wrote 16 x 32-byte segments of 16 different 128-byte rows, four instructions per row and plane.  The
// 16 x 64 tile of a plane is staged in the wave's own LDS instead (the image layout: 128-B rows, 16-B chunk XOR (row & 7); LDS
// operations of a wave execute in order, so no barrier) and leaves as whole rows: lane -> (row 8 it + (l >> 3), chunk l & 7), one
// 16-byte store per lane, 8 lanes per 128-byte row.  -DEGV_TMF_OLD_STORES: the direct 8-byte stores (A/B builds).
__device__ __forceinline__ void stage4(char* sh, char* sl, int p, int col, const f32x4_t& v, float scale, int fmt = 0) {
  uint32_t h0, h1, l0, l1;
  att_out2(v[0] * scale, v[1] * scale, fmt, h0, l0);
  att_out2(v[2] * scale, v[3] * scale, fmt, h1, l1);
  const int off = att_off(p, col);
  *(u32x2_t*)(sh + off) = (u32x2_t){h0, h1};
  if (sl) *(u32x2_t*)(sl + off) = (u32x2_t){l0, l1};
}
// rows of the staged tile -> plane rows: element offset of row r = (tok0 + row_token(r)) * ts + col0
template <int TP, int SITE>
__device__ __forceinline__ void flush_rows(const char* sh, const char* sl, bf16_t* __restrict__ ph, bf16_t* __restrict__ pl, long tok0, long ts,
                                           long col0, int T, int n, int i0, int lane) {
  // the tile was written by OTHER lanes of this wave (stage4) with plain C++ stores: LDS operations of a wave execute in order, but
  // nothing would stop the compiler from moving this lane's reads above another lane's writes -- or the next stage4 (the rows are
  // reused three times in the backward) above these reads -- if its alias analysis proves per-thread disjointness.  A wave barrier
  // is a scheduling fence only (no instruction): it pins the order the hardware already keeps.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int row = 8 * it + (lane >> 3), chunk = lane & 7;
    if (!row_valid<TP>(row, T, n, i0)) continue;
    const long o = (tok0 + row_token<TP>(row, T, n, i0)) * ts + col0 + chunk * 8;
    const int lo = row * ATT_ROW_BYTES + ((chunk ^ (row & 7)) << 4);
    egv_store<SITE>(ph + o, *(const u32x4_t*)(sh + lo));
    if (pl) egv_store<SITE>(pl + o, *(const u32x4_t*)(sl + lo));
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();           // ... and the staging rows may be overwritten only behind these reads
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------------------------------ forward
template <int PASSES, int TP, bool F16 = false>
__global__ __launch_bounds__(256) void attn_time_mfma_fwd_kernel(const bf16_t* __restrict__ qh, const bf16_t* __restrict__ ql, int B, int T,
                                                                 int n, int H, bf16_t* __restrict__ out_hi, bf16_t* __restrict__ out_lo,
                                                                 float* __restrict__ lse, float* __restrict__ cls_ws, int out_fmt) {
  constexpr int NPL = PASSES == 3 ? 2 : 1;
  constexpr int LOCS = 16 / TP;                      // locations per wave
  constexpr int WAVE_LDS = NPL * 2 * IMG17;          // Q and K images; the V image (NPL * IMG20, smaller or equal... see static_assert) reuses the space
  static_assert(NPL * IMG20 + NPL * 2048 <= WAVE_LDS + 1024, "V images + the output staging rows");
  __shared__ __attribute__((aligned(128))) char smem[4][WAVE_LDS + 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int units = (n + LOCS - 1) / LOCS;
  const long gid = (long)blockIdx.x * 4 + wave;
  if (gid >= (long)B * units * H) return;
  const int h = (int)(gid % H);
  const long r = gid / H;
  const int i0 = (int)(r % units) * LOCS, b = (int)(r / units);
  const int g = lane >> 4, p = lane & 15;
  const long S = 1 + (long)T * n, HD = (long)H * HD64, ts = 3 * HD;
  const long cbase = (long)b * S * ts + (long)h * HD64;         // the clip's CLS token, q part
  char* base = smem[wave];
  const bf16_t* planes[2] = {qh, ql};

  Tile qt[NPL], kt[NPL], vt[NPL];
  u32x4_t ct[NPL];                                   // the CLS token's q | k | v rows: lanes 0-7 | 8-15 | 16-23
#pragma unroll
  for (int pl = 0; pl < NPL; ++pl) {
    qt[pl] = load_tile<TP>(planes[pl], cbase, T, n, i0, ts, lane);
    kt[pl] = load_tile<TP>(planes[pl], cbase + HD, T, n, i0, ts, lane);
    vt[pl] = load_tile<TP>(planes[pl], cbase + 2 * HD, T, n, i0, ts, lane);
    ct[pl] = (u32x4_t){0u, 0u, 0u, 0u};
    if (lane < 24) ct[pl] = *(const u32x4_t*)(planes[pl] + cbase + (lane >> 3) * HD + (lane & 7) * 8);
  }
#if defined(EGV_TMF_DBG) && EGV_TMF_DBG == 3      // diagnostics build: the global loads alone (no LDS traffic)
#pragma unroll
  for (int pl = 0; pl < NPL; ++pl)
    asm volatile("" ::"v"(qt[pl].r[0]), "v"(qt[pl].r[1]), "v"(kt[pl].r[0]), "v"(kt[pl].r[1]), "v"(vt[pl].r[0]), "v"(vt[pl].r[1]), "v"(ct[pl]));
  return;
#endif
  bf16x8_t k0[NPL][2], q0[NPL][2], kc[NPL][2], qc[NPL][2];
#pragma unroll
  for (int pl = 0; pl < NPL; ++pl) {
    char* qim = base + (2 * pl) * IMG17;
    char* kim = base + (2 * pl + 1) * IMG17;
    put_tile(qim, qt[pl], lane);
    put_tile(kim, kt[pl], lane);
    put_cls_rows(qim, ct[pl], 0, lane, 1, false);
    put_cls_rows(kim, ct[pl], 1, lane, 1, false);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      q0[pl][ks] = att_frag_cols(qim, 0, ks, lane);
      k0[pl][ks] = att_frag_cols(kim, 0, ks, lane);
      qc[pl][ks] = frag_cls(qim, ks, lane, LOCS);      // column c < LOCS: the CLS query as seen from location i0 + c
      kc[pl][ks] = frag_cls(kim, ks, lane, 1);         // row 0: the CLS key
    }
  }
#if defined(EGV_TMF_DBG) && EGV_TMF_DBG == 2      // diagnostics build: the loads and the LDS round trip alone
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
    asm volatile("" ::"v"(k0[0][ks]), "v"(q0[0][ks]), "v"(kc[NPL - 1][ks]), "v"(qc[NPL - 1][ks]), "v"(vt[0].r[ks]), "v"(vt[NPL - 1].r[ks]));
  return;
#endif
  constexpr int LO = NPL - 1;
  // S' = K Q^T: rows = keys (4g + j), columns = queries (p)
  const f32x4_t s00 = mma2<PASSES, F16>(k0[0], k0[LO], q0[0], q0[LO]);      // frame keys x frame queries
  const f32x4_t s10 = mma2<PASSES, F16>(kc[0], kc[LO], q0[0], q0[LO]);      // CLS key (row 0: group 0, j = 0) x frame queries
  const f32x4_t s01 = mma2<PASSES, F16>(k0[0], k0[LO], qc[0], qc[LO]);      // frame keys x CLS query (columns c < LOCS)
  const f32x4_t s11 = mma2<PASSES, F16>(kc[0], kc[LO], qc[0], qc[LO]);      // CLS key x CLS query
  // the V image takes the place of the Q / K images (LDS operations of a wave execute in order)
#pragma unroll
  for (int pl = 0; pl < NPL; ++pl) {
    char* vim = base + pl * IMG20;
    put_tile(vim, vt[pl], lane);
    put_cls_rows(vim, ct[pl], 2, lane, 1, true);
  }

  const bool qv = row_valid<TP>(p, T, n, i0);          // this lane's query column
  bool kv[4], same[4], mine[4];                        // key row 4g + j: valid / same location as the query / in location i0 + p
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int kr = 4 * g + j;
    kv[j] = row_valid<TP>(kr, T, n, i0);
    same[j] = kv[j] && kr / TP == p / TP;
    mine[j] = kv[j] && kr / TP == p && p < LOCS;
  }
  bf16x8_t b0h, b0l, b1h, b1l;
  float m0, l0, m1, l1;
  {   // frame queries: softmax over the CLS key + the T frame keys of the query's location
    float a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = same[j] ? s00[j] * 0.125f : -3e38f;
    const float c = (g == 0) ? s10[0] * 0.125f : -3e38f;
    m0 = allg_max(fmaxf(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])), c));
    float e[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) e[j] = same[j] ? __expf(a[j] - m0) : 0.f;
    const float ec = (g == 0) ? __expf(c - m0) : 0.f;
    l0 = allg_sum(e[0] + e[1] + e[2] + e[3] + ec);
    pack_b<F16>(e, ec, b0h, b0l);
  }
  {   // the clip's CLS query against the keys of location i0 + p (+ the CLS key, counted in location 0 only): un-normalised partial
    float a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = mine[j] ? s01[j] * 0.125f : -3e38f;
    const bool own = (g == 0) && (i0 + p == 0);
    const float c = own ? s11[0] * 0.125f : -3e38f;
    m1 = allg_max(fmaxf(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])), c));
    float e[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) e[j] = mine[j] ? __expf(a[j] - m1) : 0.f;
    const float ec = own ? __expf(c - m1) : 0.f;
    l1 = allg_sum(e[0] + e[1] + e[2] + e[3] + ec);
    pack_b<F16>(e, ec, b1h, b1l);
  }
  const float inv0 = 1.0f / l0;
  const long otok = (long)b * S + row_token<TP>(p, T, n, i0);
  const bool cls_col = p < LOCS && i0 + p < n;
  float* w = cls_ws + (((long)b * H + h) * n + (cls_col ? i0 + p : 0)) * 68;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    // O^T tile c: rows = channels 16c + 4g + j, columns = queries; contraction over the keys (rows 0..15 + the CLS row 16 of the image)
    const bf16x8_t ah = frag_rows20(base, 16 * c, lane);
    const bf16x8_t al = PASSES == 3 ? frag_rows20(base + IMG20, 16 * c, lane) : ah;
    f32x4_t o0 = {0.f, 0.f, 0.f, 0.f}, o1 = o0;
    o0 = att_mma<PASSES, F16>(ah, al, b0h, b0l, o0);
    o1 = att_mma<PASSES, F16>(ah, al, b1h, b1l, o1);
#ifdef EGV_TMF_OLD_STORES
    if (qv) store4(out_hi, out_lo, otok * HD + (long)h * HD64 + 16 * c + 4 * g, o0, inv0, out_fmt);
#else
    stage4(base + NPL * IMG20, PASSES == 3 ? base + NPL * IMG20 + 2048 : nullptr, p, 16 * c + 4 * g, o0, inv0, out_fmt);
#endif
    if (cls_col) *(f32x4_t*)(w + 16 * c + 4 * g) = o1;
  }
#ifndef EGV_TMF_OLD_STORES
  flush_rows<TP, EGV_NT_ATTN_OUT>(base + NPL * IMG20, base + NPL * IMG20 + 2048, out_hi, PASSES == 3 ? out_lo : nullptr, (long)b * S, HD, (long)h * HD64, T, n, i0, lane);
#endif
  if (g == 0 && qv && lse) lse[((long)b * H + h) * S + row_token<TP>(p, T, n, i0)] = m0 + __logf(l0);
  if (g == 0 && cls_col) {
    w[64] = m1;
    w[65] = l1;
  }
}

// ------------------------------------------------------------------------------------------------------------ backward
// A workgroup = WPB consecutive units of one (clip, head) (four; two in the three-product mode, whose images are twice as big):
// the CLS token's raw dq / dk / dv partials of its waves are summed in LDS and leave as one round of 192 atomics.
template <int PASSES, int WPB, int TP, bool F16 = false>
__global__ __launch_bounds__(64 * WPB) void attn_time_mfma_bwd_kernel(const bf16_t* __restrict__ qh, const bf16_t* __restrict__ ql,
                                                                 const bf16_t* __restrict__ doh, const bf16_t* __restrict__ dol,
                                                                 const float* __restrict__ lse, const float* __restrict__ delta, int B, int T,
                                                                 int n, int H, bf16_t* __restrict__ gh, bf16_t* __restrict__ gl,
                                                                 float* __restrict__ dcls, int gfmt) {
  constexpr int NPL = PASSES == 3 ? 2 : 1;
  constexpr int LOCS = 16 / TP;
  constexpr int WAVE_LDS = NPL * (3 * IMG20 + IMG17);       // K, Q, dO (transpose-read too) and V images
  __shared__ __attribute__((aligned(128))) char smem[WPB][WAVE_LDS];
  __shared__ float red[WPB][192];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int units = (n + LOCS - 1) / LOCS;
  const int chunks = (units + WPB - 1) / WPB;
  const int ic = blockIdx.x % chunks;
  const int bh = blockIdx.x / chunks;
  const int h = bh % H, b = bh / H;
  const int unit = ic * WPB + wave;
  const int i0 = unit * LOCS;
  const int g = lane >> 4, p = lane & 15;
  for (int x = lane; x < 192; x += 64) red[wave][x] = 0.f;
  if (unit < units) {
    const long S = 1 + (long)T * n, HD = (long)H * HD64, ts = 3 * HD;
    const long cbase = (long)b * S * ts + (long)h * HD64;
    const long cob = (long)b * S * HD + (long)h * HD64;            // dO of the CLS token
    const float* lb = lse + ((long)b * H + h) * S;
    char* base = smem[wave];
    const bf16_t* planes[2] = {qh, ql};
    const bf16_t* gplanes[2] = {doh, dol};

    Tile qt[NPL], kt[NPL], vt[NPL], gt[NPL];
    u32x4_t ct[NPL];                                   // the CLS token's q | k | v | dO rows: lanes 0-7 | 8-15 | 16-23 | 24-31
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
      qt[pl] = load_tile<TP>(planes[pl], cbase, T, n, i0, ts, lane);
      kt[pl] = load_tile<TP>(planes[pl], cbase + HD, T, n, i0, ts, lane);
      vt[pl] = load_tile<TP>(planes[pl], cbase + 2 * HD, T, n, i0, ts, lane);
      gt[pl] = load_tile<TP>(gplanes[pl], cob, T, n, i0, HD, lane);
      ct[pl] = (u32x4_t){0u, 0u, 0u, 0u};
      if (lane < 32) {
        const bf16_t* src = lane < 24 ? planes[pl] + cbase + (lane >> 3) * HD : gplanes[pl] + cob;
        ct[pl] = *(const u32x4_t*)(src + (lane & 7) * 8);
      }
    }
    char *kim[NPL], *qim[NPL], *gim[NPL];
    bf16x8_t k0[NPL][2], q0[NPL][2], v0[NPL][2], g0[NPL][2], kc[NPL][2], qc[NPL][2], vc[NPL][2], gc[NPL][2];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
      char* wb = base + pl * (3 * IMG20 + IMG17);
      kim[pl] = wb; qim[pl] = wb + IMG20; gim[pl] = wb + 2 * IMG20;
      char* vim = wb + 3 * IMG20;
      // the CLS query / its dO: LOCS copies (one per location of the unit: rows 16.. of the images the transpose read walks, columns
      // c < LOCS of the column fragments); the CLS key / value: one row shared by all queries
      put_tile(qim[pl], qt[pl], lane); put_cls_rows(qim[pl], ct[pl], 0, lane, LOCS, true);
      put_tile(kim[pl], kt[pl], lane); put_cls_rows(kim[pl], ct[pl], 1, lane, 1, true);
      put_tile(vim, vt[pl], lane);     put_cls_rows(vim, ct[pl], 2, lane, 1, false);
      put_tile(gim[pl], gt[pl], lane); put_cls_rows(gim[pl], ct[pl], 3, lane, LOCS, true);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        q0[pl][ks] = att_frag_cols(qim[pl], 0, ks, lane); qc[pl][ks] = frag_cls(qim[pl], ks, lane, LOCS);
        k0[pl][ks] = att_frag_cols(kim[pl], 0, ks, lane); kc[pl][ks] = frag_cls(kim[pl], ks, lane, 1);
        v0[pl][ks] = att_frag_cols(vim, 0, ks, lane);     vc[pl][ks] = frag_cls(vim, ks, lane, 1);
        g0[pl][ks] = att_frag_cols(gim[pl], 0, ks, lane); gc[pl][ks] = frag_cls(gim[pl], ks, lane, LOCS);
      }
    }
    constexpr int LO = NPL - 1;
    const float Lc = lb[0], dlc = delta[((long)b * H + h) * S];
    const bool pv_ = row_valid<TP>(p, T, n, i0);                   // this lane's column as a frame row
    const long ptok = row_token<TP>(p, T, n, i0);
    bool rv[4], same[4], mine[4];                                 // row 4g + j: valid / same location as column p / in location i0 + p
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int rr = 4 * g + j;
      rv[j] = row_valid<TP>(rr, T, n, i0);
      same[j] = rv[j] && pv_ && rr / TP == p / TP;
      mine[j] = rv[j] && rr / TP == p && p < LOCS;
    }

    // ---- orientation 1: rows = keys (4g + j), columns = queries (p)  ->  dQ (contraction over keys)
    bf16x8_t dq0h, dq0l, dq1h, dq1l;
    {
      const f32x4_t s00 = mma2<PASSES, F16>(k0[0], k0[LO], q0[0], q0[LO]), s10 = mma2<PASSES, F16>(kc[0], kc[LO], q0[0], q0[LO]);
      const f32x4_t s01 = mma2<PASSES, F16>(k0[0], k0[LO], qc[0], qc[LO]), s11 = mma2<PASSES, F16>(kc[0], kc[LO], qc[0], qc[LO]);
      const f32x4_t d00 = mma2<PASSES, F16>(v0[0], v0[LO], g0[0], g0[LO]), d10 = mma2<PASSES, F16>(vc[0], vc[LO], g0[0], g0[LO]);
      const f32x4_t d01 = mma2<PASSES, F16>(v0[0], v0[LO], gc[0], gc[LO]), d11 = mma2<PASSES, F16>(vc[0], vc[LO], gc[0], gc[LO]);
      const float Lq = lb[ptok];
      float pr[4], pc, ds[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) pr[j] = same[j] ? __expf(s00[j] * 0.125f - Lq) : 0.f;
      pc = (g == 0 && pv_) ? __expf(s10[0] * 0.125f - Lq) : 0.f;
      const float dl = allg_sum(pr[0] * d00[0] + pr[1] * d00[1] + pr[2] * d00[2] + pr[3] * d00[3] + pc * d10[0]);
#pragma unroll
      for (int j = 0; j < 4; ++j) ds[j] = pr[j] * (d00[j] - dl);
      pack_b<F16>(ds, pc * (d10[0] - dl), dq0h, dq0l);
      // the CLS query as seen from location i0 + p (columns p < LOCS)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float pj = mine[j] ? __expf(s01[j] * 0.125f - Lc) : 0.f;
        ds[j] = pj * (d01[j] - dlc);
      }
      const float pcc = (g == 0 && i0 + p == 0) ? __expf(s11[0] * 0.125f - Lc) : 0.f;
      pack_b<F16>(ds, pcc * (d11[0] - dlc), dq1h, dq1l);
    }
    // ---- orientation 2: rows = queries (4g + j), columns = keys (p)  ->  dK, dV (contraction over queries)
    bf16x8_t dk0h, dk0l, dk1h, dk1l, pv0h, pv0l, pv1h, pv1l;
    {
      const f32x4_t t00 = mma2<PASSES, F16>(q0[0], q0[LO], k0[0], k0[LO]), t10 = mma2<PASSES, F16>(qc[0], qc[LO], k0[0], k0[LO]);
      const f32x4_t t01 = mma2<PASSES, F16>(q0[0], q0[LO], kc[0], kc[LO]), t11 = mma2<PASSES, F16>(qc[0], qc[LO], kc[0], kc[LO]);
      const f32x4_t e00 = mma2<PASSES, F16>(g0[0], g0[LO], v0[0], v0[LO]), e10 = mma2<PASSES, F16>(gc[0], gc[LO], v0[0], v0[LO]);
      const f32x4_t e01 = mma2<PASSES, F16>(g0[0], g0[LO], vc[0], vc[LO]), e11 = mma2<PASSES, F16>(gc[0], gc[LO], vc[0], vc[LO]);
      float p00[4], p01[4], ds0[4], ds1[4], p10[4], d10[4], p11[4], d11[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float Lr = lb[row_token<TP>(4 * g + j, T, n, i0)];
        p00[j] = same[j] ? __expf(t00[j] * 0.125f - Lr) : 0.f;              // frame query x frame key of its location
        p01[j] = (rv[j] && p == 0) ? __expf(t01[j] * 0.125f - Lr) : 0.f;    // frame query x CLS key (column 0)
        const float dl = row16_sum(p00[j] * e00[j] + p01[j] * e01[j]);
        ds0[j] = p00[j] * (e00[j] - dl);
        ds1[j] = p01[j] * (e01[j] - dl);
        // the CLS query copy j (rows 16 + j of the query images = element 4 + j; lane group 0 holds the tile): x the frame keys of
        // location i0 + j, x the CLS key (location 0 only)
        const bool cj = g == 0 && j < LOCS;
        p10[j] = (cj && pv_ && p / TP == j) ? __expf(t10[j] * 0.125f - Lc) : 0.f;
        p11[j] = (cj && p == 0 && i0 + j == 0) ? __expf(t11[j] * 0.125f - Lc) : 0.f;
        d10[j] = p10[j] * (e10[j] - dlc);
        d11[j] = p11[j] * (e11[j] - dlc);
      }
      pack_b<F16>(ds0, d10, dk0h, dk0l);
      pack_b<F16>(ds1, d11, dk1h, dk1l);
      pack_b<F16>(p00, p10, pv0h, pv0l);
      pack_b<F16>(p01, p11, pv1h, pv1l);
    }
    const long gtok = ((long)b * S + ptok) * ts + (long)h * HD64;
    const bool cls_col = p < LOCS;                                  // columns of the CLS query's partial (masked columns hold zeros)
#ifndef EGV_TMF_OLD_STORES
    // dQ, dK, dV one after the other through the (no longer needed) V images as staging rows; whole 128-byte rows out (flush_rows)
    char* const sth = base + 3 * IMG20;
    char* const stl = PASSES == 3 ? base + (3 * IMG20 + IMG17) + 3 * IMG20 : nullptr;
    static_assert(IMG17 >= 2048, "");
    const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bf16x8_t kth = frag_rows20(kim[0], 16 * c, lane), ktl = PASSES == 3 ? frag_rows20(kim[LO], 16 * c, lane) : kth;
      const f32x4_t dq = att_mma<PASSES, F16>(kth, ktl, dq0h, dq0l, z);      // rows = channels 16c + 4g + j, columns = frame queries
      const f32x4_t dqc = att_mma<PASSES, F16>(kth, ktl, dq1h, dq1l, z);     // columns c < LOCS: the CLS query's partials, one per location
      stage4(sth, stl, p, 16 * c + 4 * g, dq, 0.125f, gfmt);
      if (cls_col) {
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(&red[wave][16 * c + 4 * g + j], dqc[j]);
      }
    }
    flush_rows<TP, EGV_NT_TIME_BWD>(sth, stl, gh, gl, (long)b * S, ts, (long)h * HD64, T, n, i0, lane);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bf16x8_t qth = frag_rows20(qim[0], 16 * c, lane), qtl = PASSES == 3 ? frag_rows20(qim[LO], 16 * c, lane) : qth;
      const f32x4_t dk = att_mma<PASSES, F16>(qth, qtl, dk0h, dk0l, z);      // columns = frame keys
      const f32x4_t dkc = att_mma<PASSES, F16>(qth, qtl, dk1h, dk1l, z);     // column 0: the CLS key's partial
      stage4(sth, stl, p, 16 * c + 4 * g, dk, 0.125f, gfmt);
      if (p == 0) *(f32x4_t*)&red[wave][64 + 16 * c + 4 * g] = dkc;
    }
    flush_rows<TP, EGV_NT_TIME_BWD>(sth, stl, gh, gl, (long)b * S, ts, (long)h * HD64 + HD, T, n, i0, lane);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bf16x8_t gth = frag_rows20(gim[0], 16 * c, lane), gtl = PASSES == 3 ? frag_rows20(gim[LO], 16 * c, lane) : gth;
      const f32x4_t dv = att_mma<PASSES, F16>(gth, gtl, pv0h, pv0l, z);
      const f32x4_t dvc = att_mma<PASSES, F16>(gth, gtl, pv1h, pv1l, z);
      stage4(sth, stl, p, 16 * c + 4 * g, dv, 1.0f, gfmt);
      if (p == 0) *(f32x4_t*)&red[wave][128 + 16 * c + 4 * g] = dvc;
    }
    flush_rows<TP, EGV_NT_TIME_BWD>(sth, stl, gh, gl, (long)b * S, ts, (long)h * HD64 + 2 * HD, T, n, i0, lane);
#else
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bf16x8_t kth = frag_rows20(kim[0], 16 * c, lane), ktl = PASSES == 3 ? frag_rows20(kim[LO], 16 * c, lane) : kth;
      const bf16x8_t qth = frag_rows20(qim[0], 16 * c, lane), qtl = PASSES == 3 ? frag_rows20(qim[LO], 16 * c, lane) : qth;
      const bf16x8_t gth = frag_rows20(gim[0], 16 * c, lane), gtl = PASSES == 3 ? frag_rows20(gim[LO], 16 * c, lane) : gth;
      const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
      const f32x4_t dq = att_mma<PASSES, F16>(kth, ktl, dq0h, dq0l, z);      // rows = channels 16c + 4g + j, columns = frame queries
      const f32x4_t dqc = att_mma<PASSES, F16>(kth, ktl, dq1h, dq1l, z);     // columns c < LOCS: the CLS query's partials, one per location
      const f32x4_t dk = att_mma<PASSES, F16>(qth, qtl, dk0h, dk0l, z);      // columns = frame keys
      const f32x4_t dkc = att_mma<PASSES, F16>(qth, qtl, dk1h, dk1l, z);     // column 0: the CLS key's partial
      const f32x4_t dv = att_mma<PASSES, F16>(gth, gtl, pv0h, pv0l, z);
      const f32x4_t dvc = att_mma<PASSES, F16>(gth, gtl, pv1h, pv1l, z);
      if (pv_) {
        const long o = gtok + 16 * c + 4 * g;
        store4(gh, gl, o, dq, 0.125f, gfmt);
        store4(gh, gl, o + HD, dk, 0.125f, gfmt);
        store4(gh, gl, o + 2 * HD, dv, 1.0f, gfmt);
      }
      if (cls_col) {
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(&red[wave][16 * c + 4 * g + j], dqc[j]);
      }
      if (p == 0) {
        *(f32x4_t*)&red[wave][64 + 16 * c + 4 * g] = dkc;
        *(f32x4_t*)&red[wave][128 + 16 * c + 4 * g] = dvc;
      }
    }
#endif
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

template <int TP>
static int launch_fwd(const bf16_t* qh, const bf16_t* ql, int B, int T, int n, int H, bf16_t* oh, bf16_t* ol, float* lse, float* ws,
                      int out_fmt, int f16, hipStream_t s) {
  constexpr int LOCS = 16 / TP;
  const long waves = (long)B * ((n + LOCS - 1) / LOCS) * H;
  const dim3 grid((unsigned)((waves + 3) / 4));
  if (f16) {
    if (!ql) return EGV_ERR_ARG;        // the fp16 forward is the three-product one
    EGV_LAUNCH((attn_time_mfma_fwd_kernel<3, TP, true>), grid, dim3(256), 0, s, qh, ql, B, T, n, H, oh, ol, lse, ws, out_fmt);
  } else if (ql)
    EGV_LAUNCH((attn_time_mfma_fwd_kernel<3, TP>), grid, dim3(256), 0, s, qh, ql, B, T, n, H, oh, ol, lse, ws, out_fmt);
  else
    EGV_LAUNCH((attn_time_mfma_fwd_kernel<1, TP>), grid, dim3(256), 0, s, qh, ql, B, T, n, H, oh, ol, lse, ws, 0);
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}

template <int TP>
static int launch_bwd(const bf16_t* qh, const bf16_t* ql, const bf16_t* doh, const bf16_t* dol, const float* lse, const float* delta, int B,
                      int T, int n, int H, bf16_t* gh, bf16_t* gl, float* dcls, int gfmt, int f16, hipStream_t s) {
  constexpr int LOCS = 16 / TP;
  const int units = (n + LOCS - 1) / LOCS;
  if (f16)
    EGV_LAUNCH((attn_time_mfma_bwd_kernel<1, 4, TP, true>), dim3((unsigned)((long)B * H * ((units + 3) / 4))), dim3(256), 0, s, qh, nullptr, doh,
               nullptr, lse, delta, B, T, n, H, gh, gl, dcls, gfmt);
  else if (ql && dol)
    EGV_LAUNCH((attn_time_mfma_bwd_kernel<3, 2, TP>), dim3((unsigned)((long)B * H * ((units + 1) / 2))), dim3(128), 0, s, qh, ql, doh, dol,
               lse, delta, B, T, n, H, gh, gl, dcls, gfmt);
  else
    EGV_LAUNCH((attn_time_mfma_bwd_kernel<1, 4, TP>), dim3((unsigned)((long)B * H * ((units + 3) / 4))), dim3(256), 0, s, qh, nullptr, doh,
               nullptr, lse, delta, B, T, n, H, gh, gl, dcls, gfmt);
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}

int egv_attn_time_mfma_fwd_impl(const bf16_t* qh, const bf16_t* ql, int B, int T, int n, int H, bf16_t* oh, bf16_t* ol, float* lse,
                                float* ws, int out_fmt, int f16, hipStream_t s) {
  if (T <= 4) return launch_fwd<4>(qh, ql, B, T, n, H, oh, ol, lse, ws, out_fmt, f16, s);
  if (T <= 8) return launch_fwd<8>(qh, ql, B, T, n, H, oh, ol, lse, ws, out_fmt, f16, s);
  return launch_fwd<16>(qh, ql, B, T, n, H, oh, ol, lse, ws, out_fmt, f16, s);
}

int egv_attn_time_mfma_bwd_impl(const bf16_t* qh, const bf16_t* ql, const bf16_t* doh, const bf16_t* dol, const float* lse,
                                const float* delta, int B, int T, int n, int H, bf16_t* gh, bf16_t* gl, float* dcls, int gfmt, int f16,
                                hipStream_t s) {
  if (T <= 4) return launch_bwd<4>(qh, ql, doh, dol, lse, delta, B, T, n, H, gh, gl, dcls, gfmt, f16, s);
  if (T <= 8) return launch_bwd<8>(qh, ql, doh, dol, lse, delta, B, T, n, H, gh, gl, dcls, gfmt, f16, s);
  return launch_bwd<16>(qh, ql, doh, dol, lse, delta, B, T, n, H, gh, gl, dcls, gfmt, f16, s);
}
