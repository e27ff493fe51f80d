// The "thin" attention shapes of divided space-time attention, on the vector ALU in fp32:
//   * time attention (model/video_transformer.py:114-124, '(b n) f d'): per (b, location, head) only
//     T queries x (CLS + T) keys (4 x 5 at T=4, 16 x 17 at T=16) -- far below an MFMA tile; HBM-bound (it streams the
//     qkv planes once).  One wave owns one (b, location) and HPW heads: with HPW = 4 a lane holds 4 channels of one head,
//     so every load is 8 B per lane / 512 B per wave and one 4-step shuffle tree reduces four heads' dot products at once
//     (HPW = 1, lane = channel, is kept for head counts that are not a multiple of 4).
//   * the CLS query row (:109-112, 1 query x all S keys per (b, head)) no longer has kernels of its own: each group of
//     the space (MFMA) / time kernel carries the clip's CLS query as one more query against ITS keys -- forward as an
//     un-normalised softmax partial merged by egv_attn_cls_combine, backward with the global log-sum-exp and delta, so
//     every dK / dV row leaves the kernel complete (patch queries + CLS query) and is written ONCE, as bf16 planes.
//     Only the CLS token's own gradients (shared by all groups of a clip) go through fp32 atomics + a finish kernel.
// All operands and results are split-bf16 planes of the fused [B, S, 3, H, 64] buffer (lo plane optional).
#include <cstdlib>

#include "common.h"
#include "egovlp_hip.h"

namespace {

constexpr int D = 64;

// sum over the 16 lanes of a DPP row, result in every lane: quad_perm xor 1, xor 2, then row rotations by 4 and 8.
// Full-rate VALU (DPP) instead of four ds_bpermute round trips through the LDS crossbar -- the time-attention backward
// does ~250 of these reductions per location and was shuffle-bound (296 us per call with __shfl_xor).
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
  return v;
}

template <int LPH>
__device__ __forceinline__ float redh(float v) {   // sum over the LPH lanes that share a head
  v = row16_sum(v);
  if (LPH >= 32) v += __shfl_xor(v, 16, 64);
  if (LPH == 64) v += __shfl_xor(v, 32, 64);
  return v;
}

template <int CPL>
__device__ __forceinline__ void ldp(const bf16_t* __restrict__ ph, const bf16_t* __restrict__ pl, long off, float (&x)[CPL]) {
  if (CPL == 4) {
    const u32x2_t a = *(const u32x2_t*)(ph + off);
    x[0] = __uint_as_float(a[0] << 16);
    x[1] = __uint_as_float(a[0] & 0xffff0000u);
    x[2] = __uint_as_float(a[1] << 16);
    x[3] = __uint_as_float(a[1] & 0xffff0000u);
    if (pl) {
      const u32x2_t b = *(const u32x2_t*)(pl + off);
      x[0] += __uint_as_float(b[0] << 16);
      x[1] += __uint_as_float(b[0] & 0xffff0000u);
      x[2] += __uint_as_float(b[1] << 16);
      x[3] += __uint_as_float(b[1] & 0xffff0000u);
    }
  } else if (CPL == 2) {
    const uint32_t a = *(const uint32_t*)(ph + off);
    x[0] = __uint_as_float(a << 16);
    x[1] = __uint_as_float(a & 0xffff0000u);
    if (pl) {
      const uint32_t b = *(const uint32_t*)(pl + off);
      x[0] += __uint_as_float(b << 16);
      x[1] += __uint_as_float(b & 0xffff0000u);
    }
  } else {
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      x[c] = bf16_to_f32(ph[off + c]);
      if (pl) x[c] += bf16_to_f32(pl[off + c]);
    }
  }
}

template <int CPL, int SITE = 0>
__device__ __forceinline__ void stp(bf16_t* __restrict__ ph, bf16_t* __restrict__ pl, long off, const float (&x)[CPL]) {
  if (CPL == 4) {
    uint32_t h0, h1, l0, l1;
    split_bf16x2(x[0], x[1], h0, l0);
    split_bf16x2(x[CPL > 2 ? 2 : 0], x[CPL > 3 ? 3 : 0], h1, l1);
    egv_store<SITE>(ph + off, (u32x2_t){h0, h1});
    if (pl) egv_store<SITE>(pl + off, (u32x2_t){l0, l1});
    return;
  }
  bf16_t h[CPL], l[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) split_bf16(x[c], h[c], l[c]);
  if (CPL == 2) {
    *(uint32_t*)(ph + off) = pack2(h[0], h[1]);
    if (pl) *(uint32_t*)(pl + off) = pack2(l[0], l[1]);
  } else {
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      ph[off + c] = h[c];
      if (pl) pl[off + c] = l[c];
    }
  }
}

template <int CPL>
__device__ __forceinline__ float dotc(const float (&a)[CPL], const float (&b)[CPL]) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CPL; ++c) s += a[c] * b[c];
  return s;
}

// ------------------------------------------------------------------------------------------- time fwd
template <int TMAX, int HPW>
__global__ __launch_bounds__(256) void attn_time_fwd_kernel(const bf16_t* __restrict__ qh, const bf16_t* __restrict__ ql,
                                                            int B, int T, int n, int H, bf16_t* __restrict__ out_hi,
                                                            bf16_t* __restrict__ out_lo, float* __restrict__ lse,
                                                            float* __restrict__ cls_ws) {
  constexpr int LPH = 64 / HPW, CPL = HPW;
  const int lane = threadIdx.x & 63;
  const int HQ = H / HPW;
  const long gid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gid >= (long)B * n * HQ) return;
  const int hq = (int)(gid % HQ);
  const long r = gid / HQ;
  const int i = (int)(r % n);
  const int b = (int)(r / n);
  const int head = hq * HPW + lane / LPH;
  const int ch = (lane % LPH) * CPL;
  const long S = 1 + (long)T * n;
  const long HD = (long)H * D;
  const long ts = 3 * HD;
  const long base = (long)b * S * ts + (long)head * D + ch;   // token 0, q part
  float qc[CPL], kc[CPL], vc[CPL];
  ldp<CPL>(qh, ql, base, qc);
  ldp<CPL>(qh, ql, base + HD, kc);
  ldp<CPL>(qh, ql, base + 2 * HD, vc);
  // T <= 4: q, k, v of the location all live in registers.  T = 8 / 16 with four heads per wave (8-byte lanes): k and v stay
  // resident (2 x T x 4 registers), the queries are STREAMED through a two-deep register ring (the next query's load is in
  // flight while the current one is multiplied) -- 3 x 16 x 4 resident values would not fit two waves per SIMD.
  constexpr bool QS = TMAX > 4 && CPL >= 4;
  float q[QS ? 2 : TMAX][CPL], k[TMAX][CPL], v[TMAX][CPL];
#pragma unroll
  for (int f = 0; f < TMAX; ++f) {
#pragma unroll
    for (int c = 0; c < CPL; ++c) k[f][c] = v[f][c] = 0.f;
    if (!QS) {
#pragma unroll
      for (int c = 0; c < CPL; ++c) q[QS ? 0 : f][c] = 0.f;
    }
    if (f < T) {
      const long p = base + (1 + (long)f * n + i) * ts;
      if (!QS) ldp<CPL>(qh, ql, p, q[QS ? 0 : f]);
      ldp<CPL>(qh, ql, p + HD, k[f]);
      ldp<CPL>(qh, ql, p + 2 * HD, v[f]);
    }
  }
  if (QS) ldp<CPL>(qh, ql, base + (1 + (long)i) * ts, q[0]);
#pragma unroll
  for (int f = 0; f < TMAX; ++f) {
    if (f < T) {
      if (QS && f + 1 < T) ldp<CPL>(qh, ql, base + (1 + (long)(f + 1) * n + i) * ts, q[(f + 1) & 1]);
      const float (&qf)[CPL] = q[QS ? (f & 1) : f];
      float s[TMAX + 1];
      s[0] = redh<LPH>(dotc<CPL>(qf, kc)) * 0.125f;
      float m = s[0];
#pragma unroll
      for (int j = 0; j < TMAX; ++j) {
        s[j + 1] = -3e38f;
        if (j < T) {
          s[j + 1] = redh<LPH>(dotc<CPL>(qf, k[j])) * 0.125f;
          m = fmaxf(m, s[j + 1]);
        }
      }
      const float p0 = __expf(s[0] - m);
      float l = p0;
      float o[CPL];
#pragma unroll
      for (int c = 0; c < CPL; ++c) o[c] = p0 * vc[c];
#pragma unroll
      for (int j = 0; j < TMAX; ++j) {
        if (j < T) {
          const float pj = __expf(s[j + 1] - m);
          l += pj;
#pragma unroll
          for (int c = 0; c < CPL; ++c) o[c] += pj * v[j][c];
        }
      }
      const float inv = 1.0f / l;
#pragma unroll
      for (int c = 0; c < CPL; ++c) o[c] *= inv;
      const long tok = (long)b * S + 1 + (long)f * n + i;
      stp<CPL, EGV_NT_ATTN_OUT>(out_hi, out_lo, tok * HD + (long)head * D + ch, o);
      if (lane % LPH == 0 && lse) lse[((long)b * H + head) * S + 1 + (long)f * n + i] = m + __logf(l);
    }
  }
  // the clip's CLS query against this location's T keys (+ the CLS key, counted in location-group 0 only)
  {
    float s[TMAX + 1];
    s[0] = (i == 0) ? redh<LPH>(dotc<CPL>(qc, kc)) * 0.125f : -1e30f;
    float m = s[0];
#pragma unroll
    for (int j = 0; j < TMAX; ++j) {
      s[j + 1] = -3e38f;
      if (j < T) {
        s[j + 1] = redh<LPH>(dotc<CPL>(qc, k[j])) * 0.125f;
        m = fmaxf(m, s[j + 1]);
      }
    }
    const float p0 = (i == 0) ? __expf(s[0] - m) : 0.f;
    float l = p0;
    float o[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) o[c] = p0 * vc[c];
#pragma unroll
    for (int j = 0; j < TMAX; ++j) {
      if (j < T) {
        const float pj = __expf(s[j + 1] - m);
        l += pj;
#pragma unroll
        for (int c = 0; c < CPL; ++c) o[c] += pj * v[j][c];
      }
    }
    float* w = cls_ws + (((long)b * H + head) * n + i) * 68;
#pragma unroll
    for (int c = 0; c < CPL; ++c) w[ch + c] = o[c];
    if (lane % LPH == 0) {
      w[64] = m;
      w[65] = l;
    }
  }
}

// ------------------------------------------------------------------------------------------- time fwd, 16-byte lanes
// The same computation with a lane holding EIGHT channels: one wave = two neighbouring locations x four heads (8 lanes per
// head), so every load / store moves 16 B per lane (1 KiB per wave instruction, two 512-B runs) and a head's dot product is
// reduced with three DPP steps (quad xor 1, xor 2, row_half_mirror).  PMC on the 8-byte version: HBM traffic = algorithmic,
// 80 % of wave cycles in s_waitcnt at 2.9 TB/s (profiles/r02_i_pmc_attention.txt) -- half as many, twice as wide requests.
__device__ __forceinline__ float row8_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  return v;
}
__device__ __forceinline__ void ldp8(const bf16_t* __restrict__ ph, const bf16_t* __restrict__ pl, long off, float (&x)[8]) {
  const u32x4_t a = egv_load<EGV_NT_TIME_LD, u32x4_t>(ph + off);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    x[2 * e] = __uint_as_float(a[e] << 16);
    x[2 * e + 1] = __uint_as_float(a[e] & 0xffff0000u);
  }
  if (pl) {
    const u32x4_t b = egv_load<EGV_NT_TIME_LD, u32x4_t>(pl + off);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      x[2 * e] += __uint_as_float(b[e] << 16);
      x[2 * e + 1] += __uint_as_float(b[e] & 0xffff0000u);
    }
  }
}
__device__ __forceinline__ void stp8(bf16_t* __restrict__ ph, bf16_t* __restrict__ pl, long off, const float (&x)[8]) {
  uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
  split_bf16x2(x[0], x[1], h0, l0);
  split_bf16x2(x[2], x[3], h1, l1);
  split_bf16x2(x[4], x[5], h2, l2);
  split_bf16x2(x[6], x[7], h3, l3);
  egv_store16<EGV_NT_ATTN_OUT>(ph + off, (u32x4_t){h0, h1, h2, h3});
  if (pl) egv_store16<EGV_NT_ATTN_OUT>(pl + off, (u32x4_t){l0, l1, l2, l3});
}

// DBG (diagnostics build only): 1 = everything but the output stores, 2 = the loads alone
template <int TMAX, int DBG = 0>
__global__ __launch_bounds__(256) void attn_time_fwd8_kernel(const bf16_t* __restrict__ qh, const bf16_t* __restrict__ ql,
                                                             int B, int T, int n, int H, bf16_t* __restrict__ out_hi,
                                                             bf16_t* __restrict__ out_lo, float* __restrict__ lse,
                                                             float* __restrict__ cls_ws) {
  constexpr int CPL = 8;
  const int lane = threadIdx.x & 63;
  const int HQ = H / 4;
  const int NPAIR = (n + 1) / 2;
  const long gid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gid >= (long)B * NPAIR * HQ) return;
  const int hq = (int)(gid % HQ);
  const long r = gid / HQ;
  const int ip = (int)(r % NPAIR);
  const int b = (int)(r / NPAIR);
  const int i_raw = 2 * ip + (lane >> 5);
  const bool live = i_raw < n;                 // odd n: the second half of the last pair repeats location n - 1, stores off
  const int i = live ? i_raw : n - 1;
  const int head = hq * 4 + ((lane & 31) >> 3);
  const int ch = (lane & 7) * CPL;
  const long S = 1 + (long)T * n;
  const long HD = (long)H * D;
  const long ts = 3 * HD;
  const long base = (long)b * S * ts + (long)head * D + ch;   // token 0, q part
  float qc[CPL], kc[CPL], vc[CPL];
  ldp8(qh, ql, base, qc);
  ldp8(qh, ql, base + HD, kc);
  ldp8(qh, ql, base + 2 * HD, vc);
  float q[TMAX][CPL], k[TMAX][CPL], v[TMAX][CPL];
#pragma unroll
  for (int f = 0; f < TMAX; ++f) {
#pragma unroll
    for (int c = 0; c < CPL; ++c) q[f][c] = k[f][c] = v[f][c] = 0.f;
    if (f < T) {
      const long p = base + (1 + (long)f * n + i) * ts;
      ldp8(qh, ql, p, q[f]);
      ldp8(qh, ql, p + HD, k[f]);
      ldp8(qh, ql, p + 2 * HD, v[f]);
    }
  }
  if (DBG == 2) {
#pragma unroll
    for (int f = 0; f < TMAX; ++f)
#pragma unroll
      for (int c = 0; c < CPL; ++c) asm volatile("" ::"v"(q[f][c]), "v"(k[f][c]), "v"(v[f][c]));
#pragma unroll
    for (int c = 0; c < CPL; ++c) asm volatile("" ::"v"(qc[c]), "v"(kc[c]), "v"(vc[c]));
    return;
  }
  const bool store_on = (DBG != 1) || B < 0;
#pragma unroll
  for (int f = 0; f < TMAX; ++f) {
    if (f < T) {
      float s[TMAX + 1];
      s[0] = row8_sum(dotc<CPL>(q[f], kc)) * 0.125f;
      float m = s[0];
#pragma unroll
      for (int j = 0; j < TMAX; ++j) {
        s[j + 1] = -3e38f;
        if (j < T) {
          s[j + 1] = row8_sum(dotc<CPL>(q[f], k[j])) * 0.125f;
          m = fmaxf(m, s[j + 1]);
        }
      }
      const float p0 = __expf(s[0] - m);
      float l = p0;
      float o[CPL];
#pragma unroll
      for (int c = 0; c < CPL; ++c) o[c] = p0 * vc[c];
#pragma unroll
      for (int j = 0; j < TMAX; ++j) {
        if (j < T) {
          const float pj = __expf(s[j + 1] - m);
          l += pj;
#pragma unroll
          for (int c = 0; c < CPL; ++c) o[c] += pj * v[j][c];
        }
      }
      const float inv = 1.0f / l;
#pragma unroll
      for (int c = 0; c < CPL; ++c) o[c] *= inv;
      const long tok = (long)b * S + 1 + (long)f * n + i;
      if (live && store_on) {
        stp8(out_hi, out_lo, tok * HD + (long)head * D + ch, o);
        if ((lane & 7) == 0 && lse) lse[((long)b * H + head) * S + 1 + (long)f * n + i] = m + __logf(l);
      }
    }
  }
  // the clip's CLS query against this location's T keys (+ the CLS key, counted in location-group 0 only)
  {
    float s[TMAX + 1];
    s[0] = (i == 0) ? row8_sum(dotc<CPL>(qc, kc)) * 0.125f : -1e30f;
    float m = s[0];
#pragma unroll
    for (int j = 0; j < TMAX; ++j) {
      s[j + 1] = -3e38f;
      if (j < T) {
        s[j + 1] = row8_sum(dotc<CPL>(qc, k[j])) * 0.125f;
        m = fmaxf(m, s[j + 1]);
      }
    }
    const float p0 = (i == 0) ? __expf(s[0] - m) : 0.f;
    float l = p0;
    float o[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) o[c] = p0 * vc[c];
#pragma unroll
    for (int j = 0; j < TMAX; ++j) {
      if (j < T) {
        const float pj = __expf(s[j + 1] - m);
        l += pj;
#pragma unroll
        for (int c = 0; c < CPL; ++c) o[c] += pj * v[j][c];
      }
    }
    if (live && store_on) {
      float* w = cls_ws + (((long)b * H + head) * n + i) * 68;
      *(f32x4_t*)(w + ch) = (f32x4_t){o[0], o[1], o[2], o[3]};
      *(f32x4_t*)(w + ch + 4) = (f32x4_t){o[4], o[5], o[6], o[7]};
      if ((lane & 7) == 0) {
        w[64] = m;
        w[65] = l;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------- time bwd
// one workgroup = (b, head group, 16 consecutive locations); each wave walks 4 locations.  Per location the T keys /
// values and their gradient accumulators live in registers and ONE rolled loop walks the T patch queries plus, as
// iteration T, the clip's CLS query (global log-sum-exp / delta, its dq partial accumulated instead of stored) -- the
// first version unrolled everything, needed > 256 VGPRs (1 wave per SIMD) and ran at 229 us per call.
// The CLS token's raw dq / dk / dv partials stay in registers across locations: one LDS reduction + one atomicAdd per
// channel per workgroup.
// WPB = waves per workgroup.  A workgroup always covers 16 consecutive locations (one LDS reduction + one round of atomics for
// the CLS token's partials per 16 locations); with WPB = 4 a wave walks four of them one after the other, with WPB = 16 every
// wave has ONE location.  Each location is a dependent chain of ~T + 2 load round trips, so the kernel's time is (chains a
// wave slot runs in sequence) x (chain latency): at B = 32 there are 18 816 chains for 4 096 wave slots -- 1.15 waves per slot
// = two rounds of four chains with WPB = 4, 4.6 waves per slot = five rounds of one chain with WPB = 16.
template <int TMAX, int HPW, int WPB = 4>
__global__ __launch_bounds__(64 * WPB, (TMAX <= 4) ? 4 : (TMAX <= 8 ? 3 : 2)) void attn_time_bwd_kernel(const bf16_t* __restrict__ qh, const bf16_t* __restrict__ ql,
                                                               const bf16_t* __restrict__ doh,
                                                               const bf16_t* __restrict__ dol,
                                                               const float* __restrict__ lse,
                                                               const float* __restrict__ delta, int B, int T, int n,
                                                               int H, bf16_t* __restrict__ gh, bf16_t* __restrict__ gl,
                                                               float* __restrict__ dcls) {
  constexpr int LPH = 64 / HPW, CPL = HPW;
  __shared__ float red[3][WPB][64 * CPL];
  constexpr int LPW = 16 / WPB;              // locations per wave
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int HQ = H / HPW;
  const int chunks = (n + 15) / 16;
  const int ic = blockIdx.x % chunks;
  const int bh = blockIdx.x / chunks;
  const int hq = bh % HQ, b = bh / HQ;
  const int head = hq * HPW + lane / LPH;
  const int ch = (lane % LPH) * CPL;
  const long S = 1 + (long)T * n;
  const long HD = (long)H * D;
  const long ts = 3 * HD;
  // per-clip bases (wave-uniform) + 32-bit per-lane element offsets
  const bf16_t* qb = qh + (long)b * S * ts;
  const bf16_t* qbl = ql ? ql + (long)b * S * ts : nullptr;
  bf16_t* gb = gh + (long)b * S * ts;
  bf16_t* gbl = gl ? gl + (long)b * S * ts : nullptr;
  const bf16_t* ob = doh + (long)b * S * HD;
  const bf16_t* obl = dol ? dol + (long)b * S * HD : nullptr;
  const unsigned hc = (unsigned)(head * D + ch);
  const float* lb = lse + ((long)b * H + head) * S;
  float qc[CPL], kc[CPL], vc[CPL], goc[CPL];
  ldp<CPL>(qb, qbl, hc, qc);
  ldp<CPL>(qb, qbl, hc + HD, kc);
  ldp<CPL>(qb, qbl, hc + 2 * HD, vc);
  ldp<CPL>(ob, obl, hc, goc);
  const float Lc = lb[0], dc = delta[((long)b * H + head) * S];
  float dqc[CPL], dkc[CPL], dvc[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) dqc[c] = dkc[c] = dvc[c] = 0.f;

#pragma unroll 1
  for (int ii = 0; ii < LPW; ++ii) {
    const int i = ic * 16 + wave * LPW + ii;
    if (i >= n) break;
    float k[TMAX][CPL], v[TMAX][CPL], dk[TMAX][CPL], dv[TMAX][CPL];
#pragma unroll
    for (int j = 0; j < TMAX; ++j) {
#pragma unroll
      for (int c = 0; c < CPL; ++c) k[j][c] = v[j][c] = dk[j][c] = dv[j][c] = 0.f;
      if (j < T) {
        const unsigned to = (unsigned)((1 + j * n + i) * ts) + hc;
        ldp<CPL>(qb, qbl, to + HD, k[j]);
        ldp<CPL>(qb, qbl, to + 2 * HD, v[j]);
      }
    }
#pragma unroll 1
    for (int f = 0; f <= T; ++f) {
      const bool is_cls = (f == T);
      const int tok = is_cls ? 0 : 1 + f * n + i;
      float qf[CPL], gof[CPL];
      ldp<CPL>(qb, qbl, (unsigned)(tok * ts) + hc, qf);
      ldp<CPL>(ob, obl, (unsigned)(tok * HD) + hc, gof);
      const float L = lb[tok];
      // CLS key: every patch query sees it; the CLS query only in location-group 0
      float p0 = __expf(redh<LPH>(dotc<CPL>(qf, kc)) * 0.125f - L);
      if (is_cls && i != 0) p0 = 0.f;
      const float dp0 = redh<LPH>(dotc<CPL>(gof, vc));
      float p[TMAX], dp[TMAX];
      float dl = p0 * dp0;
#pragma unroll
      for (int j = 0; j < TMAX; ++j) {
        p[j] = dp[j] = 0.f;
        if (j < T) {
          p[j] = __expf(redh<LPH>(dotc<CPL>(qf, k[j])) * 0.125f - L);
          dp[j] = redh<LPH>(dotc<CPL>(gof, v[j]));
          dl += p[j] * dp[j];
        }
      }
      if (is_cls) dl = dc;                  // the CLS row's delta spans all locations: precomputed
      const float ds0 = p0 * (dp0 - dl);
      float dq[CPL];
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        dq[c] = ds0 * kc[c];
        dkc[c] += ds0 * qf[c];
        dvc[c] += p0 * gof[c];
      }
#pragma unroll
      for (int j = 0; j < TMAX; ++j) {
        if (j < T) {
          const float ds = p[j] * (dp[j] - dl);
#pragma unroll
          for (int c = 0; c < CPL; ++c) {
            dq[c] += ds * k[j][c];
            dk[j][c] += ds * qf[c];
            dv[j][c] += p[j] * gof[c];
          }
        }
      }
      if (is_cls) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) dqc[c] += dq[c];
      } else {
#pragma unroll
        for (int c = 0; c < CPL; ++c) dq[c] *= 0.125f;
        stp<CPL, (CPL >= 4 ? EGV_NT_TIME_BWD : 0)>(gb, gbl, (unsigned)(tok * ts) + hc, dq);
      }
    }
#pragma unroll
    for (int j = 0; j < TMAX; ++j) {
      if (j < T) {
        const unsigned to = (unsigned)((1 + j * n + i) * ts) + hc;
#pragma unroll
        for (int c = 0; c < CPL; ++c) dk[j][c] *= 0.125f;
        stp<CPL, (CPL >= 4 ? EGV_NT_TIME_BWD : 0)>(gb, gbl, to + HD, dk[j]);
        stp<CPL, (CPL >= 4 ? EGV_NT_TIME_BWD : 0)>(gb, gbl, to + 2 * HD, dv[j]);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    red[0][wave][lane * CPL + c] = dqc[c];
    red[1][wave][lane * CPL + c] = dkc[c];
    red[2][wave][lane * CPL + c] = dvc[c];
  }
  __syncthreads();
  if (wave == 0) {
    float* a = dcls + ((long)b * H + head) * 192 + ch;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int x = lane * CPL + c;
      float r0 = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll
      for (int w = 0; w < WPB; ++w) {
        r0 += red[0][w][x];
        r1 += red[1][w][x];
        r2 += red[2][w][x];
      }
      atomicAdd(a + c, r0);
      atomicAdd(a + 64 + c, r1);
      atomicAdd(a + 128 + c, r2);
    }
  }
}

// ------------------------------------------------------------------------------------------- CLS row helpers
// forward: merge the G softmax partials (o[64], m, l) of one (clip, head) -> output planes of token 0 + its lse.
// 256 threads: thread = (sub = t >> 4, 4 channels); the 16 subs walk the groups interleaved with an online merge, then
// one LDS round merges the subs.
__global__ __launch_bounds__(256) void attn_cls_combine_kernel(const float* __restrict__ ws, int G, int S, int H,
                                                              bf16_t* __restrict__ out_hi, bf16_t* __restrict__ out_lo,
                                                              float* __restrict__ lse) {
  __shared__ float sm[16], sl[16], so[16][64];
  const int t = threadIdx.x;
  const int sub = t >> 4, c4 = (t & 15) * 4;
  const int h = blockIdx.x % H, b = blockIdx.x / H;
  const float* w = ws + (long)blockIdx.x * G * 68;
  float m = -3e38f, l = 0.f;
  f32x4_t o = {0.f, 0.f, 0.f, 0.f};
  for (int g = sub; g < G; g += 16) {
    const float mg = w[(long)g * 68 + 64], lg = w[(long)g * 68 + 65];
    const f32x4_t og = *(const f32x4_t*)(w + (long)g * 68 + c4);
    const float mn = fmaxf(m, mg);
    const float ea = __expf(m - mn), eb = __expf(mg - mn);
    l = l * ea + lg * eb;
    o = o * ea + og * eb;
    m = mn;
  }
  if ((t & 15) == 0) {
    sm[sub] = m;
    sl[sub] = l;
  }
  *(f32x4_t*)&so[sub][c4] = o;
  __syncthreads();
  if (t < 64) {
    float mm = -3e38f;
#pragma unroll
    for (int i = 0; i < 16; ++i) mm = fmaxf(mm, sm[i]);
    float ll = 0.f, oo = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float e = __expf(sm[i] - mm);   // subs that saw no group: m = -3e38 -> e = 0
      ll += sl[i] * e;
      oo += so[i][t] * e;
    }
    oo /= ll;
    bf16_t hh, lo2;
    split_bf16(oo, hh, lo2);
    const long off = (long)b * S * H * D + (long)h * D + t;
    out_hi[off] = hh;
    if (out_lo) out_lo[off] = lo2;
    if (t == 0) lse[((long)b * H + h) * S] = mm + __logf(ll);
  }
}

// backward prologue: delta of the CLS query row = sum_d dO[d] * O[d] (== sum_j P_j dP_j over ALL keys)
__global__ __launch_bounds__(64) void attn_cls_delta_kernel(const bf16_t* __restrict__ oh, const bf16_t* __restrict__ ol,
                                                            const bf16_t* __restrict__ doh,
                                                            const bf16_t* __restrict__ dol, int S, int H,
                                                            float* __restrict__ delta, float* __restrict__ dcls) {
  const int lane = threadIdx.x;
  const int h = blockIdx.x % H, b = blockIdx.x / H;
  if (dcls) {   // this (clip, head)'s raw dq / dk / dv accumulators start at zero: saves a memset node per attention backward
    float* a = dcls + (long)blockIdx.x * 192;
    a[lane] = 0.f; a[64 + lane] = 0.f; a[128 + lane] = 0.f;
  }
  const long off = (long)b * S * H * D + (long)h * D + lane;
  float o = bf16_to_f32(oh[off]), g = bf16_to_f32(doh[off]);
  if (ol) o += bf16_to_f32(ol[off]);
  if (dol) g += bf16_to_f32(dol[off]);
  const float d = wave_sum(o * g);
  if (lane == 0) delta[((long)b * H + h) * S] = d;
}

// backward epilogue: the CLS token's accumulated raw dq / dk / dv -> gradient planes of token 0 (q and k carry 64^-0.5)
__global__ __launch_bounds__(64) void attn_cls_finish_kernel(const float* __restrict__ dcls, int S, int H,
                                                             bf16_t* __restrict__ gh, bf16_t* __restrict__ gl) {
  const int lane = threadIdx.x;
  const int h = blockIdx.x % H, b = blockIdx.x / H;
  const float* a = dcls + (long)blockIdx.x * 192;
  const long HD = (long)H * D;
  const long off = (long)b * S * 3 * HD + (long)h * D + lane;
  const float v[3] = {a[lane] * 0.125f, a[64 + lane] * 0.125f, a[128 + lane]};
#pragma unroll
  for (int part = 0; part < 3; ++part) {
    bf16_t hh, ll;
    split_bf16(v[part], hh, ll);
    gh[off + part * HD] = hh;
    if (gl) gl[off + part * HD] = ll;
  }
}

template <int TMAX>
int launch_time_fwd(const bf16_t* qh, const bf16_t* ql, int B, int T, int n, int H, bf16_t* oh, bf16_t* ol, float* lse,
                    float* ws, hipStream_t s) {
  bool done = false;
  if constexpr (TMAX <= 4) {   // 2 locations x 4 heads per wave, 16 B per lane (the per-lane q/k/v arrays fit up to T = 4)
    if (H % 4 == 0) {
      const long ngroups = (long)B * ((n + 1) / 2) * (H / 4);
#ifdef EGV_DIAG
      static const int dbg = getenv("EGV_TIME_DBG") ? atoi(getenv("EGV_TIME_DBG")) : 0;
      if (dbg == 1)
        EGV_LAUNCH((attn_time_fwd8_kernel<TMAX, 1>), dim3((unsigned)((ngroups + 3) / 4)), dim3(256), 0, s, qh, ql, B, T, n, H,
                   oh, ol, lse, ws);
      else if (dbg == 2)
        EGV_LAUNCH((attn_time_fwd8_kernel<TMAX, 2>), dim3((unsigned)((ngroups + 3) / 4)), dim3(256), 0, s, qh, ql, B, T, n, H,
                   oh, ol, lse, ws);
      else
#endif
      EGV_LAUNCH((attn_time_fwd8_kernel<TMAX>), dim3((unsigned)((ngroups + 3) / 4)), dim3(256), 0, s, qh, ql, B, T, n, H, oh,
                 ol, lse, ws);
      done = true;
    }
  }
  if (done) {
  } else {
    bool four = false;
    {
      if (H % 4 == 0) {
        const long ngroups = (long)B * n * (H / 4);
        EGV_LAUNCH((attn_time_fwd_kernel<TMAX, 4>), dim3((unsigned)((ngroups + 3) / 4)), dim3(256), 0, s, qh, ql, B, T, n, H,
                   oh, ol, lse, ws);
        four = true;
      }
    }
    if (!four) {
      const long ngroups = (long)B * n * H;
      EGV_LAUNCH((attn_time_fwd_kernel<TMAX, 1>), dim3((unsigned)((ngroups + 3) / 4)), dim3(256), 0, s, qh, ql, B, T, n, H,
                 oh, ol, lse, ws);
    }
  }
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}

template <int TMAX>
int launch_time_bwd(const bf16_t* qh, const bf16_t* ql, const bf16_t* doh, const bf16_t* dol, const float* lse,
                    const float* delta, int B, int T, int n, int H, bf16_t* gh, bf16_t* gl, float* dcls, hipStream_t s) {
  const int chunks = (n + 15) / 16;
  // four heads per wave = 8-byte lanes, one full 128-B line per head and store instruction (which is what lets the outputs
  // stream, csrc/common.h EGV_NT_TIME_BWD): 128 -> 99 us per call at B=32 (profiles/r02_tb_time_attention_bwd.txt); 8 waves x 2
  // locations per workgroup (isolated, B = 32: 95 us at 4 x 4, 80-84 at 8 x 2, 87 at 16 x 1; profiles/r03_attention_time_bwd_ab.txt).
  // Only the instances that are dispatched exist (`if constexpr`: the T = 8 / 16 forms of the four-heads kernel spilled).
  if constexpr (TMAX <= 4) {
    if (H % 4 == 0) {
      EGV_LAUNCH((attn_time_bwd_kernel<TMAX, 4, 8>), dim3((unsigned)(B * (H / 4) * chunks)), dim3(512), 0, s, qh, ql, doh, dol,
                 lse, delta, B, T, n, H, gh, gl, dcls);
      EGV_CHECK_LAUNCH();
      return EGV_OK;
    }
  }
  {
    if (H % 2 == 0) {
      EGV_LAUNCH((attn_time_bwd_kernel<TMAX, 2>), dim3((unsigned)(B * (H / 2) * chunks)), dim3(256), 0, s, qh, ql, doh, dol,
                 lse, delta, B, T, n, H, gh, gl, dcls);
      EGV_CHECK_LAUNCH();
      return EGV_OK;
    }
  }
  EGV_LAUNCH((attn_time_bwd_kernel<TMAX, 1>), dim3((unsigned)(B * H * chunks)), dim3(256), 0, s, qh, ql, doh, dol, lse,
             delta, B, T, n, H, gh, gl, dcls);
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}

}  // namespace

int egv_attn_time_mfma_fwd_impl(const bf16_t* qh, const bf16_t* ql, int B, int T, int n, int H, bf16_t* oh, bf16_t* ol, float* lse,
                                float* ws, hipStream_t s);
int egv_attn_time_mfma_bwd_impl(const bf16_t* qh, const bf16_t* ql, const bf16_t* doh, const bf16_t* dol, const float* lse,
                                const float* delta, int B, int T, int n, int H, bf16_t* gh, bf16_t* gl, float* dcls, hipStream_t s);

// T <= 4: the register-resident vector-ALU kernels above (HBM-bound); 4 < T <= 16: one wave per (location, head) on the matrix
// cores (attn_time_mfma.hip)
int egv_attn_time_fwd_impl(const bf16_t* qh, const bf16_t* ql, int B, int T, int n, int H, bf16_t* oh, bf16_t* ol,
                           float* lse, float* ws, hipStream_t s) {
  if (T <= 4) return launch_time_fwd<4>(qh, ql, B, T, n, H, oh, ol, lse, ws, s);
  if (T <= 16) return egv_attn_time_mfma_fwd_impl(qh, ql, B, T, n, H, oh, ol, lse, ws, s);
  return EGV_ERR_ARG;
}

int egv_attn_time_bwd_impl(const bf16_t* qh, const bf16_t* ql, const bf16_t* doh, const bf16_t* dol, const float* lse,
                           const float* delta, int B, int T, int n, int H, bf16_t* gh, bf16_t* gl, float* dcls,
                           hipStream_t s) {
  if (T <= 4) return launch_time_bwd<4>(qh, ql, doh, dol, lse, delta, B, T, n, H, gh, gl, dcls, s);
  if (T <= 16) return egv_attn_time_mfma_bwd_impl(qh, ql, doh, dol, lse, delta, B, T, n, H, gh, gl, dcls, s);
  return EGV_ERR_ARG;
}

int egv_attn_cls_combine_impl(const float* ws, int B, int G, int S, int H, bf16_t* oh, bf16_t* ol, float* lse,
                              hipStream_t s) {
  EGV_LAUNCH(attn_cls_combine_kernel, dim3(B * H), dim3(256), 0, s, ws, G, S, H, oh, ol, lse);
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}

int egv_attn_cls_delta_impl(const bf16_t* oh, const bf16_t* ol, const bf16_t* doh, const bf16_t* dol, int B, int S, int H,
                            float* delta, float* dcls, hipStream_t s) {
  EGV_LAUNCH(attn_cls_delta_kernel, dim3(B * H), dim3(64), 0, s, oh, ol, doh, dol, S, H, delta, dcls);
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}

int egv_attn_cls_finish_impl(const float* dcls, int B, int S, int H, bf16_t* gh, bf16_t* gl, hipStream_t s) {
  EGV_LAUNCH(attn_cls_finish_kernel, dim3(B * H), dim3(64), 0, s, dcls, S, H, gh, gl);
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}
