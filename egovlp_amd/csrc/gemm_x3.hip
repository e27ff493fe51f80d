// gemm_x3: the fp32-grade ("bf16x3") NT GEMM of the parity-mode forward pass, with the three products fused per k-tile.
//
// gemm_big runs bf16x3 as three k-segments -- (A_hi,B_lo), (A_lo,B_hi), (A_hi,B_hi) -- so every k-block's A_hi and B_hi
// planes are DMA-ed and fragment-read twice: 6 plane-tiles moved per 3 products.  Here one LDS stage holds all FOUR
// planes of a 32-deep k-tile, (64 MF + 256) rows x 64 B x 2 planes = the same 72 KiB as gemm_big's stage at MF = 5, and
// each fragment pair feeds three MFMAs:
//     acc += B_lo.A_hi ; acc += B_hi.A_lo ; acc += B_hi.A_hi          (small terms first)
// -> 26 ds_read_b128 and 72 KiB of LDS-DMA per 120 MFMAs (segments: 39 reads / 108 KiB), one barrier per 120 MFMAs.
// Same skeleton as gemm_big.hip: 8 waves as 4 x 2, wave tile (16 MF) x 128, two LDS stages filled by LDS-DMA, persistent
// workgroups, LDS-staged rolled epilogue, next output tile's first k-tile in flight under the epilogue.  Differences:
//   * 64-B LDS rows: 16-B chunk position XOR g[(row >> 2) & 3], g = {0,2,3,1} (gemm_nt.hip's conflict-free scheme);
//   * 4 phases per k-tile (2 B-fragments hi+lo each, register double-buffered); the A fragments (hi+lo, 40 VGPRs) are
//     single-buffered and reloaded ROLLING in the last phase: its MFMAs run row-major, and as soon as row i is done its
//     A_hi / A_lo fragments are re-read from the next k-tile's stage, so they have the rest of the phase to land.
#include <cstdlib>

#include "common.h"
#include "egovlp_hip.h"
#include "gemm_epi.h"

namespace {

constexpr int KT = 32;
constexpr int NFW = 8;
constexpr int BNX = 256;

__device__ __forceinline__ int swz4x(int x) { return (0x78 >> (2 * x)) & 3; }

__device__ __forceinline__ void glds16x(const void* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int MF, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_x3_kernel(const egv_gemm_desc p, const int dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NC = (MF == 5) ? 1 : 2;        // B fragments (hi + lo each) per phase: 160 accumulators leave room for one
  constexpr int NCH = NFW / NC;
  constexpr int BM = MF * 64;
  constexpr int A_PLANE = BM * 64;             // bytes of one A plane tile [BM][32] bf16
  constexpr int B_PLANE = BNX * 64;
  constexpr int OFF_AL = A_PLANE, OFF_BH = 2 * A_PLANE, OFF_BL = 2 * A_PLANE + B_PLANE;
  constexpr int STAGE = 2 * A_PLANE + 2 * B_PLANE;
  constexpr int NPA = 2 * BM / 16;             // DMA pieces (16 rows x 64 B) of [A_hi ; A_lo]: 40 (MF = 5) / 32
  constexpr int GA = NPA / 8;                  // per wave: 5 / 4
  constexpr int GB = 4;                        // [B_hi ; B_lo]: 32 pieces
  constexpr int EP_LD = 20;
  constexpr int EP_WAVE = MF * 16 * EP_LD * 4;
  static_assert(NPA % 8 == 0, "A pieces must split evenly over 8 waves");
  static_assert(8 * EP_WAVE <= STAGE, "epilogue staging must fit in one LDS stage");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;

  const int tiles_n = (p.N + BNX - 1) / BNX;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int nwg = tiles_m * tiles_n;
  const int nt = p.K / KT;                     // k-tiles per output tile

  // DMA source offsets (elements): piece = 16 rows; lane -> row (lane >> 2), LDS chunk position lane & 3 holding
  // source chunk pos ^ g[(row >> 2) & 3]
  const int srcchunk = (lane & 3) ^ swz4x((lane >> 4) & 3);
  const long a_lane = (long)(lane >> 2) * p.lda + srcchunk * 8;
  const long b_lane = (long)(lane >> 2) * p.ldb + srcchunk * 8;

  const int frow = lane & 15;
  const int foff = frow * 64 + (((lane >> 4) ^ swz4x(frow >> 2)) * 16);
  const int a_rd = (wm * MF * 16) * 64 + foff;            // + OFF_AL for the lo plane, + f * 1024 per fragment
  const int b_rd = OFF_BH + (wn * 128) * 64 + foff;       // + B_PLANE for the lo plane, + j * 1024

  bf16x8_t Ah[MF], Al[MF], Bh[2][NC], Bl[2][NC];
  auto load_a1 = [&](int sb, int f) {
    Ah[f] = *(const bf16x8_t*)(smem + sb + a_rd + f * 1024);
    Al[f] = *(const bf16x8_t*)(smem + sb + OFF_AL + a_rd + f * 1024);
  };
  auto load_b = [&](int sb, int c, int buf) {
#pragma unroll
    for (int jj = 0; jj < NC; ++jj) {
      Bh[buf][jj] = *(const bf16x8_t*)(smem + sb + b_rd + (c * NC + jj) * 1024);
      Bl[buf][jj] = *(const bf16x8_t*)(smem + sb + B_PLANE + b_rd + (c * NC + jj) * 1024);
    }
  };

  int m0, n0, sm0, sn0, st_kt;
  auto decode = [&](int v, int& om0, int& on0) {
    const int wg = xcd_remap(v, nwg);
    const int tm = wg / tiles_n;
    const int tnn = wg - tm * tiles_n;
    om0 = min(tm * BM, p.M - BM);
    on0 = min(tnn * BNX, p.N - BNX);
  };
  auto stage = [&](int buf) {
    char* lds = smem + buf * STAGE;
#pragma unroll
    for (int q = 0; q < GA; ++q) {
      const int P = wave * GA + q;                           // piece of [A_hi ; A_lo]
      const bool lo = P >= NPA / 2;
      const int blk = lo ? P - NPA / 2 : P;
      const bf16_t* src = (lo ? p.a_lo : p.a_hi) + (long)(sm0 + blk * 16) * p.lda + (long)st_kt * KT + a_lane;
      glds16x(src, lds + P * 1024);
    }
#pragma unroll
    for (int q = 0; q < GB; ++q) {
      const int P = wave * GB + q;                           // piece of [B_hi ; B_lo]
      const bool lo = P >= 16;
      const int blk = lo ? P - 16 : P;
      const bf16_t* src = (lo ? p.b_lo : p.b_hi) + (long)(sn0 + blk * 16) * p.ldb + (long)st_kt * KT + b_lane;
      glds16x(src, lds + OFF_BH + P * 1024);
    }
    ++st_kt;
  };

  int v = blockIdx.x;
  if (v >= nwg) return;
  decode(v, m0, n0);
  sm0 = m0; sn0 = n0; st_kt = 0;
  if (nt > 0) stage(0);

  for (;;) {
    if (dbg == 200) ts0 = __builtin_amdgcn_s_memrealtime();
    f32x4_t acc[MF][NFW];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
      for (int j = 0; j < NFW; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    if (nt > 0) {
      if (nt > 1) {
        stage(1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GA + GB) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      load_b(0, 0, 0);
#pragma unroll
      for (int f = 0; f < MF; ++f) load_a1(0, f);
    }
    if (dbg == 200) ts1 = __builtin_amdgcn_s_memrealtime();

    for (int t = 0; t < nt; ++t) {
      const int sb = (t & 1) * STAGE;
      const bool more = t + 1 < nt;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const bool last = (c == NCH - 1);
        const int cur = c & 1;
        // the three products of one fragment pair, small terms first: term 0 = B_lo.A_hi, 1 = B_hi.A_lo, 2 = B_hi.A_hi
        auto mma = [&](int term, int i, int jj) {
          f32x4_t& a = acc[i][c * NC + jj];
          if (term == 0) a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Bl[cur][jj], Ah[i], a, 0, 0, 0);
          else if (term == 1) a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Bh[cur][jj], Al[i], a, 0, 0, 0);
          else a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Bh[cur][jj], Ah[i], a, 0, 0, 0);
        };
        // first MFMA, then the prefetch of the next phase (see gemm_big.hip for why after)
        mma(0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (last) {
          if (more) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // k-tile t+1 landed; reads of stage t&1 done
            __builtin_amdgcn_s_barrier();
            if (t + 2 < nt) stage(t & 1);
            load_b(STAGE - sb, 0, 0);
          }
        } else {
          load_b(sb, c + 1, cur ^ 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!last) {
          // term-major: dependent accumulations are MF * NC MFMAs apart
#pragma unroll
          for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int jj = 0; jj < NC; ++jj)
#pragma unroll
              for (int i = 0; i < MF; ++i)
                if (term + jj + i > 0) mma(term, i, jj);
        } else {
          // last phase: row-major, and as soon as row i is finished for this k-tile its A_hi / A_lo fragments of k-tile
          // t+1 are fetched -- they have the rest of the phase to land
#pragma unroll
          for (int i = 0; i < MF; ++i) {
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
              for (int jj = 0; jj < NC; ++jj)
                if (term + jj + i > 0) mma(term, i, jj);
            __builtin_amdgcn_sched_barrier(0);
            if (more) load_a1(STAGE - sb, i);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (dbg == 200) ts2 = __builtin_amdgcn_s_memrealtime();

    // ---- hand-over -----------------------------------------------------------------------------------------------
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int vn = v + gridDim.x;
    const bool has_next = vn < nwg;
    int nm0 = 0, nn0 = 0;
    if (has_next) {
      decode(vn, nm0, nn0);
      sm0 = nm0; sn0 = nn0; st_kt = 0;
      if (nt > 0) stage(0);
    }
    {
      char* ep = smem + STAGE + wave * EP_WAVE;
      const int wr_off = ((lane & 15) * EP_LD + 4 * (lane >> 4)) * 4;
      const int rd_row = lane >> 2, rd_c4 = (lane & 3) * 4;
      const int mw = m0 + wm * MF * 16;
      const int nw = n0 + wn * 128;
#pragma unroll
      for (int j = 0; j < NFW; ++j) {
#pragma unroll
        for (int i = 0; i < MF; ++i) *(f32x4_t*)(ep + wr_off + i * 16 * EP_LD * 4) = acc[i][j];
        if (dbg >= 100) continue;
#pragma unroll 1
        for (int r = 0; r < MF; ++r) {
          const int row = r * 16 + rd_row;
          const f32x4_t val = *(const f32x4_t*)(ep + (row * EP_LD + rd_c4) * 4);
          epilogue4<EPI>(p, val, mw + row, nw + j * 16 + rd_c4, 0, 1);
        }
      }
    }
    if (dbg == 200 && tid == 0) {
      unsigned long long* tsb = (unsigned long long*)p.aux_out + (long)v * 4;
      tsb[0] = ts0; tsb[1] = ts1; tsb[2] = ts2; tsb[3] = __builtin_amdgcn_s_memrealtime();
    }
    if (!has_next) break;
    v = vn; m0 = nm0; n0 = nn0;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
}

template <int MF, int EPI>
int launch_x3(const egv_gemm_desc& p, hipStream_t s) {
  constexpr int BM = MF * 64;
  constexpr int lds = 2 * (2 * BM * 64 + 2 * BNX * 64);
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BNX - 1) / BNX);
  auto k = gemm_x3_kernel<MF, EPI>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return EGV_ERR_LAUNCH + (int)hipGetLastError();
    attr_set = true;
  }
  static const int dbg = getenv("EGV_GEMM_DBG") ? atoi(getenv("EGV_GEMM_DBG")) : 0;
  const int grid = tiles < 256 ? tiles : 256;
  EGV_LAUNCH(k, dim3(grid), dim3(512), lds, s, p, dbg);
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}

template <int MF>
int launch_x3_epi(const egv_gemm_desc& p, hipStream_t s) {
  if (p.alpha == 1.0f && p.act == EGV_ACT_NONE) {
    if (!p.bias && !p.residual && !p.out_hi && p.out_f32) return launch_x3<MF, EPI_RAW>(p, s);
    return launch_x3<MF, EPI_LINEAR>(p, s);
  }
  if (p.alpha == 1.0f && p.act == EGV_ACT_GELU) return launch_x3<MF, EPI_GELU>(p, s);
  if (p.alpha == 1.0f && p.act == EGV_ACT_GELU_BWD && !p.bias) return launch_x3<MF, EPI_GELU_BWD>(p, s);
  return launch_x3<MF, EPI_GENERIC>(p, s);
}

}  // namespace

bool egv_gemm_x3_supports(const egv_gemm_desc& p) {
  return !p.trans && p.passes == 3 && p.ksplit <= 1 && p.M >= 256 && p.N >= BNX && p.N % 4 == 0 && p.K % KT == 0 &&
         p.lda % 8 == 0 && p.ldb % 8 == 0;
}

int egv_gemm_x3_launch(const egv_gemm_desc& p, hipStream_t s, int mf) {
  if (mf == 5 && p.M >= 320) return launch_x3_epi<5>(p, s);
  return launch_x3_epi<4>(p, s);
}
