// gemm_duo: NT GEMM with TWO independent workgroups per CU, for the short-K (K = 768) GEMMs of the step.
//
// Why: gemm_big keeps one 8-wave workgroup per CU (144 KiB of LDS), so a tile's epilogue -- HBM-bound: up to 460 MB
// per launch for fc1 (pre-activation + GELU planes), 154 MB for proj (fp32 out + residual read) -- runs with the CU's
// matrix pipes idle.  rocprofv3 / tools/gemm_trace.py: qkv forward 75 us with the stores compiled out, 126-131 us
// with them; over a step ~5.8 ms of epilogue is exposed like that.  Two co-resident workgroups overlap one's epilogue
// (and barrier stalls, and prologue) with the other's MFMAs.
//   * 4 waves (2 x 2), block tile 160 x 256, wave tile 80 x 128 = 5 x 8 MFMA 16x16x32 fragments -- the SAME wave tile
//     (LDS read bytes per MFMA) as gemm_big's MF = 5; 240 VGPRs -> 2 waves per SIMD = the two workgroups;
//   * k-tile 32 (one MFMA k-step), 3-stage LDS ring of (160 + 256) x 64 B = 26 KiB per stage = 78 KiB per workgroup
//     (2 x 78 KiB of the CU's 160 KiB), filled by LDS-DMA; prefetch distance TWO k-tiles: at the boundary of tile u the
//     wave waits `vmcnt(G)` (tile u+1 landed, tile u+2 still in flight across the raw s_barrier) and issues tile u+3;
//   * 64-B LDS rows, 16-B chunk position XOR g[(row >> 2) & 3], g = {0,2,3,1} (gemm_nt.hip's scheme): conflict-free
//     ds_read_b128 fragment reads; the permutation is applied to the DMA SOURCE address (lane-linear destination);
//   * fragments register double-buffered in 4 phases per k-tile (the prefetch follows the first MFMA of a phase, see
//     gemm_big.hip); the tile loop is unrolled by two so the A-fragment buffers alternate statically;
//   * persistent workgroups (grid = min(#tiles, 512)), LDS-staged rolled epilogue (gemm_epi.h) through ring stage 2
//     while the next output tile's first two k-tiles land in stages 0 and 1.
// 157 x 3 = 471 tiles of 160 x 256 cover the M = 25 120, N = 768 GEMMs in one round of 512 slots (92 %).
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "egovlp_hip.h"
#include "gemm_epi.h"

namespace {

constexpr int KT = 32;
constexpr int MF = 5;                 // 16-row fragments per wave
constexpr int NFW = 8;                // 16-column fragments per wave
constexpr int NC = 2;                 // B fragments per phase
constexpr int NCH = NFW / NC;         // phases per k-tile
constexpr int BM = 2 * MF * 16;       // 160
constexpr int BN = 256;
constexpr int A_BYTES = BM * 64;      // 10 KiB
constexpr int B_BYTES = BN * 64;      // 16 KiB
constexpr int STAGE = A_BYTES + B_BYTES;
constexpr int NPA = BM / 16;          // 10 DMA pieces (16 rows x 64 B) for A, 16 for B
constexpr int GA = (NPA + 3) / 4;     // 3 per wave (waves 2, 3 re-load one piece: same bytes to the same place)
constexpr int GB = 4;
constexpr int G = GA + GB;            // DMA instructions per wave per k-tile
constexpr int EP_LD = 20;
constexpr int EP_WAVE = MF * 16 * EP_LD * 4;
static_assert(4 * EP_WAVE <= STAGE, "epilogue staging must fit in one ring stage");

__device__ __forceinline__ int swz4(int x) { return (0x78 >> (2 * x)) & 3; }

__device__ __forceinline__ void glds16d(const void* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_duo_kernel(const egv_gemm_desc p, const int dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;

  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int nwg = tiles_m * tiles_n;
  const int nkt = p.K / KT;
  const int nseg = (p.passes == 3) ? 3 : 1;
  const int nt = nkt * nseg;                       // k-tiles per output tile (no split-K in this kernel)

  auto seg_a = [&](int s) -> const bf16_t* { return (nseg == 3 && s == 1) ? p.a_lo : p.a_hi; };
  auto seg_b = [&](int s) -> const bf16_t* { return (nseg == 3 && s == 0) ? p.b_lo : p.b_hi; };

  // DMA: a piece = 16 rows x 64 B; lane -> row (lane >> 2), LDS chunk position lane & 3, which holds source chunk
  // pos ^ g[(row >> 2) & 3]
  const int srcchunk = (lane & 3) ^ swz4((lane >> 4) & 3);
  const long a_lane = (long)(lane >> 2) * p.lda + srcchunk * 8;
  const long b_lane = (long)(lane >> 2) * p.ldb + srcchunk * 8;

  // fragment read offsets within a stage
  const int frow = lane & 15;
  const int foff = frow * 64 + (((lane >> 4) ^ swz4(frow >> 2)) * 16);
  const int a_rd = (wm * MF * 16) * 64 + foff;
  const int b_rd = A_BYTES + (wn * 128) * 64 + foff;

  bf16x8_t A[2][MF], Bq[2][NC];
  auto load_a = [&](int sb, bf16x8_t (&dst)[MF]) {
#pragma unroll
    for (int f = 0; f < MF; ++f) dst[f] = *(const bf16x8_t*)(smem + sb + a_rd + f * 1024);
  };
  auto load_b = [&](int sb, int c, bf16x8_t (&dst)[NC]) {
#pragma unroll
    for (int jj = 0; jj < NC; ++jj) dst[jj] = *(const bf16x8_t*)(smem + sb + b_rd + (c * NC + jj) * 1024);
  };

  int m0, n0;                  // current output tile
  int sm0, sn0, st_seg, st_kt; // tile being staged / its next k-tile
  auto decode = [&](int v, int& om0, int& on0) {
    const int wg = xcd_remap(v, nwg);
    const int tm = wg / tiles_n;
    const int tnn = wg - tm * tiles_n;
    om0 = min(tm * BM, p.M - BM);
    on0 = min(tnn * BN, p.N - BN);
  };
  auto stage = [&](int buf) {
    char* lds = smem + buf * STAGE;
    const bf16_t* ab = seg_a(st_seg) + (long)sm0 * p.lda + (long)st_kt * KT;
    const bf16_t* bb = seg_b(st_seg) + (long)sn0 * p.ldb + (long)st_kt * KT;
#pragma unroll
    for (int q = 0; q < GA; ++q) {
      int pc = wave + 4 * q;
      if (pc >= NPA) pc -= 4;                       // waves 2, 3: duplicate of their previous piece
      glds16d(ab + a_lane + (long)pc * 16 * p.lda, lds + pc * 1024);
    }
#pragma unroll
    for (int q = 0; q < GB; ++q) {
      const int pc = wave + 4 * q;
      glds16d(bb + b_lane + (long)pc * 16 * p.ldb, lds + A_BYTES + pc * 1024);
    }
    if (++st_kt == nkt) {
      st_kt = 0;
      ++st_seg;
    }
  };

  int v = blockIdx.x;
  if (v >= nwg) return;
  decode(v, m0, n0);
  sm0 = m0; sn0 = n0; st_seg = 0; st_kt = 0;
  if (nt > 0) stage(0);
  if (nt > 1) stage(1);

  for (;;) {
    if (dbg == 200) ts0 = __builtin_amdgcn_s_memrealtime();
    f32x4_t acc[MF][NFW];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
      for (int j = 0; j < NFW; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // ---- ring prologue: k-tiles 0, 1 are staged (stages 0, 1); add k-tile 2, wait for k-tile 0 ------------------
    if (nt > 2) stage(2);
    if (nt > 2) wait_vm<2 * G>(); else if (nt > 1) wait_vm<G>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (nt > 0) {
      load_b(0, 0, Bq[0]);
      load_a(0, A[0]);
    }
    if (dbg == 200) ts1 = __builtin_amdgcn_s_memrealtime();

    // ---- one k-tile: A fragments in A[PAR], the next tile's go to A[PAR ^ 1] ------------------------------------
    int slot = 0;                                    // ring stage of k-tile t
    auto ktile = [&](auto par_tag, int t) {
      constexpr int PAR = decltype(par_tag)::value;
      const int sb = slot * STAGE;
      const int slot1 = (slot == 2) ? 0 : slot + 1;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const bool last = (c == NCH - 1);
        acc[0][c * NC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Bq[c & 1][0], A[PAR][0], acc[0][c * NC], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (last) {
          if (t + 1 < nt) {
            // k-tile t+1 landed (k-tile t+2 may stay in flight); this wave's reads of stage `slot` have returned
            if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(G) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (t + 3 < nt) stage(slot);             // k-tile t+3 reuses the stage k-tile t just vacated
            load_b(slot1 * STAGE, 0, Bq[0]);
            load_a(slot1 * STAGE, A[PAR ^ 1]);
          }
        } else {
          load_b(sb, c + 1, Bq[(c + 1) & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jj = 0; jj < NC; ++jj)
#pragma unroll
          for (int i = 0; i < MF; ++i)
            if (jj + i > 0)
              acc[i][c * NC + jj] =
                  __builtin_amdgcn_mfma_f32_16x16x32_bf16(Bq[c & 1][jj], A[PAR][i], acc[i][c * NC + jj], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      slot = slot1;
    };
    for (int t = 0; t < nt; t += 2) {
      ktile(std::integral_constant<int, 0>{}, t);
      if (t + 1 < nt) ktile(std::integral_constant<int, 1>{}, t + 1);
    }
    if (dbg == 200) ts2 = __builtin_amdgcn_s_memrealtime();

    // ---- hand-over: every wave is done with the ring; next output tile's k-tiles 0, 1 -> stages 0, 1 ----------------
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int vn = v + gridDim.x;
    const bool has_next = vn < nwg;
    int nm0 = 0, nn0 = 0;
    if (has_next) {
      decode(vn, nm0, nn0);
      sm0 = nm0; sn0 = nn0; st_seg = 0; st_kt = 0;
      if (nt > 0) stage(0);
      if (nt > 1) stage(1);
    }

    // ---- epilogue through ring stage 2 (see gemm_big.hip) -----------------------------------------------------------
    {
      char* ep = smem + 2 * STAGE + wave * EP_WAVE;
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
    // stages 0, 1 of the new tile have landed and this wave's epilogue stores have drained (vmcnt counts them too);
    // after the barrier stage 2 (every wave's epilogue staging rows) may be overwritten by k-tile 2.
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
}

template <int EPI>
int launch_duo(const egv_gemm_desc& p, hipStream_t s) {
  constexpr int lds = 3 * STAGE;
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  auto k = gemm_duo_kernel<EPI>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return EGV_ERR_LAUNCH + (int)hipGetLastError();
    attr_set = true;
  }
  static const int dbg = getenv("EGV_GEMM_DBG") ? atoi(getenv("EGV_GEMM_DBG")) : 0;
  const int grid = tiles < 512 ? tiles : 512;      // two persistent workgroups per CU; 512 keeps v % 8 == blockIdx % 8
  EGV_LAUNCH(k, dim3(grid), dim3(256), lds, s, p, dbg);
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}

}  // namespace

bool egv_gemm_duo_supports(const egv_gemm_desc& p) {
  return !p.trans && p.ksplit <= 1 && p.M >= BM && p.N >= BN && p.N % 4 == 0 && p.K % KT == 0 && p.lda % 8 == 0 &&
         p.ldb % 8 == 0;
}

int egv_gemm_duo_launch(const egv_gemm_desc& p, hipStream_t s) {
  if (p.alpha == 1.0f && p.act == EGV_ACT_NONE) {
    if (!p.bias && !p.residual && !p.out_hi && p.out_f32) return launch_duo<EPI_RAW>(p, s);
    return launch_duo<EPI_LINEAR>(p, s);
  }
  if (p.alpha == 1.0f && p.act == EGV_ACT_GELU) return launch_duo<EPI_GELU>(p, s);
  if (p.alpha == 1.0f && p.act == EGV_ACT_GELU_BWD && !p.bias) return launch_duo<EPI_GELU_BWD>(p, s);
  return launch_duo<EPI_GENERIC>(p, s);
}
