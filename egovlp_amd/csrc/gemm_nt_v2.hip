// gemm_nt v2: the large-M variant of gemm_nt.hip -- same math, same epilogue, deeper pipeline.
//
// Why: rocprofv3 (profiles/r01_a_*) showed the 2-stage 128x128 kernel at ~750 TF of MFMA issue in bf16x3 mode:
// with one tile of prefetch distance every k-step ends in `s_waitcnt vmcnt(0)` + barrier, so HBM/L2 latency
// (~1-2 us under load) is exposed once per 32-deep step.  v2 keeps TWO tiles in flight:
//   * 256x128 block tile, 8 waves (4 along M x 2 along N, 64x64 per wave, two waves per SIMD), BK = 32;
//   * 3-slot LDS ring (bf16x3: 3 x 48 KiB = 144 KiB of the CU's 160 KiB, one workgroup per CU; bf16: 3 x 24 KiB,
//     two workgroups per CU), filled by LDS-DMA (global_load_lds_dwordx4);
//   * counted waits: at step t the wave waits `vmcnt(G)` (G = its DMA instructions per tile) -- i.e. for tile t only,
//     tile t+1 stays in flight ACROSS the raw s_barrier -- then issues tile t+2 and multiplies tile t.  A plain
//     __syncthreads() would drain the DMA queue (it carries vmcnt(0) whenever LDS-DMA is outstanding).
// Bank swizzle, swapped-operand MFMA, epilogue and XCD-aware tile order are identical to gemm_nt.hip.
#include <cstdlib>

#include "common.h"
#include "egovlp_hip.h"

namespace {

constexpr int BM2 = 256, BN2 = 128, BK2 = 32;
constexpr int A_PLANE = BM2 * BK2 * 2;  // 16 KiB
constexpr int B_PLANE = BN2 * BK2 * 2;  //  8 KiB

__device__ __forceinline__ int swz_g2(int x) { return (0x78 >> (2 * x)) & 3; }

__device__ __forceinline__ void glds16b(const void* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int PASSES, bool PINGPONG>
__global__ __launch_bounds__(512, (PASSES == 3 || PINGPONG) ? 2 : 4) void gemm_nt_v2_kernel(const egv_gemm_desc p, const int blocked) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = (PASSES == 3) ? 2 * (A_PLANE + B_PLANE) : (A_PLANE + B_PLANE);
  constexpr int OFF_AH = 0;
  constexpr int OFF_AL = A_PLANE;                                    // PASSES == 3 only
  constexpr int OFF_BH = (PASSES == 3) ? 2 * A_PLANE : A_PLANE;
  constexpr int OFF_BL = OFF_BH + B_PLANE;                           // PASSES == 3 only
  constexpr int G = (PASSES == 3) ? 6 : 3;                           // DMA instructions per wave per tile

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int tiles_n = (p.N + BN2 - 1) / BN2;
  const int tiles_m = (p.M + BM2 - 1) / BM2;
  const int nwg = tiles_m * tiles_n;
  const int wg = xcd_remap(blockIdx.x, nwg);
  const int tm = wg / tiles_n, tn = wg % tiles_n;
  const int m0 = tm * BM2, n0 = tn * BN2;

  const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
  const int z = blockIdx.y;
  const int ksteps_total = p.K / BK2;
  const int ksteps_per = (ksteps_total + ksplit - 1) / ksplit;
  const int ks_begin = z * ksteps_per;
  const int ks_end = min(ksteps_total, ks_begin + ksteps_per);
  const int nk = ks_end - ks_begin;

  // per-lane DMA sources: A rows (2 instr/plane/wave), B rows (1 instr/plane/wave)
  // EXPERIMENT (blocked != 0): operands stored K-blocked [K/32][rows][32] (lda/ldb = rows): a tile's k-slice is one
  // contiguous run of 64-B rows, so every DMA instruction touches 8 full 128-B lines instead of 16 half lines.
  const int srcchunk = (lane & 3) ^ swz_g2((lane >> 4) & 3);
  const long koff = blocked ? (long)srcchunk * 8 : (long)ks_begin * BK2 + srcchunk * 8;
  const long a_step = blocked ? p.lda * 32 : BK2;
  const long b_step = blocked ? p.ldb * 32 : BK2;
  const bf16_t* a_src[2][2];
  const bf16_t* b_src[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const long ar = min(m0 + (wave * 2 + q) * 16 + (lane >> 2), p.M - 1);
    const long off = blocked ? ((long)ks_begin * p.lda + ar) * 32 + koff : ar * p.lda + koff;
    a_src[0][q] = p.a_hi + off;
    if (PASSES == 3) a_src[1][q] = p.a_lo + off;
  }
  {
    const long br = min(n0 + wave * 16 + (lane >> 2), p.N - 1);
    const long off = blocked ? ((long)ks_begin * p.ldb + br) * 32 + koff : br * p.ldb + koff;
    b_src[0] = p.b_hi + off;
    if (PASSES == 3) b_src[1] = p.b_lo + off;
  }

  auto stage = [&](int slot) {
    char* base = smem + slot * STAGE;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      glds16b(a_src[0][q], base + OFF_AH + (wave * 2 + q) * 1024);
      a_src[0][q] += a_step;
      if (PASSES == 3) {
        glds16b(a_src[1][q], base + OFF_AL + (wave * 2 + q) * 1024);
        a_src[1][q] += a_step;
      }
    }
    glds16b(b_src[0], base + OFF_BH + wave * 1024);
    b_src[0] += b_step;
    if (PASSES == 3) {
      glds16b(b_src[1], base + OFF_BL + wave * 1024);
      b_src[1] += b_step;
    }
  };

  const int frow = lane & 15;
  const int foff = frow * 64 + (((lane >> 4) ^ swz_g2(frow >> 2)) * 16);
  const int a_off = (wm * 64) * 64 + foff;
  const int b_off = (wn * 64) * 64 + foff;

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  if (nk > 0) stage(0);
  if (nk > 1) stage(1);
  int slot = 0;
  if (!PINGPONG) {
    for (int t = 0; t < nk; ++t) {
      // tile t has landed for THIS wave's DMA pieces; one more tile (t+1) may stay in flight
      if (t + 1 < nk) wait_vmcnt<G>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();   // every wave's pieces of tile t landed; slot (t+2)%3 no longer read by anyone
      if (t + 2 < nk) {
        int s2 = slot + 2;
        if (s2 >= 3) s2 -= 3;
        stage(s2);
      }
      const char* sb = smem + slot * STAGE;
      bf16x8_t ah[4], bh[4], al[4], bl[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        ah[f] = *(const bf16x8_t*)(sb + OFF_AH + a_off + f * 16 * 64);
        bh[f] = *(const bf16x8_t*)(sb + OFF_BH + b_off + f * 16 * 64);
        if (PASSES == 3) {
          al[f] = *(const bf16x8_t*)(sb + OFF_AL + a_off + f * 16 * 64);
          bl[f] = *(const bf16x8_t*)(sb + OFF_BL + b_off + f * 16 * 64);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (PASSES == 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
        }
      slot = (slot == 2) ? 0 : slot + 1;
    }
  } else {
    // ---- ping-pong schedule: the two waves that share a SIMD (wave w and w+4) run half a step apart, so one is in
    // its LDS-read phase (ds_read_b128 fragments + DMA issue) while the other is in its MFMA phase.  Two barriers per
    // k-step; "odd" barriers B(2t+1) carry the DMA wait for tile t.  Group 0 = waves 0-3: B R0 B M0 B R1 B M1 ...;
    // group 1 = waves 4-7: B -- B R0 B M0 B R1 ...  (both groups execute 2*nk+2 barriers).
    const int grp = wave >> 2;
    bf16x8_t ah[4], bh[4], al[4], bl[4];
    auto read_phase = [&](int t, int sl) {
      if (t + 2 < nk) {
        int s2 = sl + 2;
        if (s2 >= 3) s2 -= 3;
        stage(s2);
      }
      const char* sb = smem + sl * STAGE;
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        ah[f] = *(const bf16x8_t*)(sb + OFF_AH + a_off + f * 16 * 64);
        bh[f] = *(const bf16x8_t*)(sb + OFF_BH + b_off + f * 16 * 64);
        if (PASSES == 3) {
          al[f] = *(const bf16x8_t*)(sb + OFF_AL + a_off + f * 16 * 64);
          bl[f] = *(const bf16x8_t*)(sb + OFF_BL + b_off + f * 16 * 64);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto mfma_phase = [&]() {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (PASSES == 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
        }
      __builtin_amdgcn_s_setprio(0);
    };
    // barrier index b = 1 .. 2*nk+2.  odd b = 2t+1 (t < nk): everyone waits for its DMA share of tile t first.
    // group 0: after odd b=2t+1 -> R_t, after even b=2t+2 -> M_t.   group 1: after even b=2t+2 -> R_t, after odd b=2t+3 -> M_t.
    for (int t = 0; t <= nk; ++t) {
      if (t < nk) {
        if (t + 1 < nk) wait_vmcnt<G>(); else wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();                 // b = 2t+1
      __builtin_amdgcn_sched_barrier(0);
      if (grp == 0) {
        if (t < nk) read_phase(t, slot);
      } else {
        if (t > 0) mfma_phase();                    // M_{t-1}
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();                 // b = 2t+2
      __builtin_amdgcn_sched_barrier(0);
      if (grp == 0) {
        if (t < nk) mfma_phase();                   // M_t
      } else {
        if (t < nk) read_phase(t, slot);
      }
      __builtin_amdgcn_sched_barrier(0);
      slot = (slot == 2) ? 0 : slot + 1;
    }
  }

  // ---- epilogue (identical to gemm_nt.hip) ---------------------------------------------------
  const int lm = lane & 15;
  const int ln = (lane >> 4) * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + lm;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + ln;
      if (n >= p.N) continue;
      f32x4_t v = acc[i][j];
      if (ksplit > 1) {
        *(f32x4_t*)(p.partial + ((long)z * p.M + m) * p.N + n) = v;
        continue;
      }
      if (p.alpha != 1.0f) v *= p.alpha;
      if (p.bias) v += *(const f32x4_t*)(p.bias + n);
      if (p.act == EGV_ACT_GELU) {
        if (p.aux_out) *(f32x4_t*)(p.aux_out + (long)m * p.ldaux + n) = v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
      } else if (p.act == EGV_ACT_GELU_BWD) {
        const f32x4_t zv = *(const f32x4_t*)(p.aux_in + (long)m * p.ldaux + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= gelu_grad_f(zv[e]);
      } else if (p.act == EGV_ACT_RELU_BWD) {
        const f32x4_t zv = *(const f32x4_t*)(p.aux_in + (long)m * p.ldaux + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = zv[e] > 0.f ? v[e] : 0.f;
      }
      if (p.residual) v += *(const f32x4_t*)(p.residual + (long)m * p.ldr + n);
      if (p.out_f32) *(f32x4_t*)(p.out_f32 + (long)m * p.ldo + n) = v;
      if (p.out_hi) {
        bf16_t h[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split_bf16(v[e], h[e], l[e]);
        *(u32x2_t*)(p.out_hi + (long)m * p.ldoh + n) = (u32x2_t){pack2(h[0], h[1]), pack2(h[2], h[3])};
        if (p.out_lo)
          *(u32x2_t*)(p.out_lo + (long)m * p.ldoh + n) = (u32x2_t){pack2(l[0], l[1]), pack2(l[2], l[3])};
      }
    }
  }
}

}  // namespace

int egv_gemm_nt_v2_launch(const egv_gemm_desc& p, hipStream_t s) {
  static const int blocked = getenv("EGV_BLOCKED") ? atoi(getenv("EGV_BLOCKED")) : 0;
  static const int pingpong = getenv("EGV_PINGPONG") ? atoi(getenv("EGV_PINGPONG")) : 0;
  const int tiles = ((p.M + BM2 - 1) / BM2) * ((p.N + BN2 - 1) / BN2);
  const int ks = p.ksplit > 1 ? p.ksplit : 1;
  dim3 grid(tiles, ks), block(512);
  if (p.passes == 3) {
    constexpr int lds = 3 * 2 * (A_PLANE + B_PLANE);
    auto k = pingpong ? gemm_nt_v2_kernel<3, true> : gemm_nt_v2_kernel<3, false>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    EGV_LAUNCH(k, grid, block, lds, s, p, blocked);
  } else {
    constexpr int lds = 3 * (A_PLANE + B_PLANE);
    auto k = pingpong ? gemm_nt_v2_kernel<1, true> : gemm_nt_v2_kernel<1, false>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    EGV_LAUNCH(k, grid, block, lds, s, p, blocked);
  }
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}
