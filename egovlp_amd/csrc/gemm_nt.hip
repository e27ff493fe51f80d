// gemm_nt: C[M,N] = A[M,K] * B[N,K]^T (+ epilogue) on bf16 MFMA, fp32 accumulate, gfx950.
//
// This one kernel carries ~95 % of the step's FLOPs (qkv / proj / fc1 / fc2 /
// patch-embed / DistilBERT linears, and -- with pre-transposed operands -- every dgrad and wgrad).
//
// Operand format: "split-bf16" planes.  An fp32-grade tensor X is stored as X_hi = bf16(X) and
// X_lo = bf16(X - X_hi).  PASSES = 1 reads only the hi planes (plain bf16 GEMM); PASSES = 3 reads
// both and issues Ahi*Blo + Alo*Bhi + Ahi*Bhi into the same fp32 accumulator (error ~2^-16, i.e.
// fp32-grade, at 1/3 of the bf16 MFMA rate = 5x the fp32-MFMA peak of 157 TF).  The 3-pass loop
// does 3 MFMAs per 2 fragment loads, so it is less LDS-bound than the 1-pass loop.
//
// Tiling (designed for CDNA4, wave64): 128x128 block tile, BK = 32, 256 threads = 4 waves as 2x2,
// each wave 64x64 = 4x4 MFMA 16x16x32 fragments (64 fp32 accumulator VGPRs).  128^2 rather than
// 256^2 because the dominant shapes have N = 768: 197x6 = 1182 tiles fill 256 CUs x 2 blocks in
// 2.3 rounds, where 256^2 tiles (297) would run 1.16 rounds, i.e. at 58 % tail efficiency.
// Global -> LDS goes through the LDS-DMA path (global_load_lds_dwordx4, 1 KiB per wave-instr,
// no VGPR round trip), double-buffered, ONE barrier per k-step: the DMA of tile t+1 is in flight
// while tile t is multiplied.  The LDS image of a [128][32] bf16 plane has 64-B rows; since the
// DMA destination is lane-linear the bank swizzle is applied to the per-lane SOURCE address and,
// as the same involution, to the ds_read_b128 address (chunk' = chunk ^ g[(row>>2)&3],
// g = {0,2,3,1}: conflict-free for the four 16-lane groups ds_read_b128 is serviced in).
// The MFMA is issued "swapped" (weights as the A operand, activations as B) so that each lane ends
// up with 4 CONSECUTIVE output columns of one row: bias / residual / GELU epilogues then run on
// float4 and the stores are 16 B (fp32) or 8 B (bf16 planes) per lane.
// Block ids are remapped XCD-aware (common.h) so the ~6-24 blocks that share an A row-panel, and
// the weight panel they all stream, sit in one XCD's L2.
#include "common.h"
#include "egovlp_hip.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int PLANE_BYTES = BM * BK * 2;  // 8 KiB

struct Frag4 {
  bf16x8_t v[4];
};

__device__ __forceinline__ int swz_g(int x) { return (0x78 >> (2 * x)) & 3; }

__device__ __forceinline__ void glds16(const void* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// One 4-column piece of one output row: alpha, + bias, activation, + residual, stores (include/egovlp_hip.h order).
__device__ __forceinline__ void nt_epilogue4(const egv_gemm_desc& p, f32x4_t v, int m, int n) {
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
    if (p.out_lo) *(u32x2_t*)(p.out_lo + (long)m * p.ldoh + n) = (u32x2_t){pack2(l[0], l[1]), pack2(l[2], l[3])};
  }
}

template <int PASSES>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const egv_gemm_desc p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NPL = (PASSES == 3) ? 4 : 2;  // planes per stage: Ahi,(Alo),Bhi,(Blo)
  constexpr int STAGE_BYTES = NPL * PLANE_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int nwg = tiles_m * tiles_n;
  const int wg = xcd_remap(blockIdx.x, nwg);
  const int tm = wg / tiles_n, tn = wg % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // split-K range
  const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
  const int z = blockIdx.y;
  const int ksteps_total = p.K / BK;
  const int ksteps_per = (ksteps_total + ksplit - 1) / ksplit;
  const int ks_begin = z * ksteps_per;
  const int ks_end = min(ksteps_total, ks_begin + ksteps_per);
  const int nk = ks_end - ks_begin;

  // ---- per-lane DMA source pointers (2 wave-instructions per plane per k-step) -------------
  const int srcchunk = (lane & 3) ^ swz_g((lane >> 4) & 3);
  const bf16_t* asrc[2][2];  // [hi/lo][q]
  const bf16_t* bsrc[2][2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = (wave * 2 + q) * 16 + (lane >> 2);
    const long ar = min(m0 + row, p.M - 1);
    const long br = min(n0 + row, p.N - 1);
    const long koff = (long)ks_begin * BK + srcchunk * 8;
    asrc[0][q] = p.a_hi + ar * p.lda + koff;
    bsrc[0][q] = p.b_hi + br * p.ldb + koff;
    if (PASSES == 3) {
      asrc[1][q] = p.a_lo + ar * p.lda + koff;
      bsrc[1][q] = p.b_lo + br * p.ldb + koff;
    }
  }

  auto stage = [&](int buf) {
    char* base = smem + buf * STAGE_BYTES + wave * 2048;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      glds16(asrc[0][q], base + 0 * PLANE_BYTES + q * 1024);
      glds16(bsrc[0][q], base + (NPL / 2) * PLANE_BYTES + q * 1024);
      if (PASSES == 3) {
        glds16(asrc[1][q], base + 1 * PLANE_BYTES + q * 1024);
        glds16(bsrc[1][q], base + 3 * PLANE_BYTES + q * 1024);
      }
      asrc[0][q] += BK;
      bsrc[0][q] += BK;
      if (PASSES == 3) {
        asrc[1][q] += BK;
        bsrc[1][q] += BK;
      }
    }
  };

  // ---- per-lane fragment read offset (bytes within a plane) ---------------------------------
  const int frow = lane & 15;
  const int foff = frow * 64 + (((lane >> 4) ^ swz_g(frow >> 2)) * 16);
  const int a_off = (wm * 64) * 64 + foff;
  const int b_off = (wn * 64) * 64 + foff;

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  if (nk > 0) stage(0);
  for (int t = 0; t < nk; ++t) {
    __syncthreads();  // tile t landed (vmcnt(0) is part of the barrier) and buffer (t+1)&1 is free
    if (t + 1 < nk) stage((t + 1) & 1);
    const char* sb = smem + (t & 1) * STAGE_BYTES;
    Frag4 ah, bh, al, bl;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      ah.v[f] = *(const bf16x8_t*)(sb + 0 * PLANE_BYTES + a_off + f * 16 * 64);
      bh.v[f] = *(const bf16x8_t*)(sb + (NPL / 2) * PLANE_BYTES + b_off + f * 16 * 64);
      if (PASSES == 3) {
        al.v[f] = *(const bf16x8_t*)(sb + 1 * PLANE_BYTES + a_off + f * 16 * 64);
        bl.v[f] = *(const bf16x8_t*)(sb + 3 * PLANE_BYTES + b_off + f * 16 * 64);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // swapped operands: D[n][m] -> lane holds 4 consecutive n for one m
        if (PASSES == 3) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl.v[j], ah.v[i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh.v[j], al.v[i], acc[i][j], 0, 0, 0);
        }
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh.v[j], ah.v[i], acc[i][j], 0, 0, 0);
      }
  }

  // ---- epilogue ---------------------------------------------------------------------------
  const int lm = lane & 15;
  const int ln = (lane >> 4) * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + lm;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + ln;
      if (n >= p.N) continue;  // N is required to be a multiple of 4
      f32x4_t v = acc[i][j];
      if (ksplit > 1) {
        *(f32x4_t*)(p.partial + ((long)z * p.M + m) * p.N + n) = v;
        continue;
      }
      nt_epilogue4(p, v, m, n);
    }
  }
}

// out[m][n] = sum_z partial[z][m][n] (+bias); fp32, vectorised, HBM-bound.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                            long mn, int ksplit, int accumulate,
                                                            const float* __restrict__ partial2, float* __restrict__ out2,
                                                            long n2, int blocks1) {
  // second segment (blocks >= blocks1): the column-sum slab of a wgrad (bias gradient) rides in the same launch
  if ((int)blockIdx.x >= blocks1) {
    partial = partial2;
    out = out2;
    mn = n2;
    accumulate = 0;
  }
  const long i4 = ((long)((int)blockIdx.x >= blocks1 ? (int)blockIdx.x - blocks1 : (int)blockIdx.x) * blockDim.x + threadIdx.x) * 4;
  if (i4 >= mn) return;
  f32x4_t s = egv_load<EGV_NT_REDUCE_LD, f32x4_t>(partial + i4);
  for (int z = 1; z < ksplit; ++z) s += egv_load<EGV_NT_REDUCE_LD, f32x4_t>(partial + (long)z * mn + i4);
  if (accumulate) s += *(const f32x4_t*)(out + i4);
  egv_store<EGV_NT_WGRAD>(out + i4, s);
}

// The same reduction for SEVERAL weight gradients in one launch (egv_splitk_reduce_multi): the six TN GEMMs of a SpaceTimeBlock backward
// leave their slabs un-reduced (egv_gemm_desc.accumulate == 2) and the block call finishes them together -- 12 reduce launches per
// step on the wgrad stream instead of 72.
constexpr int RED_MAX_T = 8;
struct ReduceTable {
  const float* partial[RED_MAX_T];
  float* out[RED_MAX_T];
  float* colsum[RED_MAX_T];
  long mn[RED_MAX_T];
  int m[RED_MAX_T], ks[RED_MAX_T];
  int blk_start[RED_MAX_T + 1];     // first block of entry i (its product slab blocks, then its column-sum blocks)
  int blocks1[RED_MAX_T];           // product-slab blocks of entry i
  int count;
};
__global__ __launch_bounds__(256) void splitk_reduce_multi_kernel(const ReduceTable t) {
  int e = 0;
  while (e + 1 < t.count && (int)blockIdx.x >= t.blk_start[e + 1]) ++e;
  const int local = (int)blockIdx.x - t.blk_start[e];
  const int ks = t.ks[e];
  const float* partial = t.partial[e];
  float* out = t.out[e];
  long mn = t.mn[e];
  long i4;
  if (local >= t.blocks1[e]) {      // the ksplit x M column-sum slab behind the product slab
    partial += (long)ks * mn;
    out = t.colsum[e];
    mn = t.m[e];
    i4 = ((long)(local - t.blocks1[e]) * 256 + threadIdx.x) * 4;
  } else {
    i4 = ((long)local * 256 + threadIdx.x) * 4;
  }
  if (i4 >= mn) return;
  f32x4_t s = egv_load<EGV_NT_REDUCE_LD, f32x4_t>(partial + i4);
  for (int z = 1; z < ks; ++z) s += egv_load<EGV_NT_REDUCE_LD, f32x4_t>(partial + (long)z * mn + i4);
  egv_store<EGV_NT_WGRAD>(out + i4, s);
}

}  // namespace

extern "C" int egv_splitk_reduce_multi(int32_t count, const float* const* partial, float* const* out, const int64_t* mn, const int32_t* ksplit,
                                       float* const* colsum, const int32_t* m, void* stream) {
  if (count < 1 || count > RED_MAX_T || !partial || !out || !mn || !ksplit) return EGV_ERR_ARG;
  ReduceTable t = {};
  int nb = 0;
  for (int i = 0; i < count; ++i) {
    if (!partial[i] || !out[i] || mn[i] <= 0 || mn[i] % 4 != 0 || ksplit[i] < 2) return EGV_ERR_ARG;
    const bool cs = colsum && colsum[i];
    if (cs && (!m || m[i] <= 0 || m[i] % 4 != 0)) return EGV_ERR_ARG;
    t.partial[i] = partial[i]; t.out[i] = out[i]; t.colsum[i] = cs ? colsum[i] : nullptr;
    t.mn[i] = mn[i]; t.m[i] = cs ? m[i] : 0; t.ks[i] = ksplit[i];
    t.blk_start[i] = nb;
    t.blocks1[i] = (int)((mn[i] / 4 + 255) / 256);
    nb += t.blocks1[i] + (cs ? (m[i] / 4 + 255) / 256 : 0);
  }
  t.blk_start[count] = nb;
  t.count = count;
  EGV_LAUNCH(splitk_reduce_multi_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, t);
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}

int egv_gemm_big_launch(const egv_gemm_desc& p, hipStream_t s, int variant);  // gemm_big.hip
bool egv_gemm_big_supports(const egv_gemm_desc& p);

// Split-K for the small-M NT problems (DistilBERT: M = 1024 -> 48..192 tiles of 128x128 on 256 CUs, 24..96 k-steps each): the
// k-slices run as separate workgroups and this kernel sums their slabs and applies the FULL fused epilogue (bias, GELU / GELU'
// / ReLU', residual, fp32 and / or plane outputs), one 4-column piece per thread.
__global__ __launch_bounds__(256) void splitk_reduce_epilogue_kernel(const egv_gemm_desc p, int ksplit) {
  const long i4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const long mn = (long)p.M * p.N;
  if (i4 >= mn) return;
  f32x4_t s = *(const f32x4_t*)(p.partial + i4);
  for (int z = 1; z < ksplit; ++z) s += *(const f32x4_t*)(p.partial + (long)z * mn + i4);
  const int m = (int)(i4 / p.N), n = (int)(i4 - (long)m * p.N);
  nt_epilogue4(p, s, m, n);
}

// kernel choice: gemm_big (320/256 x 256 tile, k-tile 64) for the token-major GEMMs and every TN (wgrad) problem; the
// 128x128 two-stage kernel of this file where M is too small to fill big tiles (DistilBERT, projections) or K is not a
// multiple of 64.  -> 3 = big, 1 = small, -1 = unsupported.
static int gemm_variant(const egv_gemm_desc& p) {
  const bool big_ok = egv_gemm_big_supports(p);
  if (p.trans) return big_ok ? 3 : -1;
  if (p.passes == 2 || p.passes == 4) return big_ok ? 3 : -1;   // fp16 operands (f16x2 / one plain plane): the big-tile kernel is the only one that multiplies them
  // big tiles only when there are enough of them to occupy the chip (DistilBERT's M = 1024 GEMMs make 12)
  const long big_tiles = (long)((p.M + 255) / 256) * ((p.N + 255) / 256) * (p.ksplit > 1 ? p.ksplit : 1);
  if (big_ok && big_tiles >= 128) return 3;
  return 1;
}

extern "C" int egv_gemm_nt(const egv_gemm_desc* d, void* stream) {
  const egv_gemm_desc& p = *d;
  if (!p.a_hi || !p.b_hi) return EGV_ERR_ARG;
  if (p.passes < 1 || p.passes > 4) return EGV_ERR_ARG;
  if ((p.passes == 2 || p.passes == 3) && (!p.a_lo || !p.b_lo)) return EGV_ERR_ARG;
  if (p.out_fmt < 0 || p.out_fmt > 4) return EGV_ERR_ARG;
  if (p.M <= 0 || p.N <= 0 || p.K <= 0) return EGV_ERR_ARG;
  if (p.N % 4 != 0 || p.lda % 8 != 0 || p.ldb % 8 != 0) return EGV_ERR_ARG;
  if (!p.trans && p.K % BK != 0) return EGV_ERR_ARG;
  if (p.ksplit > 1 && !p.partial) return EGV_ERR_ARG;
  if (p.colsum && !p.trans) return EGV_ERR_ARG;
  if (p.grid_cap != 0 && (p.grid_cap < 8 || p.grid_cap > 256 || p.grid_cap % 8 != 0)) return EGV_ERR_ARG;
  const int variant = gemm_variant(p);
  if (variant < 0) return EGV_ERR_ARG;
  if (p.aux_bf16 < 0 || p.aux_bf16 > 3) return EGV_ERR_ARG;
  if (p.aux_bf16 && (variant < 3 || p.act == EGV_ACT_RELU_BWD)) return EGV_ERR_ARG;   // 16-bit aux: gemm_big GELU epilogues only
  if (p.aux_bf16 == 3 && p.passes != 4 && p.passes != 2) return EGV_ERR_ARG;           // fp16 gelu': fp16-product launches only
  // fp16 operands / outputs (csrc/f16x2.h; passes 2 = f16x2, 4 = one plain fp16 plane): the big-tile kernel only; NT un-split, or
  // (passes 4) the TN weight gradient of the fp16 backward with its k-slices
  if ((p.passes == 2 || p.passes == 4 || p.out_fmt != 0) && (variant < 3 || (p.passes != 2 && p.passes != 4))) return EGV_ERR_ARG;
  if ((p.passes == 2 || p.out_fmt != 0) && (p.trans || p.ksplit > 1)) return EGV_ERR_ARG;
  if (p.passes == 4 && !p.trans && p.ksplit > 1) return EGV_ERR_ARG;
  if (p.alpha != 1.0f && p.trans && !(p.alpha > 0.f)) return EGV_ERR_ARG;
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  const int ks = p.ksplit > 1 ? p.ksplit : 1;
  dim3 grid(tiles, ks), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (variant >= 3) {
    const int rc = egv_gemm_big_launch(p, s, variant);
    if (rc) return rc;
  } else if (p.passes == 3) {
    EGV_LAUNCH(gemm_nt_kernel<3>, grid, block, 2 * 4 * PLANE_BYTES, s, p);
  } else {
    EGV_LAUNCH(gemm_nt_kernel<1>, grid, block, 2 * 2 * PLANE_BYTES, s, p);
  }
  EGV_CHECK_LAUNCH();
  if (ks > 1 && !p.trans && variant < 3) {
    // small-tile NT kernels: slabs -> sum -> fused epilogue (any combination of outputs)
    if (p.accumulate) return EGV_ERR_ARG;
    const long mn4 = (long)p.M * p.N / 4;
    EGV_LAUNCH(splitk_reduce_epilogue_kernel, dim3((int)((mn4 + 255) / 256)), dim3(256), 0, s, p, ks);
    EGV_CHECK_LAUNCH();
  } else if (ks > 1 && p.accumulate == 2) {
    // the caller reduces the slabs itself (egv_splitk_reduce_multi: several weight gradients in one launch)
    if (!p.trans || !p.out_f32 || p.ldo != p.N) return EGV_ERR_ARG;
  } else if (ks > 1) {
    if (!p.out_f32 || p.ldo != p.N) return EGV_ERR_ARG;
    const long mn = (long)p.M * p.N;
    const int blocks = (int)((mn / 4 + 255) / 256);
    // the ksplit x M column-sum slab (bias gradient) sits behind the ksplit x M x N product slab: one launch reduces both
    if (p.colsum && p.M % 4 != 0) return EGV_ERR_ARG;
    const int blocks2 = p.colsum ? (p.M / 4 + 255) / 256 : 0;
    EGV_LAUNCH(splitk_reduce_kernel, dim3(blocks + blocks2), dim3(256), 0, s, p.partial, p.out_f32, mn, ks, p.accumulate,
               p.partial + (long)ks * mn, p.colsum, (long)p.M, blocks);
    EGV_CHECK_LAUNCH();
  }
  return EGV_OK;
}
