// Diagnostics (not on the product path): the sustained MFMA issue rate of this chip under its power limit.
// egv_diag_mfma_peak runs `iters` rounds of 40 independent v_mfma_f32_16x16x32_bf16 per wave (the accumulator footprint of
// a gemm_big wave, 160 registers), no memory traffic in the loop, `waves` waves per workgroup (8 = two per SIMD, the
// gemm_big occupancy), one workgroup per CU.  tools/mfma_peak.py turns the time into TFLOP/s: the number to read the
// GEMM main-loop rate against when the nominal 2.5 PFLOP/s assumes the peak engine clock.
#include "common.h"
#include "egovlp_hip.h"

namespace {

__global__ __launch_bounds__(512, 2) void mfma_peak_kernel(int iters, float* out) {
  f32x4_t acc[40];
#pragma unroll
  for (int i = 0; i < 40; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 av = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, (short)(0x3F80 + (threadIdx.x & 1))};
  s16x8 bv = av;
  if (iters < 0) {   // random bf16 in (-2, 2): realistic multiplier toggling (the sustained rate is power-limited and data-dependent)
    iters = -iters;
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      h = h * 1664525u + 1013904223u;
      av[e] = (short)((h >> 16 & 0x807F) | 0x3F00 | ((h >> 8) & 0x0080));
      h = h * 1664525u + 1013904223u;
      bv[e] = (short)((h >> 16 & 0x807F) | 0x3F00 | ((h >> 8) & 0x0080));
    }
  }
  bf16x8_t a = __builtin_bit_cast(bf16x8_t, av), b = __builtin_bit_cast(bf16x8_t, bv);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 40; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  f32x4_t s = acc[0];
#pragma unroll
  for (int i = 1; i < 40; ++i) s += acc[i];
  if (s[0] + s[1] + s[2] + s[3] == -1.0f) out[threadIdx.x] = s[0];   // never true: keeps the loop alive
}

// The gemm_big wave loop without DMA and without barriers: per round 4 phases of 10 MFMAs over 5 A x 2 B fragment
// registers into 40 accumulators.  MODE 1: fragments loaded once (register operand pattern only); MODE 2: the B fragments of
// the phase after next are re-read from LDS in every phase and the A fragments once per round (asm reads, counted waits):
// the LDS traffic of the GEMM main loop (13 ds_read_b128 per 40 MFMAs).
template <int MODE>
__global__ __launch_bounds__(512, 2) void mfma_loop_kernel(int iters, float* out) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  typedef __attribute__((ext_vector_type(4))) unsigned u4;
  for (int i = threadIdx.x; i < 65536 / 16; i += blockDim.x) ((u4*)lds)[i] = (u4){0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
  __syncthreads();
  const unsigned base = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)lds + (threadIdx.x & 63) * 16 +
                        (threadIdx.x >> 6) * 4096;
  f32x4_t acc[5][8];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  bf16x8_t A[5], B[3][2];
  auto rd = [&](unsigned off) {
    u4 r;
    asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(base + off));
    return __builtin_bit_cast(bf16x8_t, r);
  };
#pragma unroll
  for (int i = 0; i < 5; ++i) A[i] = rd(i * 1024 % 4096);
#pragma unroll
  for (int q = 0; q < 3; ++q) { B[q][0] = rd(0); B[q][1] = rd(1024); }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ph = 0; ph < 12; ++ph) {   // 12 phases = 3 rounds so that the B ring index (ph % 3) is static
      const int c = ph & 3;
      if (MODE == 2) {
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2) : "memory");
        asm volatile("" : "+v"(B[ph % 3][0]), "+v"(B[ph % 3][1]));
        if (c == 0) {
#pragma unroll
          for (int i = 0; i < 5; ++i) asm volatile("" : "+v"(A[i]));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 10; ++k) {
        const int jj = k / 5, i = k % 5;
        acc[i][c * 2 + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B[ph % 3][jj], A[i], acc[i][c * 2 + jj], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 2 && k == 0) {
          B[(ph + 2) % 3][0] = rd(((ph * 2) & 3) * 1024);
          B[(ph + 2) % 3][1] = rd(((ph * 2 + 1) & 3) * 1024);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 2 && c == 3 && k >= 1 && k <= 5) {   // A fragments of the next round, one per MFMA, behind their last use
          // (A[k-1] was last read by MFMA k-1+5 at the latest in this phase only when jj == 1; keep it simple: reload at the end)
        }
      }
      if (MODE == 2 && c == 3) {
#pragma unroll
        for (int i = 0; i < 5; ++i) A[i] = rd(i * 1024 % 4096);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(4) : "memory");   // the A reads have landed; two B sets may be in flight
      }
    }
  }
  f32x4_t s = acc[0][0];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[i][j];
  if (s[0] + s[1] + s[2] + s[3] == -1.0f) out[threadIdx.x] = s[0];
}

// Calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on the GEMM's own access paths (tools/traffic_calib.py):
// a copy over a KNOWN byte count.  Loads: LDS-DMA `global_load_lds` dwordx4 (the operand path of gemm_big: 1 KiB per wave
// instruction, 16 B per lane) or plain 16-byte global loads; stores: 16 B per lane, write-back or non-temporal (the two
// flavours the epilogues use, common.h egv_store).  mode bits: 1 = LDS-DMA loads, 2 = nt stores, 4 = no stores (read only),
// 8 = no loads (write only: fills dst).  Every byte of src is read once and every byte of dst written once per launch.
template <int MODE>
__global__ __launch_bounds__(256) void traffic_calib_kernel(const char* __restrict__ src, char* __restrict__ dst, long pieces) {
  __shared__ __attribute__((aligned(16))) char lds[4 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  char* my = lds + wave * 1024;
  typedef __attribute__((ext_vector_type(4))) unsigned u4;
  u4 keep = (u4){0u, 0u, 0u, 0u};
  const long stride = (long)gridDim.x * 4;
  for (long p = (long)blockIdx.x * 4 + wave; p < pieces; p += stride) {
    u4 v = (u4){(unsigned)p, 1u, 2u, 3u};
    if constexpr ((MODE & 8) == 0) {
      if constexpr (MODE & 1) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + p * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)my, 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        v = *(const u4*)(my + lane * 16);
      } else {
        v = *(const u4*)(src + p * 1024 + lane * 16);
      }
    }
    if constexpr (MODE & 4) {
      keep ^= v;
    } else if constexpr (MODE & 2) {
      __builtin_nontemporal_store(v, (u4*)(dst + p * 1024 + lane * 16));
    } else {
      *(u4*)(dst + p * 1024 + lane * 16) = v;
    }
  }
  if constexpr (MODE & 4) {
    if ((keep[0] ^ keep[1] ^ keep[2] ^ keep[3]) == 0x9E3779B9u) *(u4*)(dst + lane * 16) = keep;   // practically never: keeps the loads
  }
}

}  // namespace

extern "C" int egv_diag_traffic_calib(int32_t mode, const void* src, void* dst, int64_t bytes, void* stream) {
  if (!src || !dst || bytes < 1024 || bytes % 1024) return EGV_ERR_ARG;
  const long pieces = bytes / 1024;
  const dim3 grid(2048), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (mode) {
    case 0: EGV_LAUNCH(traffic_calib_kernel<0>, grid, block, 0, s, (const char*)src, (char*)dst, pieces); break;
    case 1: EGV_LAUNCH(traffic_calib_kernel<1>, grid, block, 0, s, (const char*)src, (char*)dst, pieces); break;
    case 2: EGV_LAUNCH(traffic_calib_kernel<2>, grid, block, 0, s, (const char*)src, (char*)dst, pieces); break;
    case 3: EGV_LAUNCH(traffic_calib_kernel<3>, grid, block, 0, s, (const char*)src, (char*)dst, pieces); break;
    case 4: EGV_LAUNCH(traffic_calib_kernel<4>, grid, block, 0, s, (const char*)src, (char*)dst, pieces); break;
    case 5: EGV_LAUNCH(traffic_calib_kernel<5>, grid, block, 0, s, (const char*)src, (char*)dst, pieces); break;
    case 8: EGV_LAUNCH(traffic_calib_kernel<8>, grid, block, 0, s, (const char*)src, (char*)dst, pieces); break;
    case 10: EGV_LAUNCH(traffic_calib_kernel<10>, grid, block, 0, s, (const char*)src, (char*)dst, pieces); break;
    default: return EGV_ERR_ARG;
  }
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}

extern "C" int egv_diag_mfma_peak(int32_t iters, int32_t waves, float* out, void* stream) {
  if (iters == 0 || waves < 1 || waves > 8 + 200 || !out) return EGV_ERR_ARG;   // iters < 0: random operands (plain kernel)
  if (waves >= 200) {        // 200 + w: LDS-fed loop;  100 + w: register-operand pattern only (iters = rounds of 120 MFMAs)
    EGV_LAUNCH(mfma_loop_kernel<2>, dim3(256), dim3(64 * (waves - 200)), 0, (hipStream_t)stream, iters, out);
    EGV_CHECK_LAUNCH();
    return EGV_OK;
  }
  if (waves >= 100) {
    EGV_LAUNCH(mfma_loop_kernel<1>, dim3(256), dim3(64 * (waves - 100)), 0, (hipStream_t)stream, iters, out);
    EGV_CHECK_LAUNCH();
    return EGV_OK;
  }
  EGV_LAUNCH(mfma_peak_kernel, dim3(256), dim3(64 * waves), 0, (hipStream_t)stream, iters, out);
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}
