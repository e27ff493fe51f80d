// mx_probe: what gfx950's block-scaled MFMA computes, and what a mixed fp16 + MXFP6 product stream costs.
// Standalone diagnostic (NOT part of libegovlp_hip.so):   hipcc --offload-arch=gfx950 -O3 -o tools/mx_probe tools/mx_probe.hip
// 1. v_mfma_scale_f32_16x16x128_f8f6f4 with FP6 (E2M3) operands: lane -> (row, 32-element k-block), bit packing of the 32 codes in
//    6 dwords, E8M0 scale byte + op_sel, zero-filled lane groups -- against a host dot product;
// 2. v_mfma_f32_16x16x32_f16 against a host dot product (fragment layout = the bf16 one);
// 3. the fp32 -> E2M3 code conversion built on the hardware fp32 -> E4M3 converter (v_cvt_pk_fp8_f32 of y * 2^-6: the low
//    binades of E4M3 have exactly E2M3's grid), against a host round-to-nearest-even quantiser;
// 4. issue rate of the product schemes per 16x16 output fragment and 32 k-elements: 3 x bf16 (today's bf16x3), 1 x bf16,
//    fp16 + fp6 (K = 128 instruction, half of its k-blocks used), fp16 + fp8, and the 32x32 forms (2 x f16 32x32x16 + fp6 32x32x64).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static float e2m3_value(int code) {
  const int s = code >> 5, e = (code >> 3) & 3, m = code & 7;
  const float v = e == 0 ? m * 0.125f : (1.0f + m * 0.125f) * (float)(1 << (e - 1));
  return s ? -v : v;
}
static int e2m3_rne(float y) {     // host reference quantiser, saturating at 7.5
  float best = 1e30f; int bc = 0;
  for (int c = 0; c < 32; ++c) {
    const float d = fabsf(fabsf(y) - e2m3_value(c));
    if (d < best || (d == best && (c & 1) == 0 && (bc & 1) == 1)) { best = d; bc = c; }
  }
  return bc | (y < 0 ? 32 : 0);
}

// ---- 1. semantic probe of the scaled MFMA --------------------------------------------------------------------------------------
// codesA [16][128] bytes (6-bit codes), codesB [16][128], scaleA [16][4] bytes (E8M0), scaleB [16][4].  groups: bit mask of the
// lane groups (lane >> 4) that carry data; the others hold zeros.  out [16][16]: D as the C/D map col = lane & 15, row = 4 (lane >> 4) + r.
template <int OPSEL>
__global__ void mx_semantic(const uint8_t* codesA, const uint8_t* codesB, const uint8_t* scaleA, const uint8_t* scaleB, int groups,
                            float* out) {
  const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
  unsigned a[8] = {0, 0, 0, 0, 0, 0, 0, 0}, b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if ((groups >> g) & 1) {
    for (int i = 0; i < 32; ++i) {
      const unsigned ca = codesA[r * 128 + g * 32 + i] & 63u, cb = codesB[r * 128 + g * 32 + i] & 63u;
      const int bit = 6 * i, w = bit >> 5, o = bit & 31;
      a[w] |= ca << o; b[w] |= cb << o;
      if (o > 26) { a[w + 1] |= ca >> (32 - o); b[w + 1] |= cb >> (32 - o); }
    }
  }
  // the scale byte of this lane's (row, k-block) in byte OPSEL of the scale register, garbage in the other bytes
  const unsigned sa = ((unsigned)scaleA[r * 4 + g] << (8 * OPSEL)) | (0x7b7b7b7bu & ~(0xffu << (8 * OPSEL)));
  const unsigned sb = ((unsigned)scaleB[r * 4 + g] << (8 * OPSEL)) | (0x7b7b7b7bu & ~(0xffu << (8 * OPSEL)));
  const i32x8 av = {(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)a[4], (int)a[5], 0, 0};
  const i32x8 bv = {(int)b[0], (int)b[1], (int)b[2], (int)b[3], (int)b[4], (int)b[5], 0, 0};
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 2, 2, OPSEL, (int)sa, OPSEL, (int)sb);
  for (int q = 0; q < 4; ++q) out[(4 * g + q) * 16 + r] = c[q];
}

// ---- 2. f16 MFMA ---------------------------------------------------------------------------------------------------------------
__global__ void f16_semantic(const _Float16* A, const _Float16* B, float* out) {   // A [16][32], B [16][32] (rows = output index)
  const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = A[r * 32 + g * 8 + i]; b[i] = B[r * 32 + g * 8 + i]; }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  for (int q = 0; q < 4; ++q) out[(4 * g + q) * 16 + r] = c[q];
}

// ---- 3. fp32 -> E2M3 code through the E4M3 converter -----------------------------------------------------------------------------
__device__ __forceinline__ unsigned e2m3x4_from_f32(float y0, float y1, float y2, float y3, float pre) {
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(y0 * pre, y1 * pre, w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(y2 * pre, y3 * pre, w, true);
  const unsigned u = (unsigned)w;
  return (u & 0x1f1f1f1fu) | ((u >> 2) & 0x20202020u);     // s 0 0 e e m m m -> s e e m m m, one code per byte
}
__global__ void cvt_probe(const float* y, int n, float pre, uint8_t* codes) {
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    const unsigned c = e2m3x4_from_f32(y[i], y[i + 1], y[i + 2], y[i + 3], pre);
    *(unsigned*)(codes + i) = c;
  }
}

// ---- 4. issue rate ---------------------------------------------------------------------------------------------------------------
// One workgroup per CU, 8 waves, MI x NJ accumulator fragments per wave as in gemm_big (5 x 8); operands stay in registers.
// `units`: the work of one "k-tile" = 32 k-elements for every fragment.
enum { V_BF16X3 = 0, V_BF16 = 1, V_F16_FP6 = 2, V_F16_FP8 = 3, V_F16 = 4, V_FP6_ONLY = 5, V_32_F16_FP6 = 6, V_32_BF16X3 = 7 };
template <int VAR>
__global__ __launch_bounds__(512, 2) void rate_kernel(const unsigned* seed, int iters, float* out) {
  constexpr int MI = 5, NJ = 8;
  const int lane = threadIdx.x & 63;
  unsigned s = seed[threadIdx.x & 255];
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
  if constexpr (VAR < V_32_F16_FP6) {
    f32x4 acc[MI][NJ];
    for (int i = 0; i < MI; ++i) for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // operand registers: small finite numbers in every format (exponent bits kept mid-range)
    bf16x8 ah[MI], al[MI], bh[2], bl[2];
    i32x8 a6[MI], b6[2];
    for (int i = 0; i < MI; ++i) {
      unsigned t[4];
      for (int q = 0; q < 4; ++q) t[q] = (rnd() & 0x83ff83ffu) | 0x38003800u;
      ah[i] = __builtin_bit_cast(bf16x8, *(uint4*)t);
      for (int q = 0; q < 4; ++q) t[q] = (rnd() & 0x83ff83ffu) | 0x38003800u;
      al[i] = __builtin_bit_cast(bf16x8, *(uint4*)t);
      for (int q = 0; q < 8; ++q) a6[i][q] = (int)(rnd() & 0x77777777u);
      if (lane >= 32) for (int q = 0; q < 8; ++q) a6[i][q] = 0;
    }
    for (int i = 0; i < 2; ++i) {
      unsigned t[4];
      for (int q = 0; q < 4; ++q) t[q] = (rnd() & 0x83ff83ffu) | 0x38003800u;
      bh[i] = __builtin_bit_cast(bf16x8, *(uint4*)t);
      for (int q = 0; q < 4; ++q) t[q] = (rnd() & 0x83ff83ffu) | 0x38003800u;
      bl[i] = __builtin_bit_cast(bf16x8, *(uint4*)t);
      for (int q = 0; q < 8; ++q) b6[i][q] = (int)(rnd() & 0x77777777u);
      if (lane >= 32) for (int q = 0; q < 8; ++q) b6[i][q] = 0;
    }
    const int sc = 0x7f7f7f7f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          if constexpr (VAR == V_BF16X3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j & 1], al[i], acc[i][j], 0, 0, 0);
          } else if constexpr (VAR == V_BF16) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j & 1], ah[i], acc[i][j], 0, 0, 0);
          } else if constexpr (VAR == V_F16_FP6 || VAR == V_F16_FP8 || VAR == V_F16) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bh[j & 1]), __builtin_bit_cast(f16x8, ah[i]),
                                                               acc[i][j], 0, 0, 0);
          }
        }
        if constexpr (VAR == V_BF16X3) {
#pragma unroll
          for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j & 1], ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j & 1], ah[i], acc[i][j], 0, 0, 0);
        } else if constexpr (VAR == V_F16_FP6 || VAR == V_FP6_ONLY) {
#pragma unroll
          for (int i = 0; i < MI; ++i)
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b6[j & 1], a6[i], acc[i][j], 2, 2, 0, sc, 0, sc);
        } else if constexpr (VAR == V_F16_FP8) {
#pragma unroll
          for (int i = 0; i < MI; ++i)
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b6[j & 1], a6[i], acc[i][j], 0, 0, 0, sc, 0, sc);
        }
      }
    }
    float r = 0.f;
    for (int i = 0; i < MI; ++i) for (int j = 0; j < NJ; ++j) r += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (r == 123.456f) out[0] = r;
  } else {
    // 32x32 fragments: the same 80 x 128 wave tile is not a whole number of them; use 2 x 4 fragments of 32x32 (64 x 128, MF = 4's tile)
    constexpr int M2 = 2, N2 = 4;
    f32x16 acc[M2][N2];
    for (int i = 0; i < M2; ++i) for (int j = 0; j < N2; ++j) for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    bf16x8 ah[M2][2], bh[2][2], al[M2][2], bl[2][2];
    i32x8 a6[M2], b6[2];
    for (int i = 0; i < M2; ++i) {
      for (int k = 0; k < 2; ++k) {
        unsigned t[4];
        for (int q = 0; q < 4; ++q) t[q] = (rnd() & 0x83ff83ffu) | 0x38003800u;
        ah[i][k] = __builtin_bit_cast(bf16x8, *(uint4*)t);
        for (int q = 0; q < 4; ++q) t[q] = (rnd() & 0x83ff83ffu) | 0x38003800u;
        al[i][k] = __builtin_bit_cast(bf16x8, *(uint4*)t);
      }
      for (int q = 0; q < 8; ++q) a6[i][q] = (int)(rnd() & 0x77777777u);
    }
    for (int i = 0; i < 2; ++i) {
      for (int k = 0; k < 2; ++k) {
        unsigned t[4];
        for (int q = 0; q < 4; ++q) t[q] = (rnd() & 0x83ff83ffu) | 0x38003800u;
        bh[i][k] = __builtin_bit_cast(bf16x8, *(uint4*)t);
        for (int q = 0; q < 4; ++q) t[q] = (rnd() & 0x83ff83ffu) | 0x38003800u;
        bl[i][k] = __builtin_bit_cast(bf16x8, *(uint4*)t);
      }
      for (int q = 0; q < 8; ++q) b6[i][q] = (int)(rnd() & 0x77777777u);
    }
    const int sc = 0x7f7f7f7f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < N2; ++j) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int i = 0; i < M2; ++i) {
            if constexpr (VAR == V_32_F16_FP6)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bh[j & 1][k]),
                                                                 __builtin_bit_cast(f16x8, ah[i][k]), acc[i][j], 0, 0, 0);
            else {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j & 1][k], al[i][k], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[j & 1][k], ah[i][k], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j & 1][k], ah[i][k], acc[i][j], 0, 0, 0);
            }
          }
        if constexpr (VAR == V_32_F16_FP6) {
#pragma unroll
          for (int i = 0; i < M2; ++i)
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b6[j & 1], a6[i], acc[i][j], 2, 2, 0, sc, 0, sc);
        }
      }
    }
    float r = 0.f;
    for (int i = 0; i < M2; ++i) for (int j = 0; j < N2; ++j) for (int q = 0; q < 16; ++q) r += acc[i][j][q];
    if (r == 123.456f) out[0] = r;
  }
}

template <int VAR>
static void run_rate(const char* name, const unsigned* seed, float* out, double frag_units_per_iter, double flops_per_iter_wave) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(rate_kernel<VAR>, dim3(256), dim3(512), 0, 0, seed, 200, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(rate_kernel<VAR>, dim3(256), dim3(512), 0, 0, seed, iters, out);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  // one wave-iteration = frag_units_per_iter "16x16 fragment x 32 k" units of the product scheme
  const double ns_per_unit = ms * 1e6 / iters / frag_units_per_iter;
  const double tf = flops_per_iter_wave * iters * 8.0 * 256.0 / (ms * 1e-3) * 1e-12;
  printf("rate %-34s %8.3f ms  %7.2f ns per (16x16 fragment x 32 k) per wave   algorithmic %7.1f TF\n", name, ms, ns_per_unit, tf);
}

int main() {
  // ---- 1 --------------------------------------------------------------------------------------------------------------------
  std::vector<uint8_t> cA(16 * 128), cB(16 * 128), sA(64), sB(64);
  srand(7);
  for (auto& c : cA) c = rand() & 63;
  for (auto& c : cB) c = rand() & 63;
  for (auto& s : sA) s = 127 - 3 + rand() % 7;
  for (auto& s : sB) s = 127 - 3 + rand() % 7;
  uint8_t *dA, *dB, *dsA, *dsB; float* dout;
  CK(hipMalloc(&dA, cA.size())); CK(hipMalloc(&dB, cB.size())); CK(hipMalloc(&dsA, 64)); CK(hipMalloc(&dsB, 64)); CK(hipMalloc(&dout, 1024));
  CK(hipMemcpy(dA, cA.data(), cA.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dB, cB.data(), cB.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dsA, sA.data(), 64, hipMemcpyHostToDevice)); CK(hipMemcpy(dsB, sB.data(), 64, hipMemcpyHostToDevice));
  for (int opsel = 0; opsel < 2; ++opsel)
    for (int groups : {15, 3, 1}) {
      if (opsel == 0) hipLaunchKernelGGL(mx_semantic<0>, dim3(1), dim3(64), 0, 0, dA, dB, dsA, dsB, groups, dout);
      else hipLaunchKernelGGL(mx_semantic<2>, dim3(1), dim3(64), 0, 0, dA, dB, dsA, dsB, groups, dout);
      std::vector<float> got(256);
      CK(hipMemcpy(got.data(), dout, 1024, hipMemcpyDeviceToHost));
      double err_ab = 0, err_ba = 0, ref2 = 0;
      for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
          double s = 0;     // D[i][j] = sum_k A[i][k] B[j][k] with `a` (first operand) rows = i
          for (int g = 0; g < 4; ++g) {
            if (!((groups >> g) & 1)) continue;
            double sg = 0;
            for (int k = 0; k < 32; ++k) sg += (double)e2m3_value(cA[i * 128 + g * 32 + k]) * e2m3_value(cB[j * 128 + g * 32 + k]);
            s += sg * ldexp(1.0, sA[i * 4 + g] - 127) * ldexp(1.0, sB[j * 4 + g] - 127);
          }
          err_ab += (got[i * 16 + j] - s) * (got[i * 16 + j] - s);
          err_ba += (got[j * 16 + i] - s) * (got[j * 16 + i] - s);
          ref2 += s * s;
        }
      printf("mx fp6 semantic  opsel byte %d groups 0x%x : rel err (first operand = rows) %.3e   (first operand = cols) %.3e\n",
             opsel ? 2 : 0, groups, sqrt(err_ab / ref2), sqrt(err_ba / ref2));
    }
  // ---- 2 --------------------------------------------------------------------------------------------------------------------
  {
    std::vector<_Float16> A(512), B(512);
    for (auto& v : A) v = (_Float16)((rand() % 2001 - 1000) / 512.0f);
    for (auto& v : B) v = (_Float16)((rand() % 2001 - 1000) / 512.0f);
    _Float16 *fA, *fB;
    CK(hipMalloc(&fA, 1024)); CK(hipMalloc(&fB, 1024));
    CK(hipMemcpy(fA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(fB, B.data(), 1024, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(f16_semantic, dim3(1), dim3(64), 0, 0, fA, fB, dout);
    std::vector<float> got(256);
    CK(hipMemcpy(got.data(), dout, 1024, hipMemcpyDeviceToHost));
    double e1 = 0, e2 = 0, r2 = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double s = 0;
        for (int k = 0; k < 32; ++k) s += (double)(float)A[i * 32 + k] * (double)(float)B[j * 32 + k];
        e1 += (got[i * 16 + j] - s) * (got[i * 16 + j] - s); e2 += (got[j * 16 + i] - s) * (got[j * 16 + i] - s); r2 += s * s;
      }
    printf("f16 16x16x32 semantic: rel err (first operand = rows) %.3e   (first operand = cols) %.3e\n", sqrt(e1 / r2), sqrt(e2 / r2));
  }
  // ---- 3 --------------------------------------------------------------------------------------------------------------------
  {
    const int n = 1 << 16;
    std::vector<float> y(n);
    for (int i = 0; i < n; ++i) y[i] = (i & 1 ? -1.f : 1.f) * 7.5f * (float)(i >> 1) / (float)(n / 2 - 1);
    for (int i = 0; i < 64; ++i) y[i] = ((i & 1) ? -1.f : 1.f) * (i >> 1) * 0.0625f;         // the exact ties of the low binades
    float* dy; uint8_t* dc;
    CK(hipMalloc(&dy, n * 4)); CK(hipMalloc(&dc, n));
    CK(hipMemcpy(dy, y.data(), n * 4, hipMemcpyHostToDevice));
    for (float pre : {0.015625f, 0.0078125f}) {
      hipLaunchKernelGGL(cvt_probe, dim3(n / 4 / 256), dim3(256), 0, 0, dy, n, pre, dc);
      std::vector<uint8_t> c(n);
      CK(hipMemcpy(c.data(), dc, n, hipMemcpyDeviceToHost));
      int bad = 0, first = -1;
      for (int i = 0; i < n; ++i) {
        const int want = e2m3_rne(y[i]);
        const bool same = e2m3_value(want) == e2m3_value(c[i]) || (e2m3_value(want) == 0.f && e2m3_value(c[i]) == 0.f);
        if (!same) { if (first < 0) first = i; ++bad; }
      }
      printf("fp32 -> e2m3 through cvt_pk_fp8_f32(y * %g): %d of %d values differ from host RNE", pre, bad, n);
      if (first >= 0) printf("  (first: y = %g -> code 0x%02x = %g, host 0x%02x = %g)", y[first], c[first], e2m3_value(c[first]), e2m3_rne(y[first]), e2m3_value(e2m3_rne(y[first])));
      printf("\n");
    }
  }
  // ---- 4 --------------------------------------------------------------------------------------------------------------------
  {
    std::vector<unsigned> seed(256);
    for (auto& s : seed) s = rand();
    unsigned* dseed;
    CK(hipMalloc(&dseed, 1024));
    CK(hipMemcpy(dseed, seed.data(), 1024, hipMemcpyHostToDevice));
    const double unit = 2.0 * 16 * 16 * 32;     // algorithmic flops of one (fragment x 32 k) unit
    for (int rep = 0; rep < 2; ++rep) {
      run_rate<V_BF16>("1 x bf16 16x16x32", dseed, dout, 40, 40 * unit);
      run_rate<V_BF16X3>("3 x bf16 16x16x32 (bf16x3 today)", dseed, dout, 40, 40 * unit);
      run_rate<V_F16>("1 x f16 16x16x32", dseed, dout, 40, 40 * unit);
      run_rate<V_FP6_ONLY>("1 x fp6 16x16x128 (lanes 32+ zero)", dseed, dout, 40, 40 * unit);
      run_rate<V_F16_FP6>("f16 + fp6 16x16x128 (half used)", dseed, dout, 40, 40 * unit);
      run_rate<V_F16_FP8>("f16 + fp8 16x16x128 (half used)", dseed, dout, 40, 40 * unit);
      run_rate<V_32_BF16X3>("32x32: 3 x (2 x bf16 32x32x16)", dseed, dout, 32, 32 * unit);
      run_rate<V_32_F16_FP6>("32x32: 2 x f16 32x32x16 + fp6 x64", dseed, dout, 32, 32 * unit);
    }
  }
  return 0;
}
