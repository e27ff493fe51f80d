// Issue rates of the candidate forward products on one MI355X (standalone; hipcc --offload-arch=gfx950 tools/f8_probe.hip -o tools/f8_probe):
// per (16x16 fragment pair, 32-deep k-step): three bf16 MFMAs (bf16x3) against one fp16 MFMA + two PLAIN fp8 (e5m2) MFMAs 16x16x32
// (the "f16b8" product of DESIGN 4.1), and a semantics check of the fp8 instruction (operand layout = 8 consecutive k per lane group,
// as the bf16 / fp16 forms; element 0 in the low byte).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(int iters, float* out) {
  f32x4_t acc[40];
  for (int i = 0; i < 40; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const float s = 1.0f + threadIdx.x * 1e-3f;
  bf16x8_t a, b;
  f16x8_t ah, bh;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(s + e); b[e] = (__bf16)(s - e); ah[e] = (_Float16)(s + e); bh[e] = (_Float16)(s - e); }
  long a8 = 0x3c3c3c3c3c3c3c3cL + threadIdx.x, b8 = 0x3838383838383838L + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 40; ++i) {
      if (MODE == 0) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
      } else if (MODE == 1) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8(a8, b8, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8(b8, a8, acc[i], 0, 0, 0);
      } else if (MODE == 2) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
      } else {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8(a8, b8, acc[i], 0, 0, 0);
      }
    }
  }
  float r = 0.f;
  for (int i = 0; i < 40; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

// semantics: D = A B^T with A [16 rows][32 k], B [16 cols][32 k] as e5m2 bytes
__global__ void sem_kernel(const unsigned char* A, const unsigned char* B, float* D) {
  const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
  long a = 0, b = 0;
  for (int e = 0; e < 8; ++e) {
    a |= (long)A[r * 32 + 8 * g + e] << (8 * e);
    b |= (long)B[r * 32 + 8 * g + e] << (8 * e);
  }
  f32x4_t c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8(a, b, c, 0, 0, 0);
  for (int j = 0; j < 4; ++j) D[(4 * g + j) * 16 + r] = c[j];       // D[row m = 4g + j][col n = lane & 15] with A supplying m
}

static float e5m2(unsigned char x) {
  const int s = x >> 7, e = (x >> 2) & 31, m = x & 3;
  float v = e == 0 ? ldexpf(m / 4.0f, -14) : ldexpf(1.0f + m / 4.0f, e - 15);
  return s ? -v : v;
}

int main() {
  float* out;
  hipMalloc(&out, 4 * 256 * 256 * 4);
  const char* names[4] = {"3 x bf16 (bf16x3)", "fp16 + 2 x bf8 (f16b8)", "1 x bf16", "1 x bf8"};
  for (int mode = 0; mode < 4; ++mode) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    auto launch = [&](int it) {
      if (mode == 0) rate_kernel<0><<<256 * 2, 256>>>(it, out);
      else if (mode == 1) rate_kernel<1><<<256 * 2, 256>>>(it, out);
      else if (mode == 2) rate_kernel<2><<<256 * 2, 256>>>(it, out);
      else rate_kernel<3><<<256 * 2, 256>>>(it, out);
    };
    launch(10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // 2 workgroups per CU x 4 waves = 2 waves per SIMD; per wave iters x 40 fragment-k-steps
    printf("%-26s %8.2f ns per fragment-k-step per SIMD (2 waves interleaved)\n", names[mode], ms * 1e6 / (iters * 40.0 * 2));
  }
  // semantics
  std::vector<unsigned char> A(16 * 32), B(16 * 32);
  srand(3);
  for (auto& x : A) { x = rand() & 0xff; if (((x >> 2) & 31) == 31) x &= 0x83 | (30 << 2); }
  for (auto& x : B) { x = rand() & 0xff; if (((x >> 2) & 31) == 31) x &= 0x83 | (30 << 2); }
  for (auto& x : A) if (((x >> 2) & 31) > 20) x = (x & 0x83) | (17 << 2);
  for (auto& x : B) if (((x >> 2) & 31) > 20) x = (x & 0x83) | (16 << 2);
  unsigned char *dA, *dB; float* dD;
  hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 1024);
  hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
  sem_kernel<<<1, 64>>>(dA, dB, dD);
  std::vector<float> D(256);
  hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int m = 0; m < 16; ++m)
    for (int n = 0; n < 16; ++n) {
      double ref = 0;
      for (int k = 0; k < 32; ++k) ref += (double)e5m2(A[m * 32 + k]) * e5m2(B[n * 32 + k]);
      worst = fmax(worst, fabs(ref - D[m * 16 + n]) / (fabs(ref) + 1e-30));
    }
  printf("bf8 16x16x32 semantics (A rows -> D rows, 8 consecutive k per lane group, element 0 = low byte): max rel err %.2e\n", worst);
  return 0;
}
