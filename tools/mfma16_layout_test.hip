#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float* A, const float* B, float* C, int K) {
  const int lane = threadIdx.x, row = lane & 31, kb = lane >> 5;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
      a[j] = (__bf16)A[row * K + k0 + kb * 8 + j];
      b[j] = (__bf16)B[(k0 + kb * 8 + j) * 32 + row];
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) {
    const int m = (r & 3) + 8 * (r >> 2) + 4 * kb;
    C[m * 32 + row] = acc[r];
  }
}
int main() {
  const int K = 64;
  std::vector<float> A(32 * K), B(K * 32), C(32 * 32), R(32 * 32, 0.f);
  for (auto& v : A) v = (float)((rand() % 17) - 8) / 8.f;
  for (auto& v : B) v = (float)((rand() % 17) - 8) / 4.f;
  for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { float s = 0; for (int kk = 0; kk < K; ++kk) s += A[m * K + kk] * B[kk * 32 + n]; R[m * 32 + n] = s; }
  float *dA, *dB, *dC;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
  hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
  double err = 0; for (int i = 0; i < 1024; ++i) err = fmax(err, fabs(C[i] - R[i]));
  printf("max err %g (%s)\n", err, err == 0 ? "layout OK" : "LAYOUT MISMATCH");
  return 0;
}
