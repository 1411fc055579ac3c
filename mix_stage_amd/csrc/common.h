// Shared helpers for the gfx950 kernels of libmixstage_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "mixstage.h"

namespace ms {

typedef float f32x16 __attribute__((ext_vector_type(16)));

int set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error("%s: %s", what, hipGetErrorString(e));
  return 0;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// wave-level sum over the 32 lanes that share (lane>>5); xor masks < 32 never cross the halves
__device__ inline float half_wave_sum(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 1);
  return v;
}

__device__ inline float wave_sum(float v) {
  v += __shfl_xor(v, 32);
  return half_wave_sum(v);
}

__device__ inline double wave_sum_d(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// block-wide sum for 256-thread blocks; result valid in every thread. `red` = 4 floats of LDS.
__device__ inline float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// XCD-aware workgroup id: MI355X deals consecutive workgroup ids round-robin over its 8 XCDs (each with a private
// L2).  This bijective remap gives every XCD one CONTIGUOUS range of logical ids, so workgroups that share operand
// tiles (neighbouring logical ids) hit the same L2.  Speed only, never correctness.
__device__ inline int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

__device__ inline float lrelu(float z, float slope) { return z > 0.f ? z : z * slope; }

}  // namespace ms
