// Shared helpers for the gfx950 kernels of libmixstage_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "mixstage.h"
#include "mixstage_aux.h"

namespace ms {

typedef float f32x16 __attribute__((ext_vector_type(16)));

int set_error(const char* fmt, ...);

// persistent, zero-initialised arrival counters for in-launch split reductions (ms_set_counter_buffer); nullptr: the
// reductions run as separate kernels
extern int* g_counters;
extern int g_counters_n;
enum { CNT_CONV = 0, CNT_WGRAD = 1 };   // halves of the buffer: kernels of the two kinds may run concurrently
inline int* counter_region(int kind, int need) {
  const int half = g_counters_n / 2;
  return (g_counters && need <= half) ? g_counters + kind * half : nullptr;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error("%s: %s", what, hipGetErrorString(e));
  return 0;
}

// One process may drive several devices (the launchers' residency decisions and hipFuncSetAttribute results are per device):
// CU count of the CURRENT device, and "has this been done on the current device yet" for a per-call-site mask.
int current_device_cus();
int current_device_index();
inline bool first_time_on_device(unsigned long long& mask) {
  const int dev = current_device_index();
  return dev < 0 || dev > 63 || !((mask >> dev) & 1ull);
}
inline void done_on_device(unsigned long long& mask) {
  const int dev = current_device_index();
  if (dev >= 0 && dev <= 63) mask |= 1ull << dev;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ inline int cdiv_dev(int a, int b) { return (a + b - 1) / b; }

// Cross-lane sums on the VALU (DPP row rotations inside a 16-lane row, v_permlane16_swap / v_permlane32_swap across rows):
// the __shfl_xor forms go through ds_bpermute, which measured ~200 ns per exchange on MI355X (64 exchanges = 14 us in a
// conv epilogue).  Every lane ends up with the total; the order of the additions is fixed.
template <int CTRL>
__device__ inline float dpp_rot(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
// value of lane ^ 1 / lane ^ 2 (inside a quad)
__device__ inline float lane_xor1(float x) { return dpp_rot<0xB1>(x); }   // quad_perm:[1,0,3,2]
__device__ inline float lane_xor2(float x) { return dpp_rot<0x4E>(x); }   // quad_perm:[2,3,0,1]
__device__ inline float row_sum16(float v) {
  v += dpp_rot<0x128>(v);   // row_ror:8
  v += dpp_rot<0x124>(v);   // row_ror:4
  v += dpp_rot<0x122>(v);   // row_ror:2
  v += dpp_rot<0x121>(v);   // row_ror:1
  return v;
}
// sum over the 32 lanes that share (lane>>5)
__device__ inline float half_wave_sum(float v) {
  v = row_sum16(v);
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const unsigned long long sw = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_permlane16_swap(u, u, false, false));
  return __builtin_bit_cast(float, (unsigned)sw) + __builtin_bit_cast(float, (unsigned)(sw >> 32));
}

__device__ inline float wave_sum(float v) {
  v = half_wave_sum(v);
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const unsigned long long sw = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_permlane32_swap(u, u, false, false));
  return __builtin_bit_cast(float, (unsigned)sw) + __builtin_bit_cast(float, (unsigned)(sw >> 32));
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

// e / d for a wave-uniform divisor by multiply-high with floor(2^32 / d) + 1 (exact while e * d < 2^32, which `max_e` vouches
// for; otherwise, and for d == 1, the plain division): the generic 32-bit division is ~25 instructions, and the small kernels'
// address arithmetic sits in front of their first load.
struct FastDiv {
  unsigned m, d;
  bool ok;
  __device__ __forceinline__ FastDiv(int d_, int max_e)
      : m(d_ > 1 ? 0xFFFFFFFFu / (unsigned)d_ + 1u : 0u), d((unsigned)d_), ok(d_ > 1 && (unsigned long long)max_e * (unsigned)d_ < (1ull << 32)) {}
  __device__ __forceinline__ int div(int e) const { return ok ? (int)__umulhi((unsigned)e, m) : e / (int)d; }
};

// Kernel arguments are a few hundred bytes of struct; the compiler fetches the fields lazily, next to their first use, so a
// kernel start walks through 4-5 scalar-cache misses ONE AFTER THE OTHER (each a trip to L2).  This touches every 64-byte
// line of the first `BYTES` bytes of the argument block at once and waits for them once: the later s_loads hit the scalar cache.
// (Early-clobber outputs: a destination must not share a register with the base pointer the following loads still read.)
template <int BYTES>
__device__ __forceinline__ void prefetch_kernargs(int byte_offset = 0) {
  const auto ka = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + (byte_offset & ~63);
  unsigned d0, d1, d2, d3, d4, d5;
  if constexpr (BYTES > 320)
    asm volatile("s_load_dword %0, %6, 0x0\n\ts_load_dword %1, %6, 0x40\n\ts_load_dword %2, %6, 0x80\n\ts_load_dword %3, %6, 0xc0\n\t"
                 "s_load_dword %4, %6, 0x100\n\ts_load_dword %5, %6, 0x140\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4), "=&s"(d5) : "s"(ka) : "memory");
  else if constexpr (BYTES > 192)
    asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80\n\ts_load_dword %3, %4, 0xc0\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3) : "s"(ka) : "memory");
  else
    asm volatile("s_load_dword %0, %2, 0x0\n\ts_load_dword %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(d0), "=&s"(d1) : "s"(ka) : "memory");
}

// XCD-aware workgroup id: MI355X deals consecutive workgroup ids round-robin over its 8 XCDs (each with a private
// L2).  This bijective remap gives every XCD one CONTIGUOUS range of logical ids, so workgroups that share operand
// tiles (neighbouring logical ids) hit the same L2.  Speed only, never correctness.
__device__ inline int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// Split-K / split-reduction hand-off inside one launch (cdna_hip_programming.md Guideline 16, counter form): every
// workgroup of a tile stores its partial slab, then arrives on the tile's counter; the workgroup that arrives LAST
// (returns true) sees all slabs and finishes the tile.  It sums the slabs in slice order, so the result does not depend
// on the arrival order (bitwise reproducible).  The last arriver resets the counter for the next launch.
// `s_flag` is one word of LDS that is no longer in use.
__device__ inline bool splitk_arrive_last(int* counter, int expected, int* s_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's slab stores have left
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // write back this XCD's L2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int prev = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = prev == expected - 1;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");    // drop this CU's stale L1 lines
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    *s_flag = last;
  }
  __syncthreads();
  return *s_flag != 0;
}

// Staging loads are raw buffer loads: one 32-bit byte offset per element against a buffer descriptor in SGPRs (the chunk
// base travels in the scalar offset), and padding / out-of-range elements use an offset past the descriptor's range, for
// which the hardware returns 0 -- no pointer selects, no branches, no 64-bit address arithmetic in the K loop.
constexpr unsigned BUF_OOB = 0x80000000u;
__device__ inline __amdgpu_buffer_rsrc_t buf_rsrc(const void* ptr) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, 0x7fffffff, 0x00020000);
}
__device__ inline float buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ inline float4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  // (bit_cast of the whole vector: element-wise access of the builtin's result is miscompiled to a dword load by this clang)
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

// Running BatchNorm statistics (layers.py:60-66, momentum 0.1): a batch whose statistics are not finite -- the output of a launch
// whose in-launch meeting timed out is poisoned with NaN, and every block downstream of it sees NaN -- must not reach the running
// buffers, which outlive the step (the optimizer skips such a step too: adam_prep*).  Bit-identical for finite statistics.
__device__ inline void running_stats_update(float* rm, float* rv, float rm_old, float rv_old, float momentum, float mean, float unbiased) {
  const float nm = (1.f - momentum) * rm_old + momentum * mean;
  const float nv = (1.f - momentum) * rv_old + momentum * unbiased;
  const bool ok = (fabsf(nm) <= 3.0e38f) && (fabsf(nv) <= 3.0e38f);
  *rm = ok ? nm : rm_old;
  *rv = ok ? nv : rv_old;
}

__device__ inline float2 buf_load2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
}

__device__ inline float lrelu(float z, float slope) { return z > 0.f ? z : z * slope; }

}  // namespace ms
