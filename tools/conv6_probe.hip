// Feasibility probe (not part of the library): 1-D k3 s1 grouped conv on the bf16 matrix pipe with both fp32 operands split
// EXACTLY into three bf16 parts (truncation: x = hi + mid + lo, 3 x 8 mantissa bits) and 6 of the 9 cross products kept
// (dropped: mid*lo, lo*mid, lo*lo <= 2^-24 relative) -- fp32-equivalent accuracy at 16/6 of the fp32 MFMA rate.
//   hipcc -O3 --offload-arch=gfx950 tools/conv6_probe.hip -o /tmp/conv6 && /tmp/conv6
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned BUF_OOB = 0x80000000u;
__device__ inline __amdgpu_buffer_rsrc_t buf_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ inline float buf_load(__amdgpu_buffer_rsrc_t r, unsigned v, unsigned s) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)v, (int)s, 0));
}
__device__ inline u32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned v, unsigned s) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)v, (int)s, 0));
}

// ---- weights: w[g*M + m][ci][tap] fp32 -> planes[3][g][m][chunk][tap][16 ci] bf16 (exact 3-way split)
__global__ void split_weights(const float* __restrict__ w, unsigned short* __restrict__ planes, int rows, int Cin, int KHW) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = rows * Cin * KHW;
  if (i >= total) return;
  const int tapci = i % (Cin * KHW), row = i / (Cin * KHW);
  const int chunk = tapci / (16 * KHW), rem = tapci % (16 * KHW), tap = rem / 16, cil = rem % 16;
  const float x = w[(size_t)row * Cin * KHW + (chunk * 16 + cil) * KHW + tap];
  const unsigned xb = __float_as_uint(x);
  const float hi = __uint_as_float(xb & 0xffff0000u), r1 = x - hi;
  const float mid = __uint_as_float(__float_as_uint(r1) & 0xffff0000u), r2 = r1 - mid;
  planes[(size_t)0 * total + i] = (unsigned short)(xb >> 16);
  planes[(size_t)1 * total + i] = (unsigned short)(__float_as_uint(r1) >> 16);
  planes[(size_t)2 * total + i] = (unsigned short)(__float_as_uint(r2) >> 16);
}

// ---- conv: workgroup = 64 output channels x 128 pixels (2 batch rows of 64 steps); 4 waves, each 64 x 32
constexpr int KW = 3, CK = 16, BM = 64, TW = 64, TH = 2, PC = TW + KW - 1, RP = PC, CP = TH * RP + 1, PA = 56;
constexpr int A_STAGE = 3 * BM * PA;        // bf16 elements
constexpr int P_STAGE = CK * CP + 4;        // floats
constexpr int NPE = CK * TH * PC, NP = (NPE + 255) / 256;
constexpr int NAV = 3 * BM * (CK * KW / 8), NA = (NAV + 255) / 256;   // 16-byte slots of the three A planes

__device__ inline void split3(const float (&x)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float x0 = x[2 * j], x1 = x[2 * j + 1];
    const unsigned b0 = __float_as_uint(x0), b1 = __float_as_uint(x1);
    const float r0 = x0 - __uint_as_float(b0 & 0xffff0000u), r1 = x1 - __uint_as_float(b1 & 0xffff0000u);
    const unsigned c0 = __float_as_uint(r0), c1 = __float_as_uint(r1);
    const float s0 = r0 - __uint_as_float(c0 & 0xffff0000u), s1 = r1 - __uint_as_float(c1 & 0xffff0000u);
    h[j] = __builtin_amdgcn_perm(b1, b0, 0x07060302);
    m[j] = __builtin_amdgcn_perm(c1, c0, 0x07060302);
    l[j] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302);
  }
  hi = __builtin_bit_cast(bf16x8, u32x4{h[0], h[1], h[2], h[3]});
  mid = __builtin_bit_cast(bf16x8, u32x4{m[0], m[1], m[2], m[3]});
  lo = __builtin_bit_cast(bf16x8, u32x4{l[0], l[1], l[2], l[3]});
}

__global__ __launch_bounds__(256, 2) void conv6_kernel(const float* __restrict__ x, const unsigned short* __restrict__ planes,
                                                        const float* __restrict__ bias, float* __restrict__ y, int B, int G,
                                                        int Cin, int Cout, int T) {
  __shared__ __attribute__((aligned(16))) unsigned short sA[2 * A_STAGE];
  __shared__ float sP[2 * P_STAGE];
  const int t = threadIdx.x, lane = t & 63, wn = t >> 6, kb = lane >> 5;
  const int mtiles = Cout / BM;
  const int by_ = blockIdx.x % mtiles, bx_ = (blockIdx.x / mtiles) % (B / TH), g = blockIdx.x / (mtiles * (B / TH));
  const int m0 = by_ * BM, row0 = bx_ * TH;
  const int nchunks = Cin / CK, Ktot = Cin * KW;
  const size_t plane_elems = (size_t)G * Cout * Ktot;

  // chunk-invariant staging offsets
  unsigned goff[NP];
  int loff[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int e = t + i * 256;
    const int ci = e / (TH * PC), rem = e - ci * (TH * PC), r = rem / PC, c = rem - r * PC;
    const int ix = c - 1;
    const bool ok = (e < NPE) & ((unsigned)ix < (unsigned)T);
    goff[i] = ok ? 4u * (unsigned)(ci * T + (row0 + r) * (G * Cin * T) + ix) : BUF_OOB;
    loff[i] = e < NPE ? ci * CP + r * RP + c : CK * CP;
  }
  unsigned aoff[NA];
  int alds[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int idx = t + i * 256;                 // (plane, row, 16-byte slot of the 48-element chunk row)
    const int pl = idx / (BM * 6), rem = idx - pl * (BM * 6), row = rem / 6, q = rem - row * 6;
    aoff[i] = idx < NAV ? 2u * (unsigned)(pl * plane_elems + (size_t)(g * Cout + m0 + row) * Ktot + q * 8) : BUF_OOB;
    alds[i] = idx < NAV ? (pl * BM + row) * PA + q * 8 : 3 * BM * PA - 8;
  }
  const __amdgpu_buffer_rsrc_t rsA = buf_rsrc(planes), rsX = buf_rsrc(x);

  u32x4 ra0[NA], ra1[NA];
  float rb0[NP], rb1[NP];
  auto load_chunk = [&](int ch, u32x4 (&ra)[NA], float (&rb)[NP]) {
    const unsigned sa = __builtin_amdgcn_readfirstlane(2u * (unsigned)(ch * CK * KW));
#pragma unroll
    for (int i = 0; i < NA; ++i) ra[i] = buf_load4(rsA, aoff[i], sa);
    const unsigned sx = __builtin_amdgcn_readfirstlane(4u * (unsigned)((g * Cin + ch * CK) * T));
#pragma unroll
    for (int i = 0; i < NP; ++i) rb[i] = buf_load(rsX, goff[i], sx);
  };
  auto store_chunk = [&](int buf, const u32x4 (&ra)[NA], const float (&rb)[NP]) {
    unsigned short* As = sA + buf * A_STAGE;
    float* Ps = sP + buf * P_STAGE;
#pragma unroll
    for (int i = 0; i < NA; ++i) *reinterpret_cast<u32x4*>(As + alds[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < NP; ++i) Ps[loff[i]] = rb[i];
  };

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int nloc = wn * 32 + (lane & 31);
  const int ty = nloc / TW, tx = nloc - ty * TW;
  const int b_base = (kb * 8) * CP + ty * RP + tx;          // + j*CP + tap
  const int a_base = (lane & 31) * PA + kb * 8;             // + (plane*BM + mi*32)*PA + tap*16

  auto compute_chunk = [&](int buf) {
    const unsigned short* As = sA + buf * A_STAGE;
    const float* Ps = sP + buf * P_STAGE;
#pragma unroll
    for (int tap = 0; tap < KW; ++tap) {
      float xb[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) xb[j] = Ps[b_base + j * CP + tap];
      bf16x8 bh, bm, bl;
      split3(xb, bh, bm, bl);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(As + a_base + (0 * BM + mi * 32) * PA + tap * 16);
        const bf16x8 am = *reinterpret_cast<const bf16x8*>(As + a_base + (1 * BM + mi * 32) * PA + tap * 16);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(As + a_base + (2 * BM + mi * 32) * PA + tap * 16);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[mi], 0, 0, 0);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[mi], 0, 0, 0);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[mi], 0, 0, 0);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[mi], 0, 0, 0);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[mi], 0, 0, 0);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[mi], 0, 0, 0);
      }
    }
  };

  load_chunk(0, ra0, rb0);
  if (nchunks > 1) load_chunk(1, ra1, rb1);
  store_chunk(0, ra0, rb0);
  __syncthreads();
  for (int ch = 0; ch < nchunks; ch += 2) {
    if (ch + 2 < nchunks) load_chunk(ch + 2, ra0, rb0);
    compute_chunk(0);
    if (ch + 1 < nchunks) store_chunk(1, ra1, rb1);
    __syncthreads();
    if (ch + 1 >= nchunks) break;
    if (ch + 3 < nchunks) load_chunk(ch + 3, ra1, rb1);
    compute_chunk(1);
    if (ch + 2 < nchunks) store_chunk(0, ra0, rb0);
    __syncthreads();
  }

  const int ob = row0 + ty;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb;
      const int chn = g * Cout + m;
      y[((size_t)ob * G * Cout + chn) * T + tx] = acc[mi][r] + bias[chn];
    }
}

int main() {
  const int B = 32, G = 8, Cin = 256, Cout = 256, T = 64, KHW = 3;
  const size_t nx = (size_t)B * G * Cin * T, nw = (size_t)G * Cout * Cin * KHW, ny = (size_t)B * G * Cout * T;
  std::vector<float> hx(nx), hw(nw), hb(G * Cout), hy(ny);
  srand(1);
  for (auto& v : hx) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  for (auto& v : hw) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f;
  for (auto& v : hb) v = (float)rand() / RAND_MAX - 0.5f;
  float *dx, *dw, *db, *dy;
  unsigned short* dp;
  hipMalloc(&dx, nx * 4); hipMalloc(&dw, nw * 4); hipMalloc(&db, hb.size() * 4); hipMalloc(&dy, ny * 4); hipMalloc(&dp, 3 * nw * 2 + 64);
  hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice);
  hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(split_weights, dim3((nw + 255) / 256), dim3(256), 0, 0, dw, dp, G * Cout, Cin, KHW);
  const int grid = (Cout / BM) * (B / TH) * G;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(conv6_kernel, dim3(grid), dim3(256), 0, 0, dx, dp, db, dy, B, G, Cin, Cout, T);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int iters = 20;
  for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(conv6_kernel, dim3(grid), dim3(256), 0, 0, dx, dp, db, dy, B, G, Cin, Cout, T);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("hip error: %s\n", hipGetErrorString(hipGetLastError()));
  hipMemcpy(hy.data(), dy, ny * 4, hipMemcpyDeviceToHost);
  const double flops = 2.0 * B * T * (double)G * Cout * Cin * KHW;
  printf("conv6: %.1f us per launch, %.1f TF (useful flops), %d workgroups\n", ms / iters * 1e3, flops / (ms / iters * 1e-3) / 1e12, grid);
  double maxerr = 0, maxref = 0;
  for (int s = 0; s < 3000; ++s) {
    const int b = rand() % B, c = rand() % (G * Cout), tt = rand() % T, g = c / Cout;
    double ref = hb[c];
    for (int ci = 0; ci < Cin; ++ci)
      for (int k = 0; k < 3; ++k) {
        const int ix = tt + k - 1;
        if (ix < 0 || ix >= T) continue;
        ref += (double)hw[((size_t)c * Cin + ci) * 3 + k] * (double)hx[((size_t)b * G * Cin + g * Cin + ci) * T + ix];
      }
    maxerr = fmax(maxerr, fabs(ref - hy[((size_t)b * G * Cout + c) * T + tt]));
    maxref = fmax(maxref, fabs(ref));
  }
  printf("max |err| vs fp64 over 3000 samples: %.3e (max |ref| %.3f)\n", maxerr, maxref);
  return 0;
}
