// Single-input-channel 3x3 convolution (the AudioEncoder's first block: 1 -> 64 channels over the (time, mel) plane).
// K = 9 is far too short for the matrix pipe (the patch kernel spends its time in per-tile fixed costs: 55 us for 151 MFLOP);
// here one thread computes one output pixel for all 64 channels on the VALU (weights come in through scalar loads) and
// the launch is bound by writing the 67 MB output.  BN statistics: each wave passes its 64 x 64 tile through LDS and one
// lane per channel accumulates (sum, M2 about the tile mean) in a fixed order -- same partials as the patch kernels.
// Rows of W % 4 == 0 pixels (the path's: 128 mel bins) take conv_c1_3x3_v4_kernel: a thread owns four consecutive pixels (16-byte
// stores), a workgroup's 1024 pixels are one statistics tile (4x fewer partials for bn_finalize), both statistics passes are
// constant-trip LDS reductions; the conv is recomputed for the second pass instead of keeping 256 values per thread.
#include <stdint.h>
#include <algorithm>
#include <stdlib.h>

#include "kernels.h"

namespace ms {

template <int CO>
__global__ __launch_bounds__(256) void conv_c1_3x3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          const float* __restrict__ bn_g, const float* __restrict__ bn_b,
                                                          const float* __restrict__ bn_m, const float* __restrict__ bn_v,
                                                          float* __restrict__ stats, float* __restrict__ counts, int B, int H,
                                                          int W, int ep, float slope, float eps) {
  __shared__ float tile[4][CO * 65];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int npix = B * H * W, hw = H * W;
  const int pix = blockIdx.x * 256 + t;
  const bool pv = pix < npix;
  const int b = pv ? pix / hw : 0, rem = pix - b * hw, oy = rem / W, ox = rem - oy * W;
  float xin[9];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int iy = oy + kh - 1, ix = ox + kw - 1;
      const bool ok = pv & ((unsigned)iy < (unsigned)H) & ((unsigned)ix < (unsigned)W);
      xin[kh * 3 + kw] = ok ? x[(size_t)b * hw + iy * W + ix] : 0.f;
    }
  float v[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) {
    float a = bias ? bias[c] : 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) a = fmaf(w[c * 9 + k], xin[k], a);       // uniform addresses: scalar loads
    v[c] = a;
  }
  float* op = out + (size_t)b * CO * hw + rem;
#pragma unroll
  for (int c = 0; c < CO; ++c) {
    float o = v[c];
    if (ep == EP_BN_EVAL) {
      const float sc = bn_g[c] * (1.0f / sqrtf(bn_v[c] + eps));
      o = lrelu(fmaf(o, sc, bn_b[c] - bn_m[c] * sc), slope);
    }
    if (ep == EP_LRELU) o = lrelu(o, slope);
    if (pv) op[(size_t)c * hw] = o;
  }
  if (ep != EP_RAW_STATS) return;
  // per-wave tile statistics: [channel][pixel] through LDS (pitch 65: conflict-free both ways), lane c owns channel c
  float* tl = tile[wv];
#pragma unroll
  for (int c = 0; c < CO; ++c) tl[c * 65 + lane] = pv ? v[c] : 0.f;
  // (a wave's LDS traffic is ordered: no barrier needed inside one wave; the compiler's waitcnt covers the hazard)
  __builtin_amdgcn_wave_barrier();
  const int tile_id = blockIdx.x * 4 + wv;
  const int first = tile_id * 64;
  const int cnt = min(64, npix - first);
  if (cnt <= 0) return;
  if (lane < CO) {
    float s = 0.f;
    for (int i = 0; i < 64; ++i) s += tl[lane * 65 + i];
    const float mean = s / (float)cnt;
    float m2 = 0.f;
    for (int i = 0; i < cnt; ++i) {
      const float d = tl[lane * 65 + i] - mean;
      m2 += d * d;
    }
    float* st = stats + ((size_t)tile_id * CO + lane) * 2;
    st[0] = s;
    st[1] = m2;
  }
  if (lane == 0) counts[tile_id] = (float)cnt;
}


// NT threads per workgroup = NT / 64 waves; the reduction tile holds 65 floats per (channel, wave) + 1 per channel
template <int CO, int NT>
__global__ __launch_bounds__(NT) void conv_c1_3x3_v4_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ out,
                                                             const float* __restrict__ bn_g, const float* __restrict__ bn_b,
                                                             const float* __restrict__ bn_m, const float* __restrict__ bn_v,
                                                             float* __restrict__ stats, float* __restrict__ counts, int B, int H,
                                                             int W, int ep, float slope, float eps) {
  constexpr int NWV = NT / 64, C1V_PITCH = NWV * 65 + 1;
  extern __shared__ float c1_smem[];
  float* tl = c1_smem;                       // [CO][waves][65]
  float* tmean = tl + CO * C1V_PITCH;        // [CO]
  float* wsm = tmean + CO;                   // [CO][9] weights, [CO] bias: broadcast LDS reads (with one wave per SIMD the 576
                                             // scalar loads of the weights were the kernel: nothing hides their latency)
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  for (int i = t; i < CO * 10; i += NT) wsm[i] = i < CO * 9 ? w[i] : (bias ? bias[i - CO * 9] : 0.f);
  if (ep == EP_BN_EVAL && t < CO) {          // eval-mode scale / shift per channel, once
    const float sc = bn_g[t] * (1.0f / sqrtf(bn_v[t] + eps));
    wsm[CO * 10 + t] = sc;
    wsm[CO * 11 + t] = bn_b[t] - bn_m[t] * sc;
  }
  const int W4 = W >> 2, hw = H * W, rows4 = H * W4, nq = B * rows4;
  const int q = blockIdx.x * NT + t;
  const bool pv = q < nq;
  const int qq = pv ? q : nq - 1;
  const int b = qq / rows4, r = qq - b * rows4, oy = r / W4, ox = 4 * (r - oy * W4);
  float xin[3][6];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int iy = oy + kh - 1;
    const bool rok = (unsigned)iy < (unsigned)H;
    const float* row = x + (size_t)b * hw + (size_t)(rok ? iy : 0) * W + ox;
    const float4 m = *reinterpret_cast<const float4*>(row);
    const float l = ox > 0 ? row[-1] : 0.f, rr = ox + 4 < W ? row[4] : 0.f;
    xin[kh][0] = rok ? l : 0.f; xin[kh][1] = rok ? m.x : 0.f; xin[kh][2] = rok ? m.y : 0.f; xin[kh][3] = rok ? m.z : 0.f;
    xin[kh][4] = rok ? m.w : 0.f; xin[kh][5] = rok ? rr : 0.f;
  }
  __syncthreads();
  auto conv4 = [&](int c, float (&a)[4]) {
    const float bs = wsm[CO * 9 + c];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = bs;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const float wk = wsm[c * 9 + kh * 3 + kw];
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = fmaf(wk, xin[kh][j + kw], a[j]);
      }
  };
  float* op = out + (size_t)b * CO * hw + (size_t)oy * W + ox;
#pragma unroll 16
  for (int c = 0; c < CO; ++c) {
    float a[4];
    conv4(c, a);
    if (ep == EP_RAW_STATS) tl[c * C1V_PITCH + wv * 65 + lane] = pv ? (a[0] + a[1]) + (a[2] + a[3]) : 0.f;
    if (ep == EP_BN_EVAL) {
      const float sc = wsm[CO * 10 + c], sh = wsm[CO * 11 + c];
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = lrelu(fmaf(a[j], sc, sh), slope);
    }
    if (ep == EP_LRELU) {
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = lrelu(a[j], slope);
    }
    if (pv) *reinterpret_cast<float4*>(op + (size_t)c * hw) = float4{a[0], a[1], a[2], a[3]};
  }
  if (ep != EP_RAW_STATS) return;
  // statistics of the workgroup's tile: thread (channel t >> 2, quarter t & 3) adds 64 per-thread sums, the quarters by DPP
  const int cnt = 4 * min(NT, nq - (int)blockIdx.x * NT);
  const int cc = t / NWV, part = t % NWV;
  __syncthreads();
  float s = 0.f;
#pragma unroll 16
  for (int i = 0; i < 64; ++i) s += tl[cc * C1V_PITCH + part * 65 + i];
  if (NWV >= 2) s += dpp_rot<0xB1>(s);
  if (NWV >= 4) s += dpp_rot<0x4E>(s);
  if (part == 0) tmean[cc] = s / (float)cnt;
  __syncthreads();
#pragma unroll 8
  for (int c = 0; c < CO; ++c) {
    float a[4];
    conv4(c, a);
    const float mn = tmean[c];
    float m2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float d = a[j] - mn; m2 = fmaf(d, d, m2); }
    tl[c * C1V_PITCH + wv * 65 + lane] = pv ? m2 : 0.f;
  }
  __syncthreads();
  float m2s = 0.f;
#pragma unroll 16
  for (int i = 0; i < 64; ++i) m2s += tl[cc * C1V_PITCH + part * 65 + i];
  if (NWV >= 2) m2s += dpp_rot<0xB1>(m2s);
  if (NWV >= 4) m2s += dpp_rot<0x4E>(m2s);
  if (part == 0) {
    float* st = stats + ((size_t)blockIdx.x * CO + cc) * 2;
    st[0] = s;
    st[1] = m2s;
  }
  if (t == 0) counts[blockIdx.x] = (float)cnt;
}

static bool c1_v4(int W) { return (W & 3) == 0; }

bool conv_c1_ok(int groups, int Cin, int Cout, int KH, int KW, int SH, int SW, int PH, int PW, int H, int in_plain) {
  return groups == 1 && Cin == 1 && Cout == 64 && KH == 3 && KW == 3 && SH == 1 && SW == 1 && PH == 1 && PW == 1 && H > 1 && in_plain;
}
static int c1_nt() {
  static int nt = 0;
  if (!nt) { const char* e = getenv("MS_C1_NT"); nt = e ? atoi(e) : 256; if (nt != 64 && nt != 128 && nt != 256) nt = 256; }
  return nt;
}
int conv_c1_tiles(int B, int H, int W) { return c1_v4(W) ? cdiv(B * H * (W >> 2), c1_nt()) : cdiv(B * H * W, 256) * 4; }

int launch_conv_c1(const float* x, const float* w, const float* bias, float* out, const float* bn_g, const float* bn_b,
                   const float* bn_m, const float* bn_v, float* stats, float* counts, int B, int H, int W, int ep, float slope,
                   float eps, hipStream_t s) {
  const double npix = (double)B * H * W;
  TimingScope ts(s, 2.0 * npix * 64 * 9, 4.0 * npix * 65, "conv_c1_3x3_kernel<64>|conv_fwd_c1 k3x3 Cout64 N%.0f%s", npix,
                 ep == EP_RAW_STATS ? " +bnstats" : "");
  if (ts.skip()) return 0;
  if (c1_v4(W) && (((uintptr_t)x | (uintptr_t)out) & 15) == 0) {
    const int nt = c1_nt(), nq = B * H * (W >> 2);
#define MS_C1V(NT)                                                                                                                 \
    do {                                                                                                                           \
      constexpr int lds = (64 * ((NT / 64) * 65 + 1) + 64 + 768) * 4;                                                                   \
      static int attr_done = 0;                                                                                                    \
      if (!attr_done) {                                                                                                            \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_c1_3x3_v4_kernel<64, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) \
          return set_error("conv_c1: cannot raise the dynamic LDS limit");                                                         \
        attr_done = 1;                                                                                                             \
      }                                                                                                                            \
      hipLaunchKernelGGL((conv_c1_3x3_v4_kernel<64, NT>), dim3(cdiv(nq, NT)), dim3(NT), lds, s, x, w, bias, out, bn_g, bn_b, bn_m,   \
                         bn_v, stats, counts, B, H, W, ep, slope, eps);                                                            \
    } while (0)
    if (nt == 64) MS_C1V(64); else if (nt == 128) MS_C1V(128); else MS_C1V(256);
#undef MS_C1V
    return check_launch("conv_c1_3x3_v4_kernel");
  }
  hipLaunchKernelGGL(conv_c1_3x3_kernel<64>, dim3(cdiv(B * H * W, 256)), dim3(256), 0, s, x, w, bias, out, bn_g, bn_b, bn_m, bn_v,
                     stats, counts, B, H, W, ep, slope, eps);
  return check_launch("conv_c1_3x3_kernel");
}

// ---- weight gradient of the same block: dw[c][kh][kw] = sum over (b, h, w) of dy_raw[b][c][h][w] * x[b][0][h + kh - 1][w + kw - 1].
// 576 numbers reduced over every pixel: on the MFMA kernel a 64 x 64 x 9 tile whose input-channel side is 63/64 zeros (37 us, 8 TF).
// Here it is a stream over dy_raw (67 MB at the headline size) on the vector unit: a workgroup owns `rows_per_wg` rows of the
// flattened (image, row) space; thread (cs = t / 32, v = t % 32) takes the 16-byte column groups v, v + 32, ... of a row for the
// 8 channels cs, cs + 8, ..., keeps the 3 x 6 input window of its four pixels in registers (shared by all channels) and 8 x 9
// running sums.  Fixed-order finish: the 32 lanes of a channel set by xor-shuffles, one slab of 576 partial sums per workgroup,
// summed across workgroups by the caller's slab reduction like every other split weight gradient (bit-reproducible).
__global__ __launch_bounds__(256) void wgrad_c1_3x3_kernel(const float* __restrict__ dyr, const float* __restrict__ x,
                                                           float* __restrict__ part, int B, int H, int W, int rows_per_wg) {
  const int t = threadIdx.x, v = t & 31, cs = t >> 5;
  const int W4 = W >> 2, rows = B * H;
  const int r0 = blockIdx.x * rows_per_wg, r1 = min(rows, r0 + rows_per_wg);
  float acc[8][9];
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int q = 0; q < 9; ++q) acc[k][q] = 0.f;
  for (int r = r0; r < r1; ++r) {
    const int b = r / H, h = r - b * H;
    for (int v0 = v; v0 < W4; v0 += 32) {
      // dy_raw of this thread's four pixels, 8 channels: all eight 16-byte loads go out before anything is consumed
      float4 d[8];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        d[k] = *reinterpret_cast<const float4*>(dyr + (((size_t)b * 64 + cs + 8 * k) * H + h) * W + 4 * v0);
      // input window: rows h-1 .. h+1, columns 4 v0 - 1 .. 4 v0 + 4 (zero padding)
      float xw[3][6];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int hh = h + kh - 1;
        const bool rv = (unsigned)hh < (unsigned)H;
        const float* xr = x + ((size_t)b * H + (rv ? hh : h)) * W + 4 * v0;
        const float4 m = *reinterpret_cast<const float4*>(xr);
        const float lf = v0 > 0 ? xr[-1] : 0.f, rt = v0 + 1 < W4 ? xr[4] : 0.f;
        xw[kh][0] = rv ? lf : 0.f; xw[kh][1] = rv ? m.x : 0.f; xw[kh][2] = rv ? m.y : 0.f;
        xw[kh][3] = rv ? m.z : 0.f; xw[kh][4] = rv ? m.w : 0.f; xw[kh][5] = rv ? rt : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            float a = acc[k][kh * 3 + kw];
            a = fmaf(d[k].x, xw[kh][kw], a);
            a = fmaf(d[k].y, xw[kh][kw + 1], a);
            a = fmaf(d[k].z, xw[kh][kw + 2], a);
            a = fmaf(d[k].w, xw[kh][kw + 3], a);
            acc[k][kh * 3 + kw] = a;
          }
    }
  }
  // the 32 lanes of a channel set (lane bits 0..4), fixed butterfly
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      float a = acc[k][q];
#pragma unroll
      for (int m = 1; m < 32; m <<= 1) a += __shfl_xor(a, m, 64);
      acc[k][q] = a;
    }
  if (v == 0) {
    float* o = part + (size_t)blockIdx.x * 576;
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int q = 0; q < 9; ++q) o[(cs + 8 * k) * 9 + q] = acc[k][q];
  }
}

bool wgrad_c1_ok(int groups, int Cin, int Cout, int KH, int KW, int SH, int SW, int PH, int PW, int H, int W, int in_plain) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("MS_WGRAD_C1"); on = e ? atoi(e) : 1; }
  return on && conv_c1_ok(groups, Cin, Cout, KH, KW, SH, SW, PH, PW, H, in_plain) && (W & 3) == 0;
}
// slabs (= workgroups) of the launch: four rows per workgroup, at most 1024 slabs
int wgrad_c1_splits(int B, int H) {
  const int rows = B * H;
  return std::max(2, std::min(1024, cdiv(rows, 4)));
}
int launch_wgrad_c1(const float* dyr, const float* x, float* part, int B, int H, int W, hipStream_t s) {
  const int S = wgrad_c1_splits(B, H), rows_per_wg = cdiv(B * H, S);
  const double npix = (double)B * H * W;
  TimingScope ts(s, 2.0 * npix * 64 * 9, 4.0 * npix * 65, "wgrad_c1_3x3_kernel|conv_wgrad_c1 k3x3 Cout64 N%.0f slabs%d", npix, S);
  if (ts.skip()) return 0;
  if ((((uintptr_t)dyr | (uintptr_t)x) & 15) != 0) return set_error("wgrad_c1: unaligned tensors");
  hipLaunchKernelGGL(wgrad_c1_3x3_kernel, dim3(S), dim3(256), 0, s, dyr, x, part, B, H, W, rows_per_wg);
  return check_launch("wgrad_c1_3x3_kernel");
}

}  // namespace ms
