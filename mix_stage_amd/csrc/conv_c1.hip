// Single-input-channel 3x3 convolution (the AudioEncoder's first block: 1 -> 64 channels over the (time, mel) plane).
// K = 9 is far too short for the matrix pipe (the patch kernel spends its time in per-tile fixed costs: 55 us for 151 MFLOP);
// here one thread computes one output pixel for all 64 channels on the VALU (weights come in through scalar loads) and
// the launch is bound by writing the 67 MB output.  BN statistics: each wave passes its 64 x 64 tile through LDS and one
// lane per channel accumulates (sum, M2 about the tile mean) in a fixed order -- same partials as the patch kernels.
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

bool conv_c1_ok(int groups, int Cin, int Cout, int KH, int KW, int SH, int SW, int PH, int PW, int H, int in_plain) {
  return groups == 1 && Cin == 1 && Cout == 64 && KH == 3 && KW == 3 && SH == 1 && SW == 1 && PH == 1 && PW == 1 && H > 1 && in_plain;
}
int conv_c1_tiles(int B, int H, int W) { return cdiv(B * H * W, 256) * 4; }

int launch_conv_c1(const float* x, const float* w, const float* bias, float* out, const float* bn_g, const float* bn_b,
                   const float* bn_m, const float* bn_v, float* stats, float* counts, int B, int H, int W, int ep, float slope,
                   float eps, hipStream_t s) {
  const double npix = (double)B * H * W;
  TimingScope ts(s, 2.0 * npix * 64 * 9, 4.0 * npix * 65, "conv_c1_3x3_kernel<64>|conv_fwd_c1 k3x3 Cout64 N%.0f%s", npix,
                 ep == EP_RAW_STATS ? " +bnstats" : "");
  if (ts.skip()) return 0;
  hipLaunchKernelGGL(conv_c1_3x3_kernel<64>, dim3(cdiv(B * H * W, 256)), dim3(256), 0, s, x, w, bias, out, bn_g, bn_b, bn_m, bn_v,
                     stats, counts, B, H, W, ep, slope, eps);
  return check_launch("conv_c1_3x3_kernel");
}

}  // namespace ms
