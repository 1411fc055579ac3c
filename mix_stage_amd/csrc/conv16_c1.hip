// 16-bit arithmetic, single-input-channel 3x3 convolution (the AudioEncoder's first block, 1 -> 64 channels over the (time, mel)
// plane; layers.py:167).  K = 9 (72 with the channel padded to a cb8 vector) is far too short for the matrix pipe: the MFMA kernel
// spends 35-41 us on 151 MFLOP.  Here a thread computes one output pixel for all 64 channels on the vector unit -- 16-bit operands
// (the weights rounded exactly as the prepared operand stages round them), fp32 products and sums -- and stores eight 16-byte cb8
// vectors (a wave instruction = 1 KB of one channel block's row); weights, bias and eval-mode scale / shift are broadcast from
// LDS.  BatchNorm statistics (train): the workgroup's 256 pixels are one tile, (sum, M2 about the tile mean) per channel from the
// fp32 values, second pass by recomputing the conv -- the same partials bn_finalize takes from the MFMA kernels.
#include <stdint.h>

#include "conv16_kernel.h"

namespace ms {

template <typename DT>
__global__ __launch_bounds__(256) void conv16_c1_3x3_kernel(const u32x4* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ wscale,      // per-channel factor folded into the weights before rounding (inference: eval BatchNorm), or NULL
                                                            const float* __restrict__ bias, u32x4* __restrict__ out,
                                                            const float* __restrict__ bn_g, const float* __restrict__ bn_b,
                                                            const float* __restrict__ bn_m, const float* __restrict__ bn_v,
                                                            float* __restrict__ stats, float* __restrict__ counts, int B, int H,
                                                            int W, int ep, float slope, float eps) {
  constexpr int CO = 64, PITCH = 4 * 65 + 1;
  extern __shared__ float c1h_smem[];
  float* tl = c1h_smem;                      // [CO][4 waves][65] (statistics)
  float* tmean = tl + CO * PITCH;            // [CO]
  float* wsm = tmean + CO;                   // [CO][9] weights (rounded to the operand type), [CO] bias, [CO] scale, [CO] shift
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  for (int i = t; i < CO * 10; i += 256) {
    float v = i < CO * 9 ? w[i] : (bias ? bias[i - CO * 9] : 0.f);
    if (i < CO * 9) v = DT::lo(DT::pack2(wscale ? v * wscale[i / 9] : v, 0.f));
    wsm[i] = v;
  }
  if (ep == EP_BN_EVAL && t < CO) {
    const float sc = bn_g[t] * (1.0f / sqrtf(bn_v[t] + eps));
    wsm[CO * 10 + t] = sc;
    wsm[CO * 11 + t] = bn_b[t] - bn_m[t] * sc;
  }
  const int hw = H * W, npix = B * hw;
  const int pix = blockIdx.x * 256 + t;
  const bool pv = pix < npix;
  const int pp = pv ? pix : npix - 1;
  const int b = pp / hw, rem = pp - b * hw, oy = rem / W, ox = rem - oy * W;
  // the input's cb8 vectors hold the one channel in their first 16 bits
  const unsigned short* xs = reinterpret_cast<const unsigned short*>(x);
  float xin[9];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int iy = oy + kh - 1, ix = ox + kw - 1;
      const bool ok = ((unsigned)iy < (unsigned)H) & ((unsigned)ix < (unsigned)W);
      const unsigned short u = xs[8 * ((size_t)b * hw + (size_t)(ok ? iy : oy) * W + (ok ? ix : ox))];
      xin[kh * 3 + kw] = ok ? DT::lo((unsigned)u) : 0.f;
    }
  __syncthreads();
  auto conv8 = [&](int cb, float (&a)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = 8 * cb + j;
      float s = wsm[CO * 9 + c];
#pragma unroll
      for (int k = 0; k < 9; ++k) s = fmaf(wsm[c * 9 + k], xin[k], s);
      a[j] = s;
    }
  };
  u32x4* op = out + ((size_t)b * (CO / 8)) * hw + rem;
#pragma unroll
  for (int cb = 0; cb < CO / 8; ++cb) {
    float a[8];
    conv8(cb, a);
    if (ep == EP_RAW_STATS) {
#pragma unroll
      for (int j = 0; j < 8; ++j) tl[(8 * cb + j) * PITCH + wv * 65 + lane] = pv ? a[j] : 0.f;
    }
    if (ep == EP_BN_EVAL) {
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = lrelu(fmaf(a[j], wsm[CO * 10 + 8 * cb + j], wsm[CO * 11 + 8 * cb + j]), slope);
    }
    if (ep == EP_LRELU) {
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = lrelu(a[j], slope);
    }
    if (pv) op[(size_t)cb * hw] = pack8<DT>(a);
  }
  if (ep != EP_RAW_STATS) return;
  const int cnt = min(256, npix - (int)blockIdx.x * 256);
  const int cc = t >> 2, part = t & 3;
  __syncthreads();
  float s = 0.f;
#pragma unroll 16
  for (int i = 0; i < 64; ++i) s += tl[cc * PITCH + part * 65 + i];
  s += dpp_rot<0xB1>(s);
  s += dpp_rot<0x4E>(s);
  if (part == 0) tmean[cc] = s / (float)cnt;
  __syncthreads();
#pragma unroll
  for (int cb = 0; cb < CO / 8; ++cb) {
    float a[8];
    conv8(cb, a);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float d = a[j] - tmean[8 * cb + j];
      tl[(8 * cb + j) * PITCH + wv * 65 + lane] = pv ? d * d : 0.f;
    }
  }
  __syncthreads();
  float m2 = 0.f;
#pragma unroll 16
  for (int i = 0; i < 64; ++i) m2 += tl[cc * PITCH + part * 65 + i];
  m2 += dpp_rot<0xB1>(m2);
  m2 += dpp_rot<0x4E>(m2);
  if (part == 0) {
    float* st = stats + ((size_t)blockIdx.x * CO + cc) * 2;
    st[0] = s;
    st[1] = m2;
  }
  if (t == 0) counts[blockIdx.x] = (float)cnt;
}

bool conv16_c1_ok(const ms_conv_desc* d) {
  return d->groups == 1 && d->Cin == 1 && d->Cout == 64 && d->KH == 3 && d->KW == 3 && d->SH == 1 && d->SW == 1 && d->PH == 1 &&
         d->PW == 1 && d->H > 1 && d->in_mode == MS_IN_PLAIN;
}
int conv16_c1_tiles(const ms_conv_desc* d) { return cdiv(d->B * d->H * d->W, 256); }

int launch_conv16_c1(int dt, const void* x, const float* w, const float* wscale, const float* bias, void* out, const float* bn_g, const float* bn_b,
                     const float* bn_m, const float* bn_v, float* stats, float* counts, int B, int H, int W, int ep, float slope,
                     float eps, hipStream_t s) {
  const double npix = (double)B * H * W;
  TimingScope ts(s, 2.0 * npix * 64 * 9, 2.0 * npix * 72, "conv16_c1_3x3_kernel|conv_fwd_cb8 c1 k3x3 Cout64 N%.0f%s", npix,
                 ep == EP_RAW_STATS ? " +bnstats" : "");
  if (ts.skip()) return 0;
  constexpr int lds = (64 * (4 * 65 + 1) + 64 + 64 * 12) * 4;
  static int attr_done = 0;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv16_c1_3x3_kernel<BF16>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv16_c1_3x3_kernel<F16>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return set_error("conv16_c1: cannot raise the dynamic LDS limit");
    attr_done = 1;
  }
  const dim3 grid(cdiv(B * H * W, 256));
  if (dt == DT_BF16)
    hipLaunchKernelGGL(conv16_c1_3x3_kernel<BF16>, grid, dim3(256), lds, s, (const u32x4*)x, w, wscale, bias, (u32x4*)out, bn_g, bn_b, bn_m, bn_v,
                       stats, counts, B, H, W, ep, slope, eps);
  else
    hipLaunchKernelGGL(conv16_c1_3x3_kernel<F16>, grid, dim3(256), lds, s, (const u32x4*)x, w, wscale, bias, (u32x4*)out, bn_g, bn_b, bn_m, bn_v,
                       stats, counts, B, H, W, ep, slope, eps);
  return check_launch("conv16_c1_3x3_kernel");
}

}  // namespace ms
