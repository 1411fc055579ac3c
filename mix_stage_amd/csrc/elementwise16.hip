// 16-bit arithmetic mode: the HBM-bound kernels around the convs, on cb8 tensors ([B][C8][HW][8] 16-bit, conv16.h):
// BatchNorm apply / backward, activation backward, eval-BN folding, and the layout converters at the path's fp32 boundaries.
// One thread moves one 16-byte vector (8 channels of a pixel); consecutive lanes take consecutive pixels, so every access
// is a 1 KiB wave transaction along the time axis.  All reductions run in a fixed order (no float atomics).
#include <algorithm>

#include "conv16_kernel.h"

namespace ms {

// y = lrelu(y_raw * scale[c] + shift[c]); y cb8, or plain fp32 (B, C, HW) when y_f32 != NULL
template <typename DT>
__global__ __launch_bounds__(256) void bn_apply16_kernel(const u32x4* __restrict__ y_raw, u32x4* __restrict__ y,
                                                         float* __restrict__ y_f32, const float* __restrict__ save, int C,
                                                         int C8, int HW, size_t total, float slope) {
  prefetch_kernargs<192>();
  const float* scale = save + 2 * (size_t)C;
  const float* shift = save + 3 * (size_t)C;
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (size_t)gridDim.x * 256) {
    const size_t bc = v / HW;
    const int cb = (int)(bc % C8), pix = (int)(v - bc * HW);
    const size_t b = bc / C8;
    float f[8];
    unpack8<DT>(y_raw[v], f);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cb * 8 + j;
      f[j] = c < C ? lrelu(fmaf(f[j], scale[c], shift[c]), slope) : 0.f;
    }
    if (y_f32) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (cb * 8 + j < C) y_f32[(b * C + cb * 8 + j) * HW + pix] = f[j];
    } else {
      y[v] = pack8<DT>(f);
    }
  }
}

int launch_bn_apply16(int dt, const void* y_raw, void* y, float* y_f32, const float* save, int B, int C, int HW, float slope,
                      hipStream_t s) {
  const int C8 = c8_of(C);
  const size_t total = (size_t)B * C8 * HW;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 8192);
  TimingScope ts(s, 0, (y_f32 ? 2.0 + 4.0 : 4.0) * 8.0 * total, "bn_apply16_kernel|bn_apply16 C%d N%d", C, B * HW);
  if (ts.skip()) return 0;
  if (dt == DT_BF16)
    hipLaunchKernelGGL(bn_apply16_kernel<BF16>, dim3(blocks), dim3(256), 0, s, (const u32x4*)y_raw, (u32x4*)y, y_f32, save, C, C8, HW,
                       total, slope);
  else
    hipLaunchKernelGGL(bn_apply16_kernel<F16>, dim3(blocks), dim3(256), 0, s, (const u32x4*)y_raw, (u32x4*)y, y_f32, save, C, C8, HW,
                       total, slope);
  return check_launch("bn_apply16_kernel");
}

// BatchNorm finalize + apply in ONE launch for layers with few statistics tiles (every 1-D layer): each workgroup combines
// the per-tile partials of its 8 channels itself (same order everywhere: identical results), chunk 0 also records them
// (save = mean | invstd | scale | shift, running statistics), then normalises its share of the channel block.
// Per-channel parameters of a channel block: unconditional loads with a clamped index (no branch between them, so the
// scalar loads go out back to back and are waited for once).
__device__ inline void load_chan8(const float* __restrict__ p, int c0, int C, float (&o)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = p[min(c0 + j, C - 1)];
}

// Every global load of the kernel -- tile statistics, BN parameters, the first FIN_PRE data vectors of each thread -- is
// issued before anything is consumed: the kernel costs ONE memory round trip plus the reduction, not four.
constexpr int FIN_PRE = 4;
template <typename DT>
__global__ __launch_bounds__(256) void bn_finalize_apply16_kernel(const float* __restrict__ stats, const float* __restrict__ counts,
                                                                  int n_tiles, int N, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float* running_mean,
                                                                  float* running_var, float* __restrict__ save, float eps,
                                                                  float momentum, const u32x4* __restrict__ y_raw,
                                                                  u32x4* __restrict__ y, float* __restrict__ y_f32, int B, int C,
                                                                  int C8, int HW, int b_per_chunk, float slope, int sg,
                                                                  int chunks_per_grp) {
  prefetch_kernargs<192>();
  const FastDiv fdHW(HW, B * HW);
  __shared__ double part[2][8][33];
  __shared__ float scsh[16];
  const int cb = blockIdx.x, ch = blockIdx.y, t = threadIdx.x;
  const int j = t >> 5, i = t & 31, c = cb * 8 + j;
  const bool cv = c < C;
  const int cc = min(c, C - 1);
  // chunk = b_per_chunk consecutive vectors of this channel block's flattened (batch item, pixel) space.  sg == 2 (MS_DT_STAT_PAIR):
  // the batch is two statistics groups of B / 2 clips; a chunk lies inside one of them (chunks_per_grp chunks each), the first
  // half of the statistics tiles belongs to the first group
  const int grp_vecs = B / sg * HW, my_grp = sg > 1 ? ch / chunks_per_grp : 0, chg = ch - my_grp * chunks_per_grp;
  const int e_base = my_grp * grp_vecs + chg * b_per_chunk, b0 = 0;
  const int n = min(b_per_chunk, grp_vecs - chg * b_per_chunk);
  // statistics of tiles i and i + 32 (n_tiles <= 64) and the channel's parameters -- sg == 2: tile i of either group
  const int ntg = n_tiles / sg;
  const int k0 = sg > 1 ? min(i, ntg - 1) : min(i, n_tiles - 1), k1 = sg > 1 ? ntg + min(i, ntg - 1) : min(i + 32, n_tiles - 1);
  const float2 st0 = *(const float2*)(stats + ((size_t)k0 * C + cc) * 2), st1 = *(const float2*)(stats + ((size_t)k1 * C + cc) * 2);
  const float cnt0 = counts[k0], cnt1 = counts[k1];
  const float g = gamma[cc], bt = beta[cc], rm = running_mean[cc], rv = running_var[cc];
  // data prefetch, issued AFTER the statistics: loads return in order, and the statistics (L2 hits, just written by the conv)
  // must not queue behind HBM reads
  __builtin_amdgcn_sched_barrier(0);
  u32x4 raw[FIN_PRE];
  size_t vofs[FIN_PRE];
#pragma unroll
  for (int q = 0; q < FIN_PRE; ++q) {
    const int e = min(t + q * 256, n - 1);
    const int bl = fdHW.div(e_base + e), pix = e_base + e - bl * HW;
    vofs[q] = ((size_t)(b0 + bl) * C8 + cb) * HW + pix;
    raw[q] = y_raw[vofs[q]];
  }
  __builtin_amdgcn_sched_barrier(0);
  const bool u0 = cv && (sg > 1 ? i < ntg : i < n_tiles), u1 = cv && (sg > 1 ? i < ntg : i + 32 < n_tiles);
  // sums of the tiles' sums: one total (sg == 1: both tiles of the thread belong to it) or one per group
  part[0][j][i] = (u0 ? (double)st0.x : 0.0) + ((u1 && sg == 1) ? (double)st1.x : 0.0);
  part[1][j][i] = (u1 && sg > 1) ? (double)st1.x : 0.0;
  __syncthreads();
  double tot0 = 0.0, tot1 = 0.0;
#pragma unroll 8
  for (int k = 0; k < 32; ++k) { tot0 += part[0][j][k]; tot1 += part[1][j][k]; }
  const double Ng = (double)(N / sg);
  const double mean0 = tot0 / Ng, mean1 = tot1 / Ng;
  __syncthreads();
  double q0 = 0.0, q1 = 0.0;
  if (u0) { const double cn = (double)cnt0, dlt = (double)st0.x / cn - mean0; q0 += (double)st0.y + cn * dlt * dlt; }
  if (u1) {
    const double cn = (double)cnt1, dlt = (double)st1.x / cn - (sg > 1 ? mean1 : mean0), q = (double)st1.y + cn * dlt * dlt;
    if (sg > 1) q1 += q; else q0 += q;
  }
  part[0][j][i] = q0;
  part[1][j][i] = q1;
  __syncthreads();
  if (i == 0) {
    double m2g[2] = {0.0, 0.0};
    for (int k = 0; k < 32; ++k) { m2g[0] += part[0][j][k]; m2g[1] += part[1][j][k]; }
    const double meang[2] = {mean0, mean1};
    float sc = 0.f, shf = 0.f;
    if (cv) {
      // chunk 0 of group 0 records every group's vector and moves the running statistics once per group, in group order
      float rmc = rm, rvc = rv;
      for (int gi = 0; gi < sg; ++gi) {
        if (gi != my_grp && ch != 0) continue;
        const float var = (float)(m2g[gi] / Ng);
        const float invstd = 1.0f / sqrtf(var + eps);
        const float fmean = (float)meang[gi];
        const float scg = g * invstd, shg = bt - fmean * scg;
        if (gi == my_grp) { sc = scg; shf = shg; }
        if (ch == 0) {
          float* sv = save + (size_t)gi * 4 * C;
          sv[c] = fmean;
          sv[C + c] = invstd;
          sv[2 * C + c] = scg;
          sv[3 * C + c] = shg;
          const float unbiased = Ng > 1.0 ? (float)(m2g[gi] / (Ng - 1.0)) : var;
          running_stats_update(&running_mean[c], &running_var[c], rmc, rvc, momentum, fmean, unbiased);
          rmc = running_mean[c]; rvc = running_var[c];
        }
      }
    }
    scsh[j] = sc; scsh[8 + j] = shf;
  }
  __syncthreads();
  float sc[8], shf[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { sc[k] = scsh[k]; shf[k] = scsh[8 + k]; }
  auto emit = [&](const u32x4& rawv, size_t v, int e) {
    float f[8];
    unpack8<DT>(rawv, f);
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = cb * 8 + k < C ? lrelu(fmaf(f[k], sc[k], shf[k]), slope) : 0.f;
    if (y_f32) {
      const int bl = fdHW.div(e_base + e), pix = e_base + e - bl * HW;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (cb * 8 + k < C) y_f32[((size_t)(b0 + bl) * C + cb * 8 + k) * HW + pix] = f[k];
    } else {
      y[v] = pack8<DT>(f);
    }
  };
#pragma unroll
  for (int q = 0; q < FIN_PRE; ++q)
    if (t + q * 256 < n) emit(raw[q], vofs[q], t + q * 256);
  for (int e0 = t + FIN_PRE * 256; e0 < n; e0 += FIN_PRE * 256) {
#pragma unroll
    for (int q = 0; q < FIN_PRE; ++q) {
      const int e = min(e0 + q * 256, n - 1);
      const int bl = fdHW.div(e_base + e), pix = e_base + e - bl * HW;
      vofs[q] = ((size_t)(b0 + bl) * C8 + cb) * HW + pix;
      raw[q] = y_raw[vofs[q]];
    }
#pragma unroll
    for (int q = 0; q < FIN_PRE; ++q)
      if (e0 + q * 256 < n) emit(raw[q], vofs[q], e0 + q * 256);
  }
}

int launch_bn_finalize_apply16(int dt, const float* stats, const float* counts, int n_tiles, int N, const float* gamma,
                               const float* beta, float* rm, float* rv, float* save, float eps, float momentum, const void* y_raw,
                               void* y, float* y_f32, int B, int C, int HW, float slope, hipStream_t s, int sg) {
  const int C8 = c8_of(C);
  int bpc;
  // (sg statistics groups of B / sg clips: whole chunks per group)
  const int cpg = bwd16_chunks(B / sg, C8, HW, &bpc), nchunk = cpg * sg;
  if (sg > 1 && (n_tiles % sg || n_tiles / sg > 32)) return set_error("bn_finalize_apply16: %d statistics tiles in %d groups", n_tiles, sg);
  const dim3 grid(C8, nchunk);
  TimingScope ts(s, 0, (y_f32 ? 6.0 : 4.0) * 8.0 * (double)B * C8 * HW, "bn_finalize_apply16_kernel|bn_finalize_apply16 C%d N%d tiles%d", C,
                 B * HW, n_tiles);
  if (ts.skip()) return 0;
  if (dt == DT_BF16)
    hipLaunchKernelGGL(bn_finalize_apply16_kernel<BF16>, grid, dim3(256), 0, s, stats, counts, n_tiles, N, gamma, beta, rm, rv, save, eps,
                       momentum, (const u32x4*)y_raw, (u32x4*)y, y_f32, B, C, C8, HW, bpc, slope, sg, cpg);
  else
    hipLaunchKernelGGL(bn_finalize_apply16_kernel<F16>, grid, dim3(256), 0, s, stats, counts, n_tiles, N, gamma, beta, rm, rv, save, eps,
                       momentum, (const u32x4*)y_raw, (u32x4*)y, y_f32, B, C, C8, HW, bpc, slope, sg, cpg);
  return check_launch("bn_finalize_apply16_kernel");
}

// ---------------------------------------------------------------------------------------------
// block-wide sums of 8 values per thread; result valid in thread 0.  red: 4*8 floats of LDS.
template <int NW = 4>
__device__ inline void block_sum8(float (&v)[8], float* red) {
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = wave_sum(v[j]);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) red[(threadIdx.x >> 6) * 8 + j] = v[j];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) a += red[w * 8 + j];
    v[j] = a;
  }
}

// block-wide sums of NV values per thread, result in EVERY thread.  red: NW*NV floats of LDS (rows read back as float4).
template <int NV, int NW>
__device__ inline void block_sum_n(float (&v)[NV], float* red) {
#pragma unroll
  for (int j = 0; j < NV; ++j) v[j] = wave_sum(v[j]);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int j = 0; j < NV; ++j) red[(threadIdx.x >> 6) * NV + j] = v[j];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NV; j += 4) {
    float4 a = *(const float4*)(red + j);
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      const float4 b = *(const float4*)(red + w * NV + j);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    v[j] = a.x; v[j + 1] = a.y; v[j + 2] = a.z; v[j + 3] = a.w;
  }
}

template <typename DT, bool DYF32>
__device__ inline void load_dy8(const u32x4* dy, const float* dy_f32, size_t v, size_t b, int cb, int pix, int C, int HW,
                                float (&f)[8]) {
  if (DYF32) {
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = cb * 8 + j < C ? dy_f32[(b * C + cb * 8 + j) * HW + pix] : 0.f;
  } else {
    unpack8<DT>(dy[v], f);
  }
}

// One form for both sources, no branch in the element loops.  v = the stored value (y_raw, or the block output y):
//   pos = v * P1 + P0 > 0          y_raw: z = v*scale + shift       y: the sign of y itself          (P1, P0) = (scale, shift) | (1, 0)
//   u   = pos ? v : v * NS         y_raw: v                          y: z = y / slope below zero       NS = 1 | 1 / slope
//   x_hat = u * A + Bc             y_raw: (v - mean) * invstd        y: (z - beta) / gamma             (A, Bc) = (invstd, -mean*invstd) | (invstd/scale, -beta*invstd/scale)
struct BnSrc { float P1, P0, NS, A, Bc; };
__device__ __forceinline__ BnSrc bn_src(bool raw, float mean, float invstd, float sc, float sh, float slope) {
  BnSrc k;
  if (raw) {
    k.P1 = sc; k.P0 = sh; k.NS = 1.f; k.A = invstd; k.Bc = -mean * invstd;
  } else {
    const float ios = invstd / sc;
    k.P1 = 1.f; k.P0 = 0.f; k.NS = 1.0f / slope; k.A = ios; k.Bc = -fmaf(mean, sc, sh) * ios;
  }
  return k;
}
__device__ __forceinline__ void bn_xhat_mask(const BnSrc& k, float v, float& xh, bool& pos) {
  pos = fmaf(v, k.P1, k.P0) > 0.f;
  xh = fmaf(pos ? v : v * k.NS, k.A, k.Bc);
}

// Source constants of a channel block, formed ONCE per workgroup: each of the 8 channels has one thread (t < 8, same wave) that
// fetches its four statistics, decides whether the channel inverts safely and leaves both sets of constants (for y_raw, for y) in
// LDS; the block-wide verdict is a ballot.  One barrier; no division or predicate per data thread (with 1-4 vectors per thread
// the per-thread form of this cost a quarter of the kernel).  Returns raw; fills ks[8].
struct BnSrcShared { float k[2][8][5]; int raw; };
__device__ __forceinline__ bool bn_src_block(BnSrcShared& sh_, const float* __restrict__ save, int cb, int C, float slope, bool have_y,
                                             BnSrc (&ks)[8]) {
  const int t = threadIdx.x;
  if (t < 8) {
    const int c = min(cb * 8 + t, C - 1);
    const float mn = save[c], is = save[C + c], scv = save[2 * (size_t)C + c], shv = save[3 * (size_t)C + c];
    const bool unsafe = (cb * 8 + t < C) && bn_inv_unsafe(mn, is, scv, shv, slope);
    const unsigned long long any = __ballot(unsafe);
    if (t == 0) sh_.raw = !have_y || any != 0;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const BnSrc k = bn_src(r == 0, mn, is, scv, shv, slope);
      sh_.k[r][t][0] = k.P1; sh_.k[r][t][1] = k.P0; sh_.k[r][t][2] = k.NS; sh_.k[r][t][3] = k.A; sh_.k[r][t][4] = k.Bc;
    }
  }
  __syncthreads();
  const bool raw = sh_.raw != 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float* kp = sh_.k[raw ? 0 : 1][j];
    ks[j].P1 = kp[0]; ks[j].P0 = kp[1]; ks[j].NS = kp[2]; ks[j].A = kp[3]; ks[j].Bc = kp[4];
  }
  return raw;
}

// BatchNorm + LeakyReLU backward, pass 1.  grid (C8, nchunk); chunk = contiguous range of batch items.
//   dz = dy * lrelu'(z), z = y_raw*scale+shift;  xh = (y_raw-mean)*invstd;  partial[c][chunk] = (sum dz, sum dz*xh)
template <typename DT, bool DYF32>
__global__ __launch_bounds__(256) void bn_bwd16_reduce_kernel(const u32x4* __restrict__ dy, const float* __restrict__ dy_f32,
                                                              const u32x4* __restrict__ y_raw, const float* __restrict__ save,
                                                              float* __restrict__ partial, float* __restrict__ xsum, int B, int C, int C8,
                                                              int HW, int b_per_chunk, float slope) {
  // xsum[c][chunk] = sum of x_hat: the bias gradient (a conv bias in front of BatchNorm: zero but for rounding) is
  // -gamma invstd mean(dz x_hat) sum(x_hat), written by the apply pass -- no column sums of dy_raw, no finalize launch
  prefetch_kernargs<192>();
  const FastDiv fdHW(HW, B * HW);
  __shared__ float red[32];
  const int cb = blockIdx.x, ch = blockIdx.y, t = threadIdx.x, nchunk = gridDim.y;
  const int e_base = ch * b_per_chunk, b0 = 0;                   // chunk of the flattened (batch item, pixel) space
  float mean[8], invstd[8], sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = min(cb * 8 + j, C - 1);
    mean[j] = save[c]; invstd[j] = save[C + c]; sc[j] = save[2 * C + c]; sh[j] = save[3 * C + c];
  }
  float s1[8] = {}, s2[8] = {}, s3[8] = {};
  const int n = min(b_per_chunk, B * HW - e_base);
  // 4 vectors per thread and step, all loads issued before the first use
  for (int e0 = t; e0 < n; e0 += 1024) {
    u32x4 ry[4], rg[4];
    float gf[DYF32 ? 4 : 1][8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = min(e0 + q * 256, n - 1);
      const int bl = fdHW.div(e_base + e), pix = e_base + e - bl * HW;
      const size_t b = b0 + bl, v = (b * C8 + cb) * HW + pix;
      ry[q] = y_raw[v];
      if (DYF32) {
#pragma unroll
        for (int j = 0; j < 8; ++j) gf[q][j] = dy_f32[(b * C + min(cb * 8 + j, C - 1)) * HW + pix];
      } else {
        rg[q] = dy[v];
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (e0 + q * 256 < n) {
        float g[8], yr[8];
        if (DYF32) {
#pragma unroll
          for (int j = 0; j < 8; ++j) g[j] = cb * 8 + j < C ? gf[q][j] : 0.f;
        } else {
          unpack8<DT>(rg[q], g);
        }
        unpack8<DT>(ry[q], yr);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float z = fmaf(yr[j], sc[j], sh[j]);
          const float dz = g[j] * (z > 0.f ? 1.f : slope);
          const float xh = (yr[j] - mean[j]) * invstd[j];
          s1[j] += dz;
          s2[j] = fmaf(dz, xh, s2[j]);
          s3[j] += xh;
        }
      }
    }
  }
  block_sum8(s1, red);
  block_sum8(s2, red);
  block_sum8(s3, red);
  if (t == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cb * 8 + j;
      if (c < C) {
        partial[((size_t)c * nchunk + ch) * 2] = s1[j];
        partial[((size_t)c * nchunk + ch) * 2 + 1] = s2[j];
        xsum[(size_t)c * nchunk + ch] = s3[j];
      }
    }
  }
}

//   pass 2: dyr = gamma*invstd*(dz - s1/N - xh*s2/N); colsum partial of dyr; dgamma = s2, dbeta = s1
template <typename DT, bool DYF32>
__global__ __launch_bounds__(256) void bn_bwd16_apply_kernel(const u32x4* __restrict__ dy, const float* __restrict__ dy_f32,
                                                             const u32x4* __restrict__ y_raw, const float* __restrict__ save,
                                                             const float* __restrict__ gamma, const float* __restrict__ partial,
                                                             u32x4* __restrict__ dyr, const float* __restrict__ xsum, float* dbias,
                                                             float* dgamma, float* dbeta, int B, int C, int C8, int HW, int b_per_chunk,
                                                             float slope) {
  prefetch_kernargs<192>();
  const FastDiv fdHW(HW, B * HW);
  const int cb = blockIdx.x, ch = blockIdx.y, t = threadIdx.x, nchunk = gridDim.y;
  const float invN = 1.0f / (float)((size_t)B * HW);
  // prologue in one memory round trip: a 32-lane group sums one channel's chunk partials, the 40 per-channel parameters
  // arrive through LDS
  __shared__ float prm[64];
  {
    const int jj = t >> 5, ii = t & 31, cj = min(cb * 8 + jj, C - 1);
    float a = 0.f, b2 = 0.f, c3 = 0.f;
    for (int k = ii; k < nchunk; k += 32) {
      const float2 pp = *(const float2*)(partial + ((size_t)cj * nchunk + k) * 2);
      a += pp.x; b2 += pp.y;
      c3 += xsum[(size_t)cj * nchunk + k];
    }
    const int tp = min(t, 39);
    const float pv = (tp < 32 ? save + (size_t)(tp >> 3) * C : gamma)[min(cb * 8 + (tp & 7), C - 1)];
    a = half_wave_sum(a); b2 = half_wave_sum(b2); c3 = half_wave_sum(c3);
    if (ii == 0) { prm[40 + jj] = a; prm[48 + jj] = b2; prm[56 + jj] = c3; }
    if (t < 40) prm[t] = pv;
  }
  __syncthreads();
  float mean[8], invstd[8], sc[8], sh[8], gi[8], m1[8], m2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    mean[j] = prm[j]; invstd[j] = prm[8 + j]; sc[j] = prm[16 + j]; sh[j] = prm[24 + j];
    gi[j] = cb * 8 + j < C ? prm[32 + j] * invstd[j] : 0.f;
    m1[j] = prm[40 + j] * invN; m2[j] = prm[48 + j] * invN;
  }
  if (t < 8 && ch == 0 && cb * 8 + t < C) {
    if (dgamma) { dgamma[cb * 8 + t] = prm[48 + t]; dbeta[cb * 8 + t] = prm[40 + t]; }
    if (dbias) dbias[cb * 8 + t] = -(prm[32 + t] * prm[8 + t]) * (prm[48 + t] * invN) * prm[56 + t];      // sum of dy_raw over the batch
  }
  // chunk = b_per_chunk consecutive vectors of this channel block's flattened (batch item, pixel) space
  const int e_base = ch * b_per_chunk, b0 = 0;
  const int n = min(b_per_chunk, B * HW - e_base);
  for (int e0 = t; e0 < n; e0 += 1024) {
    u32x4 ry[4], rg[4];
    float gf[DYF32 ? 4 : 1][8];
    size_t vofs[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = min(e0 + q * 256, n - 1);
      const int bl = fdHW.div(e_base + e), pix = e_base + e - bl * HW;
      const size_t b = b0 + bl;
      vofs[q] = (b * C8 + cb) * HW + pix;
      ry[q] = y_raw[vofs[q]];
      if (DYF32) {
#pragma unroll
        for (int j = 0; j < 8; ++j) gf[q][j] = dy_f32[(b * C + min(cb * 8 + j, C - 1)) * HW + pix];
      } else {
        rg[q] = dy[vofs[q]];
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (e0 + q * 256 < n) {
        float g[8], yr[8], o[8];
        if (DYF32) {
#pragma unroll
          for (int j = 0; j < 8; ++j) g[j] = cb * 8 + j < C ? gf[q][j] : 0.f;
        } else {
          unpack8<DT>(rg[q], g);
        }
        unpack8<DT>(ry[q], yr);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float z = fmaf(yr[j], sc[j], sh[j]);
          const float dz = g[j] * (z > 0.f ? 1.f : slope);
          const float xh = (yr[j] - mean[j]) * invstd[j];
          o[j] = gi[j] * (dz - m1[j] - xh * m2[j]);
        }
        dyr[vofs[q]] = pack8<DT>(o);
      }
    }
  }
}

// Fused form for channel blocks with <= 256*NE vectors (every 1-D layer of the path at B*T <= 2048): one workgroup owns a
// channel block, keeps dz / x_hat in registers across the reduction and writes dy_raw, dgamma, dbeta and the bias gradient
// in ONE launch.
template <typename DT, bool DYF32, int NE, int NT>
__global__ __launch_bounds__(NT) void bn_bwd16_fused_kernel(const u32x4* __restrict__ dy, const float* __restrict__ dy_f32,
                                                             const u32x4* __restrict__ y_raw, const u32x4* __restrict__ y_out,
                                                             const float* __restrict__ save,
                                                             const float* __restrict__ gamma, u32x4* __restrict__ dyr, float* dbias,
                                                             float* dgamma, float* dbeta, int B, int C, int C8, int HW, float slope,
                                                             int sg) {
  prefetch_kernargs<192>();
  const FastDiv fdHW(HW, B * HW);
  __shared__ float red[(NT / 64) * 16];
  const int cb = blockIdx.x, t = threadIdx.x;
  const int n = B * HW;
  // (sg > 1, MS_DT_STAT_PAIR: B clips per statistics group, the groups one after the other; parameter gradients summed in group order)
  float tot_cs = 0.f, tot_s1 = 0.f, tot_s2 = 0.f;
  for (int grp = 0; grp < sg; ++grp) {
  if (grp) {
    __syncthreads();
    const size_t adv = (size_t)B * C8 * HW;
    save += 4 * C; dyr += adv; y_raw += adv;
    if (y_out) y_out += adv;
    if (DYF32) dy_f32 += (size_t)B * C * HW; else dy += adv;
  }
  // all loads first (clamped indices, no branches): one memory round trip for the whole kernel
  // (the 40 per-channel parameters travel through LDS: as scalar loads the compiler sinks them behind the data and waits
  // for them in four more round trips)
  // The data loads go out at once from the tensor the block normally reads: its output y (conv16.h: bn_inv_unsafe).  Whether one
  // of the 8 channels is unsafe to invert is known only once the parameters have arrived; such a block -- the exception -- then
  // reloads from y_raw.
  const u32x4* ysrc = y_out ? y_out : y_raw;
  u32x4 ry[NE], rg[NE];
  float gf[DYF32 ? NE : 1][8];
  size_t vofs[NE];
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int e = min(t + i * NT, n - 1);
    const int bl = fdHW.div(e), pix = e - bl * HW;
    vofs[i] = ((size_t)bl * C8 + cb) * HW + pix;
    ry[i] = ysrc[vofs[i]];
    if (DYF32) {
#pragma unroll
      for (int j = 0; j < 8; ++j) gf[i][j] = dy_f32[((size_t)bl * C + min(cb * 8 + j, C - 1)) * HW + pix];
    } else {
      rg[i] = dy[vofs[i]];
    }
  }
  __shared__ BnSrcShared kshared;
  __shared__ float gish[8];
  if (t < 8) gish[t] = cb * 8 + t < C ? gamma[cb * 8 + t] * save[C + cb * 8 + t] : 0.f;
  BnSrc ks[8];
  const bool raw = bn_src_block(kshared, save, cb, C, slope, y_out != nullptr, ks);      // (one barrier: covers gish too)
  float gi[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) gi[j] = gish[j];
  if (raw && y_out) {
#pragma unroll
    for (int i = 0; i < NE; ++i) ry[i] = y_raw[vofs[i]];
  }
  float dz[NE][8], xh[NE][8];
  float s12[16] = {};
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const bool ok = t + i * NT < n;
    float g[8], yr[8];
    if (DYF32) {
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] = cb * 8 + j < C ? gf[i][j] : 0.f;
    } else {
      unpack8<DT>(rg[i], g);
    }
    unpack8<DT>(ry[i], yr);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float xv; bool pos;
      bn_xhat_mask(ks[j], yr[j], xv, pos);
      dz[i][j] = ok ? g[j] * (pos ? 1.f : slope) : 0.f;
      xh[i][j] = ok ? xv : 0.f;
      s12[j] += dz[i][j];
      s12[8 + j] = fmaf(dz[i][j], xh[i][j], s12[8 + j]);
    }
  }
  block_sum_n<16, NT / 64>(s12, red);
  const float invN = 1.0f / (float)n;
  float cs[8] = {};
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      o[j] = gi[j] * (dz[i][j] - s12[j] * invN - xh[i][j] * (s12[8 + j] * invN));
      cs[j] += t + i * NT < n ? o[j] : 0.f;
    }
    if (t + i * NT < n) dyr[vofs[i]] = pack8<DT>(o);
  }
  block_sum_n<8, NT / 64>(cs, red);
  if (t < 8) {
    float csj = 0.f, s1j = 0.f, s2j = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (t == j) { csj = cs[j]; s1j = s12[j]; s2j = s12[8 + j]; }
    tot_cs = grp ? tot_cs + csj : csj; tot_s1 = grp ? tot_s1 + s1j : s1j; tot_s2 = grp ? tot_s2 + s2j : s2j;
  }
  }
  if (t < 8) {
    const int c = cb * 8 + t;
    if (c < C) {
      if (dbias) dbias[c] = tot_cs;
      if (dgamma) { dgamma[c] = tot_s2; dbeta[c] = tot_s1; }
    }
  }
}

int bwd16_chunks(int B, int C8, int HW, int* b_per_chunk) {
  // chunk = *b_per_chunk consecutive vectors of a channel block's B*HW (batch item, pixel) vectors.  ~2048 workgroups in all
  // (the 2-D layers stream tens of MB: 4 waves per CU do not hide the latency), at least one vector per thread
  const long V = (long)B * HW;
  long nchunk = std::max<long>(1, std::min<long>(cdiv(2048, std::max(1, C8)), (V + 255) / 256));
  long vpc = ((V + nchunk - 1) / nchunk + 255) / 256 * 256;
  nchunk = (V + vpc - 1) / vpc;
  if (b_per_chunk) *b_per_chunk = (int)vpc;
  return (int)nchunk;
}

int launch_bn_bwd16(int dt, const void* dy, const float* dy_f32, const void* y_raw, const void* y, const float* save, const float* gamma,
                    float* partial, void* dyr, float* colpart, float* dbias, float* dgamma, float* dbeta, int B, int C, int HW,
                    float slope, int* bias_done, hipStream_t s, int sg) {
  const int C8 = c8_of(C);
  *bias_done = 0;
  if (sg > 1) B /= sg;                  // MS_DT_STAT_PAIR: clips per statistics group (the one-launch form only)
  if (sg > 1 && (long)B * HW > BN_BWD16_FUSED_MAX) return set_error("bn_bwd16: statistics groups need the one-launch form");
  if ((long)B * HW <= BN_BWD16_FUSED_MAX) {
    *bias_done = 1;
    // 1024 threads per channel block when there is enough to share: few workgroups exist (C/8), so each one's latency counts
    const int n = B * HW;
    const int nt = 256;       // (1024-thread workgroups measured 2x slower here: 29 vs 15 us at C = 256, B*T = 2048)
    const int ne = (n + nt - 1) / nt;
    TimingScope ts(s, 0, 16.0 * (dy_f32 ? 4.0 + 1.0 + 1.0 : 3.0) * (double)B * C8 * HW, "bn_bwd16_fused_kernel|bn_bwd16 C%d N%d fused", C, B * HW);
    if (ts.skip()) return 0;
#define MS_BNF(DT, F, NE, NT)                                                                                                      \
    hipLaunchKernelGGL((bn_bwd16_fused_kernel<DT, F, NE, NT>), dim3(C8), dim3(NT), 0, s, (const u32x4*)dy, dy_f32, (const u32x4*)y_raw, \
                       (const u32x4*)y, save, gamma, (u32x4*)dyr, dbias, dgamma, dbeta, B, C, C8, HW, slope, sg)
#define MS_BNF_NE(DT, F) do { if (ne <= 1) MS_BNF(DT, F, 1, 256); else if (ne <= 2) MS_BNF(DT, F, 2, 256);                         \
                              else if (ne <= 4) MS_BNF(DT, F, 4, 256); else MS_BNF(DT, F, 8, 256); } while (0)
    if (dt == DT_BF16) { if (dy_f32) MS_BNF_NE(BF16, true); else MS_BNF_NE(BF16, false); }
    else { if (dy_f32) MS_BNF_NE(F16, true); else MS_BNF_NE(F16, false); }
#undef MS_BNF_NE
#undef MS_BNF
    return check_launch("bn_bwd16_fused_kernel");
  }
  int bpc;
  const int nchunk = bwd16_chunks(B, C8, HW, &bpc);
  const dim3 grid(C8, nchunk);
  const double vec = (double)B * C8 * HW;
  TimingScope ts(s, 0, 16.0 * (dy_f32 ? 4.0 + 2.0 + 1.0 : 5.0) * vec, "bn_bwd16_kernels|bn_bwd16 C%d N%d", C, B * HW);
  if (ts.skip()) return 0;
#define MS_BNB(DT, F)                                                                                                              \
  do {                                                                                                                             \
    hipLaunchKernelGGL((bn_bwd16_reduce_kernel<DT, F>), grid, dim3(256), 0, s, (const u32x4*)dy, dy_f32, (const u32x4*)y_raw, save,   \
                       partial, colpart, B, C, C8, HW, bpc, slope);                                                                \
    hipLaunchKernelGGL((bn_bwd16_apply_kernel<DT, F>), grid, dim3(256), 0, s, (const u32x4*)dy, dy_f32, (const u32x4*)y_raw, save,    \
                       gamma, partial, (u32x4*)dyr, colpart, dbias, dgamma, dbeta, B, C, C8, HW, bpc, slope);                      \
  } while (0)
  if (dt == DT_BF16) { if (dy_f32) MS_BNB(BF16, true); else MS_BNB(BF16, false); }
  else { if (dy_f32) MS_BNB(F16, true); else MS_BNB(F16, false); }
#undef MS_BNB
  *bias_done = 1;                       // (the apply pass wrote the bias gradient: no colsum16 launch behind it)
  return check_launch("bn_bwd16 kernels");
}

// activation backward for blocks without BN: mode 1 (LRELU): dyr = dy * (y>0 ? 1 : slope); mode 0 (BARE): dyr = dy,
// written only when dy arrives as plain fp32 (it is then also the conversion to cb8).  Per-channel colsum partials always.
template <typename DT, bool DYF32>
__global__ __launch_bounds__(256) void act_bwd16_kernel(const u32x4* __restrict__ dy, const float* __restrict__ dy_f32,
                                                        const u32x4* __restrict__ y, u32x4* __restrict__ dyr,
                                                        float* __restrict__ colpart, int B, int C, int C8, int HW, int b_per_chunk,
                                                        int mode, float slope) {
  prefetch_kernargs<192>();
  const FastDiv fdHW(HW, B * HW);
  __shared__ float red[32];
  const int cb = blockIdx.x, ch = blockIdx.y, t = threadIdx.x, nchunk = gridDim.y;
  // chunk = b_per_chunk consecutive vectors of this channel block's flattened (batch item, pixel) space
  const int e_base = ch * b_per_chunk, b0 = 0;
  const int n = min(b_per_chunk, B * HW - e_base);
  float cs[8] = {};
  for (int e = t; e < n; e += 256) {
    const int bl = fdHW.div(e_base + e), pix = e_base + e - bl * HW;
    const size_t b = b0 + bl, v = (b * C8 + cb) * HW + pix;
    float g[8];
    load_dy8<DT, DYF32>(dy, dy_f32, v, b, cb, pix, C, HW, g);
    if (mode == 1) {
      float yy[8];
      unpack8<DT>(y[v], yy);
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] *= (yy[j] > 0.f ? 1.f : slope);
    }
    if (mode == 1 || DYF32) dyr[v] = pack8<DT>(g);
#pragma unroll
    for (int j = 0; j < 8; ++j) cs[j] += g[j];
  }
  block_sum8(cs, red);
  if (t == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (cb * 8 + j < C) colpart[(size_t)(cb * 8 + j) * nchunk + ch] = cs[j];
  }
}

int launch_act_bwd16(int dt, const void* dy, const float* dy_f32, const void* y, void* dyr, float* colpart, int B, int C, int HW,
                     int mode, float slope, hipStream_t s) {
  const int C8 = c8_of(C);
  int bpc;
  const int nchunk = bwd16_chunks(B, C8, HW, &bpc);
  const dim3 grid(C8, nchunk);
  TimingScope ts(s, 0, 16.0 * 3.0 * (double)B * C8 * HW, "act_bwd16_kernel|act_bwd16 C%d N%d mode%d", C, B * HW, mode);
  if (ts.skip()) return 0;
#define MS_ACT(DT, F)                                                                                                          \
  hipLaunchKernelGGL((act_bwd16_kernel<DT, F>), grid, dim3(256), 0, s, (const u32x4*)dy, dy_f32, (const u32x4*)y, (u32x4*)dyr,   \
                     colpart, B, C, C8, HW, bpc, mode, slope)
  if (dt == DT_BF16) { if (dy_f32) MS_ACT(BF16, true); else MS_ACT(BF16, false); }
  else { if (dy_f32) MS_ACT(F16, true); else MS_ACT(F16, false); }
#undef MS_ACT
  return check_launch("act_bwd16_kernel");
}

// bias gradient = sum of the chunks' column sums: one wave per channel, lanes stride over the chunks, fixed-order wave sum
__global__ __launch_bounds__(256) void colsum16_kernel(const float* __restrict__ colpart, float* __restrict__ out, int C, int nchunk) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  float s = 0.f;
  for (int k = lane; k < nchunk; k += 64) s += colpart[(size_t)c * nchunk + k];
  s = wave_sum(s);
  if (lane == 0) out[c] = s;
}

int launch_colsum16(const float* colpart, float* out, int C, int nchunk, hipStream_t s) {
  hipLaunchKernelGGL(colsum16_kernel, dim3(cdiv(C, 4)), dim3(256), 0, s, colpart, out, C, nchunk);
  return check_launch("colsum16_kernel");
}

// eval BatchNorm folded into the conv: scale[c] = gamma/sqrt(var+eps), bias'[c] = (bias - mean)*scale + beta
__global__ void bn_fold_kernel(const float* __restrict__ bias, const float* __restrict__ g, const float* __restrict__ b,
                               const float* __restrict__ m, const float* __restrict__ v, float* __restrict__ scale,
                               float* __restrict__ bias_out, int C, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = g[c] * (1.0f / sqrtf(v[c] + eps));
  scale[c] = sc;
  bias_out[c] = fmaf((bias ? bias[c] : 0.f) - m[c], sc, b[c]);
}

int launch_bn_fold(const float* bias, const float* g, const float* b, const float* m, const float* v, float* scale,
                   float* bias_out, int C, float eps, hipStream_t s) {
  hipLaunchKernelGGL(bn_fold_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, bias, g, b, m, v, scale, bias_out, C, eps);
  return check_launch("bn_fold_kernel");
}

// ---------------------------------------------------------------------------------------------
// layout converters at the fp32 boundaries of the path
template <typename DT>
__global__ __launch_bounds__(256) void cb8_from_plain_kernel(const float* __restrict__ x, u32x4* __restrict__ y, int C, int C8,
                                                             int HW, size_t total) {
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (size_t)gridDim.x * 256) {
    const size_t bc = v / HW;
    const int cb = (int)(bc % C8), pix = (int)(v - bc * HW);
    const size_t b = bc / C8;
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = cb * 8 + j < C ? x[(b * C + cb * 8 + j) * HW + pix] : 0.f;
    y[v] = pack8<DT>(f);
  }
}

template <typename DT>
__global__ __launch_bounds__(256) void cb8_to_plain_kernel(const u32x4* __restrict__ x, float* __restrict__ y, int C, int C8, int HW,
                                                           size_t total) {
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (size_t)gridDim.x * 256) {
    const size_t bc = v / HW;
    const int cb = (int)(bc % C8), pix = (int)(v - bc * HW);
    const size_t b = bc / C8;
    float f[8];
    unpack8<DT>(x[v], f);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (cb * 8 + j < C) y[(b * C + cb * 8 + j) * HW + pix] = f[j];
  }
}

// time-major fp32 (B, T, C) -> cb8 (B, C8, T); velocity: v[t] = x[t] - x[t-1], v[0] = 0 (gan.py:47-52)
template <typename DT>
__global__ __launch_bounds__(256) void cb8_from_btc_kernel(const float* __restrict__ x, u32x4* __restrict__ y, int T, int C, int C8,
                                                           size_t total, int velocity) {
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (size_t)gridDim.x * 256) {
    const size_t bc = v / T;
    const int cb = (int)(bc % C8), t = (int)(v - bc * T);
    const size_t b = bc / C8;
    const float* row = x + (b * T + t) * C;
    const float* prow = x + (b * T + (t > 0 ? t - 1 : 0)) * C;
    // all 16 loads unconditional (clamped channel), then masked: one round trip instead of a load -> wait chain per channel
    float cur[8], prv[8], f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = min(cb * 8 + j, C - 1);
      cur[j] = row[c];
      prv[j] = velocity ? prow[c] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float val = cur[j];
      if (velocity) val = t > 0 ? val - prv[j] : 0.f;
      f[j] = cb * 8 + j < C ? val : 0.f;
    }
    y[v] = pack8<DT>(f);
  }
}

// cb8 (B, C8, T) -> time-major fp32 (B, T, C); velocity_bwd: dx[t] = dv[t] - dv[t+1] (dv[0] does not reach x: v[0] = 0)
template <typename DT>
__global__ __launch_bounds__(256) void cb8_to_btc_kernel(const u32x4* __restrict__ x, float* __restrict__ y, int T, int C, int C8,
                                                         size_t total, int velocity_bwd) {
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (size_t)gridDim.x * 256) {
    const size_t bc = v / T;
    const int cb = (int)(bc % C8), t = (int)(v - bc * T);
    const size_t b = bc / C8;
    float f[8];
    unpack8<DT>(x[v], f);
    if (velocity_bwd) {
      float nx[8] = {};
      if (t + 1 < T) unpack8<DT>(x[v + 1], nx);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = (t > 0 ? f[j] : 0.f) - nx[j];
    }
    float* row = y + (b * T + t) * C + cb * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (cb * 8 + j < C) row[j] = f[j];
  }
}

}  // namespace ms

using namespace ms;

extern "C" {

static int conv_blocks(size_t total) { return (int)std::min<size_t>((total + 255) / 256, 8192); }

int ms_cb8_from_plain(int dtype, const float* x, void* y, int B, int C, int HW, void* stream) {
  if (dtype != MS_BF16 && dtype != MS_F16) return set_error("ms_cb8_from_plain: dtype %d", dtype);
  if (!x || !y || B < 1 || C < 1 || HW < 1) return set_error("ms_cb8_from_plain: bad argument");
  const int C8 = c8_of(C);
  const size_t total = (size_t)B * C8 * HW;
  hipStream_t s = (hipStream_t)stream;
  TimingScope ts(s, 0, 6.0 * 8.0 * total, "cb8_from_plain_kernel|cb8_from_plain C%d N%d", C, B * HW);
  if (ts.skip()) return 0;
  if (dtype == MS_BF16) hipLaunchKernelGGL(cb8_from_plain_kernel<BF16>, dim3(conv_blocks(total)), dim3(256), 0, s, x, (u32x4*)y, C, C8, HW, total);
  else hipLaunchKernelGGL(cb8_from_plain_kernel<F16>, dim3(conv_blocks(total)), dim3(256), 0, s, x, (u32x4*)y, C, C8, HW, total);
  return check_launch("cb8_from_plain_kernel");
}

int ms_cb8_to_plain(int dtype, const void* x, float* y, int B, int C, int HW, void* stream) {
  if (dtype != MS_BF16 && dtype != MS_F16) return set_error("ms_cb8_to_plain: dtype %d", dtype);
  if (!x || !y || B < 1 || C < 1 || HW < 1) return set_error("ms_cb8_to_plain: bad argument");
  const int C8 = c8_of(C);
  const size_t total = (size_t)B * C8 * HW;
  hipStream_t s = (hipStream_t)stream;
  TimingScope ts(s, 0, 6.0 * 8.0 * total, "cb8_to_plain_kernel|cb8_to_plain C%d N%d", C, B * HW);
  if (ts.skip()) return 0;
  if (dtype == MS_BF16) hipLaunchKernelGGL(cb8_to_plain_kernel<BF16>, dim3(conv_blocks(total)), dim3(256), 0, s, (const u32x4*)x, y, C, C8, HW, total);
  else hipLaunchKernelGGL(cb8_to_plain_kernel<F16>, dim3(conv_blocks(total)), dim3(256), 0, s, (const u32x4*)x, y, C, C8, HW, total);
  return check_launch("cb8_to_plain_kernel");
}

int ms_cb8_from_btc(int dtype, const float* x, void* y, int B, int T, int C, int velocity, void* stream) {
  TimingScope ts((hipStream_t)stream, 0, 0, "ew|ew_cb8_from_btc");
  if (ts.skip()) return 0;
  if (dtype != MS_BF16 && dtype != MS_F16) return set_error("ms_cb8_from_btc: dtype %d", dtype);
  if (!x || !y || B < 1 || C < 1 || T < 1) return set_error("ms_cb8_from_btc: bad argument");
  const int C8 = c8_of(C);
  const size_t total = (size_t)B * C8 * T;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MS_BF16) hipLaunchKernelGGL(cb8_from_btc_kernel<BF16>, dim3(conv_blocks(total)), dim3(256), 0, s, x, (u32x4*)y, T, C, C8, total, velocity);
  else hipLaunchKernelGGL(cb8_from_btc_kernel<F16>, dim3(conv_blocks(total)), dim3(256), 0, s, x, (u32x4*)y, T, C, C8, total, velocity);
  return check_launch("cb8_from_btc_kernel");
}

int ms_cb8_to_btc(int dtype, const void* x, float* y, int B, int T, int C, int velocity_bwd, void* stream) {
  TimingScope ts((hipStream_t)stream, 0, 0, "ew|ew_cb8_to_btc");
  if (ts.skip()) return 0;
  if (dtype != MS_BF16 && dtype != MS_F16) return set_error("ms_cb8_to_btc: dtype %d", dtype);
  if (!x || !y || B < 1 || C < 1 || T < 1) return set_error("ms_cb8_to_btc: bad argument");
  const int C8 = c8_of(C);
  const size_t total = (size_t)B * C8 * T;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MS_BF16) hipLaunchKernelGGL(cb8_to_btc_kernel<BF16>, dim3(conv_blocks(total)), dim3(256), 0, s, (const u32x4*)x, y, T, C, C8, total, velocity_bwd);
  else hipLaunchKernelGGL(cb8_to_btc_kernel<F16>, dim3(conv_blocks(total)), dim3(256), 0, s, (const u32x4*)x, y, T, C, C8, total, velocity_bwd);
  return check_launch("cb8_to_btc_kernel");
}

}  // extern "C"
