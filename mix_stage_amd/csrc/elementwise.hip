// HBM-bound kernels of the path: BatchNorm statistics/apply/backward, activation backward, the
// AudioEncoder time-lerp, the softmax mixture of the M sub-generators, cross entropy, pose velocity,
// layout transposes, L1 losses, and the trainer tail (global grad norm + clipped Adam).
// All reductions run in a fixed order (no float atomics) -> bitwise reproducible.
#include <algorithm>

#include <stdint.h>

#include "kernels.h"
#include "conv16.h"

namespace ms {

// ----------------------------------------------------------------------------------------------
// BatchNorm (training): finalize tile partials -> mean/invstd/scale/shift + running stats
// stats: [n_tiles][C][2] (sum, M2 about tile mean); tile i holds min(tile_n, N - i*tile_n) values
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ stats,
                                                          const float* __restrict__ counts, int n_tiles, int tile_n,
                                                          int N, int C, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* running_mean,
                                                          float* running_var, float* __restrict__ save, float eps,
                                                          float momentum) {
  prefetch_kernargs<128>();
  __shared__ double red[4];
  const int c = blockIdx.x, t = threadIdx.x;
  double s = 0.0;
  for (int i = t; i < n_tiles; i += 256) s += (double)stats[((size_t)i * C + c) * 2];
  s = wave_sum_d(s);
  if ((t & 63) == 0) red[t >> 6] = s;
  __syncthreads();
  const double mean = (red[0] + red[1] + red[2] + red[3]) / (double)N;
  __syncthreads();
  double q = 0.0;
  for (int i = t; i < n_tiles; i += 256) {
    const float* st = stats + ((size_t)i * C + c) * 2;
    const int cnt = counts ? (int)counts[i] : min(tile_n, N - i * tile_n);
    const double d = (double)st[0] / (double)cnt - mean;
    q += (double)st[1] + (double)cnt * d * d;
  }
  q = wave_sum_d(q);
  if ((t & 63) == 0) red[t >> 6] = q;
  __syncthreads();
  if (t == 0) {
    const double m2 = red[0] + red[1] + red[2] + red[3];
    const float var = (float)(m2 / (double)N);
    const float invstd = 1.0f / sqrtf(var + eps);
    const float fmean = (float)mean;
    const float sc = gamma[c] * invstd;
    save[c] = fmean;
    save[C + c] = invstd;
    save[2 * C + c] = sc;
    save[3 * C + c] = beta[c] - fmean * sc;
    const float unbiased = N > 1 ? (float)(m2 / (double)(N - 1)) : var;
    running_stats_update(&running_mean[c], &running_var[c], running_mean[c], running_var[c], momentum, fmean, unbiased);
  }
}

// bn_finalize_kernel + bn_apply_kernel in ONE launch for layers with at most 1024 statistics tiles (every 1-D layer and every 2-D
// layer of the path but the first): each workgroup combines the per-tile partials of its channel itself -- the same operations in
// the same order as bn_finalize_kernel (thread t takes tiles t, t + 256, ...), so the statistics are bit-identical -- chunk 0
// records them, and every workgroup normalises its share of the channel (grid (C, chunks of batch items)).  All loads (partials,
// parameters, FA_PRE data vectors per thread) are issued before the first use.  VEC = 4: 16-byte loads and stores (HW % 4 == 0):
// the 2-D layers' 4-67 MB tensors move at the rate of the stand-alone bn_apply_kernel.
constexpr int FA_PRE = 8;
constexpr int FA_TILES = 4;          // partials per thread: n_tiles <= 1024
template <int VEC>
__global__ __launch_bounds__(256) void bn_finalize_apply_kernel(const float* __restrict__ stats, const float* __restrict__ counts,
                                                                int n_tiles, int tile_n, int N, int C,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* running_mean, float* running_var, float* __restrict__ save,
                                                                float eps, float momentum, const float* __restrict__ y_raw,
                                                                float* __restrict__ y, int B, int HW, int b_per_chunk, float slope) {
  prefetch_kernargs<128>();
  const int HWV = HW / VEC;
  const FastDiv fdHW(HWV, B * HWV);
  __shared__ double red[4];
  const int c = blockIdx.x, ch = blockIdx.y, t = threadIdx.x;
  const int b0 = ch * b_per_chunk, nb = min(b_per_chunk, B - b0);
  const int n = nb * HWV;                        // vectors of this workgroup
  float2 st[FA_TILES];
  float cntf[FA_TILES];
#pragma unroll
  for (int j = 0; j < FA_TILES; ++j) {
    const int tc = min(t + 256 * j, n_tiles - 1);
    st[j] = *(const float2*)(stats + ((size_t)tc * C + c) * 2);
    cntf[j] = counts ? counts[tc] : 0.f;
  }
  const float g = gamma[c], bt = beta[c], rm = running_mean[c], rv = running_var[c];
  __builtin_amdgcn_sched_barrier(0);
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  vec_t v[FA_PRE];
  size_t ofs[FA_PRE];
#pragma unroll
  for (int q = 0; q < FA_PRE; ++q) {
    const int e = min(t + q * 256, n - 1);
    const int bl = fdHW.div(e), pix = e - bl * HWV;
    ofs[q] = (((size_t)(b0 + bl) * C + c) * HWV + pix);
    v[q] = reinterpret_cast<const vec_t*>(y_raw)[ofs[q]];
  }
  __builtin_amdgcn_sched_barrier(0);
  double s = 0.0;
#pragma unroll
  for (int j = 0; j < FA_TILES; ++j)
    if (t + 256 * j < n_tiles) s += (double)st[j].x;
  s = wave_sum_d(s);
  if ((t & 63) == 0) red[t >> 6] = s;
  __syncthreads();
  const double mean = (red[0] + red[1] + red[2] + red[3]) / (double)N;
  __syncthreads();
  double q2 = 0.0;
#pragma unroll
  for (int j = 0; j < FA_TILES; ++j)
    if (t + 256 * j < n_tiles) {
      const int i = t + 256 * j;
      const int cnt = counts ? (int)cntf[j] : min(tile_n, N - i * tile_n);
      const double d = (double)st[j].x / (double)cnt - mean;
      q2 += (double)st[j].y + (double)cnt * d * d;
    }
  q2 = wave_sum_d(q2);
  if ((t & 63) == 0) red[t >> 6] = q2;
  __syncthreads();
  const double m2 = red[0] + red[1] + red[2] + red[3];
  const float var = (float)(m2 / (double)N);
  const float invstd = 1.0f / sqrtf(var + eps);
  const float fmean = (float)mean;
  const float sc = g * invstd;
  const float sh = bt - fmean * sc;
  if (t == 0 && ch == 0) {
    save[c] = fmean;
    save[C + c] = invstd;
    save[2 * C + c] = sc;
    save[3 * C + c] = sh;
    const float unbiased = N > 1 ? (float)(m2 / (double)(N - 1)) : var;
    running_stats_update(&running_mean[c], &running_var[c], rm, rv, momentum, fmean, unbiased);
  }
  auto act = [&](vec_t u) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) u[k] = lrelu(fmaf(u[k], sc, sh), slope);
    return u;
  };
#pragma unroll
  for (int q = 0; q < FA_PRE; ++q)
    if (t + q * 256 < n) reinterpret_cast<vec_t*>(y)[ofs[q]] = act(v[q]);
  for (int e0 = t + FA_PRE * 256; e0 < n; e0 += FA_PRE * 256) {
#pragma unroll
    for (int q = 0; q < FA_PRE; ++q) {
      const int e = min(e0 + q * 256, n - 1);
      const int bl = fdHW.div(e), pix = e - bl * HWV;
      ofs[q] = (((size_t)(b0 + bl) * C + c) * HWV + pix);
      v[q] = reinterpret_cast<const vec_t*>(y_raw)[ofs[q]];
    }
#pragma unroll
    for (int q = 0; q < FA_PRE; ++q)
      if (e0 + q * 256 < n) reinterpret_cast<vec_t*>(y)[ofs[q]] = act(v[q]);
  }
}

// y = lrelu(y_raw * scale[c] + shift[c]);  layout (B, C, HW)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ y_raw, float* __restrict__ y,
                                                       const float* __restrict__ save, int C, int HW, size_t total,
                                                       float slope) {
  prefetch_kernargs<128>();
  const float* scale = save + 2 * (size_t)C;
  const float* shift = save + 3 * (size_t)C;
  if ((HW & 3) == 0) {
    const size_t total4 = total >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
      const int c = (int)(((i << 2) / HW) % C);
      const float sc = scale[c], sh = shift[c];
      float4 v = reinterpret_cast<const float4*>(y_raw)[i];
      v.x = lrelu(fmaf(v.x, sc, sh), slope);
      v.y = lrelu(fmaf(v.y, sc, sh), slope);
      v.z = lrelu(fmaf(v.z, sc, sh), slope);
      v.w = lrelu(fmaf(v.w, sc, sh), slope);
      reinterpret_cast<float4*>(y)[i] = v;
    }
  } else {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
      const int c = (int)((i / HW) % C);
      y[i] = lrelu(fmaf(y_raw[i], scale[c], shift[c]), slope);
    }
  }
}

// ----------------------------------------------------------------------------------------------
// Cross-rank ("global") BatchNorm, data parallel: the statistics of a block are exchanged between the ranks so that every
// rank normalises with the statistics of the GLOBAL batch, like the single device the reference trains on (layers.py:65-70).
// bn_stats: per channel (sum, M2 about the local mean) of y_raw -- the per-tile format bn_finalize_kernel combines, so the
// all-gathered [world][C][2] array is finalized as `world` tiles of B*HW values each.
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ y_raw, float* __restrict__ out, int B, int C, int HW) {
  __shared__ float red[4];
  const int c = blockIdx.x, t = threadIdx.x, n = B * HW;
  float s = 0.f;
  for (int e = t; e < n; e += 256) {
    const int b = e / HW, pix = e - b * HW;
    s += y_raw[((size_t)b * C + c) * HW + pix];
  }
  s = block_sum_256(s, red);
  const float mean = s / (float)n;
  float q = 0.f;
  for (int e = t; e < n; e += 256) {
    const int b = e / HW, pix = e - b * HW;
    const float d = y_raw[((size_t)b * C + c) * HW + pix] - mean;
    q += d * d;
  }
  q = block_sum_256(q, red);
  if (t == 0) { out[2 * c] = s; out[2 * c + 1] = q; }
}

// sums the per-chunk partials of bn_bwd_reduce_kernel: sums[c] = (sum dz, sum dz*xhat) of this rank
__global__ void bn_bwd_sum_chunks_kernel(const float* __restrict__ partial, float* __restrict__ sums, int C, int nchunk) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s1 = 0.f, s2 = 0.f;
  for (int k = 0; k < nchunk; ++k) {
    s1 += partial[((size_t)c * nchunk + k) * 2];
    s2 += partial[((size_t)c * nchunk + k) * 2 + 1];
  }
  sums[2 * c] = s1; sums[2 * c + 1] = s2;
}

// dyr = gamma*invstd*(dz - S1/N - xhat*S2/N) with the GLOBAL sums S and the global count N
__global__ __launch_bounds__(256) void bn_bwd_apply_sums_kernel(const float* __restrict__ dy, const float* __restrict__ y_raw,
                                                                const float* __restrict__ save, const float* __restrict__ gamma,
                                                                const float* __restrict__ sums, float inv_n_global,
                                                                float* __restrict__ dyr, int B, int C, int HW, float slope) {
  const int c = blockIdx.x, t = threadIdx.x, n = B * HW;
  const float mean = save[c], invstd = save[C + c], sc = save[2 * C + c], sh = save[3 * C + c];
  const float gi = gamma[c] * invstd, m1 = sums[2 * c] * inv_n_global, m2 = sums[2 * c + 1] * inv_n_global;
  for (int e = t + blockIdx.y * 256; e < n; e += 256 * gridDim.y) {
    const int b = e / HW, pix = e - b * HW;
    const size_t off = ((size_t)b * C + c) * HW + pix;
    const float yr = y_raw[off];
    const float z = fmaf(yr, sc, sh);
    const float dz = dy[off] * (z > 0.f ? 1.f : slope);
    dyr[off] = gi * (dz - m1 - (yr - mean) * invstd * m2);
  }
}

// ----------------------------------------------------------------------------------------------
// BatchNorm + LeakyReLU backward.  grid (C, nchunk); chunk = contiguous range of batch items.
//   dz = dy * lrelu'(z), z = y_raw*scale+shift (bit-identical to forward);  xh = (y_raw-mean)*invstd
//   partial[c][chunk] = (sum dz, sum dz*xh)
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ y_raw,
                                                            const float* __restrict__ save, float* __restrict__ partial,
                                                            int B, int C, int HW, int b_per_chunk, float slope) {
  prefetch_kernargs<128>();
  __shared__ float red[4];
  const int c = blockIdx.x, ch = blockIdx.y, t = threadIdx.x;
  const int b0 = ch * b_per_chunk, nb = min(b_per_chunk, B - b0);
  const float mean = save[c], invstd = save[C + c], sc = save[2 * C + c], sh = save[3 * C + c];
  float s1 = 0.f, s2 = 0.f;
  const int n = nb * HW;
  for (int e = t; e < n; e += 256) {
    const int b = b0 + e / HW, pix = e % HW;
    const size_t off = ((size_t)b * C + c) * HW + pix;
    const float yr = y_raw[off];
    const float z = fmaf(yr, sc, sh);
    const float dz = dy[off] * (z > 0.f ? 1.f : slope);
    s1 += dz;
    s2 += dz * ((yr - mean) * invstd);
  }
  s1 = block_sum_256(s1, red);
  s2 = block_sum_256(s2, red);
  if (t == 0) {
    partial[((size_t)c * gridDim.y + ch) * 2] = s1;
    partial[((size_t)c * gridDim.y + ch) * 2 + 1] = s2;
  }
}

//   dyr = gamma*invstd*(dz - s1/N - xh*s2/N);  colsum partial of dyr;  dgamma = s2, dbeta = s1
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y_raw,
                                                           const float* __restrict__ save, const float* __restrict__ gamma,
                                                           const float* __restrict__ partial, float* __restrict__ dyr,
                                                           float* __restrict__ colpart, float* dgamma, float* dbeta,
                                                           int B, int C, int HW, int b_per_chunk, float slope) {
  prefetch_kernargs<128>();
  __shared__ float red[4];
  const int c = blockIdx.x, ch = blockIdx.y, t = threadIdx.x, nchunk = gridDim.y;
  float s1 = 0.f, s2 = 0.f;
  for (int k = 0; k < nchunk; ++k) {
    s1 += partial[((size_t)c * nchunk + k) * 2];
    s2 += partial[((size_t)c * nchunk + k) * 2 + 1];
  }
  const float invN = 1.0f / (float)((size_t)B * HW);
  const float mean = save[c], invstd = save[C + c], sc = save[2 * C + c], sh = save[3 * C + c];
  const float gi = gamma[c] * invstd, m1 = s1 * invN, m2 = s2 * invN;
  const int b0 = ch * b_per_chunk, nb = min(b_per_chunk, B - b0);
  const int n = nb * HW;
  float cs = 0.f;
  for (int e = t; e < n; e += 256) {
    const int b = b0 + e / HW, pix = e % HW;
    const size_t off = ((size_t)b * C + c) * HW + pix;
    const float yr = y_raw[off];
    const float z = fmaf(yr, sc, sh);
    const float dz = dy[off] * (z > 0.f ? 1.f : slope);
    const float xh = (yr - mean) * invstd;
    const float v = gi * (dz - m1 - xh * m2);
    dyr[off] = v;
    cs += v;
  }
  cs = block_sum_256(cs, red);
  if (t == 0) {
    colpart[(size_t)c * nchunk + ch] = cs;
    if (ch == 0 && dgamma) { dgamma[c] = s2; dbeta[c] = s1; }
  }
}

// The same two passes for rows of HW % 4 == 0 values: 16 bytes per lane and load, four pairs of loads in flight per thread (the
// 2-D layers stream 70-330 MB through these two launches; one dword and two integer divisions per element ran at 3.5 TB/s).
// `xsum` (C x chunks): the chunk's sum of x_hat -- the bias gradient is the sum of dy_raw = -gamma invstd mean(dz x_hat) sum(x_hat)
// (zero but for rounding: a conv bias in front of BatchNorm), which the apply pass then writes itself: no column-sum partials of
// dy_raw and no finalize launch behind it.
__global__ __launch_bounds__(256) void bn_bwd_reduce4_kernel(const float* __restrict__ dy, const float* __restrict__ y_raw,
                                                             const float* __restrict__ save, float* __restrict__ partial,
                                                             float* __restrict__ xsum, int B, int C, int HW, int b_per_chunk, float slope) {
  prefetch_kernargs<128>();
  __shared__ float red[4];
  const int c = blockIdx.x, ch = blockIdx.y, t = threadIdx.x;
  const int b0 = ch * b_per_chunk, nb = min(b_per_chunk, B - b0);
  const float mean = save[c], invstd = save[C + c], sc = save[2 * C + c], sh = save[3 * C + c];
  const int HW4 = HW >> 2, nv = nb * HW4;
  const FastDiv fd(HW4, nv + 1024);
  float s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (int e0 = t; e0 < nv; e0 += 1024) {
    float4 yr[4], g[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = min(e0 + 256 * u, nv - 1);
      const int b = fd.div(e), p4 = e - b * HW4;
      const size_t off = ((size_t)(b0 + b) * C + c) * HW + 4 * p4;
      yr[u] = *reinterpret_cast<const float4*>(y_raw + off);
      g[u] = *reinterpret_cast<const float4*>(dy + off);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (e0 + 256 * u < nv) {
        const float a[4] = {yr[u].x, yr[u].y, yr[u].z, yr[u].w}, d[4] = {g[u].x, g[u].y, g[u].z, g[u].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float z = fmaf(a[j], sc, sh);
          const float dz = d[j] * (z > 0.f ? 1.f : slope);
          const float xh = (a[j] - mean) * invstd;
          s1 += dz;
          s2 += dz * xh;
          s3 += xh;
        }
      }
    }
  }
  s1 = block_sum_256(s1, red);
  s2 = block_sum_256(s2, red);
  s3 = block_sum_256(s3, red);
  if (t == 0) {
    partial[((size_t)c * gridDim.y + ch) * 2] = s1;
    partial[((size_t)c * gridDim.y + ch) * 2 + 1] = s2;
    xsum[(size_t)c * gridDim.y + ch] = s3;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply4_kernel(const float* __restrict__ dy, const float* __restrict__ y_raw,
                                                            const float* __restrict__ save, const float* __restrict__ gamma,
                                                            const float* __restrict__ partial, float* __restrict__ dyr,
                                                            const float* __restrict__ xsum, float* dbias, float* dgamma, float* dbeta,
                                                            int B, int C, int HW, int b_per_chunk, float slope) {
  prefetch_kernargs<128>();
  const int c = blockIdx.x, ch = blockIdx.y, t = threadIdx.x, nchunk = gridDim.y;
  float s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (int k = 0; k < nchunk; ++k) {
    s1 += partial[((size_t)c * nchunk + k) * 2];
    s2 += partial[((size_t)c * nchunk + k) * 2 + 1];
    s3 += xsum[(size_t)c * nchunk + k];
  }
  const float invN = 1.0f / (float)((size_t)B * HW);
  const float mean = save[c], invstd = save[C + c], sc = save[2 * C + c], sh = save[3 * C + c];
  const float gi = gamma[c] * invstd, m1 = s1 * invN, m2 = s2 * invN;
  const int b0 = ch * b_per_chunk, nb = min(b_per_chunk, B - b0);
  const int HW4 = HW >> 2, nv = nb * HW4;
  const FastDiv fd(HW4, nv + 1024);
  for (int e0 = t; e0 < nv; e0 += 1024) {
    float4 yr[4], g[4];
    size_t off[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = min(e0 + 256 * u, nv - 1);
      const int b = fd.div(e), p4 = e - b * HW4;
      off[u] = ((size_t)(b0 + b) * C + c) * HW + 4 * p4;
      yr[u] = *reinterpret_cast<const float4*>(y_raw + off[u]);
      g[u] = *reinterpret_cast<const float4*>(dy + off[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (e0 + 256 * u < nv) {
        const float a[4] = {yr[u].x, yr[u].y, yr[u].z, yr[u].w}, d[4] = {g[u].x, g[u].y, g[u].z, g[u].w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float z = fmaf(a[j], sc, sh);
          const float dz = d[j] * (z > 0.f ? 1.f : slope);
          const float xh = (a[j] - mean) * invstd;
          o[j] = gi * (dz - m1 - xh * m2);
        }
        *reinterpret_cast<float4*>(dyr + off[u]) = float4{o[0], o[1], o[2], o[3]};
      }
    }
  }
  if (t == 0 && ch == 0) {
    if (dgamma) { dgamma[c] = s2; dbeta[c] = s1; }
    if (dbias) dbias[c] = -gi * m2 * s3;             // = the sum of dy_raw over the batch (as the clip kernels' EP_DGRAD_BN writes it)
  }
}

// Fused BatchNorm+LeakyReLU backward for channels with <= 256*NE values: one workgroup owns a channel, keeps dy / y_raw
// in registers across the reduction, and writes dy_raw, dgamma, dbeta and the bias gradient in one launch.
// `y` != NULL: the block's OUTPUT stands in for y_raw wherever the BatchNorm + LeakyReLU map inverts safely (conv16.h:
// bn_inv_unsafe, evaluated on the channel's save values exactly as the forward kernels do): z = y > 0 ? y : y / slope, x_hat =
// (z - beta) / gamma.  The in-launch forward forms (chain32 / clip32) then write y_raw only for the channels that fail the test.
template <int NE>
__global__ __launch_bounds__(256) void bn_bwd_fused_kernel(const float* __restrict__ dy, const float* __restrict__ y_raw,
                                                           const float* __restrict__ y, const float* __restrict__ save,
                                                           const float* __restrict__ gamma,
                                                           float* __restrict__ dyr, float* dbias, float* dgamma, float* dbeta,
                                                           int B, int C, int HW, float slope, int sg) {
  prefetch_kernargs<128>();
  const FastDiv fdHW(HW, B * HW);
  __shared__ float red[4];
  const int c = blockIdx.x, t = threadIdx.x;
  const int n = B * HW;
  // (sg > 1, MS_DT_STAT_PAIR: B clips per statistics group; the groups one after the other, the parameter gradients summed in
  // group order -- what two backward passes accumulate)
  float tot_cs = 0.f, tot_s1 = 0.f, tot_s2 = 0.f;
  for (int grp = 0; grp < sg; ++grp, save += 4 * C, dy += (size_t)B * C * HW, dyr += (size_t)B * C * HW,
           y_raw += (size_t)B * C * HW, y = y ? y + (size_t)B * C * HW : y) {
  // every load is issued before the first one is consumed (clamped indices instead of branches): the kernel pays one
  // memory round trip, not NE of them -- with one wave per SIMD nothing else hides that latency
  const float mean = save[c], invstd = save[C + c], sc = save[2 * C + c], sh = save[3 * C + c], gm = gamma[c];
  const bool from_y = y != nullptr && !bn_inv_unsafe(mean, invstd, sc, sh, slope);       // (uniform: one channel per workgroup)
  const float* src = from_y ? y : y_raw;
  float ry[NE], rg[NE];
  size_t ofs[NE];
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int e = min(t + i * 256, n - 1);
    const int b = fdHW.div(e), pix = e - b * HW;
    ofs[i] = ((size_t)b * C + c) * HW + pix;
    ry[i] = src[ofs[i]];
    rg[i] = dy[ofs[i]];
  }
  const float beta_c = fmaf(mean, sc, sh), inv_sl = 1.0f / slope, inv_g = invstd / sc;     // (from_y: slope, scale are not tiny)
  float dz[NE], xh[NE];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const bool ok = t + i * 256 < n;
    bool pos;
    float xv;
    if (from_y) {
      pos = ry[i] > 0.f;
      xv = ((pos ? ry[i] : ry[i] * inv_sl) - beta_c) * inv_g;
    } else {
      pos = fmaf(ry[i], sc, sh) > 0.f;
      xv = (ry[i] - mean) * invstd;
    }
    dz[i] = ok ? rg[i] * (pos ? 1.f : slope) : 0.f;
    xh[i] = ok ? xv : 0.f;
    s1 += dz[i];
    s2 += dz[i] * xh[i];
  }
  s1 = block_sum_256(s1, red);
  s2 = block_sum_256(s2, red);
  const float invN = 1.0f / (float)n;
  const float gi = gm * invstd, m1 = s1 * invN, m2 = s2 * invN;
  float cs = 0.f;
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    if (t + i * 256 < n) {
      const float v = gi * (dz[i] - m1 - xh[i] * m2);
      dyr[ofs[i]] = v;
      cs += v;
    }
  }
  cs = block_sum_256(cs, red);
  tot_cs = grp ? tot_cs + cs : cs; tot_s1 = grp ? tot_s1 + s1 : s1; tot_s2 = grp ? tot_s2 + s2 : s2;
  }
  if (t == 0) {
    if (dbias) dbias[c] = tot_cs;
    if (dgamma) { dgamma[c] = tot_s2; dbeta[c] = tot_s1; }
  }
}

// The same for rows of HW % 4 == 0 values: NV 16-byte vectors of dy and of y_raw per thread (channels of <= 1024 * NV values).
template <int NV>
__global__ __launch_bounds__(256) void bn_bwd_fused4_kernel(const float* __restrict__ dy, const float* __restrict__ y_raw,
                                                            const float* __restrict__ y, const float* __restrict__ save,
                                                            const float* __restrict__ gamma,
                                                            float* __restrict__ dyr, float* dbias, float* dgamma, float* dbeta,
                                                            int B, int C, int HW, float slope, int sg) {
  prefetch_kernargs<128>();
  const int HW4 = HW >> 2, nv = B * HW4;
  const FastDiv fd(HW4, nv + 256 * NV);
  __shared__ float red[4];
  const int c = blockIdx.x, t = threadIdx.x;
  float tot_cs = 0.f, tot_s1 = 0.f, tot_s2 = 0.f;          // (sg statistics groups of B clips each: bn_bwd_fused_kernel)
  for (int grp = 0; grp < sg; ++grp, save += 4 * C, dy += (size_t)B * C * HW, dyr += (size_t)B * C * HW,
           y_raw += (size_t)B * C * HW, y = y ? y + (size_t)B * C * HW : y) {
  const float mean = save[c], invstd = save[C + c], sc = save[2 * C + c], sh = save[3 * C + c], gm = gamma[c];
  const bool from_y = y != nullptr && !bn_inv_unsafe(mean, invstd, sc, sh, slope);       // (uniform: one channel per workgroup)
  const float* src = from_y ? y : y_raw;
  float4 ry[NV], rg[NV];
  size_t ofs[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int e = min(t + i * 256, nv - 1);
    const int b = fd.div(e), p4 = e - b * HW4;
    ofs[i] = ((size_t)b * C + c) * HW + 4 * p4;
    ry[i] = *reinterpret_cast<const float4*>(src + ofs[i]);
    rg[i] = *reinterpret_cast<const float4*>(dy + ofs[i]);
  }
  const float beta_c = fmaf(mean, sc, sh), inv_sl = 1.0f / slope, inv_g = invstd / sc;     // (from_y: slope, scale are not tiny)
  float dz[NV][4], xh[NV][4];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const bool ok = t + i * 256 < nv;
    const float a[4] = {ry[i].x, ry[i].y, ry[i].z, ry[i].w}, g[4] = {rg[i].x, rg[i].y, rg[i].z, rg[i].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bool pos;
      float xv;
      if (from_y) {
        pos = a[j] > 0.f;
        xv = ((pos ? a[j] : a[j] * inv_sl) - beta_c) * inv_g;
      } else {
        pos = fmaf(a[j], sc, sh) > 0.f;
        xv = (a[j] - mean) * invstd;
      }
      dz[i][j] = ok ? g[j] * (pos ? 1.f : slope) : 0.f;
      xh[i][j] = ok ? xv : 0.f;
      s1 += dz[i][j];
      s2 += dz[i][j] * xh[i][j];
    }
  }
  s1 = block_sum_256(s1, red);
  s2 = block_sum_256(s2, red);
  const float invN = 1.0f / (float)(B * HW);
  const float gi = gm * invstd, m1 = s1 * invN, m2 = s2 * invN;
  float cs = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (t + i * 256 < nv) {
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { o[j] = gi * (dz[i][j] - m1 - xh[i][j] * m2); cs += o[j]; }
      *reinterpret_cast<float4*>(dyr + ofs[i]) = float4{o[0], o[1], o[2], o[3]};
    }
  }
  cs = block_sum_256(cs, red);
  tot_cs = grp ? tot_cs + cs : cs; tot_s1 = grp ? tot_s1 + s1 : s1; tot_s2 = grp ? tot_s2 + s2 : s2;
  }
  if (t == 0) {
    if (dbias) dbias[c] = tot_cs;
    if (dgamma) { dgamma[c] = tot_s2; dbeta[c] = tot_s1; }
  }
}

// activation backward for blocks without BN: mode 1 (LRELU): dyr = dy * (y>0 ? 1 : slope); mode 0 (BARE): no
// write (dyr == dy).  Always emits per-channel colsum partials (the bias gradient).
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                      float* __restrict__ dyr, float* __restrict__ colpart, int B, int C,
                                                      int HW, int b_per_chunk, int mode, float slope) {
  prefetch_kernargs<128>();
  __shared__ float red[4];
  const int c = blockIdx.x, ch = blockIdx.y, t = threadIdx.x, nchunk = gridDim.y;
  const int b0 = ch * b_per_chunk, nb = min(b_per_chunk, B - b0);
  const int n = nb * HW;
  float cs = 0.f;
  for (int e = t; e < n; e += 256) {
    const int b = b0 + e / HW, pix = e % HW;
    const size_t off = ((size_t)b * C + c) * HW + pix;
    float v = dy[off];
    if (mode == 1) {
      v *= (y[off] > 0.f ? 1.f : slope);
      dyr[off] = v;
    }
    cs += v;
  }
  cs = block_sum_256(cs, red);
  if (t == 0) colpart[(size_t)c * nchunk + ch] = cs;
}

// The same for channels of <= 256 * NE values: one workgroup per channel, every load in flight at once, the bias gradient
// written directly (no column-sum partials, no colsum_finalize launch).
template <int NE>
__global__ __launch_bounds__(256) void act_bwd_fused_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                            float* __restrict__ dyr, float* __restrict__ dbias, int B, int C,
                                                            int HW, int mode, float slope) {
  prefetch_kernargs<128>();
  const FastDiv fdHW(HW, B * HW);
  __shared__ float red[4];
  const int c = blockIdx.x, t = threadIdx.x, n = B * HW;
  float g[NE], yv[NE];
  size_t ofs[NE];
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int e = min(t + i * 256, n - 1);
    const int b = fdHW.div(e), pix = e - b * HW;
    ofs[i] = ((size_t)b * C + c) * HW + pix;
    g[i] = dy[ofs[i]];
    yv[i] = mode == 1 ? y[ofs[i]] : 1.f;
  }
  float cs = 0.f;
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    if (t + i * 256 < n) {
      float v = g[i];
      if (mode == 1) {
        v *= (yv[i] > 0.f ? 1.f : slope);
        dyr[ofs[i]] = v;
      }
      cs += v;
    }
  }
  cs = block_sum_256(cs, red);
  if (t == 0 && dbias) dbias[c] = cs;
}

__global__ void colsum_finalize_kernel(const float* __restrict__ colpart, float* __restrict__ out, int C, int nchunk) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int k = 0; k < nchunk; ++k) s += colpart[(size_t)c * nchunk + k];
  out[c] = s;
}

// ----------------------------------------------------------------------------------------------
// AudioEncoder resize: bilinear (align_corners=False) of (Tin,F) -> (Tout,1)
__device__ inline void lerp_coords(int d, int in, int out, int& i0, int& i1, float& lam) {
  const float scale = (float)in / (float)out;
  float src = scale * ((float)d + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  lam = src - (float)i0;
}

__global__ void lerp_time_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int BC, int Tin, int F, int Tout) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= BC * Tout) return;
  const int bc = i / Tout, d = i - bc * Tout;
  int t0, t1, f0, f1;
  float lt, lf;
  lerp_coords(d, Tin, Tout, t0, t1, lt);
  lerp_coords(0, F, 1, f0, f1, lf);
  const float* xp = x + (size_t)bc * Tin * F;
  const float r0 = (1.f - lf) * xp[t0 * F + f0] + lf * xp[t0 * F + f1];
  const float r1 = (1.f - lf) * xp[t1 * F + f0] + lf * xp[t1 * F + f1];
  y[i] = (1.f - lt) * r0 + lt * r1;
}

// gather form of the backward: one thread per input element, loops over the Tout outputs
__global__ void lerp_time_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int BC, int Tin, int F, int Tout) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= BC * Tin * F) return;
  const int f = i % F, t = (i / F) % Tin, bc = i / (F * Tin);
  int f0, f1;
  float lf;
  lerp_coords(0, F, 1, f0, f1, lf);
  float wf = 0.f;
  if (f == f0) wf += 1.f - lf;
  if (f == f1) wf += lf;
  float acc = 0.f;
  if (wf != 0.f) {
    // outputs whose two source rows can include t: source coordinate (d + 0.5) * Tin/Tout - 0.5 in (t - 1, t + 1);
    // one extra output on either side covers rounding and the clamped borders (the exact test stays inside)
    const float r = (float)Tout / (float)Tin;
    const int d_lo = t == 0 ? 0 : max(0, (int)floorf(((float)t - 0.5f) * r - 0.5f) - 1);
    const int d_hi = t == Tin - 1 ? Tout - 1 : min(Tout - 1, (int)ceilf(((float)t + 1.5f) * r - 0.5f) + 1);
    for (int d = d_lo; d <= d_hi; ++d) {
      int t0, t1;
      float lt;
      lerp_coords(d, Tin, Tout, t0, t1, lt);
      float wt = 0.f;
      if (t == t0) wt += 1.f - lt;
      if (t == t1) wt += lt;
      if (wt != 0.f) acc += wt * dy[(size_t)bc * Tout + d];
    }
    acc *= wf;
  }
  dx[i] = acc;
}

// ----------------------------------------------------------------------------------------------
// softmax mixture of the M sub-generators.  One workgroup per (b, 64-step time tile); lane = time.
#define MIX_TT 64
__global__ __launch_bounds__(256) void softmax_mix_fwd_kernel(const float* __restrict__ z, const float* __restrict__ score,
                                                              float* __restrict__ soft, float* __restrict__ out, int M,
                                                              int P, int T, int FC) {
  // blockIdx.z: chunk of FC pose features (so the launch fills the chip); every chunk recomputes the softmax weights
  extern __shared__ float sm[];  // wsm[M][64] | tile[64][FC+1]
  float* wsm = sm;
  float* tile = sm + (size_t)M * MIX_TT;
  const int b = blockIdx.y, t0 = blockIdx.x * MIX_TT, t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int f_beg = blockIdx.z * FC, nf = min(FC, P - f_beg);
  const int tt = t0 + lane;
  const bool tv = tt < T;
  if (w == 0) {
    float mx = -INFINITY;
    for (int m = 0; m < M; ++m) mx = fmaxf(mx, tv ? score[((size_t)b * M + m) * T + tt] : 0.f);
    float den = 0.f;
    for (int m = 0; m < M; ++m) {
      const float e = expf((tv ? score[((size_t)b * M + m) * T + tt] : 0.f) - mx);
      wsm[m * MIX_TT + lane] = e;
      den += e;
    }
    const float inv = 1.f / den;
    for (int m = 0; m < M; ++m) {
      const float v = wsm[m * MIX_TT + lane] * inv;
      wsm[m * MIX_TT + lane] = v;
      if (tv && blockIdx.z == 0) soft[((size_t)b * T + tt) * M + m] = v;
    }
  }
  __syncthreads();
  for (int fl = w; fl < nf; fl += 4) {
    float acc = 0.f;
    for (int m = 0; m < M; ++m)
      acc += wsm[m * MIX_TT + lane] * (tv ? z[((size_t)b * M * P + (size_t)m * P + f_beg + fl) * T + tt] : 0.f);
    tile[lane * (FC + 1) + fl] = acc;
  }
  __syncthreads();
  const int nt = min(MIX_TT, T - t0);
  for (int e = t; e < nt * nf; e += 256) {
    const int r = e / nf, fl = e - r * nf;
    out[((size_t)b * T + t0 + r) * P + f_beg + fl] = tile[r * (FC + 1) + fl];
  }
}

// backward, stage 1: one workgroup per (time tile, b, sub-generator m); the 4 waves split the pose features.
//   dz[b,m,f,t] = soft[b,t,m] * dout[b,t,f];   dscore[b,m,t] <- ds[m] = sum_f dout[b,t,f] * z[b,m,f,t]  (raw, stage 2 finishes)
__global__ __launch_bounds__(256) void softmax_mix_bwd_kernel(const float* __restrict__ z, const float* __restrict__ soft,
                                                              const float* __restrict__ dout, float* __restrict__ dz,
                                                              float* __restrict__ dscore, int M, int P, int T) {
  extern __shared__ float sm[];  // dso[4][64] | tile[64][P+1]
  float* dso = sm;
  float* tile = sm + 4 * MIX_TT;
  const int b = blockIdx.y, m = blockIdx.z, t0 = blockIdx.x * MIX_TT, t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int tt = t0 + lane;
  const bool tv = tt < T;
  const int nt = min(MIX_TT, T - t0);
  for (int e = t; e < nt * P; e += 256) {
    const int r = e / P, f = e - r * P;
    tile[r * (P + 1) + f] = dout[((size_t)b * T + t0 + r) * P + f];
  }
  __syncthreads();
  const float sw = tv ? soft[((size_t)b * T + tt) * M + m] : 0.f;
  const int fq = (P + 3) / 4, f0 = w * fq, f1 = min(P, f0 + fq);
  float ds = 0.f;
  if (tv) {
    // 13 rows of z in flight per thread (one load after the other was a chain of 26 round trips: 18 us for 1.7 MB)
    constexpr int U = 13;
    for (int fb = f0; fb < f1; fb += U) {
      float zz[U];
#pragma unroll
      for (int u = 0; u < U; ++u) zz[u] = z[((size_t)b * M * P + (size_t)m * P + min(fb + u, f1 - 1)) * T + tt];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (fb + u < f1) {
          const int f = fb + u;
          const float g = tile[lane * (P + 1) + f];
          ds += g * zz[u];
          dz[((size_t)b * M * P + (size_t)m * P + f) * T + tt] = sw * g;
        }
    }
  }
  dso[w * MIX_TT + lane] = ds;
  __syncthreads();
  if (w == 0 && tv)
    dscore[((size_t)b * M + m) * T + tt] = (dso[lane] + dso[MIX_TT + lane]) + (dso[2 * MIX_TT + lane] + dso[3 * MIX_TT + lane]);
}

// stage 2, in place: dscore[b,m,t] = soft[m] * (ds[m] - sum_m' soft[m'] ds[m'])
__global__ void softmax_mix_bwd_finish_kernel(const float* __restrict__ soft, float* __restrict__ dscore, int B, int M, int T) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * T) return;
  const int b = i / T, tt = i - b * T;
  float dot = 0.f;
  for (int m = 0; m < M; ++m) dot += soft[((size_t)b * T + tt) * M + m] * dscore[((size_t)b * M + m) * T + tt];
  for (int m = 0; m < M; ++m) {
    const size_t o = ((size_t)b * M + m) * T + tt;
    dscore[o] = soft[((size_t)b * T + tt) * M + m] * (dscore[o] - dot);
  }
}

// ----------------------------------------------------------------------------------------------
// Pre-step ("next" row N1): k-means cluster labels of the raw pose and z-normalisation with joint removal, on device.
// labels[b,t] = argmin_m sum_d (c[m][d] - f[b,t][d])^2, f = [x | velocity(x)], x = pose[b,t,keep], in fp64 like the
// reference (transform.py:395-410); first minimum wins.  One wave per (b,t).
__global__ __launch_bounds__(256) void kmeans_labels_kernel(const float* __restrict__ pose, const int32_t* __restrict__ keep,
                                                            const double* __restrict__ centers, int64_t* __restrict__ labels,
                                                            int BT, int T, int P, int PK, int M, int feats) {
  // feature blocks of KMeans.get_feats (transform.py:352-378), in its order: pose (PK) | velocity (PK) | speed (PK/2)
  const bool f_pose = feats & 1, f_vel = feats & 2, f_speed = feats & 4;
  const int o_vel = f_pose ? PK : 0, o_speed = o_vel + (f_vel ? PK : 0), D = o_speed + (f_speed ? PK / 2 : 0);
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= BT) return;
  const int t = r % T;
  const float* xr = pose + (size_t)r * P;
  int best = 0;
  double bestd = 0.0;
  for (int m = 0; m < M; ++m) {
    const double* c = centers + (size_t)m * D;
    double acc = 0.0;
    if (f_pose | f_vel) {
      for (int d = lane; d < PK; d += 64) {
        const int col = keep[d];
        const double x = (double)xr[col];
        const double v = t > 0 ? x - (double)xr[col - P] : 0.0;
        if (f_pose) { const double dx = c[d] - x; acc += dx * dx; }
        if (f_vel) { const double dv = c[o_vel + d] - v; acc += dv * dv; }
      }
    }
    if (f_speed) {
      // speed of a joint = |(vx, vy)|: the x block and the y block of the kept columns are PK/2 apart
      for (int d = lane; d < PK / 2; d += 64) {
        const int cx = keep[d], cy = keep[PK / 2 + d];
        const double vx = t > 0 ? (double)xr[cx] - (double)xr[cx - P] : 0.0;
        const double vy = t > 0 ? (double)xr[cy] - (double)xr[cy - P] : 0.0;
        const double ds = c[o_speed + d] - sqrt(vx * vx + vy * vy);
        acc += ds * ds;
      }
    }
    acc = wave_sum_d(acc);
    if (m == 0 || acc < bestd) { bestd = acc; best = m; }
  }
  if (lane == 0) labels[r] = best;
}

__global__ __launch_bounds__(256) void znorm_select_kernel(const float* __restrict__ x, const int32_t* __restrict__ keep,
                                                           const double* __restrict__ mean, const double* __restrict__ inv_std,
                                                           float* __restrict__ y, size_t rows, int P, int PK) {
  const size_t total = rows * PK;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t r = i / PK;
    const int d = (int)(i - r * PK);
    const int col = keep ? keep[d] : d;
    y[i] = (float)(((double)x[r * P + col] - mean[col]) * inv_std[col]);
  }
}

// ----------------------------------------------------------------------------------------------
// Step metrics ("next" row N3): L1, VelL1 and PCK numerators of one batch, on device.  One workgroup per clip b;
// out[b][0] = sum |y - gt| over kept columns, out[b][1] = sum |vel(y) - vel(gt)|, out[b][2 + a*J + j] = PCK hits of
// joint j at alpha a (root joint moved to (0,0), poses de-normalised with std/mean; removed joints take gt values).
__global__ __launch_bounds__(256) void step_metrics_kernel(const float* __restrict__ ycap, const float* __restrict__ gt,
                                                           const int32_t* __restrict__ keep, const int32_t* __restrict__ slot_of,
                                                           const double* __restrict__ mean, const double* __restrict__ stdv,
                                                           const float* __restrict__ alphas, int n_alpha,
                                                           double* __restrict__ out, int T, int P, int PK) {
  extern __shared__ double sh[];                  // [2 + n_alpha*J] block accumulators + 8 per-wave partials
  const int b = blockIdx.x, t = threadIdx.x, J = P / 2;
  const int n_out = 2 + n_alpha * J;
  for (int i = t; i < n_out; i += 256) sh[i] = 0.0;
  __syncthreads();
  const float* yc = ycap + (size_t)b * T * PK;
  const float* g = gt + (size_t)b * T * P;
  // L1 / VelL1 over (t, kept column)
  double l1 = 0.0, v1 = 0.0;
  for (int e = t; e < T * PK; e += 256) {
    const int tt = e / PK, d = e - tt * PK, col = keep[d];
    const double y = (double)yc[(size_t)tt * PK + d], q = (double)g[(size_t)tt * P + col];
    l1 += fabs(y - q);
    if (tt > 0) {
      const double yp = (double)yc[(size_t)(tt - 1) * PK + d], qp = (double)g[(size_t)(tt - 1) * P + col];
      v1 += fabs((y - yp) - (q - qp));
    }
  }
  l1 = wave_sum_d(l1);
  v1 = wave_sum_d(v1);
  // PCK: one wave per time step (4 at a time), lanes over joints
  const int lane = t & 63, wv = t >> 6;
  for (int tt = wv; tt < T; tt += 4) {
    // de-normalised gt coordinates of this lane's joint (root forced to 0), and the pose extent
    double gx = 0.0, gy = 0.0, px = 0.0, py = 0.0;
    const bool jv = lane < J;
    if (jv && lane > 0) {
      gx = (double)g[(size_t)tt * P + lane] * stdv[lane] + mean[lane];
      gy = (double)g[(size_t)tt * P + J + lane] * stdv[J + lane] + mean[J + lane];
      const int sx = slot_of[lane], sy = slot_of[J + lane];        // column in the kept layout, -1: removed joint
      px = sx >= 0 ? (double)yc[(size_t)tt * PK + sx] * stdv[lane] + mean[lane] : gx;
      py = sy >= 0 ? (double)yc[(size_t)tt * PK + sy] * stdv[J + lane] + mean[J + lane] : gy;
    }
    double mxx = jv ? gx : -1e300, mnx = jv ? gx : 1e300, mxy = jv ? gy : -1e300, mny = jv ? gy : 1e300;
    for (int o = 32; o > 0; o >>= 1) {
      mxx = fmax(mxx, __shfl_xor(mxx, o)); mnx = fmin(mnx, __shfl_xor(mnx, o));
      mxy = fmax(mxy, __shfl_xor(mxy, o)); mny = fmin(mny, __shfl_xor(mny, o));
    }
    const double ext = fmax(mxx - mnx, mxy - mny);
    const double dist = sqrt((px - gx) * (px - gx) + (py - gy) * (py - gy));
    if (jv)
      for (int a = 0; a < n_alpha; ++a)
        if (dist < (double)alphas[a] * ext) atomicAdd(&sh[2 + a * J + lane], 1.0);     // LDS, integer-valued: order-free
  }
  if (lane == 0) { sh[n_out + wv] = l1; sh[n_out + 4 + wv] = v1; }        // fixed-order sum over the 4 waves
  __syncthreads();
  if (t < 2) sh[t] = (sh[n_out + 4 * t] + sh[n_out + 4 * t + 1]) + (sh[n_out + 4 * t + 2] + sh[n_out + 4 * t + 3]);
  __syncthreads();
  for (int i = t; i < n_out; i += 256) out[(size_t)b * n_out + i] = sh[i];
}

// ----------------------------------------------------------------------------------------------
// Evaluation accumulators ("next" row N3): FID sufficient statistics and the W1 speed / acceleration histograms of one batch,
// ADDED to running device buffers (no host round trip per step; the reference copies y_cap to the CPU after every step).
//   metrics.py:374-394  FID.__call__: rows = (b,t), columns = kept (xy, joint) of the NORMALISED poses:
//                        sum[c] += sum_rows v[r][c],  gram[c][d] += sum_rows v[r][c] * v[r][d]          (fp64, row order)
// grid (PK), block PK threads: thread d of block c walks the rows in order -- fixed summation order, coalesced over d.
__global__ void fid_accumulate_kernel(const float* __restrict__ ycap, const float* __restrict__ gt, const int32_t* __restrict__ keep,
                                      double* __restrict__ sums, double* __restrict__ gram, int rows, int P, int PK) {
  const int c = blockIdx.x, d = threadIdx.x, which = blockIdx.y;        // which: 0 = prediction, 1 = ground truth
  if (d >= PK) return;
  const int kc = keep[c], kd = keep[d];
  double g = 0.0, sm = 0.0;
  for (int r = 0; r < rows; ++r) {
    const double vc = which ? (double)gt[(size_t)r * P + kc] : (double)ycap[(size_t)r * PK + c];
    const double vd = which ? (double)gt[(size_t)r * P + kd] : (double)ycap[(size_t)r * PK + d];
    g = fma(vc, vd, g);
    sm += vd;
  }
  gram[((size_t)which * PK + c) * PK + d] += g;
  if (c == 0) sums[(size_t)which * PK + d] += sm;
}

//   metrics.py:476-520  W1: de-normalised poses (removed joints never enter), per frame transition the speed of every kept
//   joint sqrt(dx^2 + dy^2), averaged over the joints -> one value per (b, t); the same on second differences; histogram over
//   the edges k * width, k = 0 .. nbins (numpy.histogram with explicit edges: [e_k, e_k+1), last bin closed).
// One workgroup per clip; thread = frame transition; integer counts (atomicAdd: order-free).
__device__ inline int w1_bin(double v, double width, int nbins) {
  if (!(v >= 0.0) || v > (double)nbins * width) return -1;
  int k = (int)(v / width);
  if (k > nbins) k = nbins;
  while (k > 0 && v < (double)k * width) --k;                 // the edges are k * width as numpy.arange computes them
  while (k < nbins && v >= (double)(k + 1) * width) ++k;
  if (k >= nbins) k = v == (double)nbins * width ? nbins - 1 : -1;
  return k;
}
__global__ __launch_bounds__(256) void w1_accumulate_kernel(const float* __restrict__ ycap, const float* __restrict__ gt,
                                                            const int32_t* __restrict__ keep, const double* __restrict__ mean,
                                                            const double* __restrict__ stdv, unsigned long long* __restrict__ hist,
                                                            int T, int P, int PK, double width, int nbins) {
  const int b = blockIdx.x, which = blockIdx.y, JK = PK / 2;
  const float* yc = ycap + (size_t)b * T * PK;
  const float* g = gt + (size_t)b * T * P;
  auto val = [&](int t, int d) -> double {                  // de-normalised coordinate, d in the kept (xy, joint) order
    const int col = keep[d];
    const double v = which ? (double)g[(size_t)t * P + col] : (double)yc[(size_t)t * PK + d];
    return v * stdv[col] + mean[col];
  };
  for (int t = threadIdx.x; t + 1 < T; t += blockDim.x) {
    double sv = 0.0, sa = 0.0;
    for (int j = 0; j < JK; ++j) {
      const double x0 = val(t, j), x1 = val(t + 1, j), y0 = val(t, JK + j), y1 = val(t + 1, JK + j);
      const double vx = x1 - x0, vy = y1 - y0;
      sv += sqrt(vx * vx + vy * vy);
      if (t + 2 < T) {
        const double ax = (val(t + 2, j) - x1) - vx, ay = (val(t + 2, JK + j) - y1) - vy;
        sa += sqrt(ax * ax + ay * ay);
      }
    }
    const int kv = w1_bin(sv / JK, width, nbins);
    if (kv >= 0) atomicAdd(&hist[((size_t)which * 2 + 0) * nbins + kv], 1ull);
    if (t + 2 < T) {
      const int ka = w1_bin(sa / JK, width, nbins);
      if (ka >= 0) atomicAdd(&hist[((size_t)which * 2 + 1) * nbins + ka], 1ull);
    }
  }
}

// ----------------------------------------------------------------------------------------------
// content || style concat (JL:175-180) in channel-major layout: out[b,c,t] = c < C ? x[b,c,t] : E[ids[b,t]][c-C]
__global__ __launch_bounds__(256) void concat_style_fwd_kernel(const float* __restrict__ x, const float* __restrict__ emb,
                                                               const int64_t* __restrict__ ids, int ids_sb, int ids_st,
                                                               float* __restrict__ out, int C, int D, int T, size_t total) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int t = (int)(i % T);
    const size_t bc = i / T;
    const int c = (int)(bc % (C + D)), b = (int)(bc / (C + D));
    out[i] = c < C ? x[((size_t)b * C + c) * T + t] : emb[ids[(size_t)b * ids_sb + (size_t)t * ids_st] * D + (c - C)];
  }
}

// dx = dout[:, :C];  dE[s][j] = sum over (b,t) with ids[b,t] == s of dout[b, C+j, t]   (grid (D, S), fixed order)
__global__ __launch_bounds__(256) void concat_style_bwd_x_kernel(const float* __restrict__ dout, float* __restrict__ dx, int C,
                                                                 int D, int T, size_t total) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int t = (int)(i % T);
    const size_t bc = i / T;
    const int c = (int)(bc % C), b = (int)(bc / C);
    dx[i] = dout[((size_t)b * (C + D) + c) * T + t];
  }
}

__global__ __launch_bounds__(256) void concat_style_bwd_emb_kernel(const float* __restrict__ dout, const int64_t* __restrict__ ids,
                                                                   int ids_sb, int ids_st, float* __restrict__ demb, int B, int C,
                                                                   int D, int T) {
  __shared__ float red[4];
  const int j = blockIdx.x, sidx = blockIdx.y;
  float acc = 0.f;
  for (int e = threadIdx.x; e < B * T; e += 256) {
    const int b = e / T, t = e - b * T;
    if (ids[(size_t)b * ids_sb + (size_t)t * ids_st] == sidx) acc += dout[((size_t)b * (C + D) + C + j) * T + t];
  }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) demb[(size_t)sidx * D + j] = acc;
}

// loss weights (ms_loss_scale): a host constant and / or one float on the device, applied as torch applies `loss * w` and
// `grad * w` (fp32 products, in this order), so that folding them into the kernels changes no bit
__device__ inline float loss_weight(float v, float mul, const float* mul_dev) {
  if (mul != 1.0f) v *= mul;
  if (mul_dev) v *= mul_dev[0];
  return v;
}

// ----------------------------------------------------------------------------------------------
// cross entropy (mean over rows); single workgroup, fixed order
__global__ __launch_bounds__(1024) void cross_entropy_fwd_kernel(const float* __restrict__ score, const int64_t* __restrict__ target,
                                                                 float* __restrict__ loss, int n_outer, int n_inner, int C,
                                                                 int so, int sc, int si, float mul, const float* __restrict__ mul_dev) {
  __shared__ float red[16];
  const int rows = n_outer * n_inner;
  float acc = 0.f;
  for (int r = threadIdx.x; r < rows; r += 1024) {
    const int o = r / n_inner, i = r - o * n_inner;
    const float* sp = score + (size_t)o * so + (size_t)i * si;
    if (C <= 8) {
      // (the path's class counts: all of a row's scores and its target in flight at once, not one round trip per class)
      float v[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = sp[(size_t)min(c, C - 1) * sc];
      const int tg = (int)target[r];
      float mx = -INFINITY, den = 0.f, vt = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) if (c < C) mx = fmaxf(mx, v[c]);
#pragma unroll
      for (int c = 0; c < 8; ++c) if (c < C) { den += expf(v[c] - mx); vt = c == tg ? v[c] : vt; }
      acc += (logf(den) + mx) - vt;
      continue;
    }
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, sp[(size_t)c * sc]);
    float den = 0.f;
    for (int c = 0; c < C; ++c) den += expf(sp[(size_t)c * sc] - mx);
    acc += (logf(den) + mx) - sp[(size_t)target[r] * sc];
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < 16; ++w) s += red[w];          // fixed order
    loss[0] = loss_weight(s / (float)rows, mul, mul_dev);
  }
}

__global__ void cross_entropy_bwd_kernel(const float* __restrict__ score, const int64_t* __restrict__ target,
                                         const float* __restrict__ gscale, float* __restrict__ dscore, int n_outer,
                                         int n_inner, int C, int so, int sc, int si, int accumulate, float mul,
                                         const float* __restrict__ mul_dev) {
  const int rows = n_outer * n_inner;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const int o = r / n_inner, i = r - o * n_inner;
  const size_t base = (size_t)o * so + (size_t)i * si;
  const float* sp = score + base;
  float* dp = dscore + base;
  float mx = -INFINITY;
  for (int c = 0; c < C; ++c) mx = fmaxf(mx, sp[(size_t)c * sc]);
  float den = 0.f;
  for (int c = 0; c < C; ++c) den += expf(sp[(size_t)c * sc] - mx);
  const float g = loss_weight(gscale[0], mul, mul_dev) / (float)rows, inv = 1.f / den;
  const int tg = (int)target[r];
  for (int c = 0; c < C; ++c) {
    const float v = g * (expf(sp[(size_t)c * sc] - mx) * inv - (c == tg ? 1.f : 0.f));
    if (accumulate) dp[(size_t)c * sc] += v; else dp[(size_t)c * sc] = v;
  }
}

// ----------------------------------------------------------------------------------------------
// velocity (+ transpose to channel-major) and plain transposes, via a 32x32 LDS tile
// in: (B, R, Cc) row-major -> out: (B, Cc, R);  diff: out[b,c,r] = in[b,r,c] - in[b,r-1,c], out[.,.,0] = 0
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int Cc,
                                                        int diff) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* ip = in + (size_t)b * R * Cc;
  for (int j = ty; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + tx;
    float v = 0.f;
    if (r < R && c < Cc) {
      v = ip[(size_t)r * Cc + c];
      if (diff) v = r > 0 ? v - ip[(size_t)(r - 1) * Cc + c] : 0.f;
    }
    tile[j][tx] = v;
  }
  __syncthreads();
  float* op = out + (size_t)b * R * Cc;
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + tx;
    if (r < R && c < Cc) op[(size_t)c * R + r] = tile[tx][j];
  }
}

// dv (B,P,T) -> dx (B,T,P): dx[b,t,p] = (t>=1 ? dv[b,p,t] : 0) - (t+1<T ? dv[b,p,t+1] : 0)
__global__ __launch_bounds__(256) void velocity_bwd_kernel(const float* __restrict__ dv, float* __restrict__ dx, int T, int P) {
  __shared__ float tile[32][34];
  const int b = blockIdx.z, t0 = blockIdx.x * 32, p0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* ip = dv + (size_t)b * P * T;
  for (int j = ty; j < 32; j += 8) {
    const int p = p0 + j;
    for (int k = tx; k < 33; k += 32) {
      const int tt = t0 + k;
      tile[j][k] = (p < P && tt < T) ? ip[(size_t)p * T + tt] : 0.f;
    }
  }
  __syncthreads();
  float* op = dx + (size_t)b * T * P;
  for (int j = ty; j < 32; j += 8) {
    const int tt = t0 + j, p = p0 + tx;
    if (tt < T && p < P) {
      const float a = tt >= 1 ? tile[tx][j] : 0.f;
      const float c = tt + 1 < T ? tile[tx][j + 1] : 0.f;
      op[(size_t)tt * P + p] = a - c;
    }
  }
}

// ----------------------------------------------------------------------------------------------
// deterministic two-stage reductions
#define RED_MAX_BLOCKS 1024
template <int SQ>     // SQ = 0: |d| (L1Loss); 1: d*d (MSELoss)
__global__ __launch_bounds__(256) void l1_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, float target,
                                                         float* __restrict__ partials, size_t n) {
  __shared__ float red[4];
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float d = a[i] - (b ? b[i] : target);
    s += SQ ? d * d : fabsf(d);
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void sq_partial_kernel(const float* __restrict__ g, float* __restrict__ partials, size_t n) {
  __shared__ float red[4];
  float s = 0.f;
  if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
    // 16-byte loads, two in flight per thread; the tail (n % 4 elements) goes to the first threads
    const size_t n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4 a = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f};
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + stride < n4; i += 2 * stride) {
      const float4 u = g4[i], v = g4[i + stride];
      a.x = fmaf(u.x, u.x, a.x); a.y = fmaf(u.y, u.y, a.y); a.z = fmaf(u.z, u.z, a.z); a.w = fmaf(u.w, u.w, a.w);
      c.x = fmaf(v.x, v.x, c.x); c.y = fmaf(v.y, v.y, c.y); c.z = fmaf(v.z, v.z, c.z); c.w = fmaf(v.w, v.w, c.w);
    }
    if (i < n4) {
      const float4 u = g4[i];
      a.x = fmaf(u.x, u.x, a.x); a.y = fmaf(u.y, u.y, a.y); a.z = fmaf(u.z, u.z, a.z); a.w = fmaf(u.w, u.w, a.w);
    }
    s = ((a.x + c.x) + (a.y + c.y)) + ((a.z + c.z) + (a.w + c.w));
    const size_t t = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t < n) s = fmaf(g[t], g[t], s);
  } else {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += g[i] * g[i];
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// mode 0: out = sum * scale ; mode 1: out = sqrt(sum)
__global__ __launch_bounds__(256) void reduce_final_kernel(const float* __restrict__ partials, int n, float* out, float scale,
                                                           int mode, float mul = 1.0f, const float* __restrict__ mul_dev = nullptr) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)partials[i];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double tot = red[0] + red[1] + red[2] + red[3];
    out[0] = mode == 1 ? (float)sqrt(tot) : loss_weight((float)(tot * (double)scale), mul, mul_dev);
  }
}

template <int SQ>
__global__ __launch_bounds__(256) void l1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, float target,
                                                     const float* __restrict__ gscale, float* __restrict__ da, size_t n, float mul,
                                                     const float* __restrict__ mul_dev) {
  const float g = loss_weight(gscale[0], mul, mul_dev) / (float)n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float d = a[i] - (b ? b[i] : target);
    da[i] = SQ ? 2.f * g * d : (d > 0.f ? g : (d < 0.f ? -g : 0.f));
  }
}

// Two criterion terms of ONE tensor in one launch each way: halves [0, n) and [n, 2n) of `a` against the constants target0 /
// target1 with their own weights (gan.py:121,127: the D-step's fake and real terms on the paired discriminator pass).  n <= 2048
// per half: the two-stage form above then has ONE partial per term, and this kernel forms it in the same order -- same bits.
struct LossPair { float target[2], mul[2]; const float* mul_dev[2]; const float* gscale[2]; };
template <int SQ>
__global__ __launch_bounds__(256) void lp_pair_fwd_kernel(const float* __restrict__ a, float* __restrict__ loss, int n, const LossPair lp) {
  __shared__ float red[4];
  const int h = blockIdx.x;
  const float* ah = a + (size_t)h * n;
  const float target = lp.target[h];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float d = ah[i] - target;
    s += SQ ? d * d : fabsf(d);
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) loss[h] = loss_weight((float)((double)s * (double)(1.0f / (float)n)), lp.mul[h], lp.mul_dev[h]);
}
template <int SQ>
__global__ __launch_bounds__(256) void lp_pair_bwd_kernel(const float* __restrict__ a, float* __restrict__ da, int n, const LossPair lp) {
  const int h = blockIdx.y;
  const float g = lp.gscale[h] ? loss_weight(lp.gscale[h][0], lp.mul[h], lp.mul_dev[h]) / (float)n : 0.f;
  const float target = lp.target[h];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float d = a[(size_t)h * n + i] - target;
    da[(size_t)h * n + i] = SQ ? 2.f * g * d : (d > 0.f ? g : (d < 0.f ? -g : 0.f));
  }
}

// Adam (torch.optim.Adam defaults: no amsgrad, no weight decay) with clip_grad_norm_ folded in.
// state words: [0] step (int32), [1] clip coef (NaN: the gradient norm was not finite, the update is skipped), [2] step_size = lr/bc1, [3] sqrt(bc2)
__global__ void adam_prep_kernel(int32_t* state, const float* norm, float max_norm, float lr, float beta1, float beta2) {
  const int step = state[0] + 1;
  state[0] = step;
  float* f = reinterpret_cast<float*>(state);
  float coef = 1.f;
  if (norm) {
    coef = max_norm / (norm[0] + 1e-6f);
    if (coef > 1.f) coef = 1.f;
    if (!(fabsf(norm[0]) <= 3.0e38f)) coef = __builtin_nanf("");   // non-finite gradient: adam_kernel leaves p, m, v untouched
  }
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  f[1] = coef;
  f[2] = (float)((double)lr / bc1);
  f[3] = (float)sqrt(bc2);
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n, const int32_t* __restrict__ state,
                                                   float beta1, float beta2, float eps) {
  const float* f = reinterpret_cast<const float*>(state);
  const float coef = f[1], step_size = f[2], bc2s = f[3];
  if (coef != coef) return;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float gi = g[i] * coef;
    const float mi = m[i] + (1.f - beta1) * (gi - m[i]);
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= step_size * (mi / (sqrtf(vi) / bc2s + eps));
  }
}

// Segmented form: torch.optim.Adam keeps one step count PER PARAMETER, started when the parameter first receives a
// gradient, and skips parameters that never had one.  seg_first[s] = global step (1-based) of the segment's first
// gradient, or -1; seg_scratch[2s..2s+1] = (lr/bc1, sqrt(bc2)) of the segment for this step (0,0 = inactive).
__global__ void adam_prep_seg_kernel(int32_t* state, const float* norm, float max_norm, float lr, float beta1, float beta2,
                                     const int32_t* __restrict__ seg_first, float* __restrict__ seg_scratch, int n_seg) {
  const int step = state[0] + 1;
  for (int sidx = threadIdx.x; sidx < n_seg; sidx += blockDim.x) {
    const int first = seg_first[sidx];
    float ss = 0.f, b2 = 0.f;
    if (first >= 1 && first <= step) {
      const double t = (double)(step - first + 1);
      ss = (float)((double)lr / (1.0 - pow((double)beta1, t)));
      b2 = (float)sqrt(1.0 - pow((double)beta2, t));
    }
    seg_scratch[2 * sidx] = ss;
    seg_scratch[2 * sidx + 1] = b2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    state[0] = step;
    float coef = 1.f;
    if (norm) {
      coef = max_norm / (norm[0] + 1e-6f);
      if (coef > 1.f) coef = 1.f;
      // A non-finite gradient norm (a launch whose in-launch meeting timed out poisons its output with NaN, and the NaN reaches
      // every gradient behind it) must not reach the weights or the moments: the update of this step is SKIPPED on the device --
      // word 2 flags the step, word 3 counts such steps (MixStageTrainStep reads both with the losses); the step clocks advance
      // as if the step had had a zero gradient and frozen moments.
      const bool bad = !(fabsf(norm[0]) <= 3.0e38f);
      state[2] = bad ? 1 : 0;
      if (bad) state[3] += 1;
    }
    reinterpret_cast<float*>(state)[1] = coef;
  }
}

__global__ __launch_bounds__(256) void adam_seg_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, size_t n, const int32_t* __restrict__ state,
                                                       const int32_t* __restrict__ seg_of_chunk,
                                                       const float* __restrict__ seg_scratch, float beta1, float beta2,
                                                       float eps) {
  const float coef = reinterpret_cast<const float*>(state)[1];
  if (state[2]) return;                        // non-finite gradient norm: no update (adam_prep_seg_kernel)
  // 16 bytes per lane and stream, two vectors per thread in flight (7 streams over the 60 MB of live parameters: dword accesses
  // ran at 5.4 TB/s); a 64-element chunk belongs to one segment, so a vector does too.  n is a multiple of 64 (FlatAdam's layout);
  // the buffers are 256-byte aligned
  auto upd = [&](float pi, float gr, float mo, float vo, float step_size, float bc2s, float& mn, float& vn) {
    const float gi = gr * coef;
    mn = mo + (1.f - beta1) * (gi - mo);
    vn = beta2 * vo + (1.f - beta2) * gi * gi;
    return pi - step_size * (mn / (sqrtf(vn) / bc2s + eps));
  };
  if ((n & 3) == 0 && ((((size_t)p | (size_t)g | (size_t)m | (size_t)v) & 15) == 0)) {
    const size_t n4 = n >> 2, stride = (size_t)gridDim.x * 256;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += 2 * stride) {
      float4 pv[2], gv[2], mv[2], vv[2];
      float ss[2], bc[2];
      bool on[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const size_t i = i0 + u * stride;
        on[u] = i < n4;
        const int sidx = seg_of_chunk[min(i, n4 - 1) >> 4];
        ss[u] = seg_scratch[2 * sidx]; bc[u] = seg_scratch[2 * sidx + 1];
        on[u] = on[u] && bc[u] != 0.f;         // never received a gradient: torch.optim.Adam skips it
        if (on[u]) { pv[u] = p4[i]; gv[u] = g4[i]; mv[u] = m4[i]; vv[u] = v4[i]; }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (!on[u]) continue;
        const size_t i = i0 + u * stride;
        float4 mn, vn, pn;
        pn.x = upd(pv[u].x, gv[u].x, mv[u].x, vv[u].x, ss[u], bc[u], mn.x, vn.x);
        pn.y = upd(pv[u].y, gv[u].y, mv[u].y, vv[u].y, ss[u], bc[u], mn.y, vn.y);
        pn.z = upd(pv[u].z, gv[u].z, mv[u].z, vv[u].z, ss[u], bc[u], mn.z, vn.z);
        pn.w = upd(pv[u].w, gv[u].w, mv[u].w, vv[u].w, ss[u], bc[u], mn.w, vn.w);
        m4[i] = mn; v4[i] = vn; p4[i] = pn;
      }
    }
    return;
  }
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int sidx = seg_of_chunk[i >> 6];
    const float step_size = seg_scratch[2 * sidx], bc2s = seg_scratch[2 * sidx + 1];
    if (bc2s == 0.f) continue;                 // never received a gradient: torch.optim.Adam skips it
    float mn, vn;
    const float pn = upd(p[i], g[i], m[i], v[i], step_size, bc2s, mn, vn);
    m[i] = mn;
    v[i] = vn;
    p[i] = pn;
  }
}

static inline int red_blocks(size_t n) {
  size_t b = (n + 256 * 8 - 1) / (256 * 8);
  if (b < 1) b = 1;
  if (b > RED_MAX_BLOCKS) b = RED_MAX_BLOCKS;
  return (int)b;
}

// ---- launchers used by api.hip
int launch_bn_finalize(const float* stats, const float* counts, int n_tiles, int tile_n, int N, int C, const float* gamma,
                       const float* beta, float* rm, float* rv, float* save, float eps, float momentum, hipStream_t s) {
  TimingScope ts(s, 0, 8.0 * n_tiles * C, "bn_finalize C%d tiles%d", C, n_tiles);
  if (ts.skip()) return 0;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, s, stats, counts, n_tiles, tile_n, N, C, gamma, beta, rm, rv, save,
                     eps, momentum);
  return check_launch("bn_finalize_kernel");
}

// finalize + apply: one launch when the layer has at most 64 statistics tiles, else the two kernels
int launch_bn_finalize_apply(const float* stats, const float* counts, int n_tiles, int tile_n, int N, int C, const float* gamma,
                             const float* beta, float* rm, float* rv, float* save, float eps, float momentum, const float* y_raw,
                             float* y, int B, int HW, float slope, hipStream_t s) {
  static int fa_max = -1;                        // MS_BN_FA_TILES: largest tile count of the one-launch form (A/B runs; 64 = round 5)
  if (fa_max < 0) { const char* e = getenv("MS_BN_FA_TILES"); fa_max = e ? atoi(e) : 256 * FA_TILES; }
  if (n_tiles > std::min(fa_max, 256 * FA_TILES) || n_tiles < 1) {
    const int rc = launch_bn_finalize(stats, counts, n_tiles, tile_n, N, C, gamma, beta, rm, rv, save, eps, momentum, s);
    if (rc) return rc;
    return launch_bn_apply(y_raw, y, save, C, HW, (size_t)N * C, slope, s);
  }
  // chunks of batch items per channel: enough workgroups to fill the chip, and at most ~64 vectors per thread of one
  int bpc;
  int nchunk = bwd_chunks(B, C, &bpc);
  const bool vec = (HW & 3) == 0 && (((size_t)y_raw | (size_t)y) & 15) == 0;
  {
    const long per_item = vec ? HW / 4 : HW;
    const int want = (int)std::min<long>(B, std::max<long>(1, (per_item * B + 16383) / 16384));
    if (want > nchunk) { nchunk = want; bpc = cdiv(B, nchunk); nchunk = cdiv(B, bpc); }
  }
  TimingScope ts(s, 0, 8.0 * (double)N * C, "bn_finalize_apply C%d N%d tiles%d", C, N, n_tiles);
  if (ts.skip()) return 0;
  if (vec)
    hipLaunchKernelGGL(bn_finalize_apply_kernel<4>, dim3(C, nchunk), dim3(256), 0, s, stats, counts, n_tiles, tile_n, N, C, gamma, beta, rm,
                       rv, save, eps, momentum, y_raw, y, B, HW, bpc, slope);
  else
    hipLaunchKernelGGL(bn_finalize_apply_kernel<1>, dim3(C, nchunk), dim3(256), 0, s, stats, counts, n_tiles, tile_n, N, C, gamma, beta, rm,
                       rv, save, eps, momentum, y_raw, y, B, HW, bpc, slope);
  return check_launch("bn_finalize_apply_kernel");
}

int launch_bn_apply(const float* y_raw, float* y, const float* save, int C, int HW, size_t total, float slope, hipStream_t s) {
  const size_t work = (HW & 3) == 0 ? total / 4 : total;
  int blocks = (int)std::min<size_t>((work + 255) / 256, 4096);
  if (blocks < 1) blocks = 1;
  TimingScope ts(s, 0, 8.0 * total, "bn_apply C%d HW%d n%zu", C, HW, total);
  if (ts.skip()) return 0;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks), dim3(256), 0, s, y_raw, y, save, C, HW, total, slope);
  return check_launch("bn_apply_kernel");
}

int bwd_chunks(int B, int C, int* b_per_chunk) {
  int want = C >= 1024 ? 1 : (1024 + C - 1) / C;
  if (want > B) want = B;
  if (want < 1) want = 1;
  const int bpc = (B + want - 1) / want;
  *b_per_chunk = bpc;
  return (B + bpc - 1) / bpc;
}

// returns 1 if the fused single-launch form was used (dbias already final, no colsum_finalize needed)
int launch_bn_bwd(const float* dy, const float* y_raw, const float* y, const float* save, const float* gamma, float* partial, float* dyr,
                  float* colpart, float* dbias, float* dgamma, float* dbeta, int B, int C, int HW, float slope, int* fused,
                  hipStream_t s, int sg) {
  *fused = 0;
  if (sg > 1) B /= sg;                 // MS_DT_STAT_PAIR: clips per statistics group (the fused one-launch form only)
  const long n = (long)B * HW;
  if (sg > 1 && n > BN_BWD32_FUSED_MAX) return set_error("bn_bwd: statistics groups need the one-launch form (%ld values per channel)", n);
  if (n <= BN_BWD32_FUSED_MAX) {
    TimingScope ts(s, 0, 12.0 * sg * B * C * HW, "bn_bwd_fused C%d HW%d B%d%s", C, HW, B * sg, sg > 1 ? " pair" : "");
    if (ts.skip()) { *fused = 1; return 0; }
    const bool vec4 = (HW & 3) == 0 && (((uintptr_t)dy | (uintptr_t)y_raw | (uintptr_t)y | (uintptr_t)dyr) & 15) == 0;
    if (vec4 && n <= 1024)
      hipLaunchKernelGGL(bn_bwd_fused4_kernel<1>, dim3(C), dim3(256), 0, s, dy, y_raw, y, save, gamma, dyr, dbias, dgamma, dbeta, B, C, HW, slope, sg);
    else if (vec4 && n <= 2048)
      hipLaunchKernelGGL(bn_bwd_fused4_kernel<2>, dim3(C), dim3(256), 0, s, dy, y_raw, y, save, gamma, dyr, dbias, dgamma, dbeta, B, C, HW, slope, sg);
    else if (vec4)
      hipLaunchKernelGGL(bn_bwd_fused4_kernel<4>, dim3(C), dim3(256), 0, s, dy, y_raw, y, save, gamma, dyr, dbias, dgamma, dbeta, B, C, HW, slope, sg);
    else if (n <= 256 * 4)
      hipLaunchKernelGGL(bn_bwd_fused_kernel<4>, dim3(C), dim3(256), 0, s, dy, y_raw, y, save, gamma, dyr, dbias, dgamma, dbeta, B, C, HW, slope, sg);
    else if (n <= 256 * 8)
      hipLaunchKernelGGL(bn_bwd_fused_kernel<8>, dim3(C), dim3(256), 0, s, dy, y_raw, y, save, gamma, dyr, dbias, dgamma, dbeta, B, C, HW, slope, sg);
    else
      hipLaunchKernelGGL(bn_bwd_fused_kernel<16>, dim3(C), dim3(256), 0, s, dy, y_raw, y, save, gamma, dyr, dbias, dgamma, dbeta, B, C, HW, slope, sg);
    *fused = 1;
    return check_launch("bn_bwd_fused_kernel");
  }
  int bpc;
  const int nchunk = bwd_chunks(B, C, &bpc);
  TimingScope ts(s, 0, 20.0 * B * C * HW, "bn_bwd(reduce+apply) C%d HW%d B%d", C, HW, B);
  if (ts.skip()) return 0;
  const bool vec = (HW & 3) == 0 && (((uintptr_t)dy | (uintptr_t)y_raw | (uintptr_t)dyr) & 15) == 0;
  if (vec) hipLaunchKernelGGL(bn_bwd_reduce4_kernel, dim3(C, nchunk), dim3(256), 0, s, dy, y_raw, save, partial, colpart, B, C, HW, bpc, slope);
  else hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(C, nchunk), dim3(256), 0, s, dy, y_raw, save, partial, B, C, HW, bpc, slope);
  int rc = check_launch("bn_bwd_reduce_kernel");
  if (rc) return rc;
  if (vec) {
    // (the 16-byte form writes the bias gradient itself: *fused = 1 tells the caller that no colsum_finalize launch is needed)
    hipLaunchKernelGGL(bn_bwd_apply4_kernel, dim3(C, nchunk), dim3(256), 0, s, dy, y_raw, save, gamma, partial, dyr, colpart, dbias,
                       dgamma, dbeta, B, C, HW, bpc, slope);
    *fused = 1;
  } else {
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(C, nchunk), dim3(256), 0, s, dy, y_raw, save, gamma, partial, dyr, colpart,
                       dgamma, dbeta, B, C, HW, bpc, slope);
  }
  return check_launch("bn_bwd_apply_kernel");
}

// returns 1 in *fused when the bias gradient was written by this launch (no colsum_finalize needed)
int launch_act_bwd(const float* dy, const float* y, float* dyr, float* colpart, float* dbias, int B, int C, int HW, int mode, float slope,
                   int* fused, hipStream_t s) {
  *fused = 0;
  const long nn = (long)B * HW;
  if (nn <= 256 * 16 && (mode == 1 || dbias)) {
    TimingScope ts(s, 0, (mode == 1 ? 12.0 : 4.0) * B * C * HW, "act_bwd_fused C%d HW%d B%d mode%d", C, HW, B, mode);
    *fused = 1;
    if (ts.skip()) return 0;
    if (nn <= 256 * 4) hipLaunchKernelGGL(act_bwd_fused_kernel<4>, dim3(C), dim3(256), 0, s, dy, y, dyr, dbias, B, C, HW, mode, slope);
    else if (nn <= 256 * 8) hipLaunchKernelGGL(act_bwd_fused_kernel<8>, dim3(C), dim3(256), 0, s, dy, y, dyr, dbias, B, C, HW, mode, slope);
    else hipLaunchKernelGGL(act_bwd_fused_kernel<16>, dim3(C), dim3(256), 0, s, dy, y, dyr, dbias, B, C, HW, mode, slope);
    return check_launch("act_bwd_fused_kernel");
  }
  int bpc;
  const int nchunk = bwd_chunks(B, C, &bpc);
  TimingScope ts(s, 0, (mode == 1 ? 12.0 : 4.0) * B * C * HW, "act_bwd C%d HW%d B%d mode%d", C, HW, B, mode);
  if (ts.skip()) return 0;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(C, nchunk), dim3(256), 0, s, dy, y, dyr, colpart, B, C, HW, bpc, mode, slope);
  return check_launch("act_bwd_kernel");
}

int launch_colsum_finalize(const float* colpart, float* out, int B, int C, hipStream_t s) {
  int bpc;
  const int nchunk = bwd_chunks(B, C, &bpc);
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, colpart, out, C, nchunk);
  return check_launch("colsum_finalize_kernel");
}

}  // namespace ms

using namespace ms;

extern "C" {

int ms_lerp_time_fwd(const float* x, float* y, int B, int C, int Tin, int F, int Tout, void* stream) {
  TimingScope ts((hipStream_t)stream, 0, 0, "ew|ew_lerp_time_fwd");
  if (ts.skip()) return 0;
  const int n = B * C * Tout;
  hipLaunchKernelGGL(lerp_time_fwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, B * C, Tin, F, Tout);
  return check_launch("lerp_time_fwd_kernel");
}

int ms_lerp_time_bwd(const float* dy, float* dx, int B, int C, int Tin, int F, int Tout, void* stream) {
  TimingScope ts((hipStream_t)stream, 0, 0, "ew|ew_lerp_time_bwd");
  if (ts.skip()) return 0;
  const int n = B * C * Tin * F;
  hipLaunchKernelGGL(lerp_time_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dy, dx, B * C, Tin, F, Tout);
  return check_launch("lerp_time_bwd_kernel");
}

int ms_softmax_mix_fwd(const float* z, const float* score, float* soft, float* out, int B, int M, int P, int T, void* stream) {
  TimingScope ts((hipStream_t)stream, 0, 0, "ew|ew_softmax_mix_fwd");
  if (ts.skip()) return 0;
  // feature chunks: enough workgroups for the chip (B=32, T=64 alone gives 32)
  const int tiles = cdiv(T, MIX_TT) * B;
  int nch = std::max(1, std::min(P, 512 / std::max(1, tiles)));
  const int fc = cdiv(P, nch);
  nch = cdiv(P, fc);
  const size_t lds = ((size_t)M * MIX_TT + (size_t)MIX_TT * (fc + 1)) * sizeof(float);
  if (lds > 160 * 1024) return set_error("ms_softmax_mix_fwd: M=%d P=%d needs %zu B of LDS", M, P, lds);
  hipLaunchKernelGGL(softmax_mix_fwd_kernel, dim3(cdiv(T, MIX_TT), B, nch), dim3(256), lds, (hipStream_t)stream, z, score, soft,
                     out, M, P, T, fc);
  return check_launch("softmax_mix_fwd_kernel");
}

int ms_softmax_mix_bwd(const float* z, const float* soft, const float* dout, float* dz, float* dscore, int B, int M, int P,
                       int T, void* stream) {
  TimingScope ts((hipStream_t)stream, 0, 0, "ew|ew_softmax_mix_bwd");
  if (ts.skip()) return 0;
  const size_t lds = ((size_t)4 * MIX_TT + (size_t)MIX_TT * (P + 1)) * sizeof(float);
  if (lds > 160 * 1024) return set_error("ms_softmax_mix_bwd: M=%d P=%d needs %zu B of LDS", M, P, lds);
  if (M > 65535) return set_error("ms_softmax_mix_bwd: M=%d", M);
  hipLaunchKernelGGL(softmax_mix_bwd_kernel, dim3(cdiv(T, MIX_TT), B, M), dim3(256), lds, (hipStream_t)stream, z, soft, dout, dz,
                     dscore, M, P, T);
  int rc = check_launch("softmax_mix_bwd_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(softmax_mix_bwd_finish_kernel, dim3(cdiv(B * T, 256)), dim3(256), 0, (hipStream_t)stream, soft, dscore, B, M, T);
  return check_launch("softmax_mix_bwd_finish_kernel");
}

int ms_bn_stats(const float* y_raw, float* stats, int B, int C, int HW, void* stream) {
  TimingScope ts((hipStream_t)stream, 0, 0, "ew|ew_bn_stats");
  if (ts.skip()) return 0;
  if (!y_raw || !stats || B < 1 || C < 1 || HW < 1) return set_error("ms_bn_stats: bad argument");
  hipLaunchKernelGGL(bn_stats_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, y_raw, stats, B, C, HW);
  return check_launch("bn_stats_kernel");
}

int ms_bn_train_apply(const float* stats_all, int world, int n_local, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, const float* y_raw, float* y, float* save, int B, int C, int HW, float eps,
                      float momentum, float slope, void* stream) {
  TimingScope ts((hipStream_t)stream, 0, 0, "ew|ew_bn_train_apply");
  if (ts.skip()) return 0;
  if (!stats_all || world < 1 || n_local != B * HW || !gamma || !beta || !running_mean || !running_var || !y_raw || !y || !save)
    return set_error("ms_bn_train_apply: bad argument");
  hipStream_t s = (hipStream_t)stream;
  int rc = launch_bn_finalize(stats_all, nullptr, world, n_local, world * n_local, C, gamma, beta, running_mean, running_var, save, eps,
                              momentum, s);
  if (rc) return rc;
  return launch_bn_apply(y_raw, y, save, C, HW, (size_t)B * C * HW, slope, s);
}

size_t ms_bn_bwd_workspace(int B, int C) {
  int bpc;
  return (size_t)bwd_chunks(B, C, &bpc) * C * 2 * sizeof(float) + 256;
}

int ms_bn_bwd_sums(const float* dy, const float* y_raw, const float* save, float* sums, int B, int C, int HW, float slope,
                   void* workspace, size_t workspace_bytes, void* stream) {
  TimingScope ts((hipStream_t)stream, 0, 0, "ew|ew_bn_bwd_sums");
  if (ts.skip()) return 0;
  if (!dy || !y_raw || !save || !sums || !workspace || workspace_bytes < ms_bn_bwd_workspace(B, C)) return set_error("ms_bn_bwd_sums: bad argument");
  int bpc;
  const int nchunk = bwd_chunks(B, C, &bpc);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(C, nchunk), dim3(256), 0, s, dy, y_raw, save, (float*)workspace, B, C, HW, bpc, slope);
  hipLaunchKernelGGL(bn_bwd_sum_chunks_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, (const float*)workspace, sums, C, nchunk);
  return check_launch("bn_bwd_sums kernels");
}

int ms_bn_bwd_apply(const float* dy, const float* y_raw, const float* save, const float* gamma, const float* sums_global,
                    double n_global, float* dyr, int B, int C, int HW, float slope, void* stream) {
  TimingScope ts((hipStream_t)stream, 0, 0, "ew|ew_bn_bwd_apply");
  if (ts.skip()) return 0;
  if (!dy || !y_raw || !save || !gamma || !sums_global || !dyr || n_global < 1) return set_error("ms_bn_bwd_apply: bad argument");
  const int gy = std::max(1, std::min(64, (B * HW + 2047) / 2048));
  hipLaunchKernelGGL(bn_bwd_apply_sums_kernel, dim3(C, gy), dim3(256), 0, (hipStream_t)stream, dy, y_raw, save, gamma, sums_global,
                     (float)(1.0 / n_global), dyr, B, C, HW, slope);
  return check_launch("bn_bwd_apply_sums_kernel");
}

int ms_kmeans_labels(const float* pose, const int32_t* keep, const double* centers, int64_t* labels, int B, int T, int P,
                     int PK, int M, int feats, void* stream) {
  TimingScope ts((hipStream_t)stream, 0, 0, "ew|ew_kmeans_labels");
  if (ts.skip()) return 0;
  if (!pose || !keep || !centers || !labels || B < 1 || T < 1 || PK < 1 || M < 1) return set_error("ms_kmeans_labels: bad argument");
  if (!(feats & 7) || (feats & ~7) || ((feats & 4) && (PK & 1))) return set_error("ms_kmeans_labels: feats must be a subset of pose|velocity|speed (1|2|4)");
  const int BT = B * T;
  hipLaunchKernelGGL(kmeans_labels_kernel, dim3(cdiv(BT, 4)), dim3(256), 0, (hipStream_t)stream, pose, keep, centers, labels, BT,
                     T, P, PK, M, feats);
  return check_launch("kmeans_labels_kernel");
}

int ms_znorm_select(const float* x, const int32_t* keep, const double* mean, const double* inv_std, float* y, size_t rows, int P,
                    int PK, void* stream) {
  TimingScope ts((hipStream_t)stream, 0, 0, "ew|ew_znorm_select");
  if (ts.skip()) return 0;
  int blocks = (int)std::min<size_t>((rows * PK + 255) / 256, 2048);
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(znorm_select_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, keep, mean, inv_std, y, rows, P, PK);
  return check_launch("znorm_select_kernel");
}

int ms_step_metrics(const float* ycap, const float* gt, const int32_t* keep, const int32_t* slot_of, const double* mean,
                    const double* stdv, const float* alphas, int n_alpha, double* out, int B, int T, int P, int PK, void* stream) {
  TimingScope ts((hipStream_t)stream, 0, 0, "ew|ew_step_metrics");
  if (ts.skip()) return 0;
  if (P / 2 > 64) return set_error("ms_step_metrics: more than 64 joints");
  const size_t lds = (size_t)(2 + n_alpha * (P / 2) + 8) * sizeof(double);
  hipLaunchKernelGGL(step_metrics_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, ycap, gt, keep, slot_of, mean, stdv, alphas,
                     n_alpha, out, T, P, PK);
  return check_launch("step_metrics_kernel");
}

int ms_eval_accumulate(const float* ycap, const float* gt, const int32_t* keep, const double* mean, const double* stdv,
                       double* fid_sums, double* fid_gram, unsigned long long* w1_hist, int B, int T, int P, int PK,
                       double bin_width, int nbins, void* stream) {
  TimingScope ts((hipStream_t)stream, 0, 0, "ew|ew_eval_accumulate");
  if (ts.skip()) return 0;
  if (!ycap || !gt || !keep) return set_error("ms_eval_accumulate: null tensor");
  if (PK < 2 || PK > 1024 || (PK & 1) || PK > P) return set_error("ms_eval_accumulate: %d kept columns", PK);
  hipStream_t s = (hipStream_t)stream;
  if (fid_sums && fid_gram) {
    hipLaunchKernelGGL(fid_accumulate_kernel, dim3(PK, 2), dim3(((PK + 63) / 64) * 64), 0, s, ycap, gt, keep, fid_sums, fid_gram,
                       B * T, P, PK);
    const int rc = check_launch("fid_accumulate_kernel");
    if (rc) return rc;
  }
  if (w1_hist) {
    if (!mean || !stdv || nbins < 1 || !(bin_width > 0.0)) return set_error("ms_eval_accumulate: W1 needs mean / std / bins");
    hipLaunchKernelGGL(w1_accumulate_kernel, dim3(B, 2), dim3(256), 0, s, ycap, gt, keep, mean, stdv, w1_hist, T, P, PK, bin_width,
                       nbins);
    return check_launch("w1_accumulate_kernel");
  }
  return 0;
}

int ms_concat_style_fwd(const float* x, const float* emb, const int64_t* ids, int ids_stride_b, int ids_stride_t, float* out,
                        int B, int C, int D, int T, void* stream) {
  TimingScope ts((hipStream_t)stream, 0, 0, "ew|ew_concat_style_fwd");
  if (ts.skip()) return 0;
  const size_t total = (size_t)B * (C + D) * T;
  int blocks = (int)std::min<size_t>((total + 255) / 256, 2048);
  hipLaunchKernelGGL(concat_style_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, emb, ids, ids_stride_b,
                     ids_stride_t, out, C, D, T, total);
  return check_launch("concat_style_fwd_kernel");
}

int ms_concat_style_bwd(const float* dout, const int64_t* ids, int ids_stride_b, int ids_stride_t, float* dx, float* demb,
                        int B, int C, int D, int T, int S, void* stream) {
  TimingScope ts((hipStream_t)stream, 0, 0, "ew|ew_concat_style_bwd");
  if (ts.skip()) return 0;
  if (dx) {
    const size_t total = (size_t)B * C * T;
    int blocks = (int)std::min<size_t>((total + 255) / 256, 2048);
    hipLaunchKernelGGL(concat_style_bwd_x_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dout, dx, C, D, T, total);
    int rc = check_launch("concat_style_bwd_x_kernel");
    if (rc) return rc;
  }
  if (demb) {
    hipLaunchKernelGGL(concat_style_bwd_emb_kernel, dim3(D, S), dim3(256), 0, (hipStream_t)stream, dout, ids, ids_stride_b,
                       ids_stride_t, demb, B, C, D, T);
    return check_launch("concat_style_bwd_emb_kernel");
  }
  return 0;
}

static int cross_entropy_fwd(const float* score, const int64_t* target, float* loss, int n_outer, int n_inner, int C,
                             int stride_outer, int stride_c, int stride_inner, void* stream, const ms_loss_scale* ls) {
  TimingScope ts((hipStream_t)stream, 0, 0, "ew|ew_cross_entropy_fwd");
  if (ts.skip()) return 0;
  hipLaunchKernelGGL(cross_entropy_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, score, target, loss, n_outer, n_inner,
                     C, stride_outer, stride_c, stride_inner, ls ? ls->scale : 1.0f, ls ? ls->scale_dev : nullptr);
  return check_launch("cross_entropy_fwd_kernel");
}
static int cross_entropy_bwd(const float* score, const int64_t* target, const float* gscale, float* dscore, int n_outer, int n_inner,
                             int C, int stride_outer, int stride_c, int stride_inner, int accumulate, void* stream,
                             const ms_loss_scale* ls) {
  TimingScope ts((hipStream_t)stream, 0, 0, "ew|ew_cross_entropy_bwd");
  if (ts.skip()) return 0;
  const int rows = n_outer * n_inner;
  hipLaunchKernelGGL(cross_entropy_bwd_kernel, dim3(cdiv(rows, 64)), dim3(64), 0, (hipStream_t)stream, score, target, gscale,
                     dscore, n_outer, n_inner, C, stride_outer, stride_c, stride_inner, accumulate, ls ? ls->scale : 1.0f,
                     ls ? ls->scale_dev : nullptr);
  return check_launch("cross_entropy_bwd_kernel");
}

int ms_cross_entropy_fwd(const float* score, const int64_t* target, float* loss, float* row_scratch, int n_outer, int n_inner,
                         int C, int stride_outer, int stride_c, int stride_inner, void* stream) {
  (void)row_scratch;
  return cross_entropy_fwd(score, target, loss, n_outer, n_inner, C, stride_outer, stride_c, stride_inner, stream, nullptr);
}
int ms_cross_entropy_fwd_ex(const float* score, const int64_t* target, float* loss, int n_outer, int n_inner, int C,
                            int stride_outer, int stride_c, int stride_inner, void* stream, const ms_loss_scale* ls) {
  return cross_entropy_fwd(score, target, loss, n_outer, n_inner, C, stride_outer, stride_c, stride_inner, stream, ls);
}

int ms_cross_entropy_bwd(const float* score, const int64_t* target, const float* gscale, float* dscore, int n_outer, int n_inner,
                         int C, int stride_outer, int stride_c, int stride_inner, int accumulate, void* stream) {
  return cross_entropy_bwd(score, target, gscale, dscore, n_outer, n_inner, C, stride_outer, stride_c, stride_inner, accumulate,
                           stream, nullptr);
}
int ms_cross_entropy_bwd_ex(const float* score, const int64_t* target, const float* gscale, float* dscore, int n_outer, int n_inner,
                            int C, int stride_outer, int stride_c, int stride_inner, int accumulate, void* stream,
                            const ms_loss_scale* ls) {
  return cross_entropy_bwd(score, target, gscale, dscore, n_outer, n_inner, C, stride_outer, stride_c, stride_inner, accumulate,
                           stream, ls);
}

int ms_velocity_fwd(const float* x, float* v, int B, int T, int P, void* stream) {
  TimingScope ts((hipStream_t)stream, 0, 0, "ew|ew_velocity_fwd");
  if (ts.skip()) return 0;
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(P, 32), cdiv(T, 32), B), dim3(256), 0, (hipStream_t)stream, x, v, T, P, 1);
  return check_launch("transpose_kernel(velocity)");
}

int ms_velocity_bwd(const float* dv, float* dx, int B, int T, int P, void* stream) {
  TimingScope ts((hipStream_t)stream, 0, 0, "ew|ew_velocity_bwd");
  if (ts.skip()) return 0;
  hipLaunchKernelGGL(velocity_bwd_kernel, dim3(cdiv(T, 32), cdiv(P, 32), B), dim3(256), 0, (hipStream_t)stream, dv, dx, T, P);
  return check_launch("velocity_bwd_kernel");
}

int ms_transpose_btc(const float* x, float* y, int B, int T, int C, void* stream) {
  TimingScope ts((hipStream_t)stream, 0, 0, "ew|ew_transpose_btc");
  if (ts.skip()) return 0;
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(C, 32), cdiv(T, 32), B), dim3(256), 0, (hipStream_t)stream, x, y, T, C, 0);
  return check_launch("transpose_kernel");
}

int ms_transpose_bct(const float* x, float* y, int B, int C, int T, void* stream) {
  TimingScope ts((hipStream_t)stream, 0, 0, "ew|ew_transpose_bct");
  if (ts.skip()) return 0;
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(T, 32), cdiv(C, 32), B), dim3(256), 0, (hipStream_t)stream, x, y, C, T, 0);
  return check_launch("transpose_kernel");
}

size_t ms_reduce_partials_count(size_t n) { return (size_t)red_blocks(n); }

static int lp_mean_fwd(int sq, const float* a, const float* b, float target, float* loss, float* partials, size_t n, void* stream,
                       const ms_loss_scale* ls) {
  if (!a || !loss || !partials || n == 0) return set_error("ms_l%d_mean_fwd: bad argument", sq ? 2 : 1);
  TimingScope ts((hipStream_t)stream, 0, 0, sq ? "ew|ew_l2_mean_fwd" : "ew|ew_l1_mean_fwd");
  if (ts.skip()) return 0;
  const int nb = red_blocks(n);
  if (sq) hipLaunchKernelGGL(l1_partial_kernel<1>, dim3(nb), dim3(256), 0, (hipStream_t)stream, a, b, target, partials, n);
  else hipLaunchKernelGGL(l1_partial_kernel<0>, dim3(nb), dim3(256), 0, (hipStream_t)stream, a, b, target, partials, n);
  int rc = check_launch("l1_partial_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, nb, loss, 1.0f / (float)n, 0,
                     ls ? ls->scale : 1.0f, ls ? ls->scale_dev : nullptr);
  return check_launch("reduce_final_kernel");
}
static int lp_mean_bwd(int sq, const float* a, const float* b, float target, const float* gscale, float* da, size_t n, void* stream,
                       const ms_loss_scale* ls) {
  if (!a || !gscale || !da || n == 0) return set_error("ms_l%d_mean_bwd: bad argument", sq ? 2 : 1);
  TimingScope ts((hipStream_t)stream, 0, 0, sq ? "ew|ew_l2_mean_bwd" : "ew|ew_l1_mean_bwd");
  if (ts.skip()) return 0;
  const float mul = ls ? ls->scale : 1.0f;
  const float* mul_dev = ls ? ls->scale_dev : nullptr;
  if (sq) hipLaunchKernelGGL(l1_bwd_kernel<1>, dim3(red_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, target, gscale, da, n, mul, mul_dev);
  else hipLaunchKernelGGL(l1_bwd_kernel<0>, dim3(red_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, target, gscale, da, n, mul, mul_dev);
  return check_launch("l1_bwd_kernel");
}

int ms_l1_mean_fwd(const float* a, const float* b, float target, float* loss, float* partials, size_t n, void* stream) {
  return lp_mean_fwd(0, a, b, target, loss, partials, n, stream, nullptr);
}
int ms_l1_mean_bwd(const float* a, const float* b, float target, const float* gscale, float* da, size_t n, void* stream) {
  return lp_mean_bwd(0, a, b, target, gscale, da, n, stream, nullptr);
}
int ms_l2_mean_fwd(const float* a, const float* b, float target, float* loss, float* partials, size_t n, void* stream) {
  return lp_mean_fwd(1, a, b, target, loss, partials, n, stream, nullptr);
}
int ms_l2_mean_bwd(const float* a, const float* b, float target, const float* gscale, float* da, size_t n, void* stream) {
  return lp_mean_bwd(1, a, b, target, gscale, da, n, stream, nullptr);
}
int ms_lp_mean_pair_fwd(int squared, const float* a, const float* targets, float* loss, size_t n, void* stream, const ms_loss_scale* ls) {
  if (!a || !targets || !loss || n == 0 || n > 2048) return set_error("ms_lp_mean_pair_fwd: bad argument (1 <= n <= 2048 per half)");
  TimingScope ts((hipStream_t)stream, 0, 0, squared ? "ew|ew_l2_mean_pair_fwd" : "ew|ew_l1_mean_pair_fwd");
  if (ts.skip()) return 0;
  LossPair lp = {};
  for (int h = 0; h < 2; ++h) { lp.target[h] = targets[h]; lp.mul[h] = ls ? ls[h].scale : 1.0f; lp.mul_dev[h] = ls ? ls[h].scale_dev : nullptr; }
  if (squared) hipLaunchKernelGGL(lp_pair_fwd_kernel<1>, dim3(2), dim3(256), 0, (hipStream_t)stream, a, loss, (int)n, lp);
  else hipLaunchKernelGGL(lp_pair_fwd_kernel<0>, dim3(2), dim3(256), 0, (hipStream_t)stream, a, loss, (int)n, lp);
  return check_launch("lp_pair_fwd_kernel");
}
int ms_lp_mean_pair_bwd(int squared, const float* a, const float* targets, const float* gscale0, const float* gscale1, float* da, size_t n,
                        void* stream, const ms_loss_scale* ls) {
  if (!a || !targets || !da || n == 0 || n > 2048) return set_error("ms_lp_mean_pair_bwd: bad argument (1 <= n <= 2048 per half)");
  TimingScope ts((hipStream_t)stream, 0, 0, squared ? "ew|ew_l2_mean_pair_bwd" : "ew|ew_l1_mean_pair_bwd");
  if (ts.skip()) return 0;
  LossPair lp = {};
  for (int h = 0; h < 2; ++h) { lp.target[h] = targets[h]; lp.mul[h] = ls ? ls[h].scale : 1.0f; lp.mul_dev[h] = ls ? ls[h].scale_dev : nullptr; }
  lp.gscale[0] = gscale0; lp.gscale[1] = gscale1;
  const dim3 grid((unsigned)((n + 255) / 256), 2);
  if (squared) hipLaunchKernelGGL(lp_pair_bwd_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, a, da, (int)n, lp);
  else hipLaunchKernelGGL(lp_pair_bwd_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, a, da, (int)n, lp);
  return check_launch("lp_pair_bwd_kernel");
}
int ms_lp_mean_fwd_ex(int squared, const float* a, const float* b, float target, float* loss, float* partials, size_t n, void* stream,
                      const ms_loss_scale* ls) {
  return lp_mean_fwd(squared ? 1 : 0, a, b, target, loss, partials, n, stream, ls);
}
int ms_lp_mean_bwd_ex(int squared, const float* a, const float* b, float target, const float* gscale, float* da, size_t n,
                      void* stream, const ms_loss_scale* ls) {
  return lp_mean_bwd(squared ? 1 : 0, a, b, target, gscale, da, n, stream, ls);
}

// ---- several device-to-device copies in ONE launch (the train step's batch inputs into the captured step's static buffers)
enum { COPY_MULTI_MAX = 8 };
struct CopyBatch {
  int n;
  int block_end[COPY_MULTI_MAX];
  const char* src[COPY_MULTI_MAX];
  char* dst[COPY_MULTI_MAX];
  size_t bytes[COPY_MULTI_MAX];
};
__global__ __launch_bounds__(256) void copy_multi_kernel(const CopyBatch cb) {
  int j = 0;
  while (j + 1 < cb.n && (int)blockIdx.x >= cb.block_end[j]) ++j;
  const int b0 = j ? cb.block_end[j - 1] : 0;
  const size_t off = ((size_t)((int)blockIdx.x - b0) * 256 + threadIdx.x) * 64;      // 64 bytes per thread
  const char* s = cb.src[j];
  char* d = cb.dst[j];
  const size_t n = cb.bytes[j];
  if (off >= n) return;
  if (((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15) == 0 && off + 64 <= n) {
    const uint4* s4 = reinterpret_cast<const uint4*>(s + off);
    uint4 v0 = s4[0], v1 = s4[1], v2 = s4[2], v3 = s4[3];
    uint4* d4 = reinterpret_cast<uint4*>(d + off);
    d4[0] = v0; d4[1] = v1; d4[2] = v2; d4[3] = v3;
  } else {
    for (size_t i = off; i < n && i < off + 64; ++i) d[i] = s[i];
  }
}

int ms_copy_multi(int n, const void* const* src, void* const* dst, const size_t* bytes, void* stream) {
  if (n < 0 || (n > 0 && (!src || !dst || !bytes))) return set_error("ms_copy_multi: bad argument");
  TimingScope ts((hipStream_t)stream, 0, 0, "ew|ew_copy_multi");
  if (ts.skip()) return 0;
  for (int i0 = 0; i0 < n; i0 += COPY_MULTI_MAX) {
    CopyBatch cb = {};
    int blocks = 0;
    for (int i = i0; i < n && i < i0 + COPY_MULTI_MAX; ++i) {
      if (!bytes[i]) continue;
      if (!src[i] || !dst[i]) return set_error("ms_copy_multi: null buffer");
      cb.src[cb.n] = (const char*)src[i]; cb.dst[cb.n] = (char*)dst[i]; cb.bytes[cb.n] = bytes[i];
      blocks += (int)((bytes[i] + 256 * 64 - 1) / (256 * 64));
      cb.block_end[cb.n++] = blocks;
    }
    if (!cb.n) continue;
    hipLaunchKernelGGL(copy_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, cb);
    const int rc = check_launch("copy_multi_kernel");
    if (rc) return rc;
  }
  return 0;
}

// up to 8 floats handed over BY VALUE (kernel arguments are copied when the launch is enqueued): a host-side schedule value reaches
// the device without an asynchronous read of host memory that a later host write could overtake
struct FloatsArg { float v[8]; };
__global__ void write_floats_kernel(float* dst, FloatsArg a, int n) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = a.v[threadIdx.x];
}

int ms_write_floats(float* dst, const float* host_values, int n, void* stream) {
  if (!dst || !host_values || n < 1 || n > 8) return set_error("ms_write_floats: 1..8 values");
  FloatsArg a = {};
  for (int i = 0; i < n; ++i) a.v[i] = host_values[i];
  hipLaunchKernelGGL(write_floats_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dst, a, n);
  return check_launch("write_floats_kernel");
}

int ms_sqnorm(const float* g, size_t n, float* norm_out, float* partials, void* stream) {
  TimingScope ts((hipStream_t)stream, 0, 0, "ew|ew_sqnorm");
  if (ts.skip()) return 0;
  const int nb = red_blocks(n);
  hipLaunchKernelGGL(sq_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, g, partials, n);
  int rc = check_launch("sq_partial_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, nb, norm_out, 1.0f, 1);
  return check_launch("reduce_final_kernel");
}

int ms_adam_step(float* p, const float* g, float* m, float* v, size_t n, const float* norm, float max_norm, float lr,
                 float beta1, float beta2, float eps, int32_t* step_state, void* stream) {
  TimingScope ts((hipStream_t)stream, 0, 0, "ew|ew_adam_step");
  if (ts.skip()) return 0;
  hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_state, norm, max_norm, lr, beta1, beta2);
  int rc = check_launch("adam_prep_kernel");
  if (rc) return rc;
  int blocks = (int)std::min<size_t>((n + 1023) / 1024, 2048);
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, step_state, beta1, beta2, eps);
  return check_launch("adam_kernel");
}

int ms_adam_step_segmented(float* p, const float* g, float* m, float* v, size_t n, const float* norm, float max_norm, float lr,
                           float beta1, float beta2, float eps, int32_t* step_state, const int32_t* seg_of_chunk,
                           const int32_t* seg_first_step, float* seg_scratch, int n_seg, void* stream) {
  TimingScope ts((hipStream_t)stream, 0, 0, "ew|ew_adam_step_segmented");
  if (ts.skip()) return 0;
  hipLaunchKernelGGL(adam_prep_seg_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, step_state, norm, max_norm, lr, beta1,
                     beta2, seg_first_step, seg_scratch, n_seg);
  int rc = check_launch("adam_prep_seg_kernel");
  if (rc) return rc;
  int blocks = (int)std::min<size_t>((n + 1023) / 1024, 2048);
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_seg_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, step_state, seg_of_chunk,
                     seg_scratch, beta1, beta2, eps);
  return check_launch("adam_seg_kernel");
}

}  // extern "C"
