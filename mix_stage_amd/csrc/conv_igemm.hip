// Implicit-GEMM convolution kernels for gfx950 (MI355X) on the exact-fp32 matrix cores
// (v_mfma_f32_32x32x2_f32).  One workgroup = 4 waves (2x2), each wave owns TMxTN blocks of 32x32
// accumulators; tiles are staged through LDS k-major so both MFMA operands are read bank-conflict
// free with ds_read_b32, and the (B,C,T) time axis is the lane axis of every global access.
//
//   igemm_gather_kernel<.., TRANSPOSED=false>  forward conv   y[co,pix]  = sum_k w[co,k]  * im2col(x)[k,pix]
//   igemm_gather_kernel<.., TRANSPOSED=true >  data gradient  dx[ci,pix] = sum_k wt[ci,k] * col2im-gather(dy)[k,pix]
//   wgrad_kernel                               weight gradient dw[co,k]  = sum_pix dy[co,pix] * im2col(x)[k,pix]
//
// Replaces what ATen's convolution / convolution_backward compute for the reference's
// ConvNormRelu / nn.Conv1d calls (layers.py:58-78, JL:83, S2G:50-63).
#include <algorithm>
#include <mutex>
#include <vector>

#include "kernels.h"

namespace ms {

template <int TM, int TN, int KH_, int KW_, bool TRANSPOSED, bool UP2>
__global__ __launch_bounds__(256) void igemm_gather_kernel(const GatherArgs p) {
  prefetch_kernargs<sizeof(GatherArgs)>();
  constexpr int BM = 64 * TM, BN = 64 * TN, BK = (TM == 1 ? 64 : 32);
  constexpr int LDA = BM + 2, LDB = BN;
  constexpr int STAGE = BK * LDA + BK * LDB;
  constexpr int BROWS = 256 / BN, BITER = BK / BROWS;
  constexpr int AITER = BM * BK / 1024;       // float4 loads of the A tile per thread
  __shared__ float smem[2 * STAGE];

  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int KH = KH_ ? KH_ : p.KH, KW = KW_ ? KW_ : p.KW, KHW = KH * KW;
  const int Kg = p.Kg;

  // Data gradient of a strided conv: blockIdx.z also enumerates the SH*SW output-parity classes; inside one class
  // the problem is a dense stride-1 gather with ceil(K/S) taps per axis (no work on structural zeros).
  int zz = blockIdx.z;
  const int ks = zz % p.splitk;                 // split-K slice
  zz /= p.splitk;
  int g = zz, cls = 0, ry = 0, rx = 0, cy = p.PH, cx = p.PW, QH = p.OUTH, QW = p.OUTW;
  if (TRANSPOSED) {
    const int ncls = p.SH * p.SW;
    g = zz / ncls;
    cls = zz - g * ncls;
    ry = cls / p.SW;
    rx = cls - ry * p.SW;
    const int kh0 = (ry + p.PH) % p.SH, kw0 = (rx + p.PW) % p.SW;
    cy = (ry + p.PH - kh0) / p.SH;
    cx = (rx + p.PW - kw0) / p.SW;
    QH = (p.OUTH - ry + p.SH - 1) / p.SH;
    QW = (p.OUTW - rx + p.SW - 1) / p.SW;
  }
  const int qhw = QH * QW;
  const int npix = TRANSPOSED ? p.batch * qhw : p.Npix;
  if (n0 >= npix) return;                       // whole block out of range (uneven parity classes)
  const int kbeg = ks * p.k_per_split, kend = min(Kg, kbeg + p.k_per_split);
  const float* Abase = p.A + (size_t)cls * p.groups * p.Mg * Kg;

  // ---- the B-tile column (output pixel) this thread stages: decoded once.  All lanes of a wave stage the same
  // k rows (BN >= 64), so the k -> (channel, tap) decode below is wave-uniform and lives on the scalar unit.
  const int nl = t % BN;
  const int kk0 = __builtin_amdgcn_readfirstlane(t / BN);
  const int n = n0 + nl;
  const bool nvalid = n < npix;
  int pb = 0, py = 0, px = 0;
  if (nvalid) {
    pb = n / qhw;
    const int rem = n - pb * qhw;
    py = rem / QW;
    px = rem - py * QW;
  }
  const int by = TRANSPOSED ? py + cy : py * p.SH - p.PH;
  const int bx = TRANSPOSED ? px + cx : px * p.SW - p.PW;
  const int cbase = p.bcast ? 0 : g * p.Kc;
  const int chan0 = pb * p.src_ctotal + cbase;  // channel-row index of (pb, cbase)
  const int src_hw = p.SRCH * p.SRCW;
  const int lane_off = chan0 * src_hw + by * p.SRCW + bx;   // + per-k scalar offset (may be < 0 when out of range)

  float4 ra[AITER];
  float rb[BITER];

  // Loads are branch-free: out-of-range elements read a clamped (valid) address and are zeroed by a select, so the
  // whole tile's loads issue back to back.
  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int i = 0; i < AITER; ++i) {
      const int row = (t >> 2) + 64 * (i % TM), kq = (t & 3) + 4 * (i / TM);
      const int m = m0 + row, k = k0 + kq * 4;
      const bool ok = (m < p.Mg) & (k < kend);
      const unsigned off = ok ? (unsigned)((g * p.Mg + m) * Kg + k) : 0u;
      float4 v;
      if (p.a_vec) {
        v = *reinterpret_cast<const float4*>(Abase + off);
        if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        const unsigned last = (unsigned)(p.groups * p.Mg * Kg - 1);
        v.x = Abase[off];
        v.y = Abase[min(off + 1, last)];
        v.z = Abase[min(off + 2, last)];
        v.w = Abase[min(off + 3, last)];
        if (!ok) v.x = 0.f;
        if (!ok || k + 1 >= kend) v.y = 0.f;
        if (!ok || k + 2 >= kend) v.z = 0.f;
        if (!ok || k + 3 >= kend) v.w = 0.f;
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BITER; ++i) {
      const unsigned k = (unsigned)(k0 + kk0 + i * BROWS);    // wave-uniform -> scalar unit
      const unsigned kc = k / (unsigned)KHW, r = k - kc * (unsigned)KHW;
      const unsigned kh = (KH_ == 1) ? 0u : r / (unsigned)KW, kw = r - kh * (unsigned)KW;
      const int sy = TRANSPOSED ? by - (int)kh : by + (int)kh;
      const int sx = TRANSPOSED ? bx - (int)kw : bx + (int)kw;
      const bool ok = nvalid & ((int)k < kend) & ((unsigned)sy < (unsigned)p.SRCH) & ((unsigned)sx < (unsigned)p.SRCW);
      float v;
      if (UP2 && !TRANSPOSED) {
        const int crow = chan0 + (int)kc;
        const unsigned oa = ok ? (unsigned)(crow * (p.SRCW >> 1) + (sx >> 1)) : 0u;
        const unsigned orr = ok ? (unsigned)(crow * p.SRCW + sx) : 0u;
        v = p.src[oa] + p.src2[orr];
      } else {
        const int koff = (int)kc * src_hw + (TRANSPOSED ? -(int)(kh * p.SRCW + kw) : (int)(kh * p.SRCW + kw));   // scalar
        const unsigned off = ok ? (unsigned)(lane_off + koff) : 0u;
        v = p.src[off];
      }
      rb[i] = ok ? v : 0.f;
    }
  };
  auto store_tiles = [&](int buf) {
    float* As = smem + buf * STAGE;
    float* Bs = As + BK * LDA;
#pragma unroll
    for (int i = 0; i < AITER; ++i) {
      const int row = (t >> 2) + 64 * (i % TM), kq = (t & 3) + 4 * (i / TM);
      As[(kq * 4 + 0) * LDA + row] = ra[i].x;
      As[(kq * 4 + 1) * LDA + row] = ra[i].y;
      As[(kq * 4 + 2) * LDA + row] = ra[i].z;
      As[(kq * 4 + 3) * LDA + row] = ra[i].w;
    }
#pragma unroll
    for (int i = 0; i < BITER; ++i) Bs[(kk0 + i * BROWS) * LDB + nl] = rb[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int a_off = wm * TM * 32 + (lane & 31), b_off = wn * TN * 32 + (lane & 31), khalf = lane >> 5;
  // the epilogue's bias values are requested now (clamped addresses, no branches): fetched after the K loop they were 16*TM
  // dependent load -> wait rounds at the very end of the kernel
  float bias_pre[TM][16];
  {
    const bool want = p.bias != nullptr && p.splitk <= 1 && p.ep != EP_DGRAD && p.ep != EP_DGRAD_UP2;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        bias_pre[i][r] = want ? p.bias[g * p.Mg + (m < p.Mg ? m : 0)] : 0.f;
      }
  }
  const int nk = (kend - kbeg + BK - 1) / BK;
  load_tiles(kbeg);
  store_tiles(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tiles(kbeg + (kt + 1) * BK);
    const float* As = smem + cur * STAGE;
    const float* Bs = As + BK * LDA;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[(kk + khalf) * LDA + a_off + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[(kk + khalf) * LDB + b_off + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tiles(cur ^ 1);
    __syncthreads();
  }

  // ---------------- epilogue ----------------
  const int ctot = p.groups * p.Mg;
  const int ohw = p.OUTH * p.OUTW;              // full output image (channel stride)
  int ooff[TN];
  bool cval[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nc = n0 + wn * TN * 32 + j * 32 + (lane & 31);
    cval[j] = nc < npix;
    const int b = cval[j] ? nc / qhw : 0;
    const int pix = nc - b * qhw;
    if (TRANSPOSED) {
      const int qy = pix / QW, qx = pix - qy * QW;
      ooff[j] = b * ctot * ohw + (qy * p.SH + ry) * p.OUTW + qx * p.SW + rx;
    } else {
      ooff[j] = b * ctot * ohw + pix;           // + channel*ohw
    }
  }
  const int ep = p.ep;
  if (p.splitk > 1) {                           // raw partial tile; the epilogue runs in the split-K reduce kernel
    float* part = p.part + (size_t)ks * p.part_stride;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
#pragma unroll
        for (int j = 0; j < TN; ++j)
          if (m < p.Mg && cval[j]) part[(size_t)ooff[j] + (size_t)(g * p.Mg + m) * ohw] = acc[i][j][r];
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ml = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      const int m = m0 + ml;
      const bool mval = m < p.Mg;
      const int ch = g * p.Mg + (mval ? m : 0);
      if (ep == EP_DGRAD) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
          if (mval && cval[j]) p.out[(size_t)ooff[j] + (size_t)ch * ohw] = acc[i][j][r];
      } else if (ep == EP_DGRAD_UP2) {
        // 1-D, stride 1: out2 = grad of the residual (full resolution); out = grad of the half-resolution tensor
        const int hw = ohw >> 1;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const float v = acc[i][j][r];
          const float pr = lane_xor1(v);
          if (mval && cval[j]) {
            p.out2[(size_t)ooff[j] + (size_t)ch * ohw] = v;
            if (!(lane & 1)) {
              const int nc = n0 + wn * TN * 32 + j * 32 + (lane & 31);
              const int b = nc / ohw, pix = nc - b * ohw;
              p.out[(size_t)(b * ctot + ch) * hw + (pix >> 1)] = v + pr;
            }
          }
        }
      } else {
        const float bsv = bias_pre[i][r];
        float sc = 1.f, sh = 0.f;
        if (ep == EP_BN_EVAL) {
          const float inv = 1.0f / sqrtf(p.bn_v[ch] + p.eps);
          sc = p.bn_g[ch] * inv;
          sh = p.bn_b[ch] - p.bn_m[ch] * sc;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          float v = acc[i][j][r] + bsv;
          if (ep == EP_RAW_STATS) acc[i][j][r] = v;
          if (ep == EP_BN_EVAL) v = lrelu(fmaf(v, sc, sh), p.slope);
          if (ep == EP_LRELU) v = lrelu(v, p.slope);
          if (mval && cval[j]) p.out[(size_t)ooff[j] + (size_t)ch * ohw] = v;
        }
      }
    }
  }

  if (ep == EP_RAW_STATS) {
    // per-channel (sum, M2 about this tile's mean) over the tile's valid pixels, fixed order
    float* red = smem;  // [4][BM]
    const int cnt = min(BN, npix - n0);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ml = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) s += cval[j] ? acc[i][j][r] : 0.f;
        s = half_wave_sum(s);
        if ((lane & 31) == 0) red[wn * BM + ml] = s;
      }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ml = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        const float mean = (red[ml] + red[BM + ml]) / (float)cnt;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const float dlt = acc[i][j][r] - mean;
          q += cval[j] ? dlt * dlt : 0.f;
        }
        q = half_wave_sum(q);
        if ((lane & 31) == 0) red[2 * BM + wn * BM + ml] = q;
      }
    __syncthreads();
    if (t < BM && m0 + t < p.Mg) {
      float* st = p.stats + ((size_t)blockIdx.x * ctot + g * p.Mg + m0 + t) * 2;
      st[0] = red[t] + red[BM + t];
      st[1] = red[2 * BM + t] + red[3 * BM + t];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// (bx, by, bz): the workgroup's place in the job's grid (Kg tiles, Cog tiles, groups x splits)
template <int TM, int TN, int KH_, int KW_, bool UP2>
__device__ __forceinline__ void wgrad_body(const WgradArgs& p, const int bx_, const int by_, const int bz_) {
  constexpr int BM = 64 * TM, BN = 64 * TN, BR = 32;
  constexpr int LDA = BM + 1, LDB = BN + 1;
  constexpr int STAGE = BR * LDA + BR * LDB;
  constexpr int AIT = BM / 8, BIT = BN / 8;
  __shared__ float smem[2 * STAGE];

  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int g = bz_ / p.splits, sp = bz_ - g * p.splits;
  const int m0 = by_ * BM, n0 = bx_ * BN;
  const int KH = KH_ ? KH_ : p.KH, KW = KW_ ? KW_ : p.KW, KHW = KH * KW;
  const int rl = t & 31, row0 = t >> 5;
  const int ohw = p.OH * p.OW;
  const int ctot = p.groups * p.Cog;
  const int r_begin = sp * p.r_per_split;
  const int r_end = min(p.Npix, r_begin + p.r_per_split);
  const int cbase = p.bcast ? 0 : g * p.Cig;

  float ra[AIT], rb[BIT];

  auto load_tiles = [&](int r0) {
    const int r = r0 + rl;
    const bool rvalid = r < r_end;
    int pb = 0, oy = 0, ox = 0, pix = 0;
    if (rvalid) {
      pb = r / ohw;
      pix = r - pb * ohw;
      oy = pix / p.OW;
      ox = pix - oy * p.OW;
    }
    const float* ap = p.dyr + ((size_t)pb * ctot + g * p.Cog) * ohw + pix;
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
      const int m = m0 + row0 + i * 8;
      ra[i] = (rvalid && m < p.Cog) ? ap[(size_t)m * ohw] : 0.f;
    }
    const int by = oy * p.SH - p.PH, bx = ox * p.SW - p.PW;
    const int chan0 = pb * p.src_ctotal + cbase;
#pragma unroll
    for (int i = 0; i < BIT; ++i) {
      const int k = n0 + row0 + i * 8;
      float v = 0.f;
      if (rvalid && k < p.Kg) {
        const int kc = k / KHW, rr = k - kc * KHW, kh = rr / KW, kw = rr - kh * KW;
        const int sy = by + kh, sx = bx + kw;
        if ((unsigned)sy < (unsigned)p.H && (unsigned)sx < (unsigned)p.W) {
          const int crow = chan0 + kc;
          if (UP2) v = p.src[(size_t)crow * (p.W >> 1) + (sx >> 1)] + p.src2[(size_t)crow * p.W + sx];
          else v = p.src[((size_t)crow * p.H + sy) * p.W + sx];
        }
      }
      rb[i] = v;
    }
  };
  auto store_tiles = [&](int buf) {
    float* As = smem + buf * STAGE;
    float* Bs = As + BR * LDA;
#pragma unroll
    for (int i = 0; i < AIT; ++i) As[rl * LDA + row0 + i * 8] = ra[i];
#pragma unroll
    for (int i = 0; i < BIT; ++i) Bs[rl * LDB + row0 + i * 8] = rb[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nsteps = (r_end - r_begin + BR - 1) / BR;
  if (nsteps > 0) {
    load_tiles(r_begin);
    store_tiles(0);
  }
  __syncthreads();
  const int a_off = wm * TM * 32 + (lane & 31), b_off = wn * TN * 32 + (lane & 31), khalf = lane >> 5;
  for (int st = 0; st < nsteps; ++st) {
    const int cur = st & 1;
    if (st + 1 < nsteps) load_tiles(r_begin + (st + 1) * BR);
    const float* As = smem + cur * STAGE;
    const float* Bs = As + BR * LDA;
#pragma unroll
    for (int kk = 0; kk < BR; kk += 2) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[(kk + khalf) * LDA + a_off + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[(kk + khalf) * LDB + b_off + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (st + 1 < nsteps) store_tiles(cur ^ 1);
    __syncthreads();
  }

  float* outp = p.out + (size_t)sp * ctot * p.Kg;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int nc = n0 + wn * TN * 32 + j * 32 + (lane & 31);
        if (m < p.Cog && nc < p.Kg) {
          float* o = outp + (size_t)(g * p.Cog + m) * p.Kg + nc;
          *o = p.accumulate ? *o + acc[i][j][r] : acc[i][j][r];
        }
      }
    }
}

template <int TM, int TN, int KH_, int KW_, bool UP2>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs p) {
  wgrad_body<TM, TN, KH_, KW_, UP2>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Queued form (ms_bwd_options.defer_wgrad_launch / ms_wgrad_flush): the layers this kernel serves are the ones too short for the
// patch kernels (1-D layers of fewer than 16 output frames: the deep UNet levels, the tails of the style encoder and of the
// discriminator) -- 11 launches of 6-12 us each in the middle of the headline G-step's backward chain, which nothing downstream
// reads.  They wait in a queue and run side by side in one launch per kernel shape at the end of the backward pass.
template <int KH_, int KW_, bool UP2>
__global__ __launch_bounds__(256) void wgrad_multi_kernel(const WgradBatch b) {
  int j = 0;
  while (j + 1 < b.n && (int)blockIdx.x >= b.block_end[j]) ++j;
  const WgradArgs& p = b.job[j];
  const int id = (int)blockIdx.x - (j ? b.block_end[j - 1] : 0);
  const int gx = (p.Kg + 63) / 64, gy = (p.Cog + 63) / 64;
  const int bz = id / (gx * gy), r = id - bz * gx * gy;
  wgrad_body<1, 1, KH_, KW_, UP2>(p, r % gx, r / gx, bz);
}

// sum over the split slabs in ascending order (bitwise the same as the plain loop), 8 loads in flight at a time
__device__ inline float sum_splits_in_order(const float* __restrict__ p, size_t stride, int splits) {
  float s = 0.f;
  int k = 0;
  for (; k + 8 <= splits; k += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p[(size_t)(k + j) * stride];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
  }
  for (; k < splits; ++k) s += p[(size_t)k * stride];
  return s;
}

// many splits, few outputs: one wave per output element, lanes stride over the splits, fixed-order wave reduction
__global__ __launch_bounds__(256) void reduce_splits_wave_kernel(const float* __restrict__ part, float* __restrict__ out, int n,
                                                                 int splits) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  float s = 0.f;
  for (int k = lane; k < splits; k += 64) s += part[(size_t)k * n + i];
  s = wave_sum(s);
  if (lane == 0) out[i] = s;
}

// out[i] = sum_s part[s][i]   (fixed order -> bitwise reproducible)
__global__ void reduce_splits_kernel(const float* __restrict__ part, float* __restrict__ out, int n, int splits) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    out[i] = sum_splits_in_order(part + i, (size_t)n, splits);
  }
}

// many blocks' split reductions in ONE launch, ACCUMULATING into out (the caller zeroes out once per step); each job keeps
// the summation order of the single-job kernel it replaces (wave: lanes stride over the splits), so results are the same
__global__ __launch_bounds__(256) void reduce_splits_multi_kernel(const ReduceBatch rb) {
  int j = 0;
  while (j + 1 < rb.n && (int)blockIdx.x >= rb.job[j].block_end) ++j;
  const ReduceJob jb = rb.job[j];
  const int b = (int)blockIdx.x - (j ? rb.job[j - 1].block_end : 0);
  const int nb = jb.block_end - (j ? rb.job[j - 1].block_end : 0);
  if (jb.wave) {
    const int lane = threadIdx.x & 63;
    const int i = b * 4 + (threadIdx.x >> 6);
    if (i >= jb.n) return;
    float s = 0.f;
    for (int k = lane; k < jb.splits; k += 64) s += jb.part[(size_t)k * jb.n + i];
    s = wave_sum(s);
    if (lane == 0) jb.out[i] += s;
  } else if ((jb.n & 3) == 0 && ((reinterpret_cast<uintptr_t>(jb.part) | reinterpret_cast<uintptr_t>(jb.out)) & 15) == 0) {
    // 16-byte form: four neighbouring outputs per thread, up to eight slabs in flight (per output the same ascending sum as below)
    const int n4 = jb.n >> 2;
    const float4* part4 = reinterpret_cast<const float4*>(jb.part);
    float4* out4 = reinterpret_cast<float4*>(jb.out);
    for (int i = b * 256 + threadIdx.x; i < n4; i += nb * 256) {
      float4 o = out4[i];
      float4 s = {0.f, 0.f, 0.f, 0.f};
      int k = 0;
      for (; k + 8 <= jb.splits; k += 8) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = part4[(size_t)(k + j) * n4 + i];
#pragma unroll
        for (int j = 0; j < 8; ++j) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
      }
      for (; k + 4 <= jb.splits; k += 4) {
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = part4[(size_t)(k + j) * n4 + i];
#pragma unroll
        for (int j = 0; j < 4; ++j) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
      }
      for (; k < jb.splits; ++k) {
        const float4 v = part4[(size_t)k * n4 + i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w;
      out4[i] = o;
    }
  } else {
    for (int i = b * 256 + threadIdx.x; i < jb.n; i += nb * 256) {
      jb.out[i] += sum_splits_in_order(jb.part + i, (size_t)jb.n, jb.splits);
    }
  }
}

// Data-gradient weights.  One slab per output-parity class (ry, rx) of a stride-(SH,SW) conv:
//   wpar[cls][g][ci][co][jh][jw] = w[g][co][ci][kh0 + SH*jh][kw0 + SW*jw]   (0 beyond the kernel)
// with kh0 = (ry+PH)%SH, kw0 = (rx+PW)%SW, JH = ceil(KH/SH), JW = ceil(KW/SW).  Stride 1 -> one slab, all taps.
__device__ inline void transpose_weight_elems(const TransposeJob& jb, int first, int stride) {
  const int KH = jb.KH, KW = jb.KW, SH = jb.SH, SW = jb.SW, Cog = jb.Cog, Cig = jb.Cig;
  const int JH = (KH + SH - 1) / SH, JW = (KW + SW - 1) / SW, J = JH * JW;
  const int per_g = Cog * Cig * J, per_cls = jb.groups * per_g;
  const int total = SH * SW * per_cls;
  for (int i = first; i < total; i += stride) {
    const int cls = i / per_cls;
    int r = i - cls * per_cls;
    const int g = r / per_g;
    r -= g * per_g;                  // index in wpar[cls][g]: (ci, co, jh, jw)
    const int ci = r / (Cog * J);
    r -= ci * Cog * J;
    const int co = r / J, j = r - co * J;
    int jh = j / JW, jw = j - jh * JW;
    if (jb.flip) { jh = JH - 1 - jh; jw = JW - 1 - jw; }   // taps reversed: the data gradient becomes a forward conv
    const int ry = cls / SW, rx = cls - ry * SW;
    const int kh = (ry + jb.PH) % SH + SH * jh, kw = (rx + jb.PW) % SW + SW * jw;
    float v = 0.f;
    if (kh < KH && kw < KW) v = jb.w[(((size_t)(g * Cog + co) * Cig + ci) * KH + kh) * KW + kw];
    jb.wt[i] = v;
  }
}

__global__ void transpose_weight_kernel(const TransposeJob jb) {
  transpose_weight_elems(jb, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// many blocks' data-gradient weights in ONE launch: job j owns workgroups [block_end[j-1], block_end[j])
__global__ void transpose_weight_multi_kernel(const TransposeBatch tb) {
  int j = 0;
  while (j + 1 < tb.n && (int)blockIdx.x >= tb.job[j].block_end) ++j;
  const int b0 = j ? tb.job[j - 1].block_end : 0;
  transpose_weight_elems(tb.job[j], ((int)blockIdx.x - b0) * blockDim.x + threadIdx.x, (tb.job[j].block_end - b0) * blockDim.x);
}

// ---------------------------------------------------------------------------------------------
// launch helpers
GatherPlan plan_gather(int Mg, int npix, int zcount, int Kg) {
  GatherPlan pl;
  // big tile only when it still fills the chip (256 CUs) at >= 2 workgroups per CU
  const long big = (long)cdiv(Mg, 128) * cdiv(npix, 128) * zcount;
  if (Mg >= 128 && big >= 512) { pl.tm = 2; pl.tn = 2; } else { pl.tm = 1; pl.tn = 1; }
  const int bm = 64 * pl.tm, bn = 64 * pl.tn, bk = pl.tm == 1 ? 64 : 32;
  pl.n_tiles = cdiv(npix, bn);
  const long base = (long)pl.n_tiles * cdiv(Mg, bm) * zcount;
  const int nk = cdiv(Kg, bk);
  int splitk = 1;
  // few workgroups and a long reduction: slice K so the weight stream is spread over the chip
  if (base < 96 && nk >= 2) {
    splitk = (int)std::min<long>(nk, std::max<long>(1, 256 / base));
    if (splitk > 32) splitk = 32;
  }
  const int steps = cdiv(nk, splitk);
  pl.k_per_split = steps * bk;
  pl.splitk = cdiv(Kg, pl.k_per_split);
  return pl;
}

template <int TM, int TN, bool TR, bool UP2>
static void launch_gather_khw(const GatherArgs& a, dim3 grid, hipStream_t s) {
  const int kh = a.KH, kw = a.KW;
#define MS_GK(KH, KW) hipLaunchKernelGGL((igemm_gather_kernel<TM, TN, KH, KW, TR, UP2>), grid, dim3(256), 0, s, a)
  if constexpr (UP2) {
    if (kh == 1 && kw == 3) MS_GK(1, 3);
    else MS_GK(0, 0);
  } else if constexpr (TR) {      // taps per parity class
    if (kh == 1 && kw == 3) MS_GK(1, 3);
    else if (kh == 1 && kw == 1) MS_GK(1, 1);
    else if (kh == 1 && kw == 2) MS_GK(1, 2);
    else if (kh == 1 && kw == 4) MS_GK(1, 4);
    else if (kh == 2 && kw == 2) MS_GK(2, 2);
    else if (kh == 3 && kw == 3) MS_GK(3, 3);
    else if (kh == 3 && kw == 8) MS_GK(3, 8);
    else MS_GK(0, 0);
  } else {
    if (kh == 1 && kw == 3) MS_GK(1, 3);
    else if (kh == 1 && kw == 1) MS_GK(1, 1);
    else if (kh == 1 && kw == 4) MS_GK(1, 4);
    else if (kh == 3 && kw == 3) MS_GK(3, 3);
    else if (kh == 4 && kw == 4) MS_GK(4, 4);
    else if (kh == 3 && kw == 8) MS_GK(3, 8);
    else MS_GK(0, 0);
  }
#undef MS_GK
}

// Forward: a.Npix = B*OUTH*OUTW.  Transposed (data gradient): a.KH/a.KW are the taps per parity class, a.SH/a.SW
// the conv strides (= class counts), a.batch = B, a.A = the class slabs built by transpose_weight_kernel.
// plan.splitk > 1: a.part / a.part_stride must be set; the caller then runs a split-K epilogue kernel.
int launch_gather(GatherArgs a, bool transposed, bool up2, const GatherPlan& plan, hipStream_t s) {
  const int ncls = transposed ? a.SH * a.SW : 1;
  const int npix_cls = transposed ? a.batch * cdiv(a.OUTH, a.SH) * cdiv(a.OUTW, a.SW) : a.Npix;   // largest class
  const int bm = 64 * plan.tm, bn = 64 * plan.tn;
  a.splitk = plan.splitk;
  a.k_per_split = plan.k_per_split;
  dim3 grid(cdiv(npix_cls, bn), cdiv(a.Mg, bm), a.groups * ncls * plan.splitk);
  if (grid.y > 65535 || grid.z > 65535) return set_error("conv grid too large");
  if (plan.splitk > 1 && !a.part) return set_error("split-K without a partial buffer");
  const double batch = transposed ? (double)a.batch : (double)a.Npix / ((double)a.OUTH * a.OUTW);
  const double opix = batch * a.OUTH * a.OUTW;
  TimingScope ts(s, 2.0 * a.Mg * a.Kg * (double)npix_cls * ncls * a.groups,
                 4.0 * ((double)ncls * a.groups * a.Mg * a.Kg + batch * a.src_ctotal * a.SRCH * a.SRCW + opix * a.groups * a.Mg),
                 "igemm_gather_kernel<%d,%d,%d,%d,%d,%d>|%s k%dx%d s%d Mg%d Kg%d g%d N%.0f tile%d splitk%d%s", plan.tm, plan.tn,
                 a.KH, a.KW, transposed ? 1 : 0, (up2 && !transposed) ? 1 : 0, transposed ? "conv_dgrad" : "conv_fwd", a.KH, a.KW,
                 a.SW, a.Mg, a.Kg, a.groups, opix, bm, plan.splitk, a.ep == EP_RAW_STATS ? " +bnstats" : "");
  if (ts.skip()) return 0;
  if (transposed) up2 = false;  // the UP2 split store of the data gradient is a runtime epilogue (EP_DGRAD_UP2)
  if (plan.tm == 2) {
    if (transposed) launch_gather_khw<2, 2, true, false>(a, grid, s);
    else if (up2) launch_gather_khw<2, 2, false, true>(a, grid, s);
    else launch_gather_khw<2, 2, false, false>(a, grid, s);
  } else {
    if (transposed) launch_gather_khw<1, 1, true, false>(a, grid, s);
    else if (up2) launch_gather_khw<1, 1, false, true>(a, grid, s);
    else launch_gather_khw<1, 1, false, false>(a, grid, s);
  }
  return check_launch("igemm_gather_kernel");
}

// acc + p[0] + p[stride] + ... (slices in order); 4 loads in flight instead of a load -> wait -> add chain per slice
__device__ inline float sum_slices(const float* __restrict__ p, int n, size_t stride, float acc) {
  for (int k0 = 0; k0 < n; k0 += 4) {
    float tmp[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) tmp[kk] = p[(size_t)min(k0 + kk, n - 1) * stride];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      if (k0 + kk < n) acc += tmp[kk];
  }
  return acc;
}

// ---------------------------------------------------------------------------------------------
// split-K epilogues.  Forward: one workgroup per output channel sums the partial tiles, adds the bias and applies
// the block epilogue; for BN_TRAIN the workgroup owns all B*HW values of its channel, so batch statistics,
// running-stat update, normalisation and LeakyReLU happen here in one launch (two-pass variance).
template <int NE>
__global__ __launch_bounds__(256) void splitk_fwd_epilogue_kernel(const float* __restrict__ part, int splitk,
                                                                  size_t part_stride, const float* __restrict__ bias,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float* rm, float* rv,
                                                                  float* __restrict__ y_raw, float* __restrict__ y,
                                                                  float* __restrict__ save, int B, int C, int HW, int ep,
                                                                  float slope, float eps, float momentum, int sg) {
  prefetch_kernargs<192>();
  const FastDiv fdHW(HW, B * HW);
  // the channel's B*HW <= 256*NE values stay in registers between the passes
  __shared__ float red[4];
  const int c = blockIdx.x, t = threadIdx.x;
  const int N = B * HW;
  const float bsv = bias ? bias[c] : 0.f;
  // (sg > 1, MS_DT_STAT_PAIR: B clips per statistics group, the groups one after the other -- the running statistics move twice,
  // in group order, as two forward passes of the module move them)
  for (int grp = 0; grp < sg; ++grp, part += (size_t)B * C * HW, y += (size_t)B * C * HW, y_raw = y_raw ? y_raw + (size_t)B * C * HW : y_raw,
           save = save ? save + 4 * C : save) {
  float v[NE];
  size_t off[NE];
  float s1 = 0.f;
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int e = min(t + i * 256, N - 1);
    const int b = fdHW.div(e), pix = e - b * HW;
    off[i] = ((size_t)b * C + c) * HW + pix;
    v[i] = bsv;
  }
  // slices summed in order k = 0, 1, ... per element; the loads of 4 slices x NE elements are in flight together
  for (int k0 = 0; k0 < splitk; k0 += 4) {
    float tmp[4][NE];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const float* pk = part + (size_t)min(k0 + kk, splitk - 1) * part_stride;
#pragma unroll
      for (int i = 0; i < NE; ++i) tmp[kk][i] = pk[off[i]];
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      if (k0 + kk < splitk) {
#pragma unroll
        for (int i = 0; i < NE; ++i) v[i] += tmp[kk][i];
      }
  }
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    if (t + i * 256 >= N) v[i] = 0.f;
    s1 += v[i];
  }
  if (ep != EP_RAW_STATS) {
    float sc = 1.f, sh = 0.f;
    if (ep == EP_BN_EVAL) {
      const float inv = 1.0f / sqrtf(rv[c] + eps);
      sc = gamma[c] * inv;
      sh = beta[c] - rm[c] * sc;
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      float o = v[i];
      if (ep == EP_BN_EVAL) o = lrelu(fmaf(o, sc, sh), slope);
      if (ep == EP_LRELU) o = lrelu(o, slope);
      if (t + i * 256 < N) y[off[i]] = o;
    }
    continue;
  }
  const float mean = block_sum_256(s1, red) / (float)N;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const float d = v[i] - mean;
    q += (t + i * 256 < N) ? d * d : 0.f;
  }
  const float m2 = block_sum_256(q, red);
  const float var = m2 / (float)N;
  const float invstd = 1.0f / sqrtf(var + eps);
  const float sc = gamma[c] * invstd, sh = beta[c] - mean * sc;
  if (t == 0) {
    save[c] = mean;
    save[C + c] = invstd;
    save[2 * C + c] = sc;
    save[3 * C + c] = sh;
    const float unbiased = N > 1 ? m2 / (float)(N - 1) : var;
    running_stats_update(&rm[c], &rv[c], rm[c], rv[c], momentum, mean, unbiased);
  }
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    if (t + i * 256 < N) {
      y_raw[off[i]] = v[i];
      y[off[i]] = lrelu(fmaf(v[i], sc, sh), slope);
    }
  }
  }
}

// same, any channel size: values are re-read from y_raw between the passes
__global__ __launch_bounds__(256) void splitk_fwd_epilogue_big_kernel(const float* __restrict__ part, int splitk,
                                                                      size_t part_stride, const float* __restrict__ bias,
                                                                      const float* __restrict__ gamma,
                                                                      const float* __restrict__ beta, float* rm, float* rv,
                                                                      float* __restrict__ y_raw, float* __restrict__ y,
                                                                      float* __restrict__ save, int B, int C, int HW, int ep,
                                                                      float slope, float eps, float momentum) {
  prefetch_kernargs<192>();
  __shared__ float red[4];
  const int c = blockIdx.x, t = threadIdx.x;
  const int N = B * HW;
  const float bsv = bias ? bias[c] : 0.f;
  float sc = 1.f, sh = 0.f;
  if (ep == EP_BN_EVAL) {
    const float inv = 1.0f / sqrtf(rv[c] + eps);
    sc = gamma[c] * inv;
    sh = beta[c] - rm[c] * sc;
  }
  float s1 = 0.f;
  for (int e = t; e < N; e += 256) {
    const int b = e / HW, pix = e - b * HW;
    const size_t off = ((size_t)b * C + c) * HW + pix;
    const float v = sum_slices(part + off, splitk, part_stride, bsv);
    if (ep == EP_RAW_STATS) {
      y_raw[off] = v;
      s1 += v;
    } else {
      float o = v;
      if (ep == EP_BN_EVAL) o = lrelu(fmaf(o, sc, sh), slope);
      if (ep == EP_LRELU) o = lrelu(o, slope);
      y[off] = o;
    }
  }
  if (ep != EP_RAW_STATS) return;
  const float mean = block_sum_256(s1, red) / (float)N;
  float q = 0.f;
  for (int e = t; e < N; e += 256) {
    const int b = e / HW, pix = e - b * HW;
    const float d = y_raw[((size_t)b * C + c) * HW + pix] - mean;   // written by this thread above
    q += d * d;
  }
  const float m2 = block_sum_256(q, red);
  const float var = m2 / (float)N;
  const float invstd = 1.0f / sqrtf(var + eps);
  sc = gamma[c] * invstd;
  sh = beta[c] - mean * sc;
  if (t == 0) {
    save[c] = mean;
    save[C + c] = invstd;
    save[2 * C + c] = sc;
    save[3 * C + c] = sh;
    const float unbiased = N > 1 ? m2 / (float)(N - 1) : var;
    running_stats_update(&rm[c], &rv[c], rm[c], rv[c], momentum, mean, unbiased);
  }
  for (int e = t; e < N; e += 256) {
    const int b = e / HW, pix = e - b * HW;
    const size_t off = ((size_t)b * C + c) * HW + pix;
    y[off] = lrelu(fmaf(y_raw[off], sc, sh), slope);
  }
}

// data gradient: dx = sum of partials; UP2: dx2 (residual grad, width W) = sum, dx (half width) = pair sums
__global__ __launch_bounds__(256) void splitk_dgrad_epilogue_kernel(const float* __restrict__ part, int splitk,
                                                                    size_t part_stride, float* __restrict__ dx,
                                                                    float* __restrict__ dx2, size_t n, int up2) {
  prefetch_kernargs<192>();
  if (!up2) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
      dx[i] = sum_slices(part + i, splitk, part_stride, 0.f);
    }
  } else {
    const size_t half = n >> 1;   // W is even: pairs never straddle rows
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < half; i += (size_t)gridDim.x * 256) {
      float v0 = 0.f, v1 = 0.f;
      for (int k0 = 0; k0 < splitk; k0 += 4) {
        float2 tmp[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) tmp[kk] = *(const float2*)(part + (size_t)min(k0 + kk, splitk - 1) * part_stride + 2 * i);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          if (k0 + kk < splitk) { v0 += tmp[kk].x; v1 += tmp[kk].y; }
      }
      dx2[2 * i] = v0;
      dx2[2 * i + 1] = v1;
      dx[i] = v0 + v1;
    }
  }
}

int launch_splitk_fwd_epilogue(const float* part, int splitk, size_t part_stride, const float* bias, const float* gamma,
                               const float* beta, float* rm, float* rv, float* y_raw, float* y, float* save, int B, int C,
                               int HW, int ep, float slope, float eps, float momentum, hipStream_t s, int sg) {
  TimingScope ts(s, 0, 4.0 * B * C * HW * (splitk + 3), "splitk_fwd_epilogue C%d N%d splitk%d ep%d%s", C, B * HW, splitk, ep, sg > 1 ? " pair" : "");
  if (ts.skip()) return 0;
  if (ep != EP_RAW_STATS) sg = 1;
  if (sg > 1) B /= sg;                  // MS_DT_STAT_PAIR: clips per statistics group (the register-resident kernels only)
  if (sg > 1 && (long)B * HW > 4096) return set_error("splitk_fwd_epilogue: statistics groups need <= 4096 values per channel and group");
#define MS_SKE(K) hipLaunchKernelGGL(K, dim3(C), dim3(256), 0, s, part, splitk, part_stride, bias, gamma, beta, rm, rv, \
                                     y_raw, y, save, B, C, HW, ep, slope, eps, momentum, sg)
  const long n = (long)B * HW;
  if (n <= 256) MS_SKE(splitk_fwd_epilogue_kernel<1>);
  else if (n <= 512) MS_SKE(splitk_fwd_epilogue_kernel<2>);
  else if (n <= 1024) MS_SKE(splitk_fwd_epilogue_kernel<4>);
  else if (n <= 2048) MS_SKE(splitk_fwd_epilogue_kernel<8>);
  else if (n <= 4096) MS_SKE(splitk_fwd_epilogue_kernel<16>);
  else hipLaunchKernelGGL(splitk_fwd_epilogue_big_kernel, dim3(C), dim3(256), 0, s, part, splitk, part_stride, bias, gamma, beta, rm, rv,
                          y_raw, y, save, B, C, HW, ep, slope, eps, momentum);
#undef MS_SKE
  return check_launch("splitk_fwd_epilogue_kernel");
}

int launch_splitk_dgrad_epilogue(const float* part, int splitk, size_t part_stride, float* dx, float* dx2, size_t n, int W,
                                 int up2, hipStream_t s) {
  (void)W;
  TimingScope ts(s, 0, 4.0 * n * (splitk + 1), "splitk_dgrad_epilogue n%zu splitk%d", n, splitk);
  if (ts.skip()) return 0;
  const size_t work = up2 ? n / 2 : n;
  int blocks = (int)std::min<size_t>((work + 255) / 256, 2048);
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(splitk_dgrad_epilogue_kernel, dim3(blocks), dim3(256), 0, s, part, splitk, part_stride, dx, dx2, n, up2);
  return check_launch("splitk_dgrad_epilogue_kernel");
}

// wgrad: choose split count so the grid fills the chip
int wgrad_splits(int Cog, int Kg, int groups, int Npix) {
  const long tiles = (long)cdiv(Cog, 64) * cdiv(Kg, 64) * groups;
  int splits = 1;
  if (tiles < 512) splits = (int)((512 + tiles - 1) / tiles);
  const int max_splits = cdiv(Npix, 64);   // at least 2 reduction steps (32 pixels each) per split: the deep UNet levels have 64-256 pixels
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  return splits;
}

template <bool UP2>
static void launch_wgrad_khw(const WgradArgs& a, dim3 grid, hipStream_t s) {
  const int kh = a.KH, kw = a.KW;
#define MS_WK(KH, KW) hipLaunchKernelGGL((wgrad_kernel<1, 1, KH, KW, UP2>), grid, dim3(256), 0, s, a)
  if constexpr (UP2) {
    if (kh == 1 && kw == 3) MS_WK(1, 3);
    else MS_WK(0, 0);
  } else {
    if (kh == 1 && kw == 3) MS_WK(1, 3);
    else if (kh == 1 && kw == 1) MS_WK(1, 1);
    else if (kh == 1 && kw == 4) MS_WK(1, 4);
    else if (kh == 3 && kw == 3) MS_WK(3, 3);
    else if (kh == 4 && kw == 4) MS_WK(4, 4);
    else if (kh == 3 && kw == 8) MS_WK(3, 8);
    else MS_WK(0, 0);
  }
#undef MS_WK
}

int launch_reduce_splits(const float* part, float* out, int n, int splits, hipStream_t s) {
  TimingScope ts(s, 0, 4.0 * n * (splits + 1), "wgrad_reduce_splits n%d splits%d", n, splits);
  if (ts.skip()) return 0;
  if (splits >= 32 && n <= 65536)
    hipLaunchKernelGGL(reduce_splits_wave_kernel, dim3(cdiv(n, 4)), dim3(256), 0, s, part, out, n, splits);
  else
    hipLaunchKernelGGL(reduce_splits_kernel, dim3(min(cdiv(n, 256), 2048)), dim3(256), 0, s, part, out, n, splits);
  return check_launch("reduce_splits_kernel");
}

int launch_reduce_splits_multi(ReduceBatch& rb, hipStream_t s) {
  int blocks = 0;
  double bytes = 0;
  for (int j = 0; j < rb.n; ++j) {
    ReduceJob& jb = rb.job[j];
    jb.wave = (jb.splits >= 32 && jb.n <= 65536) ? 1 : 0;          // as launch_reduce_splits
    blocks += jb.wave ? cdiv(jb.n, 4) : std::max(1, std::min(cdiv(jb.n, 1024), 512));
    jb.block_end = blocks;
    bytes += 4.0 * jb.n * (jb.splits + 2);
  }
  TimingScope ts(s, 0, bytes, "wgrad_reduce_multi jobs%d", rb.n);
  if (ts.skip()) return 0;
  hipLaunchKernelGGL(reduce_splits_multi_kernel, dim3(blocks), dim3(256), 0, s, rb);
  return check_launch("reduce_splits_multi_kernel");
}

struct PendingWgrad { WgradArgs a; int up2, nwg; double flops, bytes; };
static std::vector<PendingWgrad> g_pending_wg;       // process-wide, like the patch kernels' queue (wgrad_patch.hip)
static std::mutex g_pending_wg_mu;

void wgrad_gather_discard() {
  std::lock_guard<std::mutex> lk(g_pending_wg_mu);
  g_pending_wg.clear();
}

template <int KH, int KW, bool UP2>
static void launch_wgrad_multi(const WgradBatch& b, hipStream_t s) {
  hipLaunchKernelGGL((wgrad_multi_kernel<KH, KW, UP2>), dim3(b.block_end[b.n - 1]), dim3(256), 0, s, b);
}

int wgrad_gather_flush(hipStream_t s) {
  std::vector<PendingWgrad> q;
  {
    std::lock_guard<std::mutex> lk(g_pending_wg_mu);
    q.swap(g_pending_wg);
  }
  // one launch per kernel shape: (1 x 4), (1 x 3), (1 x 3 on an upsample-add input), everything else through the run-time shape
  auto shape_of = [](const PendingWgrad& w) {
    if (w.a.KH == 1 && w.a.KW == 4 && !w.up2) return 0;
    if (w.a.KH == 1 && w.a.KW == 3) return w.up2 ? 2 : 1;
    return w.up2 ? 4 : 3;
  };
  for (int shape = 0; shape < 5; ++shape) {
    WgradBatch b;
    b.n = 0;
    long blocks = 0;
    double flops = 0, bytes = 0;
    auto launch = [&]() -> int {
      if (!b.n) return 0;
      TimingScope ts(s, flops, bytes, "wgrad_multi_kernel<%d>|conv_wgrad_gather multi shape%d jobs%d wgs%ld", shape, shape, b.n, blocks);
      int rc = 0;
      if (!ts.skip()) {
        switch (shape) {
          case 0: launch_wgrad_multi<1, 4, false>(b, s); break;
          case 1: launch_wgrad_multi<1, 3, false>(b, s); break;
          case 2: launch_wgrad_multi<1, 3, true>(b, s); break;
          case 3: launch_wgrad_multi<0, 0, false>(b, s); break;
          default: launch_wgrad_multi<0, 0, true>(b, s); break;
        }
        rc = check_launch("wgrad_multi_kernel");
      }
      b.n = 0; blocks = 0; flops = bytes = 0;
      return rc;
    };
    for (const PendingWgrad& w : q) {
      if (shape_of(w) != shape) continue;
      if (b.n == WG_MAX_JOBS) { const int rc = launch(); if (rc) return rc; }
      blocks += w.nwg;
      b.block_end[b.n] = (int)blocks;
      b.job[b.n] = w.a;
      ++b.n;
      flops += w.flops; bytes += w.bytes;
    }
    const int rc = launch();
    if (rc) return rc;
  }
  return 0;
}

// queue = true: nothing of this call reads the result (dw written in place -- ADDED to, the slot may already hold the parameter's
// other uses of the step -- or slabs left for the caller's reduction): the launch waits for wgrad_gather_flush
int launch_wgrad(WgradArgs a, bool up2, float* dw, float* partial_ws, bool defer_reduce, hipStream_t s, bool queue) {
  const int ctot = a.groups * a.Cog;
  a.splits = wgrad_splits(a.Cog, a.Kg, a.groups, a.Npix);
  a.r_per_split = cdiv(cdiv(a.Npix, a.splits), 32) * 32;
  a.out = a.splits > 1 ? partial_ws : dw;
  a.accumulate = 0;
  dim3 grid(cdiv(a.Kg, 64), cdiv(a.Cog, 64), a.groups * a.splits);
  if (grid.y > 65535 || grid.z > 65535) return set_error("wgrad grid too large");
  const double batch = (double)a.Npix / ((double)a.OH * a.OW);
  const double flops = 2.0 * a.Cog * a.Kg * (double)a.Npix * a.groups;
  const double bytes = 4.0 * ((double)a.Npix * ctot + batch * a.src_ctotal * a.H * a.W + (double)ctot * a.Kg);
  if (queue && (a.splits == 1 || defer_reduce)) {
    PendingWgrad w;
    w.a = a;
    w.a.accumulate = a.splits == 1 ? 1 : 0;
    w.up2 = up2 ? 1 : 0;
    w.nwg = (int)(grid.x * grid.y * grid.z);
    w.flops = flops; w.bytes = bytes;
    std::lock_guard<std::mutex> lk(g_pending_wg_mu);
    g_pending_wg.push_back(w);
    return 0;
  }
  int rc;
  {
    TimingScope ts(s, flops, bytes,
                   "wgrad_kernel<1,1,%d,%d,%d>|conv_wgrad k%dx%d s%d Cog%d Kg%d g%d N%d splits%d", a.KH, a.KW, up2 ? 1 : 0, a.KH,
                   a.KW, a.SW, a.Cog, a.Kg, a.groups, a.Npix,
                   a.splits);
    if (ts.skip()) return 0;
    if (up2) launch_wgrad_khw<true>(a, grid, s); else launch_wgrad_khw<false>(a, grid, s);
    rc = check_launch("wgrad_kernel");
  }
  if (rc) return rc;
  if (a.splits > 1 && !defer_reduce) rc = launch_reduce_splits(partial_ws, dw, ctot * a.Kg, a.splits, s);
  return rc;
}

size_t dgrad_weight_elems(int groups, int Cog, int Cig, int KH, int KW, int SH, int SW) {
  return (size_t)SH * SW * groups * Cog * Cig * cdiv(KH, SH) * cdiv(KW, SW);
}

static int transpose_total(const TransposeJob& jb) {
  return jb.SH * jb.SW * jb.groups * jb.Cog * jb.Cig * cdiv(jb.KH, jb.SH) * cdiv(jb.KW, jb.SW);
}

int launch_transpose_weight(const float* w, float* wt, int groups, int Cog, int Cig, int KH, int KW, int SH, int SW,
                            int PH, int PW, int flip, hipStream_t s) {
  TransposeJob jb = {w, wt, groups, Cog, Cig, KH, KW, SH, SW, PH, PW, flip, 0};
  const int total = transpose_total(jb);
  TimingScope ts(s, 0, 8.0 * total, "transpose_weight n%d", total);
  if (ts.skip()) return 0;
  hipLaunchKernelGGL(transpose_weight_kernel, dim3(min(cdiv(total, 256), 4096)), dim3(256), 0, s, jb);
  return check_launch("transpose_weight_kernel");
}

int launch_transpose_weight_multi(TransposeBatch& tb, hipStream_t s) {
  int blocks = 0;
  double total = 0;
  for (int j = 0; j < tb.n; ++j) {
    const int n = transpose_total(tb.job[j]);
    blocks += std::max(1, std::min(cdiv(n, 1024), 256));
    tb.job[j].block_end = blocks;
    total += n;
  }
  TimingScope ts(s, 0, 8.0 * total, "transpose_weight_multi jobs%d n%.0f", tb.n, total);
  if (ts.skip()) return 0;
  hipLaunchKernelGGL(transpose_weight_multi_kernel, dim3(blocks), dim3(256), 0, s, tb);
  return check_launch("transpose_weight_multi_kernel");
}

// ---------------------------------------------------------------------------------------------
// self test: C(32x32) = A(32xK) * B(Kx32) with the same fragment maps as the kernels above
__global__ void selftest_mfma_kernel(const float* A, const float* B, float* C, int K) {
  const int lane = threadIdx.x;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k = 0; k < K; k += 2) {
    const int kr = k + (lane >> 5);
    const float a = kr < K ? A[(lane & 31) * K + kr] : 0.f;
    const float b = kr < K ? B[kr * 32 + (lane & 31)] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    C[row * 32 + (lane & 31)] = acc[r];
  }
}

}  // namespace ms

extern "C" int ms_selftest_mfma(const float* A, const float* B, float* C, int K, void* stream) {
  hipLaunchKernelGGL(ms::selftest_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, C, K);
  return ms::check_launch("selftest_mfma_kernel");
}
