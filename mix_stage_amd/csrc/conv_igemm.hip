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
#include "kernels.h"

namespace ms {

template <int TM, int TN, int KH_, int KW_, bool TRANSPOSED, bool UP2>
__global__ __launch_bounds__(256) void igemm_gather_kernel(const GatherArgs p) {
  constexpr int BM = 64 * TM, BN = 64 * TN, BK = 16;
  constexpr int LDA = BM + 2, LDB = BN;
  constexpr int STAGE = BK * LDA + BK * LDB;
  constexpr int BROWS = 256 / BN, BITER = BK / BROWS;
  __shared__ float smem[2 * STAGE];

  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int g = blockIdx.z, m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int KH = KH_ ? KH_ : p.KH, KW = KW_ ? KW_ : p.KW, KHW = KH * KW;
  const int Kg = p.Kg;

  // ---- the B-tile column (output pixel) this thread stages: decoded once
  const int nl = t % BN, kk0 = t / BN;
  const int n = n0 + nl;
  const bool nvalid = n < p.Npix;
  const int ohw = p.OUTH * p.OUTW;
  int pb = 0, py = 0, px = 0;
  if (nvalid) {
    pb = n / ohw;
    const int rem = n - pb * ohw;
    py = rem / p.OUTW;
    px = rem - py * p.OUTW;
  }
  const int by = TRANSPOSED ? py + p.PH : py * p.SH - p.PH;
  const int bx = TRANSPOSED ? px + p.PW : px * p.SW - p.PW;
  const int cbase = p.bcast ? 0 : g * p.Kc;
  const int chan0 = pb * p.src_ctotal + cbase;  // channel-row index of (pb, cbase)

  float4 ra[TM];
  float rb[BITER];

  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int idx = t + i * 256, row = idx >> 2, kq = idx & 3;
      const int m = m0 + row, k = k0 + kq * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < p.Mg && k < Kg) {
        const float* ap = p.A + (size_t)(g * p.Mg + m) * Kg + k;
        if (p.a_vec) {
          v = *reinterpret_cast<const float4*>(ap);
        } else {
          v.x = ap[0];
          if (k + 1 < Kg) v.y = ap[1];
          if (k + 2 < Kg) v.z = ap[2];
          if (k + 3 < Kg) v.w = ap[3];
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BITER; ++i) {
      const int k = k0 + kk0 + i * BROWS;
      float v = 0.f;
      if (nvalid && k < Kg) {
        const int kc = k / KHW, r = k - kc * KHW, kh = r / KW, kw = r - kh * KW;
        int sy, sx;
        bool ok;
        if (!TRANSPOSED) {
          sy = by + kh;
          sx = bx + kw;
          ok = (unsigned)sy < (unsigned)p.SRCH && (unsigned)sx < (unsigned)p.SRCW;
        } else {
          const int ty = by - kh, tx = bx - kw;
          ok = ty >= 0 && tx >= 0;
          if (p.SH == 1) { sy = ty; } else if (p.SH == 2) { ok = ok && !(ty & 1); sy = ty >> 1; }
          else { sy = ty / p.SH; ok = ok && (sy * p.SH == ty); }
          if (p.SW == 1) { sx = tx; } else if (p.SW == 2) { ok = ok && !(tx & 1); sx = tx >> 1; }
          else { sx = tx / p.SW; ok = ok && (sx * p.SW == tx); }
          ok = ok && sy < p.SRCH && sx < p.SRCW;
        }
        if (ok) {
          const int crow = chan0 + kc;
          if (UP2 && !TRANSPOSED) {
            v = p.src[(size_t)crow * (p.SRCW >> 1) + (sx >> 1)] + p.src2[(size_t)crow * p.SRCW + sx];
          } else {
            v = p.src[((size_t)crow * p.SRCH + sy) * p.SRCW + sx];
          }
        }
      }
      rb[i] = v;
    }
  };
  auto store_tiles = [&](int buf) {
    float* As = smem + buf * STAGE;
    float* Bs = As + BK * LDA;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int idx = t + i * 256, row = idx >> 2, kq = idx & 3;
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

  const int nk = (Kg + BK - 1) / BK;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  const int a_off = wm * TM * 32 + (lane & 31), b_off = wn * TN * 32 + (lane & 31), khalf = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tiles((kt + 1) * BK);
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
  int ooff[TN];
  bool cval[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nc = n0 + wn * TN * 32 + j * 32 + (lane & 31);
    cval[j] = nc < p.Npix;
    const int b = cval[j] ? nc / ohw : 0;
    const int pix = nc - b * ohw;
    ooff[j] = b * ctot * ohw + pix;   // + channel*ohw
  }
  const int ep = p.ep;
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
        // out2 = grad of the residual (full resolution); out = grad of the half-resolution tensor
        const int hw = ohw >> 1;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const float v = acc[i][j][r];
          const float pr = __shfl_xor(v, 1);
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
        const float bsv = p.bias ? p.bias[ch] : 0.f;
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
    const int cnt = min(BN, p.Npix - n0);
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
template <int TM, int TN, int KH_, int KW_, bool UP2>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs p) {
  constexpr int BM = 64 * TM, BN = 64 * TN, BR = 32;
  constexpr int LDA = BM + 1, LDB = BN + 1;
  constexpr int STAGE = BR * LDA + BR * LDB;
  constexpr int AIT = BM / 8, BIT = BN / 8;
  __shared__ float smem[2 * STAGE];

  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int g = blockIdx.z / p.splits, sp = blockIdx.z - g * p.splits;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
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
        if (m < p.Cog && nc < p.Kg) outp[(size_t)(g * p.Cog + m) * p.Kg + nc] = acc[i][j][r];
      }
    }
}

// out[i] = sum_s part[s][i]   (fixed order -> bitwise reproducible)
__global__ void reduce_splits_kernel(const float* __restrict__ part, float* __restrict__ out, int n, int splits) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += part[(size_t)k * n + i];
    out[i] = s;
  }
}

// wt[g][ci][co][khw] = w[g][co][ci][khw]
__global__ void transpose_weight_kernel(const float* __restrict__ w, float* __restrict__ wt, int groups, int Cog,
                                        int Cig, int KHW) {
  const int per_g = Cog * Cig * KHW;
  const int total = groups * per_g;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int g = i / per_g;
    int r = i - g * per_g;           // index in wt[g]: (ci, co, khw)
    const int ci = r / (Cog * KHW);
    r -= ci * Cog * KHW;
    const int co = r / KHW, k = r - co * KHW;
    wt[i] = w[(size_t)g * per_g + ((size_t)co * Cig + ci) * KHW + k];
  }
}

// ---------------------------------------------------------------------------------------------
// launch helpers
struct TileCfg { int tm, tn; };

static TileCfg pick_tile(int Mg, int Npix, int groups) {
  // big tile only when it still fills the chip (256 CUs) at >= 2 workgroups per CU
  const long big = (long)cdiv(Mg, 128) * cdiv(Npix, 128) * groups;
  if (Mg >= 128 && big >= 512) return {2, 2};
  return {1, 1};
}

template <int TM, int TN, bool TR, bool UP2>
static void launch_gather_khw(const GatherArgs& a, dim3 grid, hipStream_t s) {
  const int kh = a.KH, kw = a.KW;
#define MS_GK(KH, KW) hipLaunchKernelGGL((igemm_gather_kernel<TM, TN, KH, KW, TR, UP2>), grid, dim3(256), 0, s, a)
  if constexpr (UP2) {
    if (kh == 1 && kw == 3) MS_GK(1, 3);
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

int launch_gather(const GatherArgs& a, bool transposed, bool up2, int* n_tiles_out, hipStream_t s) {
  const TileCfg c = pick_tile(a.Mg, a.Npix, a.groups);
  const int bm = 64 * c.tm, bn = 64 * c.tn;
  dim3 grid(cdiv(a.Npix, bn), cdiv(a.Mg, bm), a.groups);
  if (n_tiles_out) *n_tiles_out = grid.x;
  if (grid.y > 65535 || grid.z > 65535) return set_error("conv grid too large");
  const double batch = (double)a.Npix / ((double)a.OUTH * a.OUTW);
  TimingScope ts(s, 2.0 * a.Mg * a.Kg * (double)a.Npix * a.groups,
                 4.0 * ((double)a.groups * a.Mg * a.Kg + batch * a.src_ctotal * a.SRCH * a.SRCW + (double)a.Npix * a.groups * a.Mg),
                 "%s k%dx%d s%d Mg%d Kg%d g%d N%d tile%d%s", transposed ? "conv_dgrad" : "conv_fwd", a.KH, a.KW, a.SW, a.Mg,
                 a.Kg, a.groups, a.Npix, bm, a.ep == EP_RAW_STATS ? " +bnstats" : "");
  if (transposed) up2 = false;  // the UP2 split store of the data gradient is a runtime epilogue (EP_DGRAD_UP2)
  if (c.tm == 2) {
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

int gather_n_tiles(int Mg, int Npix, int groups) {
  const TileCfg c = pick_tile(Mg, Npix, groups);
  return cdiv(Npix, 64 * c.tn);
}

int gather_tile_n(int Mg, int Npix, int groups) { return 64 * pick_tile(Mg, Npix, groups).tn; }

// wgrad: choose split count so the grid fills the chip
int wgrad_splits(int Cog, int Kg, int groups, int Npix) {
  const long tiles = (long)cdiv(Cog, 64) * cdiv(Kg, 64) * groups;
  int splits = 1;
  if (tiles < 512) splits = (int)((512 + tiles - 1) / tiles);
  const int max_splits = cdiv(Npix, 256);  // at least 8 reduction steps per split
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

int launch_wgrad(WgradArgs a, bool up2, float* dw, float* partial_ws, hipStream_t s) {
  const int ctot = a.groups * a.Cog;
  a.splits = wgrad_splits(a.Cog, a.Kg, a.groups, a.Npix);
  a.r_per_split = cdiv(cdiv(a.Npix, a.splits), 32) * 32;
  a.out = a.splits > 1 ? partial_ws : dw;
  dim3 grid(cdiv(a.Kg, 64), cdiv(a.Cog, 64), a.groups * a.splits);
  if (grid.y > 65535 || grid.z > 65535) return set_error("wgrad grid too large");
  int rc;
  {
    const double batch = (double)a.Npix / ((double)a.OH * a.OW);
    TimingScope ts(s, 2.0 * a.Cog * a.Kg * (double)a.Npix * a.groups,
                   4.0 * ((double)a.Npix * ctot + batch * a.src_ctotal * a.H * a.W + (double)ctot * a.Kg),
                   "conv_wgrad k%dx%d s%d Cog%d Kg%d g%d N%d splits%d", a.KH, a.KW, a.SW, a.Cog, a.Kg, a.groups, a.Npix,
                   a.splits);
    if (up2) launch_wgrad_khw<true>(a, grid, s); else launch_wgrad_khw<false>(a, grid, s);
    rc = check_launch("wgrad_kernel");
  }
  if (rc) return rc;
  if (a.splits > 1) {
    const int n = ctot * a.Kg;
    TimingScope ts(s, 0, 4.0 * n * (a.splits + 1), "wgrad_reduce_splits n%d splits%d", n, a.splits);
    hipLaunchKernelGGL(reduce_splits_kernel, dim3(min(cdiv(n, 256), 2048)), dim3(256), 0, s, partial_ws, dw, n, a.splits);
    rc = check_launch("reduce_splits_kernel");
  }
  return rc;
}

int launch_transpose_weight(const float* w, float* wt, int groups, int Cog, int Cig, int KHW, hipStream_t s) {
  const int total = groups * Cog * Cig * KHW;
  TimingScope ts(s, 0, 8.0 * total, "transpose_weight n%d", total);
  hipLaunchKernelGGL(transpose_weight_kernel, dim3(min(cdiv(total, 256), 4096)), dim3(256), 0, s, w, wt, groups, Cog, Cig, KHW);
  return check_launch("transpose_weight_kernel");
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
