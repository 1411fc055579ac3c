// Patch-staged convolution for gfx950 (MI355X): the forward conv of every block whose output rows are >= 16 wide.
//
// A workgroup owns BM output channels x a TH x TW block of output pixels of one image.  Per K-chunk it stages the RAW
// input patch [CK channels][(TH-1)*S+KH rows][(TW-1)*S+KW cols] in LDS (each input element is loaded once, coalesced
// along the time/frequency axis, with no im2col replication) and the matching weight slice [CK*KH*KW][BM].  The MFMA
// B operand for k = (ci,kh,kw), pixel (ty,tx) is patch[ci][ty*S+kh][tx*S+kw] = lane_base + compile-time offset, so the
// inner loop is ds_read_b32 (immediate offsets) + v_mfma_f32_32x32x2_f32 only: no address arithmetic, no decode.
// All per-thread global/LDS offsets of the staging loads are chunk-invariant and computed once before the K loop.
//
// 1-D convs (KH == 1) treat the batch axis as the row axis of one image, so tiles span batch items.
#include <algorithm>

#include "kernels.h"

namespace ms {

__device__ __attribute__((aligned(16))) float g_zero_word[4] = {0.f, 0.f, 0.f, 0.f};   // padding loads read this instead of branching

template <int KH, int KW>
struct PatchCfg {
  static constexpr int KHW = KH * KW;
  // channels per K-chunk: K_step = CK*KHW in [36, 64], multiple of 4
  static constexpr int CK = KHW == 1 ? 32 : KHW == 2 ? 16 : KHW == 3 ? 16 : KHW == 4 ? 16 : KHW == 9 ? 4
                            : KHW == 16 ? 4 : KHW == 24 ? 2 : 4;
  static constexpr int KSTEP = CK * KHW;
};

constexpr int patch_row_pitch(int pc, int sv, int tw) {
  // lanes of one 32-lane read group cover 32/TW tile rows: rows must land 'TW' banks apart
  if (tw >= 32) return pc;
  int rp = pc;
  while ((rp * sv) % 32 != tw % 32) ++rp;
  return rp;
}

template <int TM, int TN, int KH, int KW, int S, int TW, bool UP2>
__global__ __launch_bounds__(256) void conv_patch_kernel(const PatchArgs p) {
  using Cfg = PatchCfg<KH, KW>;
  constexpr int BM = 64 * TM, BN = 64 * TN, TH = BN / TW;
  constexpr int SV = (KH == 1) ? 1 : S;                 // KH == 1: rows are independent batch items
  constexpr int CK = Cfg::CK, KHW = Cfg::KHW, KSTEP = Cfg::KSTEP;
  constexpr int PR = (TH - 1) * SV + KH, PC = (TW - 1) * S + KW;
  constexpr int RP = patch_row_pitch(PC, SV, TW), CP = PR * RP;
  constexpr int LDA = BM + 1;   // odd pitch: the staging stores (4 k-rows apart per lane group) stay <= 2-way conflicted
  constexpr int STAGE = KSTEP * LDA + CK * CP + 3 * LDA + 4;   // + pad words: out-of-range staging stores land there
  constexpr int NPE = CK * PR * PC;                     // patch elements per chunk
  constexpr int NP = (NPE + 255) / 256;
  constexpr int NAV = BM * (KSTEP / 4);                 // float4 slots of the weight slice
  constexpr int NA = (NAV + 255) / 256;
  static_assert(KSTEP % 4 == 0 && BN % TW == 0, "bad patch configuration");
  __shared__ float smem[2 * STAGE];

  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wm = wid >> 1, wn = wid & 1, khalf = lane >> 5;
  // logical block id: channel tile fastest, then pixel tile, then (group, K slice); one contiguous range per XCD, so the
  // channel tiles that share an input patch sit behind the same L2
  const int vid = xcd_remap(blockIdx.x, p.gx * p.gy * p.gz);
  const int by_ = vid % p.gy, bx_ = (vid / p.gy) % p.gx, bz_ = vid / (p.gy * p.gx);
  const int g = bz_ / p.splitk, ks = bz_ - g * p.splitk, m0 = by_ * BM;
  const int tiles_per_img = p.tiles_y * p.tiles_x;
  const int img = bx_ / tiles_per_img;
  const int trem = bx_ - img * tiles_per_img;
  const int tyi = trem / p.tiles_x, txi = trem - tyi * p.tiles_x;
  const int oy0 = tyi * TH, ox0 = txi * TW;
  const int iy0 = oy0 * SV - p.PH, ix0 = ox0 * S - p.PW;
  const int cbase = p.bcast ? 0 : g * p.Kc;
  const int Kg = p.Kg;

  // ---- chunk-invariant staging offsets
  int goff[NP], loff[NP];                               // global offset rel. to the chunk base (-1: padding) / LDS
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int e = t + i * 256;
    const int ci = e / (PR * PC), rem = e - ci * (PR * PC), r = rem / PC, c = rem - r * PC;
    const int iy = iy0 + r, ix = ix0 + c;
    const bool ok = (e < NPE) & ((unsigned)iy < (unsigned)p.SRCH) & ((unsigned)ix < (unsigned)p.SRCW);
    goff[i] = ok ? (int)(ci * p.s_chan + iy * p.s_row + ix) : -1;
    loff[i] = e < NPE ? ci * CP + r * RP + c : CK * CP;   // dummy slot
  }
  int aoff[NA];                                         // weight offset rel. to (group base + k0); -1: row out of range
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int idx = t + i * 256;
    const int row = idx / (KSTEP / 4), kq = idx - row * (KSTEP / 4);
    aoff[i] = (idx < NAV && m0 + row < p.Mg) ? (m0 + row) * Kg + kq * 4 : -1;
  }
  const float* Ag = p.A + (size_t)g * p.Mg * Kg;
  const int img_base = img * p.s_img;

  // two register sets: chunk c+2 is in flight from HBM/L2 while chunk c+1 waits in registers and chunk c computes
  float4 ra0[NA], ra1[NA];
  float rb0[NP], rb1[NP];
  auto load_chunk = [&](int ci0, float4 (&ra)[NA], float (&rb)[NP]) {
    const int k0 = ci0 * KHW;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int idx = t + i * 256;
      const int kq = idx % (KSTEP / 4);
      const int k = k0 + kq * 4;
      const bool ok = (aoff[i] >= 0) & (k < Kg);
      const float* ap = ok ? Ag + (unsigned)(aoff[i] + k0) : g_zero_word;    // out of range: read zeros, no select after
      float4 v;
      if (p.a_vec) {
        v = *reinterpret_cast<const float4*>(ap);
      } else {
        v.x = ap[0];
        v.y = (ok & (k + 1 < Kg)) ? ap[1] : 0.f;
        v.z = (ok & (k + 2 < Kg)) ? ap[2] : 0.f;
        v.w = (ok & (k + 3 < Kg)) ? ap[3] : 0.f;
      }
      ra[i] = v;
    }
    const int cb = img_base + (cbase + ci0) * p.s_chan;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int e = t + i * 256;
      const int ci = e / (PR * PC);
      const bool ok = (goff[i] >= 0) & (ci0 + ci < p.Kc);
      if (UP2) {
        // x = nearest_up2(a) + r : a has half the row length, hence half of every stride
        const int o = ok ? cb + goff[i] : 0;
        const int x = o % p.SRCW;                        // column inside the row (strides are multiples of SRCW)
        const float* pa = ok ? p.src + (unsigned)(((o - x) >> 1) + (x >> 1)) : g_zero_word;
        const float* pr = ok ? p.src2 + (unsigned)o : g_zero_word;
        rb[i] = *pa + *pr;
      } else {
        const float* ps = ok ? p.src + (unsigned)(cb + goff[i]) : g_zero_word;
        rb[i] = *ps;
      }
    }
  };
  auto store_chunk = [&](int buf, const float4 (&ra)[NA], const float (&rb)[NP]) {
    float* As = smem + buf * STAGE;
    float* Ps = As + KSTEP * LDA;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int idx = t + i * 256;
      const int row = idx / (KSTEP / 4), kq = idx - row * (KSTEP / 4);
      const int base = idx < NAV ? kq * 4 * LDA + row : KSTEP * LDA + CK * CP;   // out of range: the pad words
      As[base + 0 * LDA] = ra[i].x;
      As[base + 1 * LDA] = ra[i].y;
      As[base + 2 * LDA] = ra[i].z;
      As[base + 3 * LDA] = ra[i].w;
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) Ps[loff[i]] = rb[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- per-lane operand bases: k and k+1 of an MFMA pair sit in lanes 0-31 / 32-63
  const int a_base = khalf * LDA + wm * TM * 32 + (lane & 31);
  int b_same[TN], b_row[TN], b_chan[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nloc = wn * TN * 32 + j * 32 + (lane & 31);
    const int ty = nloc / TW, tx = nloc - ty * TW;
    const int base = ty * SV * RP + tx * S;
    b_same[j] = base + khalf;                                         // k+1 = next tap in the same row
    b_row[j] = base + khalf * (RP - (KW - 1));                        // k+1 = first tap of the next kernel row
    b_chan[j] = base + khalf * (CP - (KH - 1) * RP - (KW - 1));       // k+1 = first tap of the next channel
  }

  const int chunk_beg = ks * p.chunks_per_split;
  const int nchunks = min((p.Kc + CK - 1) / CK - chunk_beg, p.chunks_per_split);
  auto compute_chunk = [&](int cur) {
    const float* As = smem + cur * STAGE;
    const float* Ps = As + KSTEP * LDA;
    // operands of the next group of k-pairs are read from LDS while the current group's MFMAs issue
    constexpr int NPAIR = KSTEP / 2, GP = 4, NG = (NPAIR + GP - 1) / GP;
    float av[2][GP][TM], bv[2][GP][TN];
    auto read_group = [&](int gi, int slot) {
#pragma unroll
      for (int q = 0; q < GP; ++q) {
        const int jj = gi * GP + q;
        if (jj < NPAIR) {
          const int k0 = 2 * jj;
          const int ci = k0 / KHW, rr = k0 - ci * KHW, kh = rr / KW, kw = rr - kh * KW;
          const int offb = ci * CP + kh * RP + kw;
#pragma unroll
          for (int i = 0; i < TM; ++i) av[slot][q][i] = As[a_base + k0 * LDA + i * 32];
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int base = (kw + 1 < KW) ? b_same[j] : (kh + 1 < KH) ? b_row[j] : b_chan[j];
            bv[slot][q][j] = Ps[base + offb];
          }
        }
      }
    };
    read_group(0, 0);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      if (gi + 1 < NG) read_group(gi + 1, (gi + 1) & 1);
#pragma unroll
      for (int q = 0; q < GP; ++q) {
        if (gi * GP + q < NPAIR) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[gi & 1][q][i], bv[gi & 1][q][j], acc[i][j], 0, 0, 0);
        }
      }
    }
  };
  load_chunk(chunk_beg * CK, ra0, rb0);
  if (nchunks > 1) load_chunk((chunk_beg + 1) * CK, ra1, rb1);
  store_chunk(0, ra0, rb0);
  __syncthreads();
  for (int ch = 0; ch < nchunks; ch += 2) {
    if (ch + 2 < nchunks) load_chunk((chunk_beg + ch + 2) * CK, ra0, rb0);
    compute_chunk(0);
    if (ch + 1 < nchunks) store_chunk(1, ra1, rb1);
    __syncthreads();
    if (ch + 1 >= nchunks) break;
    if (ch + 3 < nchunks) load_chunk((chunk_beg + ch + 3) * CK, ra1, rb1);
    compute_chunk(1);
    if (ch + 2 < nchunks) store_chunk(0, ra0, rb0);
    __syncthreads();
  }

  // ---------------- epilogue ----------------
  const int ctot = p.groups * p.Mg;
  int ooff[TN];
  bool cval[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nloc = wn * TN * 32 + j * 32 + (lane & 31);
    const int ty = nloc / TW, tx = nloc - ty * TW;
    const int oy = oy0 + ty, ox = ox0 + tx;
    cval[j] = (oy < p.OUTH) & (ox < p.OUTW);
    ooff[j] = img * p.o_img + (oy * p.o_sh + p.o_ry) * p.o_row + ox * p.o_sw + p.o_rx;   // + channel * o_chan
  }
  const int ep = p.ep;
  if (p.splitk > 1) {                 // raw partial tile in the output layout; a split-K epilogue kernel finishes
    float* part = p.part + (size_t)ks * p.part_stride;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
#pragma unroll
        for (int j = 0; j < TN; ++j)
          if (m < p.Mg && cval[j]) part[(size_t)ooff[j] + (size_t)(g * p.Mg + m) * p.o_chan] = acc[i][j][r];
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      const bool mval = m < p.Mg;
      const int chn = g * p.Mg + (mval ? m : 0);
      const float bsv = p.bias ? p.bias[chn] : 0.f;
      float sc = 1.f, sh = 0.f;
      if (ep == EP_BN_EVAL) {
        const float inv = 1.0f / sqrtf(p.bn_v[chn] + p.eps);
        sc = p.bn_g[chn] * inv;
        sh = p.bn_b[chn] - p.bn_m[chn] * sc;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float v = acc[i][j][r] + bsv;
        if (ep == EP_RAW_STATS) acc[i][j][r] = v;
        if (ep == EP_BN_EVAL) v = lrelu(fmaf(v, sc, sh), p.slope);
        if (ep == EP_LRELU) v = lrelu(v, p.slope);
        if (ep == EP_DGRAD_UP2) {
          // 1-D stride-1 data gradient of an upsample-add input: out2 = grad of the residual (full resolution),
          // out = grad of the half-resolution tensor = sum over the pair of columns (adjacent lanes)
          const float pr = __shfl_xor(v, 1);
          if (mval && cval[j]) {
            p.out2[(size_t)ooff[j] + (size_t)chn * p.o_chan] = v;
            if (!(lane & 1)) p.out[(size_t)(ooff[j] >> 1) + (size_t)chn * (p.o_chan >> 1)] = v + pr;
          }
        } else if (mval && cval[j]) {
          p.out[(size_t)ooff[j] + (size_t)chn * p.o_chan] = v;
        }
      }
    }
  }

  if (ep == EP_RAW_STATS) {
    // per-channel (sum, M2 about this tile's mean) over the tile's valid pixels, fixed order
    float* red = smem;  // [4][BM]
    const int cnt = min(TH, p.OUTH - oy0) * min(TW, p.OUTW - ox0);
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
      float* st = p.stats + ((size_t)bx_ * ctot + g * p.Mg + m0 + t) * 2;
      st[0] = red[t] + red[BM + t];
      st[1] = red[2 * BM + t] + red[3 * BM + t];
    }
    if (t == 0 && by_ == 0 && g == 0) p.counts[bx_] = (float)cnt;
  }
}

// ---------------------------------------------------------------------------------------------
// dispatch
int g_patch_min_wgs = 96;
int g_patch_force_splitk = 0;   // tuning knob: > 0 forces this split-K factor in the patch kernel
int g_patch_wide_min = 1 << 30;  // 64x128 tiles (TN=2) when they still give this many workgroups (tuning knob)
int g_patch_big_min = 1 << 30;   // 128x128 tiles when they still give this many workgroups (test/tuning knob)   // below this many workgroups the split-K im2col path is used instead
int patch_chunk_channels(int KH, int KW) {
  const int khw = KH * KW;
  return khw == 1 ? 32 : khw == 2 ? 16 : khw == 3 ? 16 : khw == 4 ? 16 : khw == 9 ? 4 : khw == 16 ? 4 : khw == 24 ? 2 : 4;
}

PatchPlan plan_patch(int nd, int Mg, int groups, int Kc, int KH, int KW, int SH, int SW, int B, int OH, int OW) {
  PatchPlan pl = {0, 1, 64, 0, 0, 0, 1, 1 << 30, 1};
  const int S = SW;
  if (nd == 2 && SH != SW) return pl;
  const bool known = (KH == 1 && KW == 2 && S == 1) || (KH == 2 && KW == 2 && S == 1) || (KH == 1 && KW == 3 && S == 1) ||
                     (KH == 1 && KW == 4 && S == 2) || (KH == 1 && KW == 4 && S == 1) ||
                     (KH == 1 && KW == 1 && S == 1) || (KH == 3 && KW == 3 && S == 1) || (KH == 4 && KW == 4 && S == 2) ||
                     (KH == 3 && KW == 8 && S == 1);
  if (!known) return pl;
  if (nd == 1 ? OW < 16 : OW < 15) return pl;
  const int rows = nd == 1 ? B : OH, imgs = nd == 1 ? 1 : B;
  int tw;
  if (nd == 1) tw = OW > 32 ? 64 : OW > 16 ? 32 : 16;
  else tw = OW > 16 ? 32 : 16;
  // big tile (128x128) only when it still gives >= 2 workgroups per CU
  int tm = 1;
  if (Mg >= 128) {
    const int th = 128 / tw;
    const long big = (long)cdiv(Mg, 128) * groups * imgs * cdiv(rows, th) * cdiv(OW, tw);
    if (big >= g_patch_big_min) tm = 2;
  }
  int tn = tm;
  if (tm == 1) {
    const int th2 = 128 / tw;
    const long wide = (long)cdiv(Mg, 64) * groups * imgs * cdiv(rows, th2) * cdiv(OW, tw);
    if (wide >= g_patch_wide_min) tn = 2;
  }
  const int th = 64 * tn / tw;
  pl.ok = 1; pl.tm = tm; pl.tw = tw; pl.tn = tn;
  pl.tiles_y = cdiv(rows, th); pl.tiles_x = cdiv(OW, tw);
  pl.n_tiles = imgs * pl.tiles_y * pl.tiles_x;
  const long base = (long)pl.n_tiles * cdiv(Mg, 64 * tm) * groups;
  // too few workgroups: the split-K im2col path spreads the weight stream better
  if (base < g_patch_min_wgs) pl.ok = 0;
  // fewer workgroups than 1.5 per CU and a long reduction: slice the channel chunks over workgroups
  const int nchunks = cdiv(Kc, patch_chunk_channels(KH, KW));
  if (pl.ok && g_patch_force_splitk > 0 && nchunks >= g_patch_force_splitk) {
    pl.chunks_per_split = cdiv(nchunks, g_patch_force_splitk);
    pl.splitk = cdiv(nchunks, pl.chunks_per_split);
  } else if (pl.ok && base < 384 && nchunks >= 4) {
    int sk = (int)std::min<long>(nchunks / 2, (384 + base - 1) / base);
    if (sk > 4) sk = 4;
    if (sk > 1) {
      pl.chunks_per_split = cdiv(nchunks, sk);
      pl.splitk = cdiv(nchunks, pl.chunks_per_split);
    }
  }
  return pl;
}

template <int TM, int TN, int KH, int KW, int S, bool UP2>
static void launch_patch_tw(const PatchArgs& a, int tw, dim3 grid, hipStream_t s) {
#define MS_PK(TW) hipLaunchKernelGGL((conv_patch_kernel<TM, TN, KH, KW, S, TW, UP2>), grid, dim3(256), 0, s, a)
  if constexpr (KH == 1) {
    if (tw == 64) MS_PK(64);
    else if (tw == 32) MS_PK(32);
    else MS_PK(16);
  } else {
    if (tw == 32) MS_PK(32);
    else MS_PK(16);
  }
#undef MS_PK
}

template <int TM, int TN>
static void launch_patch_k(const PatchArgs& a, int kh, int kw, int s_, int tw, bool up2, dim3 grid, hipStream_t s) {
  if (kh == 1 && kw == 3 && s_ == 1) {
    if (up2) launch_patch_tw<TM, TN, 1, 3, 1, true>(a, tw, grid, s);
    else launch_patch_tw<TM, TN, 1, 3, 1, false>(a, tw, grid, s);
  } else if (kh == 1 && kw == 4 && s_ == 2) launch_patch_tw<TM, TN, 1, 4, 2, false>(a, tw, grid, s);
  else if (kh == 1 && kw == 4 && s_ == 1) launch_patch_tw<TM, TN, 1, 4, 1, false>(a, tw, grid, s);
  else if (kh == 1 && kw == 1 && s_ == 1) launch_patch_tw<TM, TN, 1, 1, 1, false>(a, tw, grid, s);
  else if (kh == 1 && kw == 2 && s_ == 1) launch_patch_tw<TM, TN, 1, 2, 1, false>(a, tw, grid, s);
  else if (kh == 2 && kw == 2 && s_ == 1) launch_patch_tw<TM, TN, 2, 2, 1, false>(a, tw, grid, s);
  else if (kh == 3 && kw == 3 && s_ == 1) launch_patch_tw<TM, TN, 3, 3, 1, false>(a, tw, grid, s);
  else if (kh == 4 && kw == 4 && s_ == 2) launch_patch_tw<TM, TN, 4, 4, 2, false>(a, tw, grid, s);
  else launch_patch_tw<TM, TN, 3, 8, 1, false>(a, tw, grid, s);
}

int launch_patch(const PatchArgs& a, const PatchPlan& pl, int KH, int KW, int S, bool up2, double flops, double bytes,
                 hipStream_t s) {
  const int bm = 64 * pl.tm;
  PatchArgs b = a;
  b.gx = pl.n_tiles; b.gy = cdiv(a.Mg, bm); b.gz = a.groups * a.splitk;
  if ((double)b.gx * b.gy * b.gz > 2.0e9) return set_error("conv grid too large");
  dim3 grid(b.gx * b.gy * b.gz);
  if (a.splitk < 1 || (a.splitk > 1 && !a.part)) return set_error("patch conv: bad split-K setup");
  TimingScope ts(s, flops, bytes, "conv_patch_kernel<%d,%d,%d,%d,%d,%d,%d>|conv_%s_patch k%dx%d s%d Mg%d Kg%d g%d tiles%d tile%d tw%d splitk%d%s",
                 pl.tm, pl.tm, KH, KW, S, pl.tw, up2 ? 1 : 0, a.is_dgrad ? "dgrad" : "fwd", KH, KW, S, a.Mg, a.Kg,
                 a.groups, pl.n_tiles, bm, pl.tw, a.splitk, a.ep == EP_RAW_STATS ? " +bnstats" : "");
  if (pl.tm == 2) launch_patch_k<2, 2>(b, KH, KW, S, pl.tw, up2, grid, s);
  else if (pl.tn == 2) launch_patch_k<1, 2>(b, KH, KW, S, pl.tw, up2, grid, s);
  else launch_patch_k<1, 1>(b, KH, KW, S, pl.tw, up2, grid, s);
  return check_launch("conv_patch_kernel");
}

}  // namespace ms

extern "C" int ms_debug_set_patch_tuning(int wide_tile_min_workgroups, int force_splitk) {
  ms::g_patch_wide_min = wide_tile_min_workgroups > 0 ? wide_tile_min_workgroups : (1 << 30);
  ms::g_patch_force_splitk = force_splitk > 0 ? force_splitk : 0;
  return 0;
}

extern "C" int ms_debug_set_patch_min_workgroups(int n) {
  const int old = ms::g_patch_min_wgs;
  ms::g_patch_min_wgs = n;
  return old;
}
