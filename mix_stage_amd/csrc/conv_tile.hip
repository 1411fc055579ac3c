// Lean patch-staged convolution for gfx950 (MI355X), fp32: the 2-D layers (AudioEncoder, reference src/model/layers.py:159-199)
// forward and data gradient on 128 x 128 (or 64 x 256) output tiles.
//
// Same decomposition as conv_patch.hip -- a workgroup owns output channels x a TH x TW block of output pixels of one image and
// stages, per K chunk, the RAW input patch and the matching weight slice in LDS; the MFMA B operand of k = (ci, kh, kw) is
// patch[ci][ty*S + kh][tx*S + kw] = per-lane base + compile-time offset -- but built on what the round-5 weight-gradient
// experiments measured (wgrad_patch.hip, "lean form"): on fp32 MFMA the instruction stream around the MFMAs is the limit, and a
// 16-byte staging load costs the CU ~60 cycles of issue whatever it hits.  So:
//   * a wave owns 64 x 64 of the tile (four 32 x 32 accumulators: an A fragment feeds two MFMAs, a B fragment two), a workgroup
//     128 x 128 or 64 x 256: 4 x the MFMAs per barrier and per staged byte of the 64 x 64 kernel;
//   * the patch is staged as ALIGNED 16-byte windows of each input row (window start rounded down to a multiple of 4 columns)
//     instead of dword by dword: 1-2 loads per thread and chunk instead of 4-6;
//   * loads run two chunks ahead in two register sets; chunk offsets travel in the scalar offset, validity of rows / columns /
//     padding is chunk-invariant (out-of-range offsets read zeros): no address arithmetic in the K loop;
//   * batch statistics of a tile (EP_RAW_STATS) are reduced in registers (DPP row sums) and Chan-combined over the waves in a
//     fixed order: no [channel][pixel] transposition through LDS.
#include <algorithm>
#include <cstdint>

#include "kernels.h"

#ifndef MS_TILE_OCC
#define MS_TILE_OCC 3
#endif
namespace ms {

int g_conv_tile = 1;          // ms_debug_set_conv_tile: 0 keeps every layer on conv_patch_kernel

template <int KH, int KW>
struct TileCfg {
  static constexpr int KHW = KH * KW;
  // channels per K chunk: KSTEP = CK*KHW in [32, 48], a multiple of 4
  static constexpr int CK = KHW == 4 ? 8 : KHW == 9 ? 4 : KHW == 16 ? 2 : KHW == 24 ? 2 : 4;
  static constexpr int KSTEP = CK * KHW;
};

constexpr int tile_row_pitch(int win, int sv, int tw) {
  // a multiple of 4 (16-byte LDS stores); rows of one 32-lane read group must land 16 banks apart when TW == 16
  int rp = win;
  if (tw >= 32) return rp;
  while ((rp * sv) % 32 != 16) rp += 4;
  return rp;
}

// AM 1: A rows [Mg][Kg], 16-byte loads.  AM 2: data gradient straight from the conv weight w[co][ci][tap] (stride-1 convs):
// A(ci, (co, tap')) = w[co][ci][KHW-1-tap'], Mg = Cin_g, Kc = Cout_g.
template <int KH, int KW, int S, int TW, int AM, int WM, int WN>
// (three workgroups per CU where the instance fits 168 registers: the 4 x 4 / 2 x 2 shapes and the 64 x 256 forms of the (3, .) shapes
// with row-major weights; the 128 x 128 3 x 3 instances and the in-place-weight instances need ~200)
__global__ __launch_bounds__(256, (KH * KW == 16 || KH * KW == 4 || (AM == 1 && WM == 1)) ? MS_TILE_OCC : 2) void conv_tile_kernel(const PatchArgs p) {
  prefetch_kernargs<sizeof(PatchArgs)>();
  using Cfg = TileCfg<KH, KW>;
  static_assert(WM * WN == 4, "four waves");
  constexpr int BM = 64 * WM, BN = 64 * WN, TH = BN / TW;
  constexpr int SV = S;
  constexpr int CK = Cfg::CK, KHW = Cfg::KHW, KSTEP = Cfg::KSTEP;
  constexpr int PR = (TH - 1) * SV + KH;
  constexpr int WIN = ((TW - 1) * S + KW + 3 + 3) / 4 * 4, W4 = WIN / 4;      // window start rounded down to a multiple of 4: shift <= 3
  constexpr int RP = tile_row_pitch(WIN, SV, TW), CP = PR * RP;
  // weight slice in LDS: [row m][k], k contiguous as it lies in memory -- pitch = 4 * odd (mod 64): the 16 lanes of a ds_read_b128
  // group (16 consecutive rows) cover the 64 banks.  The two halves of a wave take the two HALVES of the chunk's channels: lanes
  // 0-31 k = kk, lanes 32-63 k = KSTEP / 2 + kk -- the same tap of a channel CK / 2 further on, i.e. ONE constant offset in the patch.
  constexpr int KH2 = KSTEP / 2;                                               // k values per half = (CK / 2) * KHW
  constexpr int AP = KSTEP % 8 == 4 ? KSTEP : KSTEP + 4;                       // 36 -> 36, 32 -> 36, 48 -> 52
  constexpr int ABUF = BM * AP;
  constexpr int STAGE = ABUF + CK * CP + 16;                                   // + pad words: idle staging slots land there
  static_assert(CK % 2 == 0 && (AP / 4) % 2 == 1 && AP >= KSTEP, "bad weight-slice pitch");
  constexpr int NB4 = CK * PR * W4, NB = (NB4 + 255) / 256;                    // 16-byte slots of the patch
  constexpr int NAV = BM * (KSTEP / 4), NA = (NAV + 255) / 256;                // 16-byte slots of the weight slice
  constexpr int RUN4 = BM * KHW / 4;                                           // AM 2: 16-byte slots per output channel
  static_assert(KSTEP % 4 == 0 && BN % TW == 0 && (BM * KHW) % 4 == 0 && (KSTEP / 2) % 2 == 0, "bad tile configuration");
  static_assert(2 * STAGE >= 2 * WN * BM + 64, "statistics exchange does not fit the staging buffers");
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

  const int t = threadIdx.x, lane = t & 63, khalf = lane >> 5, l31 = lane & 31;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wid / WN, wn = wid - wm * WN;
  const int vid = xcd_remap(blockIdx.x, p.gx * p.gy * p.gz);
  // (strided data gradient with cls_fast: the output-parity class is the FASTEST index -- the 2 x 2 / 1 x 2 classes of a pixel tile
  // read the same dy window and write interleaved pixels of the same dx lines; next to each other in one XCD's id range they
  // share both in its L2.  Class-slowest gave every class to its own pair of XCDs: dy fetched once per class, 2.4x the
  // algorithmic bytes at the fabric, L2 hit 0.54 -- profiles/r05_sq_backward.json)
  const int cls_f = (p.ncls > 1 && p.cls_fast) ? vid % p.ncls : 0;
  const int vid2 = (p.ncls > 1 && p.cls_fast) ? vid / p.ncls : vid;
  const int by_ = vid2 % p.gy, bx_ = (vid2 / p.gy) % p.gx, bz_ = vid2 / (p.gy * p.gx);
  const int zz = p.groups * p.splitk;
  const int cls = (p.ncls > 1 && !p.cls_fast) ? bz_ / zz : cls_f, bzc = (p.ncls > 1 && !p.cls_fast) ? bz_ - cls * zz : bz_;
  const int g = bzc / p.splitk, ks = bzc - g * p.splitk, m0 = by_ * BM;
  const int PHc = p.ncls > 1 ? p.cls_PH[cls] : p.PH, PWc = p.ncls > 1 ? p.cls_PW[cls] : p.PW;
  const int OUTHc = p.ncls > 1 ? p.cls_OUTH[cls] : p.OUTH, OUTWc = p.ncls > 1 ? p.cls_OUTW[cls] : p.OUTW;
  const int o_ryc = p.ncls > 1 ? p.cls_ry[cls] : p.o_ry, o_rxc = p.ncls > 1 ? p.cls_rx[cls] : p.o_rx;
  const int tiles_per_img = p.tiles_y * p.tiles_x;
  const int img = bx_ / tiles_per_img;
  const int trem = bx_ - img * tiles_per_img;
  const int tyi = trem / p.tiles_x, txi = trem - tyi * p.tiles_x;
  const int oy0 = tyi * TH, ox0 = txi * TW;
  const int PWA = (PWc + 3) & ~3, shift = PWA - PWc;        // (PW may be negative for a parity class: then PWA <= 0, shift in 0..3)
  const int iy0 = oy0 * SV - PHc, ix0 = ox0 * S - PWA;
  const int cbase = p.bcast ? 0 : g * p.Kc;
  const int Kg = p.Kg;

  // ---- chunk-invariant staging offsets (bytes; BUF_OOB = reads as zero)
  unsigned goff[NB];
  int loff[NB], gci[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int e = t + i * 256;
    const int ci = e / (PR * W4), rem = e - ci * (PR * W4), r = rem / W4, f = rem - r * W4;
    const int iy = iy0 + r, ix = ix0 + 4 * f;
    const bool ok = (e < NB4) & ((unsigned)iy < (unsigned)p.SRCH) & ((unsigned)ix < (unsigned)p.SRCW);
    goff[i] = ok ? 4u * (unsigned)(ci * p.s_chan + iy * p.s_row + ix) : BUF_OOB;
    gci[i] = ci;
    loff[i] = e < NB4 ? ci * CP + r * RP + 4 * f : CK * CP;   // dummy slot (pad words)
  }
  unsigned aoff[NA];
  int lsto[AM == 2 ? NA : 1][4];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int idx = t + i * 256;
    if (AM == 2) {
      const int co = idx / RUN4, q4 = idx - co * RUN4;
      aoff[i] = idx < NAV ? 4u * (unsigned)((co * p.Mg + m0) * KHW + 4 * q4) : BUF_OOB;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = 4 * q4 + j, cil = e / KHW, tap = e - cil * KHW;
        lsto[i][j] = idx < NAV ? cil * AP + co * KHW + (KHW - 1 - tap) : ABUF + CK * CP + 4 + j;   // pad words
      }
    } else {
      const int row = idx / (KSTEP / 4), kq = idx - row * (KSTEP / 4);
      aoff[i] = (idx < NAV && m0 + row < p.Mg) ? 4u * (unsigned)((m0 + row) * Kg + kq * 4) : BUF_OOB;
    }
  }
  const __amdgpu_buffer_rsrc_t rsA = buf_rsrc(p.A), rsS = buf_rsrc(p.src);
  const unsigned a_group = (unsigned)cls * p.cls_a_stride + (unsigned)g * p.Mg * Kg;
  const int img_base = img * p.s_img;

  // one register set: chunk c + 2 is requested right after chunk c + 1 moved registers -> LDS and has the whole of chunk c's MFMAs
  // (72 per wave) to come back; a second set costs the third workgroup per CU
  float4 raa[1][NA], rbb[1][NB];
  auto load_chunk = [&](int ci0, const int set) {
    float4 (&ra)[NA] = raa[0];
    float4 (&rb)[NB] = rbb[0];
    (void)set;
    const int k0 = ci0 * KHW;
    const unsigned sa = __builtin_amdgcn_readfirstlane(AM == 2 ? 4u * (a_group + (unsigned)(ci0 * p.Mg * KHW)) : 4u * (a_group + (unsigned)k0));
    const bool full_k = k0 + KSTEP <= Kg;               // uniform: only the last chunk can be partial
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      if (AM == 2) {
        const int co = (t + i * 256) / RUN4;
        ra[i] = buf_load4(rsA, (full_k | (ci0 + co < p.Kc)) ? aoff[i] : BUF_OOB, sa);
      } else {
        const int kq = (t + i * 256) % (KSTEP / 4);
        ra[i] = buf_load4(rsA, (full_k | (k0 + kq * 4 < Kg)) ? aoff[i] : BUF_OOB, sa);
      }
    }
    const unsigned cb = __builtin_amdgcn_readfirstlane(4u * (unsigned)(img_base + (cbase + ci0) * p.s_chan));
    const bool full_c = ci0 + CK <= p.Kc;
#pragma unroll
    for (int i = 0; i < NB; ++i) rb[i] = buf_load4(rsS, (full_c | (ci0 + gci[i] < p.Kc)) ? goff[i] : BUF_OOB, cb);
  };
  auto store_chunk = [&](int buf, const int set) {
    const float4 (&ra)[NA] = raa[0];
    const float4 (&rb)[NB] = rbb[0];
    (void)set;
    float* As = smem + buf * STAGE;
    float* Ps = As + ABUF;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      if (AM == 2) {
        As[lsto[i][0]] = ra[i].x; As[lsto[i][1]] = ra[i].y; As[lsto[i][2]] = ra[i].z; As[lsto[i][3]] = ra[i].w;
      } else {
        const int idx = t + i * 256;
        const int row = idx / (KSTEP / 4), kq = idx - row * (KSTEP / 4);
        *reinterpret_cast<float4*>(As + (idx < NAV ? row * AP + 4 * kq : ABUF + CK * CP + 8)) = ra[i];   // (out of range: the pad words)
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<float4*>(Ps + loff[i]) = rb[i];
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // ---- per-lane operand bases: the two k values of an MFMA sit in lanes 0-31 / 32-63 = the two channel halves of the chunk
  const int a_base = (64 * wm + l31) * AP + khalf * KH2;    // + 32 mi AP + kk
  int b_base[2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int nloc = 64 * wn + 32 * ni + l31;
    const int ty = nloc / TW, tx = nloc - ty * TW;
    b_base[ni] = ty * SV * RP + tx * S + shift + khalf * (CK / 2) * CP;
  }

  const int chunk_beg = ks * p.chunks_per_split;
  const int nchunks = min((p.Kc + CK - 1) / CK - chunk_beg, p.chunks_per_split);
  auto compute_chunk = [&](int cur) {
    const float* As = smem + cur * STAGE;
    const float* Ps = As + ABUF;
    // groups of 4 k values per half (one ds_read_b128 per 32-row block; the 3 x 3 chunk ends with a group of 2): the next group's
    // operands are read while the current group's MFMAs issue
    constexpr int NG = (KH2 + 3) / 4;
    float av[2][2][4], bv[2][4][2];
    auto read_group = [&](int gi, int slot) {
      const int kk0 = 4 * gi;
      constexpr int dummy = 0; (void)dummy;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        if (kk0 + 4 <= KH2) {
          const float4 v = *reinterpret_cast<const float4*>(As + a_base + 32 * mi * AP + kk0);
          av[slot][mi][0] = v.x; av[slot][mi][1] = v.y; av[slot][mi][2] = v.z; av[slot][mi][3] = v.w;
        } else {
          const float2 v = *reinterpret_cast<const float2*>(As + a_base + 32 * mi * AP + kk0);
          av[slot][mi][0] = v.x; av[slot][mi][1] = v.y;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kk = kk0 + r;
        if (kk < KH2) {
          const int ci = kk / KHW, rr = kk - ci * KHW, kh = rr / KW, kw = rr - kh * KW;
          const int offb = ci * CP + kh * RP + kw;
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) bv[slot][r][ni] = Ps[b_base[ni] + offb];
        }
      }
    };
    read_group(0, 0);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      if (gi + 1 < NG) read_group(gi + 1, (gi + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);                  // keep the next group's LDS reads ahead of these MFMAs
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * gi + r < KH2) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[gi & 1][mi][r], bv[gi & 1][r][ni], acc[mi][ni], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // chunk i computes out of buffer i & 1 while chunk i + 1 moves registers -> the other buffer and chunk i + 3 is requested
  if (nchunks > 0) {
    load_chunk(chunk_beg * CK, 0);
    store_chunk(0, 0);
    if (nchunks > 1) load_chunk((chunk_beg + 1) * CK, 0);
  }
  __syncthreads();
  for (int ch = 0; ch < nchunks; ch += 2) {
    if (ch + 1 < nchunks) store_chunk(1, 0);
    if (ch + 2 < nchunks) load_chunk((chunk_beg + ch + 2) * CK, 0);
    compute_chunk(0);
    __syncthreads();
    if (ch + 1 >= nchunks) break;
    if (ch + 2 < nchunks) store_chunk(0, 0);
    if (ch + 3 < nchunks) load_chunk((chunk_beg + ch + 3) * CK, 0);
    compute_chunk(1);
    __syncthreads();
  }

  // ---------------- epilogue ----------------
  const int ctot = p.groups * p.Mg;
  const int ep = p.ep;
  int ooff[2];
  bool cval[2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int nloc = 64 * wn + 32 * ni + l31;
    const int oy = oy0 + nloc / TW, ox = ox0 + nloc % TW;
    cval[ni] = (oy < OUTHc) & (ox < OUTWc);
    ooff[ni] = img * p.o_img + (oy * p.o_sh + o_ryc) * p.o_row + ox * p.o_sw + o_rxc;   // + channel * o_chan
  }
  if (p.part) {                       // raw partial tile in the output layout; a split-K epilogue kernel finishes
    float* part = p.part + (size_t)ks * p.part_stride;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + 64 * wm + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * khalf;
          if (m < p.Mg && cval[ni]) part[(size_t)ooff[ni] + (size_t)(g * p.Mg + m) * p.o_chan] = acc[mi][ni][r];
        }
    return;
  }
  // (register diet: this epilogue used to hold 112 registers of per-row temporaries next to the 64 accumulators and cost the kernel
  // its third workgroup per CU.  Now: scale / folded shift of 16 rows at a time, and every row's statistics live in ONE lane --
  // lane 16 mi + r of each half-wave keeps (sum, M2) of row (mi, r) after the half-wave reductions.)
  const bool stats = ep == EP_RAW_STATS;
  int cnt_w = 0;
  if (stats) {
    const int ty_lo = (64 * wn) / TW, ty_hi = (64 * wn + 63) / TW;               // this wave's rows of the tile (TW <= 64)
    const int cols = min(TW, OUTWc - ox0);
    cnt_w = max(0, min(ty_hi + 1, OUTHc - oy0) - ty_lo) * max(0, cols);
  }
  const float inv_cnt = cnt_w > 0 ? 1.0f / (float)cnt_w : 0.f;
  float keep_s = 0.f, keep_m2 = 0.f;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    // the 16 rows' parameters of this 32-row block in one round trip (clamped addresses, no branches between the loads):
    // v = acc * sc + sh with the bias folded into sh
    float sc_r[16], sh_r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + 64 * wm + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      const int chn = g * p.Mg + (m < p.Mg ? m : 0);
      const float bsv = p.bias ? p.bias[chn] : 0.f;
      if (ep == EP_BN_EVAL) {
        const float inv = 1.0f / sqrtf(p.bn_v[chn] + p.eps);
        sc_r[r] = p.bn_g[chn] * inv;
        sh_r[r] = fmaf(bsv - p.bn_m[chn], sc_r[r], p.bn_b[chn]);
      } else {
        sc_r[r] = 1.f; sh_r[r] = bsv;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + 64 * wm + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      const bool mval = m < p.Mg;
      const int chn = g * p.Mg + (mval ? m : 0);
      float v[2], srow = 0.f;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        v[ni] = ep == EP_BN_EVAL ? fmaf(acc[mi][ni][r], sc_r[r], sh_r[r]) : acc[mi][ni][r] + sh_r[r];
        if (stats) srow += cval[ni] ? v[ni] : 0.f;
        float o = v[ni];
        if (ep == EP_BN_EVAL || ep == EP_LRELU) o = lrelu(o, p.slope);
        if (mval && cval[ni]) p.out[(size_t)ooff[ni] + (size_t)chn * p.o_chan] = o;
      }
      if (stats) {
        // per-channel (sum, M2 about this wave's mean) over the wave's valid pixels: row sums over the 32 lanes of a half-wave
        const float sw = half_wave_sum(srow);
        const float mean = sw * inv_cnt;
        float q = 0.f;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const float dlt = v[ni] - mean;
          q += cval[ni] ? dlt * dlt : 0.f;
        }
        const float m2w = half_wave_sum(q);
        if (l31 == 16 * mi + r) { keep_s = sw; keep_m2 = m2w; }
      }
    }
  }

  if (stats) {
    // the WN waves that share the rows are combined Chan-style in wave order (fixed order: bitwise reproducible)
    float* red = smem;                 // [wn][BM rows][2]
    __syncthreads();
    {
      const int mi = l31 >> 4, r = l31 & 15;
      const int ml = 64 * wm + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      red[(wn * BM + ml) * 2 + 0] = keep_s;
      red[(wn * BM + ml) * 2 + 1] = keep_m2;
    }
    int* cnts = reinterpret_cast<int*>(red + 2 * WN * BM);
    if (lane == 0 && wm == 0) cnts[wn] = cnt_w;
    __syncthreads();
    if (t < BM && m0 + t < p.Mg) {
      float n_a = 0.f, s_a = 0.f, m2_a = 0.f;
#pragma unroll
      for (int w = 0; w < WN; ++w) {
        const float n_b = (float)cnts[w];
        if (n_b > 0.f) {
          const float s_b = red[(w * BM + t) * 2], m2_b = red[(w * BM + t) * 2 + 1];
          if (n_a == 0.f) { n_a = n_b; s_a = s_b; m2_a = m2_b; }
          else {
            const float dlt = s_b / n_b - s_a / n_a;
            m2_a = m2_a + m2_b + dlt * dlt * (n_a * n_b / (n_a + n_b));
            s_a += s_b; n_a += n_b;
          }
        }
      }
      float* st = p.stats + ((size_t)bx_ * ctot + g * p.Mg + m0 + t) * 2;
      st[0] = s_a;
      st[1] = m2_a;
      if (t == 0 && by_ == 0 && g == 0) p.counts[bx_] = n_a;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// dispatch
bool conv_tile_shape_ok(int KH, int KW, int S) {
  return (KH == 3 && KW == 3 && S == 1) || (KH == 4 && KW == 4 && S == 2) || (KH == 2 && KW == 2 && S == 1) || (KH == 3 && KW == 8 && S == 1);
}

template <int KH, int KW, int S, int AM>
static void launch_tile_tw(const PatchArgs& a, int tw, int wide, dim3 grid, hipStream_t s) {
#define MS_TK(TW, WM, WN) hipLaunchKernelGGL((conv_tile_kernel<KH, KW, S, TW, AM, WM, WN>), grid, dim3(256), 0, s, a)
  if (wide) { if (tw == 32) MS_TK(32, 1, 4); else MS_TK(16, 1, 4); }
  else { if (tw == 32) MS_TK(32, 2, 2); else MS_TK(16, 2, 2); }
#undef MS_TK
}

int launch_tile(const PatchArgs& a, const PatchPlan& pl, int KH, int KW, int S, double flops, double bytes, hipStream_t s) {
  const int wide = pl.tile == 2 ? 1 : 0;
  const int bm = wide ? 64 : 128;
  PatchArgs b = a;
  if (a.ncls > 4) return set_error("tile conv: more than 4 parity classes");
  b.gx = pl.n_tiles; b.gy = cdiv(a.Mg, bm); b.gz = a.groups * a.splitk * std::max(1, a.ncls);
  { static int cf = -1; if (cf < 0) { const char* e = getenv("MS_CLS_FAST"); cf = e ? atoi(e) : 1; } b.cls_fast = cf; }   // (MS_CLS_FAST=0: class-slowest order, A/B runs)
  if ((double)b.gx * b.gy * b.gz > 2.0e9) return set_error("conv grid too large");
  dim3 grid(b.gx * b.gy * b.gz);
  if (a.splitk < 1 || (a.splitk > 1 && !a.part)) return set_error("tile conv: bad split-K setup");
  if (a.src_elems >= (1u << 29) || a.a_elems >= (1u << 29)) return set_error("tile conv: operand of 2 GiB or more");
  const int am = a.a_vec == 2 ? 2 : 1;
  if (am == 2 && a.Mg % bm) return set_error("tile conv: in-place weights need whole channel tiles");
  TimingScope ts(s, flops, bytes, "conv_tile_kernel<%d,%d,%d,%d,%d,%d>|conv_%s_tile k%dx%d s%d Mg%d Kg%d g%d tiles%d tile%dx%d tw%d splitk%d%s",
                 KH, KW, S, pl.tw, am, wide, a.is_dgrad ? "dgrad" : "fwd", KH, KW, S, a.Mg, a.Kg, a.groups, pl.n_tiles, bm, wide ? 256 : 128,
                 pl.tw, a.splitk, (a.ep == EP_RAW_STATS && !a.part) ? " +bnstats" : "");
  if (ts.skip()) return 0;
  if (am == 2) {
    if (KH == 3 && KW == 3) launch_tile_tw<3, 3, 1, 2>(b, pl.tw, wide, grid, s);
    else if (KH == 3 && KW == 8) launch_tile_tw<3, 8, 1, 2>(b, pl.tw, wide, grid, s);
    else return set_error("tile conv: no in-place instance for this shape");
  } else {
    if (KH == 3 && KW == 3) launch_tile_tw<3, 3, 1, 1>(b, pl.tw, wide, grid, s);
    else if (KH == 4 && KW == 4) launch_tile_tw<4, 4, 2, 1>(b, pl.tw, wide, grid, s);
    else if (KH == 2 && KW == 2) launch_tile_tw<2, 2, 1, 1>(b, pl.tw, wide, grid, s);
    else launch_tile_tw<3, 8, 1, 1>(b, pl.tw, wide, grid, s);
  }
  return check_launch("conv_tile_kernel");
}

}  // namespace ms

extern "C" int ms_debug_set_conv_tile(int on) {
  const int old = ms::g_conv_tile;
  if (old != (on ? 1 : 0)) ms_debug_set_patch_min_workgroups(ms::g_patch_min_wgs);      // (bumps the tuning epoch: workspace sizes follow the planner)
  ms::g_conv_tile = on ? 1 : 0;
  return old;
}
