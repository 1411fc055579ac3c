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
#include <cstdint>

#include "kernels.h"

namespace ms {

// channels per K chunk of the 2-D kernels (tuning: -DMS_CK9=8 ...)
#ifndef MS_CK9
#define MS_CK9 4
#endif
#ifndef MS_CK16
#define MS_CK16 4
#endif
#ifndef MS_CK24
#define MS_CK24 2
#endif
template <int KH, int KW>
struct PatchCfg {
  static constexpr int KHW = KH * KW;
  // channels per K-chunk: K_step = CK*KHW in [36, 64], multiple of 4
  static constexpr int CK = KHW == 1 ? 32 : KHW == 2 ? 16 : KHW == 3 ? 16 : KHW == 4 ? 16 : KHW == 9 ? MS_CK9
                            : KHW == 16 ? MS_CK16 : KHW == 24 ? MS_CK24 : 4;
  static constexpr int KSTEP = CK * KHW;
};

constexpr int patch_row_pitch(int pc, int sv, int tw) {
  // lanes of one 32-lane read group cover 32/TW tile rows: rows must land 'TW' banks apart
  if (tw >= 32) return pc;
  int rp = pc;
  while ((rp * sv) % 32 != tw % 32) ++rp;
  return rp;
}

// 1-D kernels (taps <= 3): registers capped at 128 so that 4 workgroups share a CU (the decoder launches 1024 = 4 x 256)
// AM: layout of the A operand.  0: [Mg][Kg] rows, scalar loads (Kg % 4 != 0); 1: the same, 16-B loads;
// 2: data gradient straight from the conv weight w[co][ci][tap] (stride-1 convs): A(ci, (co,tap')) = w[co][ci][KHW-1-tap'],
//    Mg = Cin_g, Kc = Cout_g; per output channel the 64 x KHW block of a channel tile is one contiguous run
// WM: 32-row wave tiles along the channels (tile = 32*WM channels x 64 pixels).  KS: wave groups that split every K chunk
// among themselves (intra-workgroup split-K: 2*WM*KS waves; the groups' accumulators are added through LDS in a fixed
// order at the end) -- for layers with too few tiles to fill the chip: no partial tiles in HBM, no second kernel.
template <int KH, int KW, int S, int TW, bool UP2, int AM, int WM = 2, int KS = 1>
__global__ __launch_bounds__(128 * WM * KS, (KS == 1 && WM == 2 && KH * KW <= 3 && TW >= 32 ? 4 : 1))
void conv_patch_kernel(const PatchArgs p) {
  prefetch_kernargs<sizeof(PatchArgs)>();
  using Cfg = PatchCfg<KH, KW>;
  constexpr int NT = 128 * WM * KS;                     // threads
  constexpr int BM = 32 * WM, BN = 64, TH = BN / TW;
  constexpr int SV = (KH == 1) ? 1 : S;                 // KH == 1: rows are independent batch items
  // intra-split workgroups run alone on their CU: 4x longer chunks keep the MFMA share of a chunk above its fixed cost
  constexpr int CK = Cfg::CK * (KS > 1 ? 4 : 1), KHW = Cfg::KHW, KSTEP = CK * KHW;
  constexpr int PR = (TH - 1) * SV + KH, PC = (TW - 1) * S + KW;
  constexpr int RP = patch_row_pitch(PC, SV, TW), CP = PR * RP;
  constexpr int LDA = BM + 1;   // odd pitch: the staging stores (4 k-rows apart per lane group) stay <= 2-way conflicted
  constexpr int STAGE = KSTEP * LDA + CK * CP + 3 * LDA + 4;   // + pad words: out-of-range staging stores land there
  constexpr int NPE = CK * PR * PC;                     // patch elements per chunk
  constexpr int NP = (NPE + NT - 1) / NT;
  constexpr int NAV = BM * (KSTEP / 4);                 // float4 slots of the weight slice
  constexpr int RUN4 = BM * KHW / 4;                    // AM 2: float4 slots per output channel
  constexpr int NA = (NAV + NT - 1) / NT;
  constexpr int LP = 68;                                // pitch of the epilogue's [channel][pixel] tile (4 mod 32)
  static_assert(KSTEP % 4 == 0 && BN % TW == 0, "bad patch configuration");
  static_assert(2 * STAGE >= BM * LP, "epilogue tile does not fit the staging buffers");
  static_assert(CK % KS == 0 && (KSTEP / KS) % 2 == 0, "the wave groups split a chunk by whole channels");
  static_assert(KS == 1 || 2 * STAGE >= (KS / 2) * 2 * WM * 1024, "accumulator exchange does not fit the staging buffers");
  __shared__ float smem[2 * STAGE];

  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int kg = wid / (2 * WM), wv = wid - kg * (2 * WM);   // wave group (K split), wave inside the group
  const int wm = wv >> 1, wn = wv & 1, khalf = lane >> 5;
  // logical block id: channel tile fastest, then pixel tile, then (group, K slice); one contiguous range per XCD, so the
  // channel tiles that share an input patch sit behind the same L2
  const int vid = xcd_remap(blockIdx.x, p.gx * p.gy * p.gz);
  // (strided data gradient with cls_fast: the output-parity class is the FASTEST index -- the 2 x 2 / 1 x 2 classes of a pixel tile
  // read the same dy window and write interleaved pixels of the same dx lines; next to each other in one XCD's id range they
  // share both in its L2.  Class-slowest gave every class to its own pair of XCDs: dy fetched once per class, 2.4x the
  // algorithmic bytes at the fabric, L2 hit 0.54 -- profiles/r05_sq_backward.json)
  const int cls_f = (p.ncls > 1 && p.cls_fast) ? vid % p.ncls : 0;
  const int vid2 = (p.ncls > 1 && p.cls_fast) ? vid / p.ncls : vid;
  const int by_ = vid2 % p.gy, bx_ = (vid2 / p.gy) % p.gx, bz_ = vid2 / (p.gy * p.gx);
  // (strided data gradient: the output-parity class comes first in z)
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
  const int iy0 = oy0 * SV - PHc, ix0 = ox0 * S - PWc;
  const int cbase = p.bcast ? 0 : g * p.Kc;
  const int Kg = p.Kg;

  // ---- chunk-invariant staging offsets (bytes; BUF_OOB = reads as zero)
  unsigned goff[NP], goff_h[UP2 ? NP : 1];
  int loff[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int e = t + i * NT;
    const int ci = e / (PR * PC), rem = e - ci * (PR * PC), r = rem / PC, c = rem - r * PC;
    const int iy = iy0 + r, ix = ix0 + c;
    const bool ok = (e < NPE) & ((unsigned)iy < (unsigned)p.SRCH) & ((unsigned)ix < (unsigned)p.SRCW);
    const int rowbase = ci * p.s_chan + iy * p.s_row;
    goff[i] = ok ? 4u * (unsigned)(rowbase + ix) : BUF_OOB;
    // x = nearest_up2(a) + r : a has half the row length, hence half of every stride (all strides are even)
    if (UP2) goff_h[i] = ok ? 4u * (unsigned)((rowbase >> 1) + (ix >> 1)) : BUF_OOB;
    loff[i] = e < NPE ? ci * CP + r * RP + c : CK * CP;   // dummy slot
  }
  unsigned aoff[NA];                                    // weight offset rel. to the chunk base
  int lsto[AM == 2 ? NA : 1][4];                        // AM 2: LDS slots of the 4 elements of a float4
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int idx = t + i * NT;
    if (AM == 2) {
      const int co = idx / RUN4, q4 = idx - co * RUN4;
      aoff[i] = idx < NAV ? 4u * (unsigned)((co * p.Mg + m0) * KHW + 4 * q4) : BUF_OOB;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = 4 * q4 + j, cil = e / KHW, tap = e - cil * KHW;
        lsto[i][j] = idx < NAV ? (co * KHW + (KHW - 1 - tap)) * LDA + cil : KSTEP * LDA + CK * CP + j;   // pad words
      }
    } else {
      const int row = idx / (KSTEP / 4), kq = idx - row * (KSTEP / 4);
      aoff[i] = (idx < NAV && m0 + row < p.Mg) ? 4u * (unsigned)((m0 + row) * Kg + kq * 4) : BUF_OOB;
    }
  }
  const __amdgpu_buffer_rsrc_t rsA = buf_rsrc(p.A), rsS = buf_rsrc(p.src), rsS2 = buf_rsrc(UP2 ? p.src2 : p.src);
  const unsigned a_group = (unsigned)cls * p.cls_a_stride + (unsigned)g * p.Mg * Kg;     // (AM 2: Kg = Kc*KHW, so this is the group's first output channel)
  const int img_base = img * p.s_img;

  // two register sets: chunk c+2 is in flight from HBM/L2 while chunk c+1 waits in registers and chunk c computes
  float4 ra0[NA], ra1[NA];
  float rb0[NP], rb1[NP];
  auto load_chunk = [&](int ci0, float4 (&ra)[NA], float (&rb)[NP]) {
    const int k0 = ci0 * KHW;
    // scalar offsets: values derived from integer divisions live in VGPRs; readfirstlane keeps the loads waterfall-free
    const unsigned sa = __builtin_amdgcn_readfirstlane(AM == 2 ? 4u * (a_group + (unsigned)(ci0 * p.Mg * KHW))
                                                               : 4u * (a_group + (unsigned)k0));
    const bool full_k = k0 + KSTEP <= Kg;               // uniform: only the last chunk can be partial
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      if (AM == 2) {
        const int co = (t + i * NT) / RUN4;
        ra[i] = buf_load4(rsA, (full_k | (ci0 + co < p.Kc)) ? aoff[i] : BUF_OOB, sa);
        continue;
      }
      const int kq = (t + i * NT) % (KSTEP / 4);
      const int k = k0 + kq * 4;
      if (AM == 1) {
        ra[i] = buf_load4(rsA, (full_k | (k < Kg)) ? aoff[i] : BUF_OOB, sa);
      } else {
        float4 v;
        v.x = buf_load(rsA, (full_k | (k < Kg)) ? aoff[i] : BUF_OOB, sa);
        v.y = buf_load(rsA, (full_k | (k + 1 < Kg)) ? aoff[i] + 4u : BUF_OOB, sa);
        v.z = buf_load(rsA, (full_k | (k + 2 < Kg)) ? aoff[i] + 8u : BUF_OOB, sa);
        v.w = buf_load(rsA, (full_k | (k + 3 < Kg)) ? aoff[i] + 12u : BUF_OOB, sa);
        ra[i] = v;
      }
    }
    const int cb = __builtin_amdgcn_readfirstlane(img_base + (cbase + ci0) * p.s_chan);
    const bool full_c = ci0 + CK <= p.Kc;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int ci = (t + i * NT) / (PR * PC);
      const bool ok = full_c | (ci0 + ci < p.Kc);
      if (UP2) rb[i] = buf_load(rsS, ok ? goff_h[i] : BUF_OOB, 4u * (unsigned)(cb >> 1)) +
                       buf_load(rsS2, ok ? goff[i] : BUF_OOB, 4u * (unsigned)cb);
      else rb[i] = buf_load(rsS, ok ? goff[i] : BUF_OOB, 4u * (unsigned)cb);
    }
  };
  auto store_chunk = [&](int buf, const float4 (&ra)[NA], const float (&rb)[NP]) {
    float* As = smem + buf * STAGE;
    float* Ps = As + KSTEP * LDA;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      if (AM == 2) {
        As[lsto[i][0]] = ra[i].x;
        As[lsto[i][1]] = ra[i].y;
        As[lsto[i][2]] = ra[i].z;
        As[lsto[i][3]] = ra[i].w;
        continue;
      }
      const int idx = t + i * NT;
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

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  // ---- per-lane operand bases: k and k+1 of an MFMA pair sit in lanes 0-31 / 32-63
  // (wave group kg owns channels [kg*CK/KS, (kg+1)*CK/KS) of every chunk: a constant shift of both operand bases)
  const int a_base = khalf * LDA + wm * 32 + (lane & 31) + kg * (KSTEP / KS) * LDA;
  int b_same, b_row, b_chan;
  {
    const int nloc = wn * 32 + (lane & 31);
    const int ty = nloc / TW, tx = nloc - ty * TW;
    const int base = ty * SV * RP + tx * S + kg * (CK / KS) * CP;
    b_same = base + khalf;                                         // k+1 = next tap in the same row
    b_row = base + khalf * (RP - (KW - 1));                        // k+1 = first tap of the next kernel row
    b_chan = base + khalf * (CP - (KH - 1) * RP - (KW - 1));       // k+1 = first tap of the next channel
  }

  const int chunk_beg = ks * p.chunks_per_split;
  const int nchunks = min((p.Kc + CK - 1) / CK - chunk_beg, p.chunks_per_split);
  auto compute_chunk = [&](int cur) {
    const float* As = smem + cur * STAGE;
    const float* Ps = As + KSTEP * LDA;
    // operands of the next group of k-pairs are read from LDS while the current group's MFMAs issue
    constexpr int NPAIR = KSTEP / 2 / KS, GP = 4, NG = (NPAIR + GP - 1) / GP;   // k-pairs of this wave group
    float av[2][GP], bv[2][GP];
    auto read_group = [&](int gi, int slot) {
#pragma unroll
      for (int q = 0; q < GP; ++q) {
        const int jj = gi * GP + q;
        if (jj < NPAIR) {
          const int k0 = 2 * jj;
          const int ci = k0 / KHW, rr = k0 - ci * KHW, kh = rr / KW, kw = rr - kh * KW;
          const int offb = ci * CP + kh * RP + kw;
          av[slot][q] = As[a_base + k0 * LDA];
          const int base = (kw + 1 < KW) ? b_same : (kh + 1 < KH) ? b_row : b_chan;
          bv[slot][q] = Ps[base + offb];
        }
      }
    };
    read_group(0, 0);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      if (gi + 1 < NG) read_group(gi + 1, (gi + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);                  // keep the next group's LDS reads ahead of these MFMAs
#pragma unroll
      for (int q = 0; q < GP; ++q)
        if (gi * GP + q < NPAIR) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[gi & 1][q], bv[gi & 1][q], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
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

  // ---------------- wave groups: add the accumulators (tree, fixed order), group 0 keeps the result ----------------
  if (KS > 1) {
    float* xch = smem;                                  // [slot][wave][r][lane]
#pragma unroll
    for (int step = KS / 2; step >= 1; step >>= 1) {
      if (kg >= step && kg < 2 * step) {
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[(((kg - step) * 2 * WM + wv) * 16 + r) * 64 + lane] = acc[r];
      }
      __syncthreads();
      if (kg < step) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += xch[((kg * 2 * WM + wv) * 16 + r) * 64 + lane];
      }
      __syncthreads();
    }
  }
  const bool lead = kg == 0;                            // the other groups only keep the barriers company from here on

  // ---------------- epilogue ----------------
  const int ctot = p.groups * p.Mg;
  const int nloc = wn * 32 + (lane & 31);
  const int oy = oy0 + nloc / TW, ox = ox0 + nloc % TW;
  const bool cval = lead & (oy < OUTHc) & (ox < OUTWc);
  const int ooff = img * p.o_img + (oy * p.o_sh + o_ryc) * p.o_row + ox * p.o_sw + o_rxc;   // + channel * o_chan
  const int ep = p.ep;
  if (p.part) {                       // raw partial tile in the output layout; a split-K epilogue kernel finishes
    float* part = p.part + (size_t)ks * p.part_stride;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      if (m < p.Mg && cval) part[(size_t)ooff + (size_t)(g * p.Mg + m) * p.o_chan] = acc[r];
    }
    return;
  }
  // the 16 bias values of this lane's rows in ONE round trip (clamped addresses, no branches between the loads): fetched one by
  // one next to their use they were 16 dependent load -> wait -> store rounds at the end of every forward launch
  float bias_r[16];
  if (p.bias) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      bias_r[r] = p.bias[g * p.Mg + (m < p.Mg ? m : 0)];
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) bias_r[r] = 0.f;
  }
  // eval-mode BatchNorm: the rows' (gamma, beta, mean, var) eight rows per round trip -- fetched next to their use they were 16
  // dependent rounds of four loads at the end of every eval-mode launch (the D-step's generator pass: +13..16 us per 2-D layer)
#pragma unroll
  for (int rh = 0; rh < 16; rh += 8) {
  float sc_r[8], sh_r[8];
  if (ep == EP_BN_EVAL) {
    float gq[8], bq[8], mq[8], vq[8];
#pragma unroll
    for (int r8 = 0; r8 < 8; ++r8) {
      const int r = rh + r8;
      const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      const int chn = g * p.Mg + (m < p.Mg ? m : 0);
      gq[r8] = p.bn_g[chn]; bq[r8] = p.bn_b[chn]; mq[r8] = p.bn_m[chn]; vq[r8] = p.bn_v[chn];
    }
#pragma unroll
    for (int r8 = 0; r8 < 8; ++r8) {
      const float inv = 1.0f / sqrtf(vq[r8] + p.eps);
      sc_r[r8] = gq[r8] * inv;
      sh_r[r8] = bq[r8] - mq[r8] * sc_r[r8];
    }
  } else {
#pragma unroll
    for (int r8 = 0; r8 < 8; ++r8) { sc_r[r8] = 1.f; sh_r[r8] = 0.f; }
  }
#pragma unroll
  for (int r = rh; r < rh + 8; ++r) {
    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
    const bool mval = m < p.Mg;
    const int chn = g * p.Mg + (mval ? m : 0);
    const float bsv = bias_r[r];
    const float sc = sc_r[r - rh], sh = sh_r[r - rh];
    float v = acc[r] + bsv;
    if (ep == EP_RAW_STATS) acc[r] = v;
    if (ep == EP_BN_EVAL) v = lrelu(fmaf(v, sc, sh), p.slope);
    if (ep == EP_LRELU) v = lrelu(v, p.slope);
    if (ep == EP_DGRAD_UP2) {
      // 1-D stride-1 data gradient of an upsample-add input: out2 = grad of the residual (full resolution),
      // out = grad of the half-resolution tensor = sum over the pair of columns (adjacent lanes)
      const float pr = lane_xor1(v);
      if (mval && cval) {
        p.out2[(size_t)ooff + (size_t)chn * p.o_chan] = v;
        if (!(lane & 1)) p.out[(size_t)(ooff >> 1) + (size_t)chn * (p.o_chan >> 1)] = v + pr;
      }
    } else if (mval && cval) {
      p.out[(size_t)ooff + (size_t)chn * p.o_chan] = v;
    }
  }
  }

  if (ep == EP_RAW_STATS) {
    // per-channel (sum, M2 about this tile's mean) over the tile's valid pixels, fixed order: the tile goes through LDS
    // as [channel][pixel]; 4 threads per channel sum 16 pixels each from registers (both passes), then combine
    float* tile = smem;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ml = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      if (lead) tile[ml * LP + nloc] = cval ? acc[r] : 0.f;
    }
    __syncthreads();
    if (!lead) return;                                  // t < 4*BM from here: 4 threads per channel
    const int ch = t >> 2, q = t & 3;
    float v[16];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      v[i] = tile[ch * LP + q + 4 * i];
      s += v[i];
    }
    s += lane_xor1(s);
    s += lane_xor2(s);
    const int cnt = min(TH, OUTHc - oy0) * min(TW, OUTWc - ox0);
    const float mean = s / (float)cnt;
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int nl = q + 4 * i;
      const bool ok = (oy0 + nl / TW < OUTHc) & (ox0 + nl % TW < OUTWc);
      const float dlt = v[i] - mean;
      m2 += ok ? dlt * dlt : 0.f;
    }
    m2 += lane_xor1(m2);
    m2 += lane_xor2(m2);
    if (q == 0 && m0 + ch < p.Mg) {
      float* st = p.stats + ((size_t)bx_ * ctot + g * p.Mg + m0 + ch) * 2;
      st[0] = s;
      st[1] = m2;
    }
    if (t == 0 && by_ == 0 && g == 0) p.counts[bx_] = (float)cnt;
  }
}

// ---------------------------------------------------------------------------------------------
// dispatch
int g_patch_min_wgs = 32;      // below this many workgroups the split-K im2col path is used instead (swept 8..192: flat from 8
                               // to 128, best at 32; 192 loses 11 %)
int g_patch_force_splitk = 0;   // tuning knob: > 0 forces this split-K factor in the patch kernel
int g_precision = 0;
int g_patch_intra = 1;          // tuning knob: intra-workgroup K split for small 1-D k3 layers
int patch_chunk_channels(int KH, int KW) {
  const int khw = KH * KW;
  return khw == 1 ? 32 : khw == 2 ? 16 : khw == 3 ? 16 : khw == 4 ? 16 : khw == 9 ? MS_CK9 : khw == 16 ? MS_CK16 : khw == 24 ? MS_CK24 : 4;
}

static PatchPlan plan_patch_impl(int nd, int Mg, int groups, int Kc, int KH, int KW, int SH, int SW, int B, int OH, int OW, int zmul, int in_w);
int g_tile_min_wgs = 256;     // layers with fewer 128 x 128 (64 x 256) tiles than this stay on the 64 x 64 kernel: measured, round 5 -- the
                              // mid layers (128 tiles) need split-K partials + an epilogue launch there, which costs what the tile kernel wins
PatchPlan plan_patch(int nd, int Mg, int groups, int Kc, int KH, int KW, int SH, int SW, int B, int OH, int OW, int zmul, int in_w) {
  PatchPlan pl = plan_patch_impl(nd, Mg, groups, Kc, KH, KW, SH, SW, B, OH, OW, zmul, in_w);
  if (pl.tile && (long)pl.n_tiles * cdiv(Mg, 64 * pl.tm) * groups * zmul < g_tile_min_wgs)
    pl = plan_patch_impl(nd, Mg, groups, Kc, KH, KW, SH, SW, B, OH, OW, zmul, 0);
  return pl;
}
static PatchPlan plan_patch_impl(int nd, int Mg, int groups, int Kc, int KH, int KW, int SH, int SW, int B, int OH, int OW, int zmul, int in_w) {
  PatchPlan pl = {0, 1, 64, 0, 0, 0, 1, 1 << 30, 1, 2, 1, 0, 0};
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
  const bool p6 = g_precision == 1 && patch6_supported(KH, KW, S);   // bf16x6 kernels: 64 x 128 tiles
  // the lean kernel (conv_tile.hip: 128 x 128 or, for at most 64 output rows, 64 x 256 tiles): 2-D layers whose input rows are
  // 16-byte aligned and whose weight rows are whole 16-byte runs
  const int tile = (g_conv_tile && !p6 && nd == 2 && conv_tile_shape_ok(KH, KW, S) && in_w > 0 && (in_w & 3) == 0 && ((Kc * KH * KW) & 3) == 0 &&
                    (Mg <= 64 || Mg % 128 == 0 || KH * KW == 4 || KH * KW == 16)) ? (Mg <= 64 ? 2 : 1) : 0;
  const int tm = tile == 1 ? 2 : 1, tn = tile == 1 ? 2 : tile == 2 ? 4 : p6 ? 2 : 1;
  const int th = 64 * tn / tw;
  pl.ok = 1; pl.tm = tm; pl.tw = tw; pl.tn = tn; pl.p6 = p6 ? 1 : 0; pl.tile = tile;
  pl.tiles_y = cdiv(rows, th); pl.tiles_x = cdiv(OW, tw);
  pl.n_tiles = imgs * pl.tiles_y * pl.tiles_x;
  const long base = (long)pl.n_tiles * cdiv(Mg, 64 * tm) * groups * zmul;   // zmul: parity classes sharing the launch
  // too few workgroups: the split-K im2col path spreads the weight stream better
  if (base * tn < g_patch_min_wgs) pl.ok = 0;   // (bf16x6: 128-pixel tiles, half as many workgroups for the same layer)
  // fewer workgroups than 1.5 per CU and a long reduction: slice the channel chunks over workgroups
  const int nchunks = cdiv(Kc, tile ? (KH * KW == 4 ? 8 : KH * KW == 9 ? 4 : 2) : p6 ? (KH * KW <= 4 ? 16 : 8) : patch_chunk_channels(KH, KW));   // (bf16x6: 16 / 8 channels; lean kernel: TileCfg)
  if (pl.ok && g_patch_force_splitk > 0 && nchunks >= g_patch_force_splitk) {
    pl.chunks_per_split = cdiv(nchunks, g_patch_force_splitk);
    pl.splitk = cdiv(nchunks, pl.chunks_per_split);
  } else if (pl.ok && base < (p6 || tile ? 256 : 384) && nchunks >= 4) {   // (bf16x6 / lean kernel: larger tiles, fewer workgroups per CU)
    const long base32 = (long)pl.n_tiles * cdiv(Mg, 32) * groups * zmul;
    // (short reductions only: with a long K -- the 2048-channel data gradient of the first decoder layer -- slices over
    // workgroups keep more of the chip busy: 91 vs 152 us)
    if (!p6 && g_patch_intra && KH == 1 && KW == 3 && S == 1 && tw >= 32 && base32 >= 192 && nchunks <= 16) {
      // 1-D k3 layers with few tiles: 32-channel tiles, every K chunk split over 4 wave groups INSIDE the workgroup
      // (no partial tiles in HBM, no reduction kernel)
      pl.wm = 1; pl.ksi = 4;
    } else {
      // one workgroup per CU (256) measured better than 1.5 for the 128-tile mid layers (both kernels of the pair: conv
      // 19.3 vs 19.8 us, epilogue 11.3 vs 13.3 us); from 256 tiles up, two slices (512 workgroups) stay
      const long target = base <= 128 ? 256 : 384;
      int sk = (int)std::min<long>(nchunks / 2, (target + base - 1) / base);
      if (sk > 4) sk = 4;
      if (sk > 1) {
        pl.chunks_per_split = cdiv(nchunks, sk);
        pl.splitk = cdiv(nchunks, pl.chunks_per_split);
      }
    }
  }
  return pl;
}

template <int KH, int KW, int S, bool UP2, int AM>
static void launch_patch_tw(const PatchArgs& a, int tw, dim3 grid, hipStream_t s) {
#define MS_PK(TW) hipLaunchKernelGGL((conv_patch_kernel<KH, KW, S, TW, UP2, AM>), grid, dim3(256), 0, s, a)
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

template <int AM>
static void launch_patch_k(const PatchArgs& a, int kh, int kw, int s_, int tw, bool up2, dim3 grid, hipStream_t s) {
  if constexpr (AM == 2) {   // data gradient of stride-1 convs only
    if (kh == 1 && kw == 3) launch_patch_tw<1, 3, 1, false, 2>(a, tw, grid, s);
    else if (kh == 1 && kw == 1) launch_patch_tw<1, 1, 1, false, 2>(a, tw, grid, s);
    else if (kh == 3 && kw == 3) launch_patch_tw<3, 3, 1, false, 2>(a, tw, grid, s);
    else launch_patch_tw<3, 8, 1, false, 2>(a, tw, grid, s);
  } else {
    if (kh == 1 && kw == 3 && s_ == 1) {
      if (up2) launch_patch_tw<1, 3, 1, true, AM>(a, tw, grid, s);
      else launch_patch_tw<1, 3, 1, false, AM>(a, tw, grid, s);
    } else if (kh == 1 && kw == 4 && s_ == 2) launch_patch_tw<1, 4, 2, false, AM>(a, tw, grid, s);
    else if (kh == 1 && kw == 4 && s_ == 1) launch_patch_tw<1, 4, 1, false, AM>(a, tw, grid, s);
    else if (kh == 1 && kw == 1 && s_ == 1) launch_patch_tw<1, 1, 1, false, AM>(a, tw, grid, s);
    else if (kh == 1 && kw == 2 && s_ == 1) launch_patch_tw<1, 2, 1, false, AM>(a, tw, grid, s);
    else if (kh == 2 && kw == 2 && s_ == 1) launch_patch_tw<2, 2, 1, false, AM>(a, tw, grid, s);
    else if (kh == 3 && kw == 3 && s_ == 1) launch_patch_tw<3, 3, 1, false, AM>(a, tw, grid, s);
    else if (kh == 4 && kw == 4 && s_ == 2) launch_patch_tw<4, 4, 2, false, AM>(a, tw, grid, s);
    else launch_patch_tw<3, 8, 1, false, AM>(a, tw, grid, s);
  }
}

template <int AM>
static void launch_patch_intra(const PatchArgs& a, int tw, bool up2, dim3 grid, hipStream_t s) {
#define MS_PI(TW, UP2) hipLaunchKernelGGL((conv_patch_kernel<1, 3, 1, TW, UP2, AM, 1, 4>), grid, dim3(512), 0, s, a)
  if constexpr (AM == 2) {
    if (tw == 64) MS_PI(64, false); else MS_PI(32, false);
  } else {
    if (up2) { if (tw == 64) MS_PI(64, true); else MS_PI(32, true); }
    else { if (tw == 64) MS_PI(64, false); else MS_PI(32, false); }
  }
#undef MS_PI
}

// the data gradient can read the conv weight in place (AM 2) when whole 64-channel tiles of contiguous runs exist
bool patch_dgrad_direct_ok(const float* w, int Cin_g, int KH, int KW, int SH, int SW, bool up2_or_bcast) {
  const bool shape = (KH == 1 && KW == 3) || (KH == 1 && KW == 1) || (KH == 3 && KW == 3) || (KH == 3 && KW == 8);
  return shape && SH == 1 && SW == 1 && !up2_or_bcast && Cin_g % 64 == 0 && ((uintptr_t)w & 15) == 0;
}

int launch_patch(const PatchArgs& a, const PatchPlan& pl, int KH, int KW, int S, bool up2, double flops, double bytes,
                 hipStream_t s) {
  if (pl.tile) {
    if (up2 || (a.a_vec != 2 && !(a.a_vec && ((uintptr_t)a.A & 15) == 0)) || ((uintptr_t)a.src & 15))
      return set_error("conv: the lean kernel needs 16-byte-aligned operands (weights %p, input %p)", (const void*)a.A, (const void*)a.src);
    return launch_tile(a, pl, KH, KW, S, flops, bytes, s);
  }
  const int bm = 32 * pl.wm;
  PatchArgs b = a;
  if (a.ncls > 4) return set_error("patch conv: more than 4 parity classes");
  b.gx = pl.n_tiles; b.gy = cdiv(a.Mg, bm); b.gz = a.groups * a.splitk * std::max(1, a.ncls);
  { static int cf = -1; if (cf < 0) { const char* e = getenv("MS_CLS_FAST"); cf = e ? atoi(e) : 1; } b.cls_fast = cf; }   // (MS_CLS_FAST=0: class-slowest order, A/B runs)
  if ((double)b.gx * b.gy * b.gz > 2.0e9) return set_error("conv grid too large");
  dim3 grid(b.gx * b.gy * b.gz);
  if (a.splitk < 1 || (a.splitk > 1 && !a.part)) return set_error("patch conv: bad split-K setup");
  if (a.src_elems >= (1u << 29) || a.a_elems >= (1u << 29)) return set_error("patch conv: operand of 2 GiB or more");
  const int am = a.a_vec == 2 ? 2 : (a.a_vec && ((uintptr_t)a.A & 15) == 0) ? 1 : 0;
  if (am == 2 && a.Mg % bm) return set_error("patch conv: in-place weights need whole channel tiles");
  if (am == 2 && (S != 1 || up2)) return set_error("patch conv: in-place weights need a stride-1 data gradient");
  TimingScope ts(s, flops, bytes, "conv_patch_kernel<%d,%d,%d,%d,%d,%d,%d,%d>|conv_%s_patch k%dx%d s%d Mg%d Kg%d g%d tiles%d tile%d tw%d splitk%d%s%s",
                 KH, KW, S, pl.tw, up2 ? 1 : 0, am, pl.wm, pl.ksi, a.is_dgrad ? "dgrad" : "fwd", KH, KW, S, a.Mg, a.Kg,
                 a.groups, pl.n_tiles, bm, pl.tw, a.splitk, pl.ksi > 1 ? " intra4" : "",
                 (a.ep == EP_RAW_STATS && !a.part) ? " +bnstats" : "");
  if (ts.skip()) return 0;
  if (pl.ksi > 1) {
    if (pl.wm != 1 || pl.ksi != 4 || KH != 1 || KW != 3 || S != 1 || pl.tw < 32) return set_error("patch conv: no such intra-split kernel");
    if (am == 2) launch_patch_intra<2>(b, pl.tw, up2, grid, s);
    else if (am == 1) launch_patch_intra<1>(b, pl.tw, up2, grid, s);
    else launch_patch_intra<0>(b, pl.tw, up2, grid, s);
  } else if (am == 2) launch_patch_k<2>(b, KH, KW, S, pl.tw, up2, grid, s);
  else if (am == 1) launch_patch_k<1>(b, KH, KW, S, pl.tw, up2, grid, s);
  else launch_patch_k<0>(b, KH, KW, S, pl.tw, up2, grid, s);
  return check_launch("conv_patch_kernel");
}

}  // namespace ms

static int g_tuning_epoch = 0;
extern "C" int ms_tuning_epoch(void) { return g_tuning_epoch; }
extern "C" int ms_set_precision(int mode) {
  if (mode != 0 && mode != 1) return ms::set_error("ms_set_precision: mode %d", mode);
  const int old = ms::g_precision;
  if (old != mode) ++g_tuning_epoch;
  ms::g_precision = mode;
  return old;
}
extern "C" int ms_get_precision(void) { return ms::g_precision; }

extern "C" int ms_debug_set_patch_tuning(int intra_split, int force_splitk) {
  ++g_tuning_epoch;
  ms::g_patch_intra = intra_split < 0 ? 0 : 1;
  ms::g_patch_force_splitk = force_splitk > 0 ? force_splitk : 0;
  return 0;
}

namespace ms { extern int g_conv16_force_wm, g_conv16_force_wn, g_conv16_dma, g_conv16_wide8, g_conv16_ring, g_conv16_dbg, g_conv16_big_stages; }
extern "C" int ms_debug_set_conv16_ring(int nstg, int wide8) {
  ++g_tuning_epoch;
  ms::g_conv16_ring = nstg; ms::g_conv16_wide8 = wide8 & 1; ms::g_conv16_dbg = ((wide8 >> 4) & 15) | (((wide8 >> 12) & 15) << 4); ms::g_conv16_big_stages = (wide8 & 256) ? 0 : (wide8 & 512) ? 1 : 2;
  return 0;
}
extern "C" int ms_debug_set_conv16_tile(int wm, int wn) {
  ++g_tuning_epoch;
  ms::g_conv16_dma = wm >= 0 ? 1 : 0;          // negative wm: register-staged path (A/B against the LDS-DMA ring)
  if (wm < 0) { wm = 0; wn = 0; }
  ms::g_conv16_force_wm = wm; ms::g_conv16_force_wn = wn;
  return 0;
}

namespace ms { extern int g_wgrad16_target_wgs, g_wgrad_patch_target_wgs; }
extern "C" int ms_debug_set_wgrad_target(int workgroups) {
  const int old = ms::g_wgrad_patch_target_wgs;
  ++g_tuning_epoch;
  ms::g_wgrad_patch_target_wgs = workgroups > 0 ? workgroups : 768;
  return old;
}
namespace ms { extern int g_wgrad16_ring; }
extern "C" int ms_debug_set_wgrad16_ring(int buffers) {
  const int old = ms::g_wgrad16_ring;
  ++g_tuning_epoch;
  ms::g_wgrad16_ring = buffers >= 2 ? (buffers > 8 ? 8 : buffers) : 2;
  return old;
}
extern "C" int ms_debug_set_wgrad16_target(int workgroups) {
  const int old = ms::g_wgrad16_target_wgs;
  ++g_tuning_epoch;
  ms::g_wgrad16_target_wgs = workgroups > 0 ? workgroups : 128;
  return old;
}

namespace ms { extern int g_clip32; }
extern "C" int ms_debug_set_clip32(int on) {
  const int prev = ms::g_clip32;
  if (prev != (on ? 1 : 0)) ++g_tuning_epoch;
  ms::g_clip32 = on ? 1 : 0;
  return prev;
}

extern "C" int ms_debug_set_patch_min_workgroups(int n) {
  const int old = ms::g_patch_min_wgs;
  ++g_tuning_epoch;
  ms::g_patch_min_wgs = n;
  return old;
}
