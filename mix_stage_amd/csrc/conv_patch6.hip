// Patch-staged convolution on the bf16 matrix pipe with fp32-equivalent accuracy ("bf16x6").
//
// Both fp32 operands are split EXACTLY into three bf16 parts by truncation (x = hi + mid + lo: 3 x 8 mantissa bits) and
// 6 of the 9 partial products are accumulated in fp32 (dropped: mid*lo, lo*mid, lo*lo <= 2^-24 relative, the size of an
// fp32 rounding error).  v_mfma_f32_32x32x16_bf16 runs 16x the flops of v_mfma_f32_32x32x2_f32 per cycle, so 6 of them
// per fp32 product are 16/6 = 2.7x faster than the exact-fp32 matrix path of conv_patch.hip at the same accuracy
// (measured against fp64: 1.2e-6 on the decoder layer, like the fp32 kernel).
//
// Same tiling idea as conv_patch.hip (raw input patch in LDS, compile-time operand offsets, raw buffer loads, XCD remap),
// with these differences:
//   * workgroup tile = 64 output channels x 128 pixels; 4 waves side by side along the pixels, each 64 x 32 (two 32x32
//     accumulators), so one converted activation fragment feeds two weight fragments;
//   * weights arrive PRE-SPLIT: three bf16 planes [rows][chunk][k-block][2 halves][8] built by split_weights_kernel (per call
//     into the workspace, or once per optimizer update by the trainer: ms_split_weights_prepare); a lane's 8 consecutive k of
//     a k-block are 8 consecutive channels at one tap, i.e. one 16-byte LDS read per plane.  (Splitting them while they are
//     staged -- 8 strided scalar loads + the split per half-block -- was measured: 71 vs 54 us on the decoder layer.)
//   * activations stay fp32 in LDS (the raw patch, as before); a lane reads its 8 channels (stride = channel pitch) and
//     splits them in registers (~36 VALU ops per 12 MFMAs).
#include <algorithm>
#include <cstdint>

#include "kernels.h"

namespace ms {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ inline u32x4 buf_load_u4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

constexpr int p6_ck(int khw) { return khw <= 4 ? 16 : 8; }              // channels per K chunk
constexpr int p6_nhb(int khw) { return khw * (p6_ck(khw) / 8); }        // half-blocks (8 channels at one tap) per chunk
constexpr int p6_nkb(int khw) { return (p6_nhb(khw) + 1) / 2; }         // 16-deep MFMA k-blocks per chunk

bool patch6_supported(int KH, int KW, int S) {
  const int khw = KH * KW;
  const bool known = (KH == 1 && KW == 2 && S == 1) || (KH == 2 && KW == 2 && S == 1) || (KH == 1 && KW == 3 && S == 1) ||
                     (KH == 1 && KW == 4 && S == 1) || (KH == 1 && KW == 1 && S == 1) ||
                     (KH == 3 && KW == 3 && S == 1) || (KH == 4 && KW == 4 && S == 2);   // (k1x4 s2: 89 KB of LDS -> fp32 kernel)
  return known;   // (3x8: 77 KB of weight planes per chunk -- stays on the fp32 kernel)
}
int patch6_row_elems(int Kc, int KH, int KW) {
  const int khw = KH * KW;
  return cdiv(Kc, p6_ck(khw)) * p6_nkb(khw) * 16;
}

// w rows [rows][Kc][KHW] fp32 -> planes[3][rows][chunk][k-block][half][8] bf16; half-block h of a chunk = (tap h / (CK/8),
// channel group h % (CK/8)); channels past Kc and the odd last half are zeros.
__global__ void split_weights_kernel(const float* __restrict__ w, unsigned short* __restrict__ planes, int rows, int Kc, int KHW,
                                     int CK, int row_elems) {
  const size_t total = (size_t)rows * row_elems;
  const int cg_per = CK / 8, nhb = KHW * cg_per, per_chunk = ((nhb + 1) / 2) * 16;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / row_elems), e = (int)(i - (size_t)row * row_elems);
    const int chunk = e / per_chunk, r = e - chunk * per_chunk, h = r / 8, j = r - h * 8;
    const int tap = h / cg_per, cg = h - tap * cg_per, ci = chunk * CK + cg * 8 + j;
    float x = 0.f;
    if (h < nhb && ci < Kc) x = w[((size_t)row * Kc + ci) * KHW + tap];
    const unsigned xb = __float_as_uint(x);
    const float r1 = x - __uint_as_float(xb & 0xffff0000u);
    const unsigned rb = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(rb & 0xffff0000u);
    planes[i] = (unsigned short)(xb >> 16);
    planes[total + i] = (unsigned short)(rb >> 16);
    planes[2 * total + i] = (unsigned short)(__float_as_uint(r2) >> 16);
  }
}

__device__ inline void split_weights_elems(const SplitJob& jb, size_t first, size_t stride) {
  const size_t total = (size_t)jb.rows * jb.row_elems;
  const int cg_per = jb.CK / 8, nhb = jb.KHW * cg_per, per_chunk = ((nhb + 1) / 2) * 16;
  for (size_t i = first; i < total; i += stride) {
    const int row = (int)(i / jb.row_elems), e = (int)(i - (size_t)row * jb.row_elems);
    const int chunk = e / per_chunk, r = e - chunk * per_chunk, h = r / 8, j = r - h * 8;
    const int tap = h / cg_per, cg = h - tap * cg_per, ci = chunk * jb.CK + cg * 8 + j;
    float x = 0.f;
    if (h < nhb && ci < jb.Kc) x = jb.w[((size_t)row * jb.Kc + ci) * jb.KHW + tap];
    const unsigned xb = __float_as_uint(x);
    const float r1 = x - __uint_as_float(xb & 0xffff0000u);
    const unsigned rb = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(rb & 0xffff0000u);
    jb.planes[i] = (unsigned short)(xb >> 16);
    jb.planes[total + i] = (unsigned short)(rb >> 16);
    jb.planes[2 * total + i] = (unsigned short)(__float_as_uint(r2) >> 16);
  }
}

// many weight tensors in ONE launch: job j owns workgroups [block_end[j-1], block_end[j])
__global__ void split_weights_multi_kernel(const SplitBatch sb) {
  int j = 0;
  while (j + 1 < sb.n && (int)blockIdx.x >= sb.job[j].block_end) ++j;
  const int b0 = j ? sb.job[j - 1].block_end : 0;
  split_weights_elems(sb.job[j], (size_t)((int)blockIdx.x - b0) * blockDim.x + threadIdx.x,
                      (size_t)(sb.job[j].block_end - b0) * blockDim.x);
}

int launch_split_weights_multi(SplitBatch& sb, hipStream_t s) {
  int blocks = 0;
  double total = 0;
  for (int j = 0; j < sb.n; ++j) {
    SplitJob& jb = sb.job[j];
    const int khw = jb.KHW;
    jb.CK = p6_ck(khw);
    jb.row_elems = cdiv(jb.Kc, jb.CK) * p6_nkb(khw) * 16;
    const size_t n = (size_t)jb.rows * jb.row_elems;
    blocks += (int)std::max<size_t>(1, std::min<size_t>((n + 1023) / 1024, 256));
    jb.block_end = blocks;
    total += (double)n;
  }
  TimingScope ts(s, 0, 10.0 * total, "split_weights_multi jobs%d n%.0f", sb.n, total);
  if (ts.skip()) return 0;
  hipLaunchKernelGGL(split_weights_multi_kernel, dim3(blocks), dim3(256), 0, s, sb);
  return check_launch("split_weights_multi_kernel");
}

int launch_split_weights(const float* w, unsigned short* planes, int rows, int Kc, int KH, int KW, hipStream_t s) {
  const int khw = KH * KW, re = patch6_row_elems(Kc, KH, KW);
  const size_t total = (size_t)rows * re;
  TimingScope ts(s, 0, 4.0 * rows * Kc * khw + 6.0 * total, "split_weights rows%d Kc%d khw%d", rows, Kc, khw);
  if (ts.skip()) return 0;
  hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, s, w, planes,
                     rows, Kc, khw, p6_ck(khw), re);
  return check_launch("split_weights_kernel");
}

// exact 3-way bf16 split of 8 floats (truncation keeps every bit: 3 x 8 mantissa bits)
__device__ inline void split3(const float (&x)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float x0 = x[2 * j], x1 = x[2 * j + 1];
    const unsigned b0 = __float_as_uint(x0), b1 = __float_as_uint(x1);
    const float r0 = x0 - __uint_as_float(b0 & 0xffff0000u), r1 = x1 - __uint_as_float(b1 & 0xffff0000u);
    const unsigned c0 = __float_as_uint(r0), c1 = __float_as_uint(r1);
    const float s0 = r0 - __uint_as_float(c0 & 0xffff0000u), s1 = r1 - __uint_as_float(c1 & 0xffff0000u);
    h[j] = __builtin_amdgcn_perm(b1, b0, 0x07060302);           // upper halves of two floats = two bf16
    m[j] = __builtin_amdgcn_perm(c1, c0, 0x07060302);
    l[j] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302);
  }
  hi = __builtin_bit_cast(bf16x8, u32x4{h[0], h[1], h[2], h[3]});
  mid = __builtin_bit_cast(bf16x8, u32x4{m[0], m[1], m[2], m[3]});
  lo = __builtin_bit_cast(bf16x8, u32x4{l[0], l[1], l[2], l[3]});
}

template <int KH, int KW, int S, int TW, bool UP2>
__global__ __launch_bounds__(256, 2) void conv_patch6_kernel(const PatchArgs p) {
  prefetch_kernargs<sizeof(PatchArgs)>();
  constexpr int KHW = KH * KW, CK = p6_ck(KHW), CG = CK / 8, NHB = p6_nhb(KHW), NKB = p6_nkb(KHW);
  constexpr int BM = 64, BN = 128, TH = BN / TW;
  constexpr int SV = (KH == 1) ? 1 : S;
  constexpr int PR = (TH - 1) * SV + KH, PC = (TW - 1) * S + KW;
  constexpr int RP = PC, CP = PR * RP + 1;              // odd-ish channel pitch: the 8 channels of a lane spread over banks
  constexpr int PA = NKB * 16 + 8;                      // bf16 per weight row (+16 B: rows 28*odd dwords apart -> b128 reads conflict-free)
  constexpr int A_STAGE = 3 * BM * PA;                  // bf16 elements
  constexpr int P_STAGE = CK * CP + 4;                  // floats
  constexpr int NPE = CK * PR * PC, NP = (NPE + 255) / 256;
  constexpr int SLOTS = NKB * 2;                        // 16-byte slots per row and plane
  constexpr int NAV = 3 * BM * SLOTS, NA = (NAV + 255) / 256;
  constexpr int LP = BN + 4;                            // pitch of the epilogue's [channel][pixel] tile
  // taps > 4: one LDS stage (the three weight planes of a chunk take 34-52 KB), two barriers per chunk; else double-buffered
  constexpr int NST = KHW > 4 ? 1 : 2;
  constexpr int SMEM_BYTES = NST * A_STAGE * 2 + NST * P_STAGE * 4;
  static_assert(SMEM_BYTES >= BM * LP * 4, "epilogue tile does not fit");
  static_assert(BN % TW == 0, "bad tile");
  __shared__ __attribute__((aligned(16))) unsigned char smem_raw[SMEM_BYTES];
  unsigned short* sA = reinterpret_cast<unsigned short*>(smem_raw);
  float* sP = reinterpret_cast<float*>(smem_raw + NST * A_STAGE * 2);

  const int t = threadIdx.x, lane = t & 63, wn = t >> 6, kb = lane >> 5;
  const int vid = xcd_remap(blockIdx.x, p.gx * p.gy * p.gz);
  const int by_ = vid % p.gy, bx_ = (vid / p.gy) % p.gx, bz_ = vid / (p.gy * p.gx);
  const int zz = p.groups * p.splitk;
  const int cls = p.ncls > 1 ? bz_ / zz : 0, bzc = bz_ - cls * zz;
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

  // ---- chunk-invariant staging offsets (bytes; BUF_OOB reads as zero)
  unsigned goff[NP], goff_h[UP2 ? NP : 1];
  int loff[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int e = t + i * 256;
    const int ci = e / (PR * PC), rem = e - ci * (PR * PC), r = rem / PC, c = rem - r * PC;
    const int iy = iy0 + r, ix = ix0 + c;
    const bool ok = (e < NPE) & ((unsigned)iy < (unsigned)p.SRCH) & ((unsigned)ix < (unsigned)p.SRCW);
    const int rowbase = ci * p.s_chan + iy * p.s_row;
    goff[i] = ok ? 4u * (unsigned)(rowbase + ix) : BUF_OOB;
    if (UP2) goff_h[i] = ok ? 4u * (unsigned)((rowbase >> 1) + (ix >> 1)) : BUF_OOB;
    loff[i] = e < NPE ? ci * CP + r * RP + c : CK * CP;
  }
  unsigned aoff[NA];
  int alds[NA];
  {
    const unsigned row_first = (unsigned)cls * p.cls_a_stride + (unsigned)(g * p.Mg + m0);   // planes row of tile row 0
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int idx = t + i * 256;                        // (plane, row, 16-byte slot)
      const int pl = idx / (BM * SLOTS), rem = idx - pl * (BM * SLOTS), row = rem / SLOTS, q = rem - row * SLOTS;
      const bool ok = (idx < NAV) & (m0 + row < p.Mg);
      aoff[i] = ok ? 2u * ((unsigned)pl * p.plane_stride + (row_first + (unsigned)row) * (unsigned)p.a_row_elems + (unsigned)q * 8u)
                   : BUF_OOB;
      alds[i] = idx < NAV ? (pl * BM + row) * PA + q * 8 : 3 * BM * PA - 8;   // out of range: a row's padding
    }
  }
  const __amdgpu_buffer_rsrc_t rsA = buf_rsrc(p.Aplanes), rsS = buf_rsrc(p.src), rsS2 = buf_rsrc(UP2 ? p.src2 : p.src);
  const int img_base = img * p.s_img;

  u32x4 ra0[NA], ra1[NA];
  float rb0[NP], rb1[NP];
  auto load_chunk = [&](int chunk, u32x4 (&ra)[NA], float (&rb)[NP]) {
    const unsigned sa = __builtin_amdgcn_readfirstlane(2u * (unsigned)(chunk * NKB * 16));
#pragma unroll
    for (int i = 0; i < NA; ++i) ra[i] = buf_load_u4(rsA, aoff[i], sa);
    const int ci0 = chunk * CK;
    const int cb = __builtin_amdgcn_readfirstlane(img_base + (cbase + ci0) * p.s_chan);
    const bool full_c = ci0 + CK <= p.Kc;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int ci = (t + i * 256) / (PR * PC);
      const bool ok = full_c | (ci0 + ci < p.Kc);
      if (UP2) rb[i] = buf_load(rsS, ok ? goff_h[i] : BUF_OOB, 4u * (unsigned)(cb >> 1)) +
                       buf_load(rsS2, ok ? goff[i] : BUF_OOB, 4u * (unsigned)cb);
      else rb[i] = buf_load(rsS, ok ? goff[i] : BUF_OOB, 4u * (unsigned)cb);
    }
  };
  auto store_chunk = [&](int buf, const u32x4 (&ra)[NA], const float (&rb)[NP]) {
    unsigned short* As = sA + buf * A_STAGE;
    float* Ps = sP + buf * P_STAGE;
#pragma unroll
    for (int i = 0; i < NA; ++i) *reinterpret_cast<u32x4*>(As + alds[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < NP; ++i) Ps[loff[i]] = rb[i];
  };

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // ---- per-lane operand bases.  k-block b, half kb -> half-block h = 2b + kb = (tap, channel group)
  const int nloc = wn * 32 + (lane & 31);
  const int pty = nloc / TW, ptx = nloc - pty * TW;
  const int pix_base = pty * SV * RP + ptx * S;
  int boff[NKB];
#pragma unroll
  for (int b = 0; b < NKB; ++b) {
    const int h = 2 * b + kb;
    const int tap = h / CG, cg = h - tap * CG, kh = tap / KW, kw = tap - kh * KW;
    boff[b] = h < NHB ? pix_base + cg * 8 * CP + kh * RP + kw : 0;    // odd last half: weights are zero there
  }
  const int a_base = (lane & 31) * PA + kb * 8;           // + (plane*BM + mi*32)*PA + b*16

  const int chunk_beg = ks * p.chunks_per_split;
  const int nchunks = min((p.Kc + CK - 1) / CK - chunk_beg, p.chunks_per_split);
  auto compute_chunk = [&](int buf) {
    const unsigned short* As = sA + buf * A_STAGE;
    const float* Ps = sP + buf * P_STAGE;
#pragma unroll
    for (int b = 0; b < NKB; ++b) {
      float xb[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) xb[j] = Ps[boff[b] + j * CP];
      bf16x8 bh, bm, bl;
      split3(xb, bh, bm, bl);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(As + a_base + (0 * BM + mi * 32) * PA + b * 16);
        const bf16x8 am = *reinterpret_cast<const bf16x8*>(As + a_base + (1 * BM + mi * 32) * PA + b * 16);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(As + a_base + (2 * BM + mi * 32) * PA + b * 16);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[mi], 0, 0, 0);     // small products first
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[mi], 0, 0, 0);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[mi], 0, 0, 0);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[mi], 0, 0, 0);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[mi], 0, 0, 0);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[mi], 0, 0, 0);
      }
    }
  };
  if (NST == 2) {
    load_chunk(chunk_beg, ra0, rb0);
    if (nchunks > 1) load_chunk(chunk_beg + 1, ra1, rb1);
    store_chunk(0, ra0, rb0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ch += 2) {
      if (ch + 2 < nchunks) load_chunk(chunk_beg + ch + 2, ra0, rb0);
      compute_chunk(0);
      if (ch + 1 < nchunks) store_chunk(1, ra1, rb1);
      __syncthreads();
      if (ch + 1 >= nchunks) break;
      if (ch + 3 < nchunks) load_chunk(chunk_beg + ch + 3, ra1, rb1);
      compute_chunk(1);
      if (ch + 2 < nchunks) store_chunk(0, ra0, rb0);
      __syncthreads();
    }
  } else {
    load_chunk(chunk_beg, ra0, rb0);
    store_chunk(0, ra0, rb0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
      if (ch + 1 < nchunks) load_chunk(chunk_beg + ch + 1, ra0, rb0);     // in flight while this chunk computes
      compute_chunk(0);
      __syncthreads();
      if (ch + 1 < nchunks) {
        store_chunk(0, ra0, rb0);
        __syncthreads();
      }
    }
  }

  // ---------------- epilogue ----------------
  const int ctot = p.groups * p.Mg;
  const int oy = oy0 + pty, ox = ox0 + ptx;
  const bool cval = (oy < OUTHc) & (ox < OUTWc);
  const int ooff = img * p.o_img + (oy * p.o_sh + o_ryc) * p.o_row + ox * p.o_sw + o_rxc;   // + channel * o_chan
  const int ep = p.ep;
  if (p.part) {                       // raw partial tile in the output layout; a split-K epilogue kernel finishes
    float* part = p.part + (size_t)ks * p.part_stride;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb;
        if (m < p.Mg && cval) part[(size_t)ooff + (size_t)(g * p.Mg + m) * p.o_chan] = acc[mi][r];
      }
    return;
  }
  float bias_r[2][16];                 // one round trip for all bias values (see conv_patch.hip)
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb;
      bias_r[mi][r] = p.bias ? p.bias[g * p.Mg + (m < p.Mg ? m : 0)] : 0.f;
    }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb;
      const bool mval = m < p.Mg;
      const int chn = g * p.Mg + (mval ? m : 0);
      const float bsv = bias_r[mi][r];
      float sc = 1.f, sh = 0.f;
      if (ep == EP_BN_EVAL) {
        const float inv = 1.0f / sqrtf(p.bn_v[chn] + p.eps);
        sc = p.bn_g[chn] * inv;
        sh = p.bn_b[chn] - p.bn_m[chn] * sc;
      }
      float v = acc[mi][r] + bsv;
      if (ep == EP_RAW_STATS) acc[mi][r] = v;
      if (ep == EP_BN_EVAL) v = lrelu(fmaf(v, sc, sh), p.slope);
      if (ep == EP_LRELU) v = lrelu(v, p.slope);
      if (ep == EP_DGRAD_UP2) {
        const float pr = lane_xor1(v);
        if (mval && cval) {
          p.out2[(size_t)ooff + (size_t)chn * p.o_chan] = v;
          if (!(lane & 1)) p.out[(size_t)(ooff >> 1) + (size_t)chn * (p.o_chan >> 1)] = v + pr;
        }
      } else if (mval && cval) {
        p.out[(size_t)ooff + (size_t)chn * p.o_chan] = v;
      }
    }

  if (ep == EP_RAW_STATS) {
    // per-channel (sum, M2 about this tile's mean) over the tile's valid pixels, fixed order: tile through LDS as
    // [channel][pixel]; 4 threads per channel take 32 pixels each
    float* tile = reinterpret_cast<float*>(smem_raw);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ml = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb;
        tile[ml * LP + nloc] = cval ? acc[mi][r] : 0.f;
      }
    __syncthreads();
    const int ch = t >> 2, q = t & 3;
    float s = 0.f;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) s += tile[ch * LP + q + 4 * i];
    s += lane_xor1(s);
    s += lane_xor2(s);
    const int cnt = min(TH, OUTHc - oy0) * min(TW, OUTWc - ox0);
    const float mean = s / (float)cnt;
    float m2 = 0.f;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) {
      const int nl = q + 4 * i;
      const bool ok = (oy0 + nl / TW < OUTHc) & (ox0 + nl % TW < OUTWc);
      const float dlt = tile[ch * LP + nl] - mean;
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
template <int KH, int KW, int S, bool UP2>
static void launch_p6_tw(const PatchArgs& a, int tw, dim3 grid, hipStream_t s) {
#define MS_P6(TW) hipLaunchKernelGGL((conv_patch6_kernel<KH, KW, S, TW, UP2>), grid, dim3(256), 0, s, a)
  if constexpr (KH == 1) {
    if (tw == 64) MS_P6(64);
    else if (tw == 32) MS_P6(32);
    else MS_P6(16);
  } else {
    if (tw == 32) MS_P6(32);
    else MS_P6(16);
  }
#undef MS_P6
}

int launch_patch6(const PatchArgs& a, const PatchPlan& pl, int KH, int KW, int S, bool up2, double flops, double bytes,
                  hipStream_t s) {
  PatchArgs b = a;
  if (!patch6_supported(KH, KW, S) || pl.tn != 2) return set_error("patch6 conv: unsupported geometry");
  if (!a.Aplanes) return set_error("patch6 conv: no split weights");
  if (a.ncls > 4) return set_error("patch conv: more than 4 parity classes");
  b.gx = pl.n_tiles; b.gy = cdiv(a.Mg, 64); b.gz = a.groups * a.splitk * std::max(1, a.ncls);
  if ((double)b.gx * b.gy * b.gz > 2.0e9) return set_error("conv grid too large");
  if (a.splitk < 1 || (a.splitk > 1 && !a.part)) return set_error("patch conv: bad split-K setup");
  if (a.src_elems >= (1u << 29) || 3.0 * a.plane_stride >= 1.0e9) return set_error("patch conv: operand of 2 GiB or more");
  dim3 grid(b.gx * b.gy * b.gz);
  TimingScope ts(s, flops, bytes, "conv_patch6_kernel<%d,%d,%d,%d,%d>|conv_%s_patch6 k%dx%d s%d Mg%d Kg%d g%d tiles%d tw%d splitk%d%s",
                 KH, KW, S, pl.tw, up2 ? 1 : 0, a.is_dgrad ? "dgrad" : "fwd", KH, KW, S, a.Mg, a.Kg, a.groups, pl.n_tiles, pl.tw,
                 a.splitk, (a.ep == EP_RAW_STATS && !a.part) ? " +bnstats" : "");
  if (ts.skip()) return 0;
  if (KH == 1 && KW == 3 && S == 1) {
    if (up2) launch_p6_tw<1, 3, 1, true>(b, pl.tw, grid, s);
    else launch_p6_tw<1, 3, 1, false>(b, pl.tw, grid, s);
  } else if (KH == 1 && KW == 4 && S == 2) launch_p6_tw<1, 4, 2, false>(b, pl.tw, grid, s);
  else if (KH == 1 && KW == 4 && S == 1) launch_p6_tw<1, 4, 1, false>(b, pl.tw, grid, s);
  else if (KH == 1 && KW == 1 && S == 1) launch_p6_tw<1, 1, 1, false>(b, pl.tw, grid, s);
  else if (KH == 1 && KW == 2 && S == 1) launch_p6_tw<1, 2, 1, false>(b, pl.tw, grid, s);
  else if (KH == 2 && KW == 2 && S == 1) launch_p6_tw<2, 2, 1, false>(b, pl.tw, grid, s);
  else if (KH == 3 && KW == 3 && S == 1) launch_p6_tw<3, 3, 1, false>(b, pl.tw, grid, s);
  else launch_p6_tw<4, 4, 2, false>(b, pl.tw, grid, s);
  return check_launch("conv_patch6_kernel");
}

}  // namespace ms

// =============================================================================================================
// Weight gradient, bf16x6:  dw[co][(ci,kh,kw)] = sum_pix dy[co][pix] * x[ci][pix*S + tap]   (see wgrad_patch.hip).
// The MFMA reduction index is the pixel.  A(co, pix) = dy: split into three bf16 planes [co][64 px] while it is staged (a
// thread converts 8 consecutive pixels and writes 16 bytes per plane); B(pix, n) = the raw fp32 patch: a lane reads its 8
// consecutive pixels of column n (stride S) and splits them in registers.  Workgroup = 64 channels x 128 weight columns,
// 4 waves side by side along the columns, each 64 x 32 (one converted B fragment feeds two A fragments).
namespace ms {

constexpr int pitch_mod32_6(int at_least, int want_mod) {
  int v = at_least;
  while (v % 32 != want_mod % 32) ++v;
  return v;
}

template <int KH, int KW, int S, int TW, bool UP2>
__global__ __launch_bounds__(256, 3) void wgrad_patch6_kernel(const WgradPatchArgs p) {
  constexpr int BM = 64, BN = 128, NPIX = 64, TH = NPIX / TW;
  constexpr int SV = (KH == 1) ? 1 : S;
  constexpr int KHW = KH * KW;
  constexpr int PR = (TH - 1) * SV + KH, PC = (TW - 1) * S + KW;
  constexpr int RP = pitch_mod32_6(PC, KW), CP = pitch_mod32_6(PR * RP, KHW);
  constexpr int NCH = (BN + 2 * KHW - 2) / KHW;          // channels spanned by BN consecutive columns
  constexpr int PAW = NPIX + 8;                          // bf16 per dy row (144 B: 16-byte reads conflict-free)
  constexpr int A_STAGE = 3 * BM * PAW;                  // bf16
  constexpr int P_STAGE = NCH * CP + 4;                  // floats
  constexpr int NPE = NCH * PR * PC, NP = (NPE + 255) / 256;
  constexpr int NA = BM * (NPIX / 8) / 256;              // 8-pixel slots of the dy tile per thread (= 2)
  constexpr int NKB = NPIX / 16;                         // 16-pixel MFMA k-blocks per tile
  static_assert(TW % 8 == 0 && NA * 256 == BM * (NPIX / 8), "bad wgrad6 configuration");
  // ONE LDS stage (the next tile waits in registers; two barriers per 48-MFMA tile step): 40-45 KB, three workgroups per CU
  __shared__ __attribute__((aligned(16))) unsigned short sA[A_STAGE];
  __shared__ float sP[P_STAGE];

  const int t = threadIdx.x, lane = t & 63, wn = t >> 6, kb = lane >> 5;
  const int vid = xcd_remap(blockIdx.x, p.gx * p.gy * p.gz);
  const int bx_ = vid % p.gx, by_ = (vid / p.gx) % p.gy, bz_ = vid / (p.gx * p.gy);
  const int g = bz_ / p.splits, sp = bz_ - g * p.splits;
  const int m0 = by_ * BM, n0 = bx_ * BN;
  const int ctot = p.groups * p.Cog;
  const int ci_first = n0 / KHW;
  const int cbase = (p.bcast ? 0 : g * p.Cig) + ci_first;
  const int tiles_per_img = p.tiles_y * p.tiles_x;
  const int tile_beg = sp * p.tiles_per_split, tile_end = min(p.n_tiles, tile_beg + p.tiles_per_split);

  int prow[NP], pcol[NP], prc[NP], ploff[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int e = t + i * 256;
    const int c = e / (PR * PC), rem = e - c * (PR * PC), r = rem / PC, col = rem - r * PC;
    const bool ok = (e < NPE) & (ci_first + c < p.Cig);
    prow[i] = ok ? r : -(1 << 20);
    pcol[i] = col;
    prc[i] = c * p.s_chan + r * p.s_row;
    ploff[i] = e < NPE ? c * CP + r * RP + col : NCH * CP;
  }
  // dy slots: (channel row, 8 consecutive pixels of the tile)
  int a_co[NA], a_ty[NA], a_tx[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int idx = t + i * 256;
    a_co[i] = idx >> 3;
    const int p8 = (idx & 7) * 8;
    a_ty[i] = p8 / TW;
    a_tx[i] = p8 - a_ty[i] * TW;
  }
  const __amdgpu_buffer_rsrc_t rsD = buf_rsrc(p.dyr), rsS = buf_rsrc(p.src), rsS2 = buf_rsrc(UP2 ? p.src2 : p.src);

  float ra[NA][8], rb[NP];
  auto load_tile = [&](int tile) {
    const int img = tile / tiles_per_img, trem = tile - img * tiles_per_img;
    const int tyi = trem / p.tiles_x, txi = trem - tyi * p.tiles_x;
    const int oy0 = tyi * TH, ox0 = txi * TW;
    const unsigned sd = __builtin_amdgcn_readfirstlane(4u * (unsigned)(img * p.o_img + oy0 * p.o_row + ox0));
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const bool rok = (m0 + a_co[i] < p.Cog) & (oy0 + a_ty[i] < p.OUTH);
      const unsigned base = 4u * (unsigned)((g * p.Cog + m0 + a_co[i]) * p.o_chan + a_ty[i] * p.o_row + a_tx[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) ra[i][j] = buf_load(rsD, (rok & (ox0 + a_tx[i] + j < p.OUTW)) ? base + 4u * j : BUF_OOB, sd);
    }
    const int iy0 = oy0 * SV - p.PH, ix0 = ox0 * S - p.PW;
    const int xrow = img * p.s_img + cbase * p.s_chan + iy0 * p.s_row;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int iy = iy0 + prow[i], ix = ix0 + pcol[i];
      const bool ok = ((unsigned)iy < (unsigned)p.SRCH) & ((unsigned)ix < (unsigned)p.SRCW);
      const int o = xrow + prc[i];
      if (UP2) rb[i] = buf_load(rsS, ok ? 4u * (unsigned)((o >> 1) + (ix >> 1)) : BUF_OOB, 0) +
                       buf_load(rsS2, ok ? 4u * (unsigned)(o + ix) : BUF_OOB, 0);
      else rb[i] = buf_load(rsS, ok ? 4u * (unsigned)(o + ix) : BUF_OOB, 0);
    }
  };
  auto store_tile = [&]() {
    unsigned short* As = sA;
    float* Ps = sP;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      bf16x8 h, m, l;
      split3(ra[i], h, m, l);
      const int o = a_co[i] * PAW + a_ty[i] * TW + a_tx[i];
      *reinterpret_cast<bf16x8*>(As + 0 * BM * PAW + o) = h;
      *reinterpret_cast<bf16x8*>(As + 1 * BM * PAW + o) = m;
      *reinterpret_cast<bf16x8*>(As + 2 * BM * PAW + o) = l;
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) Ps[ploff[i]] = rb[i];
  };

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int a_base = (lane & 31) * PAW + kb * 8;         // + (plane*BM + mi*32)*PAW + kblock*16
  int nbase;
  {
    const int n = n0 + wn * 32 + (lane & 31);
    const int c = n / KHW - ci_first, rr = n % KHW, kh = rr / KW, kw = rr - kh * KW;
    nbase = n < p.Kg ? c * CP + kh * RP + kw : 0;
  }

  const int nsteps = tile_end - tile_beg;
  if (nsteps > 0) {
    load_tile(tile_beg);
    store_tile();
  }
  __syncthreads();
  for (int st = 0; st < nsteps; ++st) {
    if (st + 1 < nsteps) load_tile(tile_beg + st + 1);
    const unsigned short* As = sA;
    const float* Ps = sP;
#pragma unroll
    for (int b = 0; b < NKB; ++b) {
      // pixels b*16 + kb*8 + j of the tile: row (b*16)/TW, columns (b*16)%TW + kb*8 + j
      const int ty = (b * 16) / TW, txb = (b * 16) % TW;
      float xb[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) xb[j] = Ps[nbase + ty * SV * RP + (txb + kb * 8 + j) * S];
      bf16x8 bh, bm, bl;
      split3(xb, bh, bm, bl);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(As + a_base + (0 * BM + mi * 32) * PAW + b * 16);
        const bf16x8 am = *reinterpret_cast<const bf16x8*>(As + a_base + (1 * BM + mi * 32) * PAW + b * 16);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(As + a_base + (2 * BM + mi * 32) * PAW + b * 16);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[mi], 0, 0, 0);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[mi], 0, 0, 0);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[mi], 0, 0, 0);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[mi], 0, 0, 0);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[mi], 0, 0, 0);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[mi], 0, 0, 0);
      }
    }
    __syncthreads();
    if (st + 1 < nsteps) {
      store_tile();
      __syncthreads();
    }
  }

  float* outp = p.out + (size_t)sp * ctot * p.Kg;
  const int nc = n0 + wn * 32 + (lane & 31);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb;
      if (m < p.Cog && nc < p.Kg) outp[(size_t)(g * p.Cog + m) * p.Kg + nc] = acc[mi][r];
    }
}

template <int KH, int KW, int S, bool UP2>
static void launch_wg6_tw(const WgradPatchArgs& a, int tw, dim3 grid, hipStream_t s) {
#define MS_W6(TW) hipLaunchKernelGGL((wgrad_patch6_kernel<KH, KW, S, TW, UP2>), grid, dim3(256), 0, s, a)
  if constexpr (KH == 1) {
    if (tw == 64) MS_W6(64);
    else if (tw == 32) MS_W6(32);
    else MS_W6(16);
  } else {
    if (tw == 32) MS_W6(32);
    else MS_W6(16);
  }
#undef MS_W6
}

bool wgrad6_supported(int KH, int KW, int S) {
  // (the others would need > 80 KB of LDS per workgroup with these tiles: they stay on the fp32 kernel)
  return (KH == 1 && KW == 3 && S == 1) || (KH == 3 && KW == 3 && S == 1) || (KH == 3 && KW == 8 && S == 1) ||
         (KH == 4 && KW == 4 && S == 2);
}

int launch_wgrad_patch6(const WgradPatchArgs& a, const WgradPatchPlan& pl, int KH, int KW, int S, bool up2, double flops,
                        double bytes, hipStream_t s) {
  WgradPatchArgs b = a;
  b.gx = cdiv(a.Kg, 128); b.gy = cdiv(a.Cog, 64); b.gz = a.groups * a.splits;
  if ((double)b.gx * b.gy * b.gz > 2.0e9) return set_error("wgrad grid too large");
  dim3 grid(b.gx * b.gy * b.gz);
  TimingScope ts(s, flops, bytes, "wgrad_patch6_kernel<%d,%d,%d,%d,%d>|conv_wgrad_patch6 k%dx%d s%d Cog%d Kg%d g%d tiles%d tw%d splits%d",
                 KH, KW, S, pl.tw, up2 ? 1 : 0, KH, KW, S, a.Cog, a.Kg, a.groups, pl.n_tiles, pl.tw, a.splits);
  if (ts.skip()) return 0;
  if (!wgrad6_supported(KH, KW, S)) return set_error("wgrad6: unsupported geometry");
  if (KH == 1 && KW == 3 && S == 1) {
    if (up2) launch_wg6_tw<1, 3, 1, true>(b, pl.tw, grid, s);
    else launch_wg6_tw<1, 3, 1, false>(b, pl.tw, grid, s);
  } else if (KH == 3 && KW == 3 && S == 1) launch_wg6_tw<3, 3, 1, false>(b, pl.tw, grid, s);
  else if (KH == 4 && KW == 4 && S == 2) launch_wg6_tw<4, 4, 2, false>(b, pl.tw, grid, s);
  else launch_wg6_tw<3, 8, 1, false>(b, pl.tw, grid, s);
  return check_launch("wgrad_patch6_kernel");
}

}  // namespace ms
