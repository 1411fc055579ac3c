// Patch-staged weight gradient for gfx950 (MI355X):  dw[co][(ci,kh,kw)] = sum_pix dy[co][pix] * x[ci][pix*S + tap].
//
// A workgroup owns BM output channels x BN consecutive weight columns n = (ci,kh,kw) and walks over TH x TW pixel
// tiles (its share of the reduction).  Per tile it stages dy^T [64 pixels][BM] and the RAW input patch
// [channels spanned by the BN columns][(TH-1)*S+KH][(TW-1)*S+KW] in LDS.  The MFMA reduction index is the pixel:
// A(co, pix) = dyT[pix][co]; B(pix, n) = patch[ci(n)][ty*S+kh(n)][tx*S+kw(n)] = per-lane column base + compile-time
// pixel offset, so the inner loop is ds_read_b32 with immediate offsets + v_mfma_f32_32x32x2_f32.  Patch pitches are
// chosen so that consecutive columns n fall in consecutive LDS banks (row pitch = KW, channel pitch = KH*KW mod 32).
#include <algorithm>
#include <mutex>
#include <vector>

#include "kernels.h"

namespace ms {

int g_wgrad_patch_target_wgs = 768;   // workgroups a layer's launch aims for through pixel splits (ms_debug_set_wgrad_target)
// ms_set_wgrad_batched(1): the caller queues the weight gradients of a backward pass and launches them side by side
// (ms_bwd_options.defer_wgrad_launch / ms_wgrad_flush), so a layer need not fill the chip alone: a workgroup then takes at
// least `g_wgrad_tiles_per_wg` 64-pixel tiles of the reduction.  Counters, round 5: with 768 workgroups per layer a
// classifier / UNet layer's workgroup ran 2 tiles -- its fixed parts (argument fetch, address set-up, first operand round
// trip, accumulator exchange, scattered stores) outweighed its matrix work, and the partial slabs of ~8 splits per layer
// made the slab reduction a 519 MB launch.
int g_wgrad_batched = 0;
int g_wgrad_tiles_per_wg = 16;
int g_wgrad_min_wgs = 48;       // (swept 16 / 48 / 128 on the headline G-step: 4.65 / 4.60 / 4.65 ms -- a lower floor means fewer partial slabs for the many small 1-D layers)
int g_wgrad_steps_per_wg = 96;
extern int g_wgrad_wave;

constexpr int pitch_mod32(int at_least, int want_mod) {
  int v = at_least;
  while (v % 32 != want_mod % 32) ++v;
  return v;
}

template <int KH, int KW, int S, int TW, bool UP2>
__device__ __forceinline__ void wgrad_patch_body(const WgradPatchArgs& p, const int bid) {
  constexpr int BM = 64, BN = 64, NPIX = 64, TH = NPIX / TW;
  constexpr int SV = (KH == 1) ? 1 : S;
  constexpr int KHW = KH * KW;
  constexpr int PR = (TH - 1) * SV + KH, PC = (TW - 1) * S + KW;
  constexpr int RP = pitch_mod32(PC, KW), CP = pitch_mod32(PR * RP, KHW);
  constexpr int NCH = (BN + 2 * KHW - 2) / KHW;          // channels spanned by BN consecutive columns
  constexpr int LDA = BM + 1;
  constexpr int STAGE = NPIX * LDA + NCH * CP + 4;
  constexpr int NPE = NCH * PR * PC, NP = (NPE + 255) / 256;
  constexpr int NA = BM * NPIX / 256;                    // dy values per thread per tile
  __shared__ float smem[2 * STAGE];

  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wm = wid >> 1, wn = wid & 1, khalf = lane >> 5;
  // logical block id: column tile fastest, then channel tile, then (group, pixel split); one contiguous range per XCD:
  // all workgroups of a pixel split (which re-read the same dy / x tiles) sit behind the same L2
  const int vid = xcd_remap(bid, p.gx * p.gy * p.gz);
  const int bx_ = vid % p.gx, by_ = (vid / p.gx) % p.gy, bz_ = vid / (p.gx * p.gy);
  const int g = bz_ / p.splits, sp = bz_ - g * p.splits;
  const int m0 = by_ * BM, n0 = bx_ * BN;
  const int ctot = p.groups * p.Cog;
  const int ci_first = n0 / KHW;
  const int cbase = (p.bcast ? 0 : g * p.Cig) + ci_first;
  const int tiles_per_img = p.tiles_y * p.tiles_x;
  const int tile_beg = sp * p.tiles_per_split, tile_end = min(p.n_tiles, tile_beg + p.tiles_per_split);

  // ---- tile-invariant parts of the staging addresses (raw buffer loads: see conv_patch.hip)
  int prow[NP], pcol[NP], prc[NP], ploff[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int e = t + i * 256;
    const int c = e / (PR * PC), rem = e - c * (PR * PC), r = rem / PC, col = rem - r * PC;
    const bool ok = (e < NPE) & (ci_first + c < p.Cig);
    prow[i] = ok ? r : -(1 << 20);                       // forces the bounds test to fail
    pcol[i] = col;
    prc[i] = c * p.s_chan + r * p.s_row;
    ploff[i] = e < NPE ? c * CP + r * RP + col : NCH * CP;   // pad word
  }
  const int apix = t & 63, am0 = t >> 6;                 // dy element: pixel apix, rows am0 + 4*i
  const int aty = apix / TW, atx = apix - aty * TW;
  const unsigned a_inv = 4u * (unsigned)((g * p.Cog + m0 + am0) * p.o_chan + aty * p.o_row + atx);
  const unsigned a_step = 16u * (unsigned)p.o_chan;
  const bool full_m = m0 + BM <= p.Cog;
  const __amdgpu_buffer_rsrc_t rsD = buf_rsrc(p.dyr), rsS = buf_rsrc(p.src), rsS2 = buf_rsrc(UP2 ? p.src2 : p.src);

  float ra[NA], rb[NP];
  auto load_tile = [&](int tile) {
    const int img = tile / tiles_per_img, trem = tile - img * tiles_per_img;
    const int tyi = trem / p.tiles_x, txi = trem - tyi * p.tiles_x;
    const int oy0 = tyi * TH, ox0 = txi * TW;
    // dy^T: the tile base is uniform and travels in the scalar offset
    const bool pok = (oy0 + aty < p.OUTH) & (ox0 + atx < p.OUTW);
    const unsigned sd = __builtin_amdgcn_readfirstlane(4u * (unsigned)(img * p.o_img + oy0 * p.o_row + ox0));
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const bool ok = pok & (full_m | (m0 + am0 + 4 * i < p.Cog));
      ra[i] = buf_load(rsD, ok ? a_inv + i * a_step : BUF_OOB, sd);
    }
    // raw input patch
    const int iy0 = oy0 * SV - p.PH, ix0 = ox0 * S - p.PW;
    const int xrow = img * p.s_img + cbase * p.s_chan + iy0 * p.s_row;   // even for UP2 (all strides are even)
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
  auto store_tile = [&](int buf) {
    float* As = smem + buf * STAGE;
    float* Ps = As + NPIX * LDA;
#pragma unroll
    for (int i = 0; i < NA; ++i) As[apix * LDA + am0 + 4 * i] = ra[i];
#pragma unroll
    for (int i = 0; i < NP; ++i) Ps[ploff[i]] = rb[i];
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  // per-lane operand bases
  const int a_base = khalf * LDA + wm * 32 + (lane & 31);
  int nbase;
  {
    const int n = n0 + wn * 32 + (lane & 31);
    const int c = n / KHW - ci_first, rr = n % KHW, kh = rr / KW, kw = rr - kh * KW;
    nbase = (n < p.Kg ? c * CP + kh * RP + kw : 0) + khalf * S;   // pixel k+1 is the next column of the same row
  }

  const int nsteps = tile_end - tile_beg;
  if (nsteps > 0) {
    load_tile(tile_beg);
    store_tile(0);
  }
  __syncthreads();
  for (int st = 0; st < nsteps; ++st) {
    const int cur = st & 1;
    if (st + 1 < nsteps) load_tile(tile_beg + st + 1);
    const float* As = smem + cur * STAGE;
    const float* Ps = As + NPIX * LDA;
    // operands of the next group of pixel pairs are read from LDS while the current group's MFMAs issue
    constexpr int NPAIR = NPIX / 2, GP = 4, NG = NPAIR / GP;
    float av[2][GP], bv[2][GP];
    auto read_group = [&](int gi, int slot) {
#pragma unroll
      for (int q = 0; q < GP; ++q) {
        const int k0 = 2 * (gi * GP + q);
        const int ty = k0 / TW, tx = k0 - ty * TW;
        av[slot][q] = As[a_base + k0 * LDA];
        bv[slot][q] = Ps[nbase + ty * SV * RP + tx * S];
      }
    };
    read_group(0, 0);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      if (gi + 1 < NG) read_group(gi + 1, (gi + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < GP; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[gi & 1][q], bv[gi & 1][q], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (st + 1 < nsteps) store_tile(cur ^ 1);
    __syncthreads();
  }

  float* outp = p.out + (size_t)sp * ctot * p.Kg;
  const int nc = n0 + wn * 32 + (lane & 31);
  float prev[16];
  if (p.accumulate) {          // queued launches that write dw themselves: all 16 reads first, then the writes
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = min(m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf, p.Cog - 1);
      prev[r] = outp[(size_t)(g * p.Cog + m) * p.Kg + min(nc, p.Kg - 1)];
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
    if (m < p.Cog && nc < p.Kg) outp[(size_t)(g * p.Cog + m) * p.Kg + nc] = p.accumulate ? prev[r] + acc[r] : acc[r];
  }
  if (p.counters == nullptr || p.splits == 1) return;
  // in-launch reduction over the pixel splits: the last workgroup to arrive for this tile sums the slabs in split order
  if (!splitk_arrive_last(p.counters + (g * p.gy + by_) * p.gx + bx_, p.splits, reinterpret_cast<int*>(smem))) return;
  const size_t slab = (size_t)ctot * p.Kg;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
    if (m < p.Cog && nc < p.Kg) {
      const size_t off = (size_t)(g * p.Cog + m) * p.Kg + nc;
      float v = 0.f;
      for (int k = 0; k < p.splits; ++k) v += p.out[(size_t)k * slab + off];
      p.final_out[off] = v;
    }
  }
}

template <int KH, int KW, int S, int TW, bool UP2>
__global__ __launch_bounds__(256) void wgrad_patch_kernel(const WgradPatchArgs p) {
  prefetch_kernargs<sizeof(WgradPatchArgs)>();
  wgrad_patch_body<KH, KW, S, TW, UP2>(p, (int)blockIdx.x);
}

// many blocks' weight gradients in one launch: a workgroup finds its job in the table of block ranges
template <int KH, int KW, int S, int TW, bool UP2>
__global__ __launch_bounds__(256) void wgrad_patch_multi_kernel(const WgradPatchBatch b) {
  prefetch_kernargs<128>();                                   // n and the table of block ranges
  int j = 0;
  while (j + 1 < b.n && (int)blockIdx.x >= b.block_end[j]) ++j;
  const int b0 = j ? b.block_end[j - 1] : 0;
  prefetch_kernargs<sizeof(WgradPatchArgs)>((int)offsetof(WgradPatchBatch, job) + j * (int)sizeof(WgradPatchArgs));
  wgrad_patch_body<KH, KW, S, TW, UP2>(b.job[j], (int)blockIdx.x - b0);
}

// =============================================================================================
// Lean form of the same kernel (round 5): `wgrad_wave_multi_kernel`.
// Counters on the kernel above (profiles/r05_sq_backward.json): 13 non-MFMA instructions per MFMA -- two integer divisions and
// a round of kernel-argument reloads per tile, 22 dword staging loads + 22 dword LDS stores per thread -- and a 64 x 64 tile
// that fetches 194 operand bytes per MFMA; three waves per SIMD keep the matrix pipe 58 % busy.  Two experiments shaped this
// form (tools/probe_wgrad32.py with the MS_WAVE_ABL builds): (1) waves that each stage their own operands and never meet
// (4 x fewer barriers, 2 non-MFMA instructions per MFMA) stop at 102 TF, and dropping their staging LOADS alone gives 130 TF:
// a 16-byte buffer load costs the CU's vector-memory path ~60 cycles of issue whatever it hits, so what matters is the number
// of load instructions per MFMA, i.e. operand reuse; (2) without loads AND LDS reads the loop runs at 143 TF = the clock
// (2.19 GHz under this load), so that is the ceiling.  Hence:
//   * a workgroup owns 128 x 128 (or 64 x 256 for layers of <= 64 output channels) of dw and its four waves share what is
//     staged: 97 operand bytes per MFMA, 4 staging loads per thread and 16-pixel step (32 MFMAs per wave);
//   * staging is 16-byte loads (dy: channels x 16 pixels; x: the aligned window of each input row the step touches), issued
//     two steps ahead of their use; dy is stored as it lies ([channel][pixel], pitch 20) and read back as ONE ds_read_b128 per
//     four MFMA k-steps -- lanes 0-31 take pixels p..p+3, lanes 32-63 pixels p+4..p+7, and MFMA r pairs pixel p+r with p+4+r
//     (the reduction does not care which two pixels share an MFMA); x rows keep the bank-skewed pitches of the kernel above;
//   * a step's address arithmetic is scalar and incremental (no divisions); invalid steps / channels / halo are out-of-range
//     buffer offsets (zeros): no branches in the loop body; double-buffered LDS, ONE workgroup barrier per step.
#ifndef MS_WAVE_ABL
#define MS_WAVE_ABL 0      // timing ablations of experimental builds only (tools/build_alt.sh): 1 no staging loads, 2 no LDS stores, 4 no operand reads
#endif
// Geometry of one kernel shape on the lean kernel (host side; the device code takes it from the job's arguments so that ONE
// launch serves every shape: only the stride and the wave arrangement are compile-time).
struct WaveGeo { int KHW, SV, WIN, W4, NCH, NX4, RP, CP, XBUF, SB; };
constexpr int WV_LDA = 20;                                                       // dy pitch: 16 lanes x 4 dwords cover the 64 banks
static WaveGeo wave_geo(int KH, int KW, int S, int BM, int BN) {
  WaveGeo g;
  g.KHW = KH * KW;
  g.SV = KH == 1 ? 1 : S;
  g.WIN = KW == 1 ? 16 : (15 * S + KW + 3 + 3) / 4 * 4;     // input columns staged per row (window start rounded down to a multiple of 4)
  g.W4 = g.WIN / 4;
  g.NCH = (BN + 2 * g.KHW - 2) / g.KHW;                     // channels spanned by BN consecutive columns
  g.NX4 = g.NCH * KH * g.W4;
  // LDS pitches of the x window (row, channel): 32 consecutive columns (ci, kh, kw) must fall into 32 different banks
  g.RP = KW == 3 ? 35 : KW == 4 ? 44 : KW == 8 ? 40 : g.WIN;
  g.CP = g.KHW == 1 ? 17 : KH * g.RP;
  g.XBUF = (g.NCH * g.CP + 3) / 4 * 4 + 4;                  // + 4 pad words: idle staging slots land there
  g.SB = BM * WV_LDA + g.XBUF;
  return g;
}
constexpr bool wave_banks_ok(int KH, int KW, int RP, int CP) {
  const int KHW = KH * KW;
  for (int k = 0; k < KHW; ++k) {
    const int ph = (32 * k) % KHW;            // tap phases a 32-column block can start at (column tiles start at multiples of 64)
    unsigned seen = 0;
    for (int i = 0; i < 32; ++i) {
      const int n = ph + i, c = n / KHW, tap = n % KHW, kh = tap / KW, kw = tap % KW;
      const unsigned bit = 1u << ((c * CP + kh * RP + kw) & 31);
      if (seen & bit) return false;
      seen |= bit;
    }
  }
  return true;
}
static_assert(wave_banks_ok(1, 3, 35, 35) && wave_banks_ok(3, 3, 35, 105) && wave_banks_ok(1, 4, 44, 44) && wave_banks_ok(4, 4, 44, 176) &&
              wave_banks_ok(1, 1, 16, 17) && wave_banks_ok(3, 8, 40, 120), "x window pitches put two columns of a read into one bank");

// MODE 0: plain.  1: dy rows are not 16-byte aligned / not whole 16-pixel runs (the (3, 8) layer: rows of 15) -- dy travels dword by
// dword, pixels past the row end are zeros.  2: the input is nearest_up2(a) + r (UNet1D's up path, layers.py:148-151): a at half
// resolution (p.src, 8-byte loads), r at full resolution (p.src2); the strides describe r.
template <int S, int WM, int WN, int MODE = 0>
__device__ __forceinline__ void wgrad_wave_body(const WgradPatchArgs& p, const int bid, float* smem) {
  static_assert(WM * WN == 4, "four waves");
  constexpr int LDA = WV_LDA, BM = 64 * WM, BN = 64 * WN, ABUF = BM * LDA;
  constexpr int NXL = WN == 2 ? 2 : 3;                         // 16-byte x slots per thread: covers every shape the planner sends (wave_setup checks)
  const int KH = p.KH, KW = p.KW, KHW = KH * KW, SV = KH == 1 ? 1 : S;
  const int W4 = p.w4, NX4 = p.nx4, RP = p.xrp, CP = p.xcp, XBUF = p.xbuf, SB = ABUF + XBUF;

  const int t = threadIdx.x, lane = t & 63, khalf = lane >> 5, l31 = lane & 31;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wid / WN, wn = wid - wm * WN;
  const int vid = xcd_remap(bid, p.gx * p.gy * p.gz);
  const int bx_ = vid % p.gx, by_ = (vid / p.gx) % p.gy, bz_ = vid / (p.gx * p.gy);
  const int g = bz_ / p.splits, sp = bz_ - g * p.splits;
  const int m0 = by_ * BM, n0 = bx_ * BN;
  const int ctot = p.groups * p.Cog;
  const int ci_first = n0 / KHW;
  const int cbase = (p.bcast ? 0 : g * p.Cig) + ci_first;
  const int PWA = (p.PW + 3) & ~3, shift = PWA - p.PW;
  const int nseg = (p.OUTW + 15) >> 4;

  // ---- the workgroup's steps [sbeg, send); load cursor (scalar): the step whose loads are issued next
  const int sbeg = sp * p.steps_per_split, send = min(p.n_steps, sbeg + p.steps_per_split);
  const int nst = max(0, send - sbeg);
  int lidx = sbeg, lseg, loy, limg;
  {
    const int rowi = sbeg / nseg;
    lseg = sbeg - rowi * nseg;
    limg = rowi / p.OUTH;
    loy = rowi - limg * p.OUTH;
    lseg = __builtin_amdgcn_readfirstlane(lseg); loy = __builtin_amdgcn_readfirstlane(loy); limg = __builtin_amdgcn_readfirstlane(limg);
  }
  auto advance = [&]() {
    lidx += 1; lseg += 1;
    const bool c1 = lseg >= nseg;
    lseg = c1 ? 0 : lseg;
    loy = c1 ? loy + 1 : loy;
    const bool c2 = loy >= p.OUTH;
    loy = c2 ? 0 : loy;
    limg = c2 ? limg + 1 : limg;
    // (uniform by construction; pinned to SGPRs -- a cursor the compiler keeps in VGPRs turns every staging load into a waterfall loop)
    lseg = __builtin_amdgcn_readfirstlane(lseg); loy = __builtin_amdgcn_readfirstlane(loy); limg = __builtin_amdgcn_readfirstlane(limg);
  };

  // ---- step-invariant parts of the staging addresses
  // dy: thread -> channels (t >> 2) + 64 i, pixels 4 (t & 3) .. + 3: every load instruction reads 16 runs of 64 bytes
  unsigned dvo[WM];
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const int co = (t >> 2) + 64 * i;
    dvo[i] = (m0 + co < p.Cog) ? 4u * (unsigned)((g * p.Cog + m0 + co) * p.o_chan + 4 * (t & 3)) : BUF_OOB;
  }
  const int dlo = (t >> 2) * LDA + 4 * (t & 3);                // + 64 i LDA
  int xinv[NXL], xr[NXL], xc4[NXL], xlo[NXL];
  bool xok[NXL];
#pragma unroll
  for (int i = 0; i < NXL; ++i) {
    const int e = t + 256 * i;
    const int c = e / (KH * W4), rem = e - c * (KH * W4), r = rem / W4, f = rem - r * W4;
    xok[i] = (e < NX4) & (ci_first + c < p.Cig);
    xinv[i] = c * p.s_chan + r * p.s_row + 4 * f;
    xr[i] = r; xc4[i] = 4 * f;
    xlo[i] = e < NX4 ? c * CP + r * RP + 4 * f : XBUF - 4;
  }
  const __amdgpu_buffer_rsrc_t rsD = buf_rsrc(p.dyr), rsS = buf_rsrc(MODE == 2 ? p.src2 : p.src), rsH = buf_rsrc(p.src);

  // two register sets: the loads of step i + 3 are issued while step i computes and step i + 1 moves registers -> LDS
  constexpr int NSET = MODE == 0 ? 2 : 1;       // (the two rarer modes keep one set: their extra registers would spill at three workgroups per CU)
  float4 rdd[NSET][WM], rxx[NSET][NXL];
  float2 rhh[NSET][MODE == 2 ? NXL : 1];
  auto issue_loads = [&](const int set) {
    float4 (&rd)[WM] = rdd[set];
    float4 (&rx)[NXL] = rxx[set];
    if (MS_WAVE_ABL & 1) {
#pragma unroll
      for (int i = 0; i < WM; ++i) rd[i] = float4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
      for (int i = 0; i < NXL; ++i) rx[i] = float4{1.f, 1.f, 1.f, 1.f};
      return;
    }
    const bool valid = lidx < send;
    const unsigned dso = 4u * (unsigned)(limg * p.o_img + loy * p.o_row + lseg * 16);
    if (MODE == 1) {
      const int left = p.OUTW - lseg * 16 - 4 * (t & 3);          // valid pixels from this thread's first one on
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const unsigned o = valid ? dvo[i] : BUF_OOB;
        rd[i].x = buf_load(rsD, left > 0 ? o : BUF_OOB, dso);
        rd[i].y = buf_load(rsD, left > 1 ? o + 4u : BUF_OOB, dso);
        rd[i].z = buf_load(rsD, left > 2 ? o + 8u : BUF_OOB, dso);
        rd[i].w = buf_load(rsD, left > 3 ? o + 12u : BUF_OOB, dso);
      }
    } else {
#pragma unroll
      for (int i = 0; i < WM; ++i) rd[i] = buf_load4(rsD, valid ? dvo[i] : BUF_OOB, dso);
    }
    const int iy0 = loy * SV - p.PH, ix0 = lseg * 16 * S - PWA;
    const int xb = limg * p.s_img + cbase * p.s_chan + iy0 * p.s_row + ix0;
#pragma unroll
    for (int i = 0; i < NXL; ++i) {
      const bool ok = valid & xok[i] & ((unsigned)(ix0 + xc4[i]) < (unsigned)p.SRCW) & ((unsigned)(iy0 + xr[i]) < (unsigned)p.SRCH);
      rx[i] = buf_load4(rsS, ok ? 4u * (unsigned)(xb + xinv[i]) : BUF_OOB, 0);
      if (MODE == 2) rhh[set][i] = buf_load2(rsH, ok ? 4u * (unsigned)((xb + xinv[i]) >> 1) : BUF_OOB, 0);   // (all strides of r are even)
    }
  };
  float* const buf0 = smem;
  float* const buf1 = smem + SB;
  auto write_step = [&](const int buf, const int set) {
    if (MS_WAVE_ABL & 2) return;
    const float4 (&rd)[WM] = rdd[set];
    const float4 (&rx)[NXL] = rxx[set];
    float* Ab = buf ? buf1 : buf0;
    float* Xb = Ab + ABUF;
#pragma unroll
    for (int i = 0; i < WM; ++i) *reinterpret_cast<float4*>(Ab + dlo + 64 * i * LDA) = rd[i];
#pragma unroll
    for (int i = 0; i < NXL; ++i) {       // (dword stores: the bank-skewed pitches of the 3-tap shapes are odd)
      float4 v = rx[i];
      if (MODE == 2) { const float2 h = rhh[set][i]; v.x += h.x; v.y += h.x; v.z += h.y; v.w += h.y; }
      Xb[xlo[i] + 0] = v.x; Xb[xlo[i] + 1] = v.y; Xb[xlo[i] + 2] = v.z; Xb[xlo[i] + 3] = v.w;
    }
  };

  // ---- per-lane operand bases
  const int a_off = (64 * wm + l31) * LDA + 4 * khalf;        // + 32 mi LDA + 8 q
  int nb[2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int n = n0 + 64 * wn + 32 * ni + l31;
    const int c = n / KHW - ci_first, rr = n % KHW, kh = rr / KW, kw = rr - kh * KW;
    nb[ni] = (n < p.Kg ? c * CP + kh * RP + kw : 0) + shift + 4 * khalf * S;     // + (8 q + r) S
  }
  float av[2][2][4], bv[2][2][4];                             // [set][mi | ni][r]
  auto read_group = [&](const int buf, int q, int set) {
    if (MS_WAVE_ABL & 4) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) { av[set][mi][r] = 1.f + (float)q; bv[set][mi][r] = 2.f; }
      return;
    }
    const float* Ab = buf ? buf1 : buf0;
    const float* Xb = Ab + ABUF;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const float4 v = *reinterpret_cast<const float4*>(Ab + a_off + 32 * mi * LDA + 8 * q);
      av[set][mi][0] = v.x; av[set][mi][1] = v.y; av[set][mi][2] = v.z; av[set][mi][3] = v.w;
    }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[set][ni][r] = Xb[nb[ni] + (8 * q + r) * S];
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
  auto mfma_rows = [&](int set, int r_beg, int r_end) {
#pragma unroll
    for (int r = r_beg; r < r_end; ++r)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[set][mi][r], bv[set][ni][r], acc[mi][ni], 0, 0, 0);
  };

  // ---- the loop.  Step i computes out of LDS buffer i & 1; meanwhile step i + 1 moves registers -> the other buffer and the
  // loads of step i + 3 are issued.  ONE barrier per step, placed between the step's two MFMA groups: by then every wave holds
  // BOTH groups' operands of step i in registers (read right behind the previous barrier), so the barrier only has to order
  // "buffer i & 1 may be overwritten" / "buffer (i + 1) & 1 is complete", and the operand reads of step i + 1 are issued
  // behind it with a whole MFMA group (1024 cycles) to land -- no LDS latency in front of an MFMA.  Steps past the last one
  // load zeros from out-of-range offsets: no branches in the body.
  if (nst > 0) {
    issue_loads(0); advance();
    write_step(0, 0);
    if (NSET == 2) { issue_loads(1); advance(); }             // step 1 -> set 1
    issue_loads(0); advance();                                // step 2 -> set 0 (one set: step 1)
  }
  __syncthreads();
  if (nst > 0) {
    read_group(0, 0, 0);
    read_group(0, 1, 1);
  }
  auto body = [&](const int cur) {
    write_step(cur ^ 1, (cur ^ 1) & (NSET - 1));              // step i + 1: registers (loaded two steps ago) -> the other buffer
    issue_loads((cur ^ 1) & (NSET - 1)); advance();           // step i + 3 (one set: i + 2) into the set just drained
    __builtin_amdgcn_sched_barrier(0);
    mfma_rows(0, 0, 4);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                          // buffer cur ^ 1 complete; nobody reads buffer cur any more
    read_group(cur ^ 1, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_rows(1, 0, 4);
    __builtin_amdgcn_sched_barrier(0);
    read_group(cur ^ 1, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
  };
  int it = 0;
  for (; it + 1 < nst; it += 2) {
    body(0);
    body(1);
  }
  if (it < nst) body(0);

  // ---- epilogue: every wave stores its four 32 x 32 blocks
  float* outp = p.out + (size_t)sp * ctot * p.Kg;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int nc = n0 + 64 * wn + 32 * ni + l31;
      const int mb = m0 + 64 * wm + 32 * mi + 4 * khalf;
      float prev[16];
      if (p.accumulate) {          // queued launches that write dw themselves: all 16 reads first, then the writes
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = min(mb + (r & 3) + 8 * (r >> 2), p.Cog - 1);
          prev[r] = outp[(size_t)(g * p.Cog + m) * p.Kg + min(nc, p.Kg - 1)];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mb + (r & 3) + 8 * (r >> 2);
        if (m < p.Cog && nc < p.Kg) outp[(size_t)(g * p.Cog + m) * p.Kg + nc] = p.accumulate ? prev[r] + acc[mi][ni][r] : acc[mi][ni][r];
      }
    }
  }
}

// One launch for every layer of every kernel shape: a workgroup finds its job in the table of block ranges and branches (uniformly)
// to the job's instance (stride x wave arrangement: the rest of the geometry is data) -- the queued weight gradients of a backward
// pass share ONE ramp and ONE tail instead of one per shape.
__global__ __launch_bounds__(256, 3) void wgrad_wave_multi_kernel(const WgradPatchBatch b) {
  extern __shared__ float wave_smem[];
  prefetch_kernargs<128>();
  int j = 0;
  while (j + 1 < b.n && (int)blockIdx.x >= b.block_end[j]) ++j;
  const int b0 = j ? b.block_end[j - 1] : 0;
  prefetch_kernargs<sizeof(WgradPatchArgs)>((int)offsetof(WgradPatchBatch, job) + j * (int)sizeof(WgradPatchArgs));
  const WgradPatchArgs& p = b.job[j];
  const int bid = (int)blockIdx.x - b0;
  switch (p.wave_kind) {            // stride * 2 + (tile == 64 x 256); 6, 7: stride 1, 128 x 128, MODE 1 / 2
    case 2: wgrad_wave_body<1, 2, 2>(p, bid, wave_smem); break;
    case 3: wgrad_wave_body<1, 1, 4>(p, bid, wave_smem); break;
    case 4: wgrad_wave_body<2, 2, 2>(p, bid, wave_smem); break;
    case 5: wgrad_wave_body<2, 1, 4>(p, bid, wave_smem); break;
    case 6: wgrad_wave_body<1, 2, 2, 1>(p, bid, wave_smem); break;     // ragged dy rows
    case 7: wgrad_wave_body<1, 2, 2, 2>(p, bid, wave_smem); break;     // upsample-add input
    default: break;
  }
}

int g_wgrad_wave = 1;          // ms_debug_set_wgrad_wave: 0 keeps every layer on the barrier-per-tile kernel above

// the wave-pipelined kernel takes whole 16-pixel runs and 16-byte-aligned rows
static bool wgrad_wave_ok(const WgradPatchArgs& a, int KH, int KW, int S, bool up2) {
  if (!g_wgrad_wave || a.counters) return false;
  const bool shape = (KH == 1 && KW == 3 && S == 1) || (KH == 1 && KW == 4 && S == 2) || (KH == 1 && KW == 1 && S == 1) ||
                     (KH == 3 && KW == 3 && S == 1) || (KH == 4 && KW == 4 && S == 2) || (KH == 3 && KW == 8 && S == 1);
  if (!shape) return false;
  if (up2 && !(KH == 1 && KW == 3)) return false;
  if (KW == 1 && a.PW != 0) return false;
  if (a.PW < 0 || a.PW > 3 || a.PH < 0) return false;
  if ((a.SRCW | a.s_img | a.s_chan | a.s_row) & 3) return false;                                  // x: 16-byte-aligned rows
  if (((uintptr_t)a.src | (uintptr_t)(up2 ? a.src2 : a.src)) & 15) return false;
  const bool dy_aligned = ((a.OUTW & 15) | ((a.o_img | a.o_chan | a.o_row) & 3) | (int)((uintptr_t)a.dyr & 15)) == 0;
  if (!dy_aligned && (up2 || S != 1 || a.Cog <= 64)) return false;                             // ragged dy: MODE 1 exists for stride 1, 128 x 128 only
  return true;
}
static int wave_mode(const WgradPatchArgs& a, bool up2) {
  if (up2) return 2;
  return (((a.OUTW & 15) | ((a.o_img | a.o_chan | a.o_row) & 3) | (int)((uintptr_t)a.dyr & 15)) == 0) ? 0 : 1;
}

static int launch_wave(const WgradPatchBatch& b, hipStream_t s) {
  static unsigned long long attr = 0;               // (per device)
  static int lds_bytes = 0;
  if (first_time_on_device(attr)) {
    // the largest image of any shape: 64 x 256 tiles of the 4 x 4 stride-2 kernel
    for (int kind = 0; kind < 5; ++kind) {
      static const int khs[5] = {1, 1, 1, 3, 4}, kws[5] = {3, 4, 1, 3, 4}, ss[5] = {1, 2, 1, 1, 2};
      lds_bytes = std::max(lds_bytes, 8 * std::max(wave_geo(khs[kind], kws[kind], ss[kind], 128, 128).SB, wave_geo(khs[kind], kws[kind], ss[kind], 64, 256).SB));
    }
    const char* e = getenv("MS_WAVE_LDS");        // experiments: a larger allocation = fewer workgroups per CU
    if (e && atoi(e) > lds_bytes) lds_bytes = atoi(e);
    if (hipFuncSetAttribute((const void*)wgrad_wave_multi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
      return set_error("wgrad_wave_multi_kernel: cannot set the LDS size");
    done_on_device(attr);
  }
  hipLaunchKernelGGL(wgrad_wave_multi_kernel, dim3(b.block_end[b.n - 1]), dim3(256), lds_bytes, s, b);
  return check_launch("wgrad_wave_multi_kernel");
}

// grid and steps (16-pixel runs of one output row) of a layer on the lean kernel
static int wave_setup(WgradPatchArgs& a, int wave, int KH, int KW, int S, bool up2) {
  const int bm = wave == 2 ? 64 : 128, bn = wave == 2 ? 256 : 128;
  const WaveGeo ge = wave_geo(KH, KW, S, bm, bn);
  const int mode = wave_mode(a, up2);
  a.wave_kind = mode ? 5 + mode : 2 * S + (wave == 2 ? 1 : 0);
  if (mode && (wave == 2 || S != 1)) return -1;
  a.KH = KH; a.KW = KW; a.w4 = ge.W4; a.nx4 = ge.NX4; a.xrp = ge.RP; a.xcp = ge.CP; a.xbuf = ge.XBUF;
  if (ge.NX4 > 256 * (wave == 2 ? 3 : 2)) return -1;        // (cannot happen for the shapes the planner sends)
  a.gx = cdiv(a.Kg, bn); a.gy = cdiv(a.Cog, bm); a.gz = a.groups * a.splits;
  const int imgs = a.n_tiles / std::max(1, a.tiles_y * a.tiles_x);
  a.n_steps = imgs * a.OUTH * cdiv(a.OUTW, 16);
  a.steps_per_split = cdiv(a.n_steps, a.splits);
  return a.gx * a.gy * a.gz;
}

// ---------------------------------------------------------------------------------------------
WgradPatchPlan plan_wgrad_patch(int nd, int Cog, int Kg, int groups, int KH, int KW, int SH, int SW, int B, int OH, int OW, int W,
                                bool up2) {
  WgradPatchPlan pl = {0, 64, 0, 0, 0, 1, 0, 0, 0};
  const int S = SW;
  if (nd == 2 && SH != SW) return pl;
  const bool known = (KH == 1 && KW == 3 && S == 1) || (KH == 1 && KW == 4 && S == 2) || (KH == 1 && KW == 4 && S == 1) ||
                     (KH == 1 && KW == 1 && S == 1) || (KH == 3 && KW == 3 && S == 1) || (KH == 4 && KW == 4 && S == 2) ||
                     (KH == 3 && KW == 8 && S == 1);
  if (!known) return pl;
  if (nd == 1 ? OW < 16 : OW < 15) return pl;
  const int rows = nd == 1 ? B : OH, imgs = nd == 1 ? 1 : B;
  if (nd == 1) pl.tw = OW > 32 ? 64 : OW > 16 ? 32 : 16;
  else pl.tw = OW > 16 ? 32 : 16;
  const int th = 64 / pl.tw;
  pl.tiles_y = cdiv(rows, th); pl.tiles_x = cdiv(OW, pl.tw);
  pl.n_tiles = imgs * pl.tiles_y * pl.tiles_x;
  if (pl.n_tiles < (g_patch_min_wgs > 0 ? 4 : 1)) return pl;
  pl.p6 = (g_precision == 1 && wgrad6_supported(KH, KW, S)) ? 1 : 0;     // bf16x6 kernel: 64 x 128 tiles
  // ---- the lean kernel (wgrad_wave_multi_kernel) where the geometry allows: whole 16-pixel runs, 16-byte-aligned rows
  const bool wshape = (KH == 1 && KW == 3 && S == 1) || (KH == 1 && KW == 4 && S == 2) || (KH == 1 && KW == 1 && S == 1) ||
                      (KH == 3 && KW == 3 && S == 1) || (KH == 4 && KW == 4 && S == 2);
  const bool wragged = (OW & 15) != 0;        // dy rows that are not whole 16-pixel runs: the kernel's dword path (stride 1, 128 x 128 tiles)
  if (g_wgrad_wave && !pl.p6 && (wshape || (KH == 3 && KW == 8 && S == 1)) && W > 0 && (W & 3) == 0 && Kg >= 64 &&
      (Cog > 64 || (Kg >= 256 && KH * KW > 1 && !up2 && !wragged)) && (!up2 || (KH == 1 && KW == 3)) && (!wragged || (S == 1 && !up2))) {
    pl.wave = Cog <= 64 ? 2 : 1;
    const int bm = pl.wave == 2 ? 64 : 128, bn = pl.wave == 2 ? 256 : 128;
    const long wbase = (long)cdiv(Cog, bm) * cdiv(Kg, bn) * groups;
    const int n_steps = imgs * rows * cdiv(OW, 16);
    int sp;
    if (g_wgrad_batched) {
      // queued launches share the chip: a workgroup takes ~g_wgrad_steps_per_wg steps of the reduction (long enough to amortise
      // its fixed parts, short enough for the launch's tail), but a layer keeps at least g_wgrad_min_wgs workgroups while it has
      // steps to split -- a backward pass of small layers only (the discriminator's step) still has to spread over the chip
      const int want = std::max(1, n_steps / std::max(1, g_wgrad_steps_per_wg));
      const int floor_ = std::min(std::max(1, n_steps / 4), (int)cdiv(g_wgrad_min_wgs, (int)std::max<long>(1, wbase)));
      sp = std::max(want, floor_);
    } else {
      sp = wbase < g_wgrad_patch_target_wgs ? std::max(1, (int)(g_wgrad_patch_target_wgs / wbase)) : 1;
      sp = std::min(sp, std::max(1, n_steps / 4));
    }
    sp = std::min(sp, std::max(1, pl.n_tiles));
    pl.tiles_per_split = cdiv(pl.n_tiles, sp);       // (the 64 x 64 kernel's unit: it takes the layer when a pointer is misaligned)
    pl.splits = cdiv(pl.n_tiles, pl.tiles_per_split);
    pl.ok = 1;
    return pl;
  }
  const long base = (long)cdiv(Cog, 64) * cdiv(Kg, pl.p6 ? 128 : 64) * groups;
  int splits = 1;
  // 3 workgroups share a CU: fill one round of 768 (the decoder layer: 384 tiles x 2); layers with few tiles take as many
  // pixel splits as that allows -- measured: 512 -> 768 target
  if (base < g_wgrad_patch_target_wgs) splits = std::max(1, (int)(g_wgrad_patch_target_wgs / base));
  splits = std::min(splits, std::max(1, pl.n_tiles / 1));      // at least 1 pixel tile per split
  if (g_wgrad_batched && !pl.p6) {
    // long workgroups -- but a layer keeps at least `g_wgrad_min_wgs` of them while it has tiles to split: a backward pass made
    // of small layers only (the discriminator's step) still has to spread over the chip
    const int want = std::max(1, pl.n_tiles / std::max(1, g_wgrad_tiles_per_wg));
    const int floor_ = std::min(std::max(1, pl.n_tiles), (int)cdiv(g_wgrad_min_wgs, (int)std::max<long>(1, base)));
    splits = std::min(splits, std::max(want, floor_));
  }
  pl.tiles_per_split = cdiv(pl.n_tiles, splits);
  pl.splits = cdiv(pl.n_tiles, pl.tiles_per_split);
  pl.ok = 1;
  return pl;
}

template <int KH, int KW, int S, bool UP2>
static void launch_wgp_tw(const WgradPatchArgs& a, int tw, dim3 grid, hipStream_t s) {
#define MS_WP(TW) hipLaunchKernelGGL((wgrad_patch_kernel<KH, KW, S, TW, UP2>), grid, dim3(256), 0, s, a)
  if constexpr (KH == 1) {
    if (tw == 64) MS_WP(64);
    else if (tw == 32) MS_WP(32);
    else MS_WP(16);
  } else {
    if (tw == 32) MS_WP(32);
    else MS_WP(16);
  }
#undef MS_WP
}

// ---- queued launches (ms_bwd_options.defer_wgrad_launch / ms_wgrad_flush)
template <int KH, int KW, int S, bool UP2>
static void launch_wgpm_tw(const WgradPatchBatch& b, int tw, hipStream_t s) {
  const dim3 grid(b.block_end[b.n - 1]);
#define MS_WPM(TW) hipLaunchKernelGGL((wgrad_patch_multi_kernel<KH, KW, S, TW, UP2>), grid, dim3(256), 0, s, b)
  if constexpr (KH == 1) {
    if (tw == 64) MS_WPM(64);
    else if (tw == 32) MS_WPM(32);
    else MS_WPM(16);
  } else {
    if (tw == 32) MS_WPM(32);
    else MS_WPM(16);
  }
#undef MS_WPM
}

struct PendingWgradPatch { int KH, KW, S, tw, up2, nwg, wave; double flops, bytes; WgradPatchArgs a; };
static std::vector<PendingWgradPatch> g_pending;     // process-wide (autograd's device thread queues, the caller's thread flushes)
static std::mutex g_pending_mu;

int queue_wgrad_patch(const WgradPatchArgs& a, const WgradPatchPlan& pl, int KH, int KW, int S, bool up2, double flops, double bytes) {
  PendingWgradPatch pw;
  pw.a = a;
  pw.a.gx = cdiv(a.Kg, 64); pw.a.gy = cdiv(a.Cog, 64); pw.a.gz = a.groups * a.splits;
  const double nwg = (double)pw.a.gx * pw.a.gy * pw.a.gz;
  if (nwg > 1.0e9) return set_error("wgrad grid too large");
  pw.a.counters = nullptr; pw.a.final_out = nullptr;
  // a queued kernel that writes dw itself ADDS to it: autograd may already have accumulated the parameter's other uses of the
  // step into the slot (which starts the step zeroed)
  pw.a.accumulate = a.splits == 1 ? 1 : 0;
  pw.KH = KH; pw.KW = KW; pw.S = S; pw.tw = pl.tw; pw.up2 = up2 ? 1 : 0; pw.nwg = (int)nwg; pw.flops = flops; pw.bytes = bytes;
  pw.wave = (pl.wave && wgrad_wave_ok(pw.a, KH, KW, S, up2)) ? pl.wave : 0;
  if (pw.wave) {
    pw.nwg = wave_setup(pw.a, pw.wave, KH, KW, S, up2);
    if (pw.nwg < 0) return set_error("wgrad: x window does not fit the lean kernel's staging slots");
    pw.KH = pw.KW = pw.S = 0; pw.up2 = 0;      // every shape shares the one launch
    pw.tw = 0;                       // one instance per kernel shape: the 16-pixel steps do not depend on the tile width
  }
  std::lock_guard<std::mutex> lk(g_pending_mu);
  g_pending.push_back(pw);
  return 0;
}

void wgrad_patch_discard() {
  std::lock_guard<std::mutex> lk(g_pending_mu);
  g_pending.clear();
}

int wgrad_patch_flush(hipStream_t s) {
  std::vector<PendingWgradPatch> q;
  {
    std::lock_guard<std::mutex> lk(g_pending_mu);
    q.swap(g_pending);
  }
  // workgroups are dispatched in id order: the layers whose workgroups run longest go first, so the launch's tail is made of
  // short ones (stable: equal lengths keep their queue order)
  std::stable_sort(q.begin(), q.end(), [](const PendingWgradPatch& x, const PendingWgradPatch& y) {
    const int lx = x.wave ? 4 * x.a.steps_per_split : 4 * x.a.tiles_per_split, ly = y.wave ? 4 * y.a.steps_per_split : 4 * y.a.tiles_per_split;
    return lx > ly;
  });
  std::vector<char> done(q.size(), 0);
  for (size_t i = 0; i < q.size(); ++i) {
    if (done[i]) continue;
    WgradPatchBatch b;
    b.n = 0;
    long blocks = 0;
    double flops = 0, bytes = 0;
    const PendingWgradPatch& h = q[i];
    auto launch = [&]() -> int {
      if (!b.n) return 0;
      TimingScope ts(s, flops, bytes, "%s<%d,%d,%d,%d,%d>|conv_wgrad_%s multi k%dx%d s%d tw%d up%d jobs%d wgs%ld",
                     h.wave ? "wgrad_wave_multi_kernel" : "wgrad_patch_multi_kernel", h.KH, h.KW, h.S, h.tw, h.up2,
                     h.wave ? "wave" : "patch", h.KH, h.KW, h.S, h.tw, h.up2, b.n, blocks);
      int rc = 0;
      if (!ts.skip()) {
        const int KH = h.KH, KW = h.KW, S = h.S;
        if (h.wave) {
          rc = launch_wave(b, s);
          b.n = 0; blocks = 0; flops = bytes = 0;
          return rc;
        }
        if (KH == 1 && KW == 3 && S == 1) {
          if (h.up2) launch_wgpm_tw<1, 3, 1, true>(b, h.tw, s);
          else launch_wgpm_tw<1, 3, 1, false>(b, h.tw, s);
        } else if (KH == 1 && KW == 4 && S == 2) launch_wgpm_tw<1, 4, 2, false>(b, h.tw, s);
        else if (KH == 1 && KW == 4 && S == 1) launch_wgpm_tw<1, 4, 1, false>(b, h.tw, s);
        else if (KH == 1 && KW == 1 && S == 1) launch_wgpm_tw<1, 1, 1, false>(b, h.tw, s);
        else if (KH == 3 && KW == 3 && S == 1) launch_wgpm_tw<3, 3, 1, false>(b, h.tw, s);
        else if (KH == 4 && KW == 4 && S == 2) launch_wgpm_tw<4, 4, 2, false>(b, h.tw, s);
        else launch_wgpm_tw<3, 8, 1, false>(b, h.tw, s);
        rc = check_launch("wgrad_patch_multi_kernel");
      }
      b.n = 0; blocks = 0; flops = bytes = 0;
      return rc;
    };
    for (size_t k = i; k < q.size(); ++k) {
      if (done[k] || q[k].KH != h.KH || q[k].KW != h.KW || q[k].S != h.S || q[k].tw != h.tw || q[k].up2 != h.up2 || (q[k].wave != 0) != (h.wave != 0)) continue;
      if (b.n == WGP_MAX_JOBS || blocks + q[k].nwg > 0x3fffffff) {
        const int rc = launch();
        if (rc) return rc;
      }
      blocks += q[k].nwg;
      b.block_end[b.n] = (int)blocks;
      b.job[b.n] = q[k].a;
      ++b.n;
      flops += q[k].flops; bytes += q[k].bytes;
      done[k] = 1;
    }
    const int rc = launch();
    if (rc) return rc;
  }
  return 0;
}

int launch_wgrad_patch(const WgradPatchArgs& a, const WgradPatchPlan& pl, int KH, int KW, int S, bool up2, double flops,
                       double bytes, hipStream_t s) {
  WgradPatchArgs b = a;
  b.gx = cdiv(a.Kg, 64); b.gy = cdiv(a.Cog, 64); b.gz = a.groups * a.splits;
  if ((double)b.gx * b.gy * b.gz > 2.0e9) return set_error("wgrad grid too large");
  if (pl.wave && wgrad_wave_ok(b, KH, KW, S, up2)) {
    WgradPatchBatch wb;
    const int wave = pl.wave;
    wb.n = 1; wb.job[0] = b;
    wb.block_end[0] = wave_setup(wb.job[0], wave, KH, KW, S, up2);
    if (wb.block_end[0] < 0) return set_error("wgrad: no lean-kernel instance for this layer");
    TimingScope ts(s, flops, bytes, "wgrad_wave_multi_kernel<%d,%d,%d,0,0>|conv_wgrad_wave k%dx%d s%d Cog%d Kg%d g%d steps%d splits%d",
                   KH, KW, S, KH, KW, S, a.Cog, a.Kg, a.groups, wb.job[0].n_steps, a.splits);
    if (ts.skip()) return 0;
    return launch_wave(wb, s);
  }
  dim3 grid(b.gx * b.gy * b.gz);
  TimingScope ts(s, flops, bytes, "wgrad_patch_kernel<%d,%d,%d,%d,%d>|conv_wgrad_patch k%dx%d s%d Cog%d Kg%d g%d tiles%d tw%d splits%d",
                 KH, KW, S, pl.tw, up2 ? 1 : 0, KH, KW, S, a.Cog, a.Kg, a.groups, pl.n_tiles, pl.tw, a.splits);
  if (ts.skip()) return 0;
  if (KH == 1 && KW == 3 && S == 1) {
    if (up2) launch_wgp_tw<1, 3, 1, true>(b, pl.tw, grid, s);
    else launch_wgp_tw<1, 3, 1, false>(b, pl.tw, grid, s);
  } else if (KH == 1 && KW == 4 && S == 2) launch_wgp_tw<1, 4, 2, false>(b, pl.tw, grid, s);
  else if (KH == 1 && KW == 4 && S == 1) launch_wgp_tw<1, 4, 1, false>(b, pl.tw, grid, s);
  else if (KH == 1 && KW == 1 && S == 1) launch_wgp_tw<1, 1, 1, false>(b, pl.tw, grid, s);
  else if (KH == 3 && KW == 3 && S == 1) launch_wgp_tw<3, 3, 1, false>(b, pl.tw, grid, s);
  else if (KH == 4 && KW == 4 && S == 2) launch_wgp_tw<4, 4, 2, false>(b, pl.tw, grid, s);
  else launch_wgp_tw<3, 8, 1, false>(b, pl.tw, grid, s);
  return check_launch("wgrad_patch_kernel");
}

}  // namespace ms

extern "C" int ms_debug_set_wgrad_wave(int on) {
  const int old = ms::g_wgrad_wave;
  if (old != (on ? 1 : 0)) ms_debug_set_wgrad_target(ms::g_wgrad_patch_target_wgs);   // (bumps the tuning epoch: the planner's splits follow the kernel)
  ms::g_wgrad_wave = on ? 1 : 0;
  return old;
}

extern "C" int ms_set_wgrad_batched(int on, int tiles_per_workgroup) {
  if (const char* e = getenv("MS_WGRAD_MINWG")) { if (atoi(e) > 0) ms::g_wgrad_min_wgs = atoi(e); }      // experiments
  const int old = ms::g_wgrad_batched;
  if (old != (on ? 1 : 0) || (tiles_per_workgroup > 0 && tiles_per_workgroup != ms::g_wgrad_tiles_per_wg)) ms_debug_set_wgrad_target(ms::g_wgrad_patch_target_wgs);   // (bumps the tuning epoch: slab sizes change)
  ms::g_wgrad_batched = on ? 1 : 0;
  if (tiles_per_workgroup > 0) { ms::g_wgrad_tiles_per_wg = tiles_per_workgroup; ms::g_wgrad_steps_per_wg = 4 * tiles_per_workgroup; }
  return old;
}
