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

// ---------------------------------------------------------------------------------------------
WgradPatchPlan plan_wgrad_patch(int nd, int Cog, int Kg, int groups, int KH, int KW, int SH, int SW, int B, int OH, int OW) {
  WgradPatchPlan pl = {0, 64, 0, 0, 0, 1, 0, 0};
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
  const long base = (long)cdiv(Cog, 64) * cdiv(Kg, pl.p6 ? 128 : 64) * groups;
  int splits = 1;
  // 3 workgroups share a CU: fill one round of 768 (the decoder layer: 384 tiles x 2); layers with few tiles take as many
  // pixel splits as that allows -- measured: 512 -> 768 target
  if (base < g_wgrad_patch_target_wgs) splits = std::max(1, (int)(g_wgrad_patch_target_wgs / base));
  splits = std::min(splits, std::max(1, pl.n_tiles / 1));      // at least 1 pixel tile per split
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

struct PendingWgradPatch { int KH, KW, S, tw, up2, nwg; double flops, bytes; WgradPatchArgs a; };
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
      TimingScope ts(s, flops, bytes, "wgrad_patch_multi_kernel<%d,%d,%d,%d,%d>|conv_wgrad_patch multi k%dx%d s%d tw%d up%d jobs%d wgs%ld",
                     h.KH, h.KW, h.S, h.tw, h.up2, h.KH, h.KW, h.S, h.tw, h.up2, b.n, blocks);
      int rc = 0;
      if (!ts.skip()) {
        const int KH = h.KH, KW = h.KW, S = h.S;
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
      if (done[k] || q[k].KH != h.KH || q[k].KW != h.KW || q[k].S != h.S || q[k].tw != h.tw || q[k].up2 != h.up2) continue;
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
