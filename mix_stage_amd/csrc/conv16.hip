// 16-bit arithmetic mode: tile planning, weight preparation and dispatch of conv16_kernel (conv16_kernel.h).
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "conv16_kernel.h"

namespace ms {

// ---------------------------------------------------------------------------------------------
// planning
static int pow2_at_least(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}
static int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

int g_conv16_force_wm = 0, g_conv16_force_wn = 0;     // tuning knob (ms_debug_set_conv16_tile)
int g_conv16_dma = 1;                                 // tuning knob: 0 = register-staged path everywhere
int g_conv16_wide8 = 0;                               // tuning knob: 8-wave form of the 128 x 128 tile (measured: no gain)
int g_conv16_dbg = 0;
int g_conv16_ring = 0;                                // tuning knob: force the LDS-DMA ring depth (0 = planner)
int g_conv16_big_stages = 2;                          // tuning knob: 0 = short stages everywhere (ms_debug_set_conv16_ring, bit 8 of the flags)

Conv16Plan plan_conv16(int nd, int Mg, int groups, int Kc, int KH, int KW, int SH, int SW, int B, int OH, int OW, int zmul,
                       bool up2) {
  Conv16Plan pl = {};
  if (!(KW == 1 || KW == 2 || KW == 3 || KW == 4 || KW == 8)) return pl;
  if (nd == 2 && SH != SW) return pl;
  if (up2 && !(KW == 3 && SW == 1 && nd == 1)) return pl;
  const int rows = nd == 1 ? B : OH, imgs = nd == 1 ? 1 : B;
  const int S = SW, SV = nd == 1 ? 1 : SH;
  const int ck8 = conv16_ck8(KW);
  // candidate tiles, largest first: the first one that fills the chip (>= 192 workgroups), else the smallest
  const int cand[3][2] = {{2, 2}, {1, 2}, {1, 1}};
  for (int c = 0; c < 3; ++c) {
    int wm = cand[c][0], wn = cand[c][1], nwn = 2;
    const int bm = 64 * wm, bn = 64 * wn;
    // the 128 x 128 tile runs as 8 waves (2 x 4, 64 x 32 each) on the LDS-DMA path: same tile, half the per-wave overhead
    if (c == 0 && g_conv16_wide8 && g_conv16_dma && !up2) { wn = 1; nwn = 4; }
    const int nt = 128 * nwn;
    if (g_conv16_force_wm && (cand[c][0] != g_conv16_force_wm || cand[c][1] != g_conv16_force_wn)) continue;
    if (c == 0 && Mg < 128) continue;          // 128-row tiles only for layers with >= 128 rows per group
    const int tw = std::min(pow2_at_least(OW), bn), th = bn / tw;
    const int pc = (tw - 1) * S + KW;
    const int tiles_y = cdiv(rows, th), tiles_x = cdiv(OW, tw);
    const long nwg = (long)imgs * tiles_y * tiles_x * cdiv(Mg, bm) * groups * zmul;
    const bool fits = ck8 * th * pc <= CONV16_NP * nt;
    if (!fits) continue;
    // the 128 x 128 tile only when it leaves >= 384 workgroups: with 256 (the grouped decoder) every CU would hold ONE
    // workgroup whose LDS reads and MFMAs alternate; 64 x 128 tiles put two on a CU and they overlap (20.2 -> 18.7 us)
    if (nwg >= (c == 0 ? 384 : 192) || c == 2 || g_conv16_force_wm) {
      pl.ok = 1; pl.wm = wm; pl.wn = wn; pl.nwn = nwn; pl.tw = tw; pl.th = th; pl.tiles_y = tiles_y; pl.tiles_x = tiles_x;
      pl.n_tiles = imgs * tiles_y * tiles_x;
      pl.ck8 = ck8; pl.nchunks = cdiv(c8_of(Kc), ck8); pl.pc = pc;
      pl.lds_bytes = 2 * (KW * ck8 * bm + ck8 * th * pc + 1) * 16;
      // 64 x 64 (and 64 x 128) tiles on at most 256 workgroups (one per CU, the small layers): stages of twice the channels.  Measured per
      // launch: 256->256 k3 T=64 12.9 -> 9.9 us, k4 s2 13.9 -> 10.8, 3x3 (8,16) 23.0 -> 16.8, 3x8 38.0 -> 30.2; the tiles that share
      // a CU (decoder, first audio-encoder layers) lose 10 % with it and keep the short stages
      if (g_conv16_dma && !up2 && (c == 2 || c == 1) && nwg <= 256 && g_conv16_big_stages && 2 * ck8 * th * pc <= CONV16_NP * nt &&
          2 * (KW * 2 * ck8 * bm + cdiv(2 * ck8 * th * pc, nt) * nt) * 16 <= 160 * 1024) {
        int mult = 2;
        // four times the channels where two such stages still fit (k <= 3 taps per row) and the reduction has more than 2 of them
        if (g_conv16_big_stages >= 2 && c == 2 && KW <= 3 && 4 * ck8 * th * pc <= CONV16_NP * nt && c8_of(Kc) > 2 * ck8 &&
            2 * (KW * 4 * ck8 * bm + cdiv(4 * ck8 * th * pc, nt) * nt) * 16 <= 160 * 1024)
          mult = 4;
        const int ck8b = mult * ck8;
        pl.ck8 = ck8b; pl.nchunks = cdiv(c8_of(Kc), ck8b);
        const int stage = (KW * ck8b * bm + cdiv(ck8b * th * pc, nt) * nt) * 16;
        const int nstages = pl.nchunks * KH;
        int nstg = std::min(std::min(CONV16_MAX_RING, nstages + 1), 160 * 1024 / stage);
        if (g_conv16_ring >= 2) nstg = std::min(nstg, g_conv16_ring);
        pl.dma = 1; pl.nstg = std::max(2, nstg); pl.lds_bytes = pl.nstg * stage;
        return pl;
      }
      if (g_conv16_dma && !up2) {
        // LDS-DMA ring: up to 4 buffers; when the grid holds more workgroups than CUs the ring is kept below half of the
        // CU's LDS so that two workgroups share a CU (one computes while the other waits for its stage)
        const int stage = (KW * ck8 * bm + cdiv(ck8 * th * pc, nt) * nt) * 16;
        const int budget = nwg > 256 ? 80 * 1024 : 160 * 1024;
        // (depth up to 8: the small layers -- 64 x 64 tiles, a handful of workgroups -- are pure latency chains, and with the
        // whole reduction in flight at once they pay one memory round trip instead of one per 3 stages)
        const int nstages = cdiv(c8_of(Kc), ck8) * KH;
        int nstg = std::min(std::min(CONV16_MAX_RING, nstages + 1), budget / stage);
        if (nstg < 2) nstg = std::min(4, 160 * 1024 / stage);
        if (g_conv16_ring >= 2) nstg = std::min(nstg, g_conv16_ring);   // knob: cap the depth
        if (nstg >= 2) { pl.dma = 1; pl.nstg = nstg; pl.lds_bytes = nstg * stage; }
      }
      (void)SV;
      return pl;
    }
  }
  return pl;
}

size_t conv16_weight_bytes(const Conv16Plan& pl, int Mg, int groups, int Kc, int KH, int KW, int ncls) {
  (void)Kc;
  const int bm = 64 * pl.wm;
  return (size_t)ncls * groups * cdiv(Mg, bm) * pl.nchunks * KH * KW * pl.ck8 * bm * 16;
}

// ---------------------------------------------------------------------------------------------
// weight preparation.  One thread per 16-byte vector of the A operand:
//   out[cls][g][mt][ch][kh][kw][ks][h][row] = 8 consecutive reduction channels (ch*CK8*8 + (2*ks+h)*8 + j) of row mt*BM+row
// forward:       row = output channel of group g, reduction channel = input channel, tap (kh, kw)
// data gradient: row = input channel, reduction channel = output channel (bcast: of ANY group, all groups sum into the
//                shared input), class (ry, rx) of a strided conv keeps taps kh0 + SH*jh, kw0 + SW*jw, reversed -- the
//                same slabs transpose_weight_kernel(flip=1) builds for the fp32 kernels.
// A workgroup converts one unit = (group, row tile, reduction chunk, block of R rows) for ALL taps (and, for the data
// gradient, all stride classes): phase 1 reads the unit's fp32 source -- per row (forward) or per reduction channel (data
// gradient) one CONTIGUOUS run of the weight tensor -- coalesced into LDS; phase 2 gathers each output vector's 8 reduction
// channels from LDS and writes the vectors in output order (consecutive lanes = consecutive rows).  Every weight is read
// from HBM once, in full cache lines.
constexpr int PREP16_LDS = 8192;     // floats
__host__ __device__ inline int prep16_rows(int BM, int khw, int nk) {      // nk = reduction channels of a chunk (CK8 * 8)
  int r = 1;
  while (r * 2 <= BM && (r * 2) * (nk * khw + 1) <= PREP16_LDS && nk * ((r * 2) * khw + 1) <= PREP16_LDS) r *= 2;
  return r;
}
__host__ __device__ inline int prep16_units(const Prep16Job& jb) {
  const int tg = (jb.dgrad && jb.bcast) ? 1 : jb.groups;
  return tg * jb.n_mt * jb.nchunks * (jb.BM / prep16_rows(jb.BM, jb.KH * jb.KW, jb.CK8 * 8));
}

template <typename DT>
__device__ inline void prep16_unit(const Prep16Job& jb, int unit, float* lds) {
  const int t = threadIdx.x;
  const int KS = jb.CK8 / 2, BM = jb.BM, KH = jb.KH, KW = jb.KW, KHW = KH * KW;
  const bool dg = jb.dgrad != 0;
  const int KHs = dg ? cdiv_dev(KH, jb.SH) : KH, KWs = dg ? cdiv_dev(KW, jb.SW) : KW;
  const int tg = (dg && jb.bcast) ? 1 : jb.groups, ncls = dg ? jb.SH * jb.SW : 1;
  const int R = prep16_rows(BM, KHW, jb.CK8 * 8), nrb = BM / R;
  int x = unit;
  const int rb = x % nrb; x /= nrb;
  const int ch = x % jb.nchunks; x /= jb.nchunks;
  const int mt = x % jb.n_mt;
  const int g = x / jb.n_mt;
  const int m0 = mt * BM + rb * R, k0 = ch * jb.CK8 * 8;
  const int nk = jb.CK8 * 8;                                   // reduction channels of a chunk
  __syncthreads();                                             // the previous unit's gathers are done
  const bool w16 = (reinterpret_cast<uintptr_t>(jb.w) & 15) == 0 && ((jb.Cig * KHW) & 3) == 0;
  if (!dg) {
    // lds[row][kk*KHW + tap], row pitch nk*KHW + 1
    const int run = nk * KHW, pitch = run + 1;
    if (w16 && (run & 3) == 0) {
      // 16-byte loads: 4 consecutive elements of a row's run per lane, 4 loads in flight per thread (the index arithmetic once
      // per vector); a vector that reaches past the layer's channels (last chunk only) falls back to masked scalar loads
      const int nv = R * (run >> 2);
      for (int v0 = t; v0 < nv; v0 += 4 * 256) {
        float4 val[4];
        int dst[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int v = v0 + q * 256;
          const int vc = min(v, nv - 1);
          const int row = vc / (run >> 2), off = (vc - row * (run >> 2)) << 2;
          const int m = m0 + row;
          const bool rowok = v < nv && m < jb.Cog;
          const int co = g * jb.Cog + min(m, jb.Cog - 1);
          const size_t src = ((size_t)co * jb.Cig + k0) * KHW + off;
          const bool full = rowok && k0 + (off + 3) / KHW < jb.Cig;
          float4 x = {0.f, 0.f, 0.f, 0.f};
          if (full) {
            x = *reinterpret_cast<const float4*>(jb.w + src);
          } else if (rowok) {
            if (k0 + (off + 0) / KHW < jb.Cig) x.x = jb.w[src + 0];
            if (k0 + (off + 1) / KHW < jb.Cig) x.y = jb.w[src + 1];
            if (k0 + (off + 2) / KHW < jb.Cig) x.z = jb.w[src + 2];
            if (k0 + (off + 3) / KHW < jb.Cig) x.w = jb.w[src + 3];
          }
          if (jb.scale) { const float sc = jb.scale[co]; x.x *= sc; x.y *= sc; x.z *= sc; x.w *= sc; }
          if (!rowok) x = float4{0.f, 0.f, 0.f, 0.f};
          val[q] = x;
          dst[q] = v < nv ? row * pitch + off : -1;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (dst[q] >= 0) { lds[dst[q]] = val[q].x; lds[dst[q] + 1] = val[q].y; lds[dst[q] + 2] = val[q].z; lds[dst[q] + 3] = val[q].w; }
      }
    } else {
    // 8 loads in flight per thread (clamped addresses, masked values): the unit is a latency chain otherwise
    for (int e0 = t; e0 < R * run; e0 += 8 * 256) {
      float val[8];
      int dst[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int e = e0 + q * 256;
        const int ec = min(e, R * run - 1);
        const int row = ec / run, off = ec - row * run;
        const int m = m0 + row, kk = off / KHW;
        const bool ok = e < R * run && m < jb.Cog && k0 + kk < jb.Cig;
        const int co = g * jb.Cog + min(m, jb.Cog - 1);
        const size_t src = ((size_t)co * jb.Cig + k0) * KHW + off;
        val[q] = jb.w[ok ? src : 0];
        if (jb.scale) val[q] *= jb.scale[co];
        if (!ok) val[q] = 0.f;
        dst[q] = e < R * run ? row * pitch + off : -1;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (dst[q] >= 0) lds[dst[q]] = val[q];
    }
    }
  } else {
    // lds[kk][row*KHW + tap], pitch R*KHW + 1; reduction channel k = output channel (bcast: of any group)
    const int run = R * KHW, pitch = run + 1;
    const int tcog = jb.bcast ? jb.groups * jb.Cog : jb.Cog;
    if (w16 && (run & 3) == 0 && ((m0 * KHW) & 3) == 0) {
      const int nv = nk * (run >> 2);
      for (int v0 = t; v0 < nv; v0 += 4 * 256) {
        float4 val[4];
        int dst[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int v = v0 + q * 256;
          const int vc = min(v, nv - 1);
          const int kk = vc / (run >> 2), off = (vc - kk * (run >> 2)) << 2;
          const int k = k0 + kk;
          const bool kok = v < nv && k < tcog;
          const int co = jb.bcast ? min(k, tcog - 1) : g * jb.Cog + min(k, tcog - 1);
          const size_t src = ((size_t)co * jb.Cig + m0) * KHW + off;
          const bool full = kok && m0 + (off + 3) / KHW < jb.Cig;
          float4 x = {0.f, 0.f, 0.f, 0.f};
          if (full) {
            x = *reinterpret_cast<const float4*>(jb.w + src);
          } else if (kok) {
            if (m0 + (off + 0) / KHW < jb.Cig) x.x = jb.w[src + 0];
            if (m0 + (off + 1) / KHW < jb.Cig) x.y = jb.w[src + 1];
            if (m0 + (off + 2) / KHW < jb.Cig) x.z = jb.w[src + 2];
            if (m0 + (off + 3) / KHW < jb.Cig) x.w = jb.w[src + 3];
          }
          val[q] = x;
          dst[q] = v < nv ? kk * pitch + off : -1;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (dst[q] >= 0) { lds[dst[q]] = val[q].x; lds[dst[q] + 1] = val[q].y; lds[dst[q] + 2] = val[q].z; lds[dst[q] + 3] = val[q].w; }
      }
    } else {
    for (int e0 = t; e0 < nk * run; e0 += 8 * 256) {
      float val[8];
      int dst[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int e = e0 + q * 256;
        const int ec = min(e, nk * run - 1);
        const int kk = ec / run, off = ec - kk * run;
        const int row = off / KHW, m = m0 + row, k = k0 + kk;
        const bool ok = e < nk * run && m < jb.Cig && k < tcog;
        const int co = jb.bcast ? k : g * jb.Cog + k;
        val[q] = jb.w[ok ? ((size_t)co * jb.Cig + m0) * KHW + off : 0];
        if (!ok) val[q] = 0.f;
        dst[q] = e < nk * run ? kk * pitch + off : -1;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (dst[q] >= 0) lds[dst[q]] = val[q];
    }
    }
  }
  __syncthreads();
  const int nvec = ncls * KHs * KWs * KS * 2 * R;
  for (int ov = t; ov < nvec; ov += 256) {
    int y = ov;
    const int row = y % R; y /= R;
    const int h = y & 1; y >>= 1;
    const int ks = y % KS; y /= KS;
    const int kw = y % KWs; y /= KWs;
    const int kh = y % KHs;
    const int cls = y / KHs;
    float f[8];
    if (!dg) {
      const float* src = lds + row * (nk * KHW + 1) + ((2 * ks + h) * 8) * KHW + kh * KW + kw;
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = src[j * KHW];
    } else {
      const int ry = cls / jb.SW, rx = cls - ry * jb.SW;
      const int jh = KHs - 1 - kh, jw = KWs - 1 - kw;              // taps reversed: the data gradient is a forward conv
      const int okh = (ry + jb.PH) % jb.SH + jb.SH * jh, okw = (rx + jb.PW) % jb.SW + jb.SW * jw;
      const bool tap_ok = okh < KH && okw < KW;
      const float* src = lds + ((2 * ks + h) * 8) * (R * KHW + 1) + row * KHW + okh * KW + okw;
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = tap_ok ? src[j * (R * KHW + 1)] : 0.f;
    }
    const size_t v = ((((((((size_t)cls * tg + g) * jb.n_mt + mt) * jb.nchunks + ch) * KHs + kh) * KWs + kw) * KS + ks) * 2 + h) * BM +
                     rb * R + row;
    reinterpret_cast<u32x4*>(jb.out)[v] = pack8<DT>(f);
  }
}

__global__ __launch_bounds__(256) void prep16_multi_kernel(const Prep16Batch pb) {
  __shared__ float lds[PREP16_LDS + 64];
  int j = 0;
  while (j + 1 < pb.n && (int)blockIdx.x >= pb.job[j].block_end) ++j;
  const Prep16Job& jb = pb.job[j];
  const int b0 = j ? pb.job[j - 1].block_end : 0, nb = jb.block_end - b0;
  const int units = prep16_units(jb);
  for (int u = (int)blockIdx.x - b0; u < units; u += nb) {
    if (jb.dt == DT_BF16) prep16_unit<BF16>(jb, u, lds);
    else prep16_unit<F16>(jb, u, lds);
  }
}

int launch_prep16_multi(Prep16Batch& pb, hipStream_t s) {
  if (pb.n <= 0) return 0;
  int blocks = 0;
  double total = 0;
  for (int i = 0; i < pb.n; ++i) {
    Prep16Job& jb = pb.job[i];
    const bool dg = jb.dgrad != 0;
    const int KHs = dg ? cdiv(jb.KH, jb.SH) : jb.KH, KWs = dg ? cdiv(jb.KW, jb.SW) : jb.KW;
    const int tg = (dg && jb.bcast) ? 1 : jb.groups, ncls = dg ? jb.SH * jb.SW : 1;
    const double n = (double)ncls * tg * jb.n_mt * jb.nchunks * KHs * KWs * jb.CK8 * jb.BM;
    total += n;
    blocks += std::min(4096, std::max(1, prep16_units(jb)));
    jb.block_end = blocks;
  }
  TimingScope ts(s, 0, 48.0 * total, "prep16_multi_kernel|prep16_multi jobs%d", pb.n);
  if (ts.skip()) return 0;
  hipLaunchKernelGGL(prep16_multi_kernel, dim3(blocks), dim3(256), 0, s, pb);
  return check_launch("prep16_multi_kernel");
}

// ---------------------------------------------------------------------------------------------
// dispatch
// resident_query != NULL: no launch; *resident_query = workgroups of this instance the device holds at once
template <typename DT, int KW, int WM, int WN, bool UP2, bool DMA, int NWN = 2, int CKX = 1>
static int launch_one(const Conv16Args& a, int lds_bytes, int nwg, hipStream_t s, int* resident_query) {
  static bool attr_done = false;          // kernels that stage more than 64 KiB need the limit raised once
  auto fn = conv16_kernel<DT, KW, WM, WN, UP2, DMA, NWN, CKX>;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return set_error("conv16: cannot raise the dynamic LDS limit");
    attr_done = true;
  }
  if (resident_query) {
    static int cached_lds = -1, cached = 0;
    if (cached_lds != lds_bytes) {
      int per_cu = 0, dev = 0, cus = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(fn), 128 * NWN, lds_bytes) != hipSuccess ||
          hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
        (void)hipGetLastError();
        per_cu = 0;
      }
      // the occupancy query can answer one workgroup per CU too many for kernels with 81+ SGPRs (MI355X_MICROARCH.md,
      // "Residency and cooperative launch"): stay at or below what 112 SGPRs admit
      per_cu = std::min(per_cu, 6 * 256 / (128 * NWN));
      // ... and leave room in the register file for foreign waves: a launch whose workgroups fill the CU's registers exactly
      // (4 x 128 per SIMD, measured with the fp32 decoder kernel) stalls for SECONDS when a second stream's kernel holds a few
      // registers of one CU while this launch's resident workgroups spin for the member that cannot be placed.  With
      // 96 registers per SIMD lane left over the foreign wave and the member coexist (tests/test_gpu_bnfused.py, the load test).
      hipFuncAttributes fa = {};
      if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(fn)) != hipSuccess || fa.numRegs <= 0) {
        (void)hipGetLastError();
        per_cu = 0;
      } else {
        const int regs = (fa.numRegs + 7) & ~7, waves_per_simd = (128 * NWN) / 256 > 0 ? (128 * NWN) / 256 : 1;
        per_cu = std::min(per_cu, (512 - 96) / (regs * waves_per_simd));
      }
      cached = per_cu * cus;
      cached_lds = lds_bytes;
      if (getenv("MS_PRINT_REGS"))
        fprintf(stderr, "conv16<KW%d WM%d WN%d UP2%d DMA%d NWN%d CKX%d> lds %d: %d workgroups per CU for the in-launch BatchNorm, %d registers\n", KW, WM,
                WN, (int)UP2, (int)DMA, NWN, CKX, lds_bytes, per_cu, fa.numRegs);
    }
    *resident_query = cached;
    return 0;
  }
  hipLaunchKernelGGL(fn, dim3(nwg), dim3(128 * NWN), lds_bytes, s, a);
  return 0;
}

template <typename DT, int KW, bool UP2>
static int launch_tile(const Conv16Args& a, const Conv16Plan& pl, int nwg, hipStream_t s, int* rq) {
  if constexpr (!UP2) {
    if (pl.dma) {
      if (pl.wm == 2 && pl.wn == 1 && pl.nwn == 4) return launch_one<DT, KW, 2, 1, false, true, 4>(a, pl.lds_bytes, nwg, s, rq);
      if (pl.wm == 2 && pl.wn == 2) return launch_one<DT, KW, 2, 2, false, true>(a, pl.lds_bytes, nwg, s, rq);
      if (pl.wm == 1 && pl.wn == 2)
        return pl.ck8 == 2 * conv16_ck8(KW) ? launch_one<DT, KW, 1, 2, false, true, 2, 2>(a, pl.lds_bytes, nwg, s, rq)
                                            : launch_one<DT, KW, 1, 2, false, true>(a, pl.lds_bytes, nwg, s, rq);
      if constexpr (KW <= 3) {
        if (pl.ck8 == 4 * conv16_ck8(KW)) return launch_one<DT, KW, 1, 1, false, true, 2, 4>(a, pl.lds_bytes, nwg, s, rq);
      }
      if (pl.ck8 == 2 * conv16_ck8(KW)) return launch_one<DT, KW, 1, 1, false, true, 2, 2>(a, pl.lds_bytes, nwg, s, rq);
      return launch_one<DT, KW, 1, 1, false, true>(a, pl.lds_bytes, nwg, s, rq);
    }
  }
  if (pl.wm == 2 && pl.wn == 2) return launch_one<DT, KW, 2, 2, UP2, false>(a, pl.lds_bytes, nwg, s, rq);
  if (pl.wm == 1 && pl.wn == 2) return launch_one<DT, KW, 1, 2, UP2, false>(a, pl.lds_bytes, nwg, s, rq);
  return launch_one<DT, KW, 1, 1, UP2, false>(a, pl.lds_bytes, nwg, s, rq);
}

template <typename DT>
static int launch_kw(const Conv16Args& a, const Conv16Plan& pl, int KW, bool up2, int nwg, hipStream_t s, int* rq = nullptr) {
  if (up2) return launch_tile<DT, 3, true>(a, pl, nwg, s, rq);
  switch (KW) {
    case 1: return launch_tile<DT, 1, false>(a, pl, nwg, s, rq);
    case 2: return launch_tile<DT, 2, false>(a, pl, nwg, s, rq);
    case 3: return launch_tile<DT, 3, false>(a, pl, nwg, s, rq);
    case 4: return launch_tile<DT, 4, false>(a, pl, nwg, s, rq);
    case 8: return launch_tile<DT, 8, false>(a, pl, nwg, s, rq);
  }
  return set_error("conv16: no kernel for %d taps per row", KW);
}

int g_bn_fused = 1;               // tuning / test knob (ms_debug_set_bn_fused): 0 = BatchNorm always as its own launch
int g_bn_fused_min_wgs = 0;       // ... and the smallest grid that takes the in-launch form (ms_debug_set_bn_fused_min_workgroups)
int* g_bn_sync = nullptr;         // ms_set_bn_sync_buffer
int g_bn_sync_n = 0;

// Can EVERY workgroup of this launch be resident at once?  The in-launch BatchNorm (EP_BN_FUSED) makes the workgroups of a
// channel tile wait for each other; with the whole grid resident that cannot deadlock under any dispatch order.
bool conv16_coresident(int dt, const Conv16Plan& pl, int KW, bool up2, int nwg) {
  if (!pl.ok) return false;
  Conv16Args a = {};
  int resident = 0;
  const int rc = dt == DT_BF16 ? launch_kw<BF16>(a, pl, KW, up2, nwg, nullptr, &resident) : launch_kw<F16>(a, pl, KW, up2, nwg, nullptr, &resident);
  return rc == 0 && nwg <= resident;
}

int launch_conv16(int dt, const Conv16Args& a, const Conv16Plan& pl, int KW, bool up2, double flops, double bytes,
                  hipStream_t s) {
  if (!pl.ok) return set_error("conv16: geometry not supported (taps per row %d)", KW);
  if (a.ncls > 4) return set_error("conv16: more than 4 parity classes");
  const int bm = 64 * pl.wm;
  Conv16Args b = a;
  b.ltw = ilog2(pl.tw); b.TH = pl.th; b.PC = pl.pc; b.nchunks = pl.nchunks; b.tiles_x = pl.tiles_x; b.tiles_y = pl.tiles_y;
  b.gx = pl.n_tiles; b.gy = cdiv(a.Mg, bm); b.gz = a.groups * std::max(1, a.ncls);
  b.nstg = pl.nstg; b.dbg = g_conv16_dbg;
  const unsigned nav = (unsigned)(KW * pl.ck8 * bm);
  b.a_mt_stride = (unsigned)(pl.nchunks * a.KH) * nav;
  b.a_group_stride = (unsigned)b.gy * b.a_mt_stride;
  b.a_cls_stride = (unsigned)a.groups * b.a_group_stride;
  if ((double)b.gx * b.gy * b.gz > 2.0e9) return set_error("conv16: grid too large");
  if ((double)b.a_cls_stride * std::max(1, a.ncls) * 16.0 >= 4.0e9) return set_error("conv16: prepared weights of 4 GB or more");
  if (a.groups > 1 && (a.Mg & 7)) return set_error("conv16: grouped blocks need a multiple of 8 output channels per group");
  const int nwg = b.gx * b.gy * b.gz;
  char ring[32] = "";
  if (pl.dma) {
    const int nt = 128 * pl.nwn, na = (KW * pl.ck8 * bm) / nt, npd = cdiv(pl.ck8 * pl.th * pl.pc, nt);
    snprintf(ring, sizeof(ring), " dma%d w%d", pl.nstg, (pl.nstg - 2) * (na + npd));
  }
  TimingScope ts(s, flops, bytes, "conv16_kernel<%s,%d,%d,%d,%d,%d,%d>|conv_%s_cb8 k%dx%d s%d Mg%d Kg%d g%d tiles%d tile%dx%d tw%d%s%s",
                 dt == DT_BF16 ? "bf16" : "f16", KW, pl.wm, pl.wn, up2 ? 1 : 0, pl.dma, pl.nwn, a.is_dgrad ? "dgrad" : "fwd", a.KH, KW, a.S, a.Mg,
                 a.Kc8g * 8 * a.KH * KW, a.groups, pl.n_tiles, bm, 32 * pl.wn * pl.nwn, pl.tw, ring,
                 a.ep == EP_RAW_STATS ? " +bnstats" : a.ep == EP_BN_FUSED ? " +bnfused" : "");
  if (ts.skip()) return 0;
  const int rc = dt == DT_BF16 ? launch_kw<BF16>(b, pl, KW, up2, nwg, s) : launch_kw<F16>(b, pl, KW, up2, nwg, s);
  if (rc) return rc;
  return check_launch("conv16_kernel");
}

}  // namespace ms
