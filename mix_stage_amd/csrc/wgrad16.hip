// 16-bit arithmetic mode: weight gradient on v_mfma_f32_32x32x16_{bf16,f16}.
//
//   dw[co][ci][kh][kw] = sum over pixels p of dy[co][p] * x[ci][p*S + (kh, kw) - pad]
//
// is a GEMM whose reduction index is the PIXEL, while the cb8 layout keeps 8 CHANNELS of one pixel together.  The MFMA
// operands (8 consecutive reduction indices per lane = 8 pixels of ONE channel) are therefore transposed on their way out of
// LDS by ds_read_b64_tr_b16: per group of 16 lanes it reads 4 rows (pixels) x 16 columns (channels, two cb8 vectors) and
// hands lane i column i -- four consecutive pixels of its channel.  Two such reads make one MFMA operand; no data is
// re-laid out anywhere, every lane supplies its own row address, so tap and stride offsets are free.
//
// A workgroup (4 waves, 2 x 2) owns 64 output channels x 64 input channels x TP taps (kw0 .. kw0+TP-1 of one kernel row) of
// one group and a range of 64-pixel tiles; per tile it stages dy [8 blocks][64 px] and the input rows
// [8 blocks][TH][(TW-1)*S + TP] as 16-byte vectors.  Pixel splits leave fp32 partial slabs that are summed in a fixed order
// by the caller (ms_wgrad_reduce_multi), like the fp32 kernels'.
#include <algorithm>
#include <mutex>
#include <vector>

#include "conv16_kernel.h"

namespace ms {

typedef short s16x4 __attribute__((ext_vector_type(4)));

// workgroups a layer's launch aims for through pixel splits (ms_debug_set_wgrad16_target).  The trainer launches the layers'
// kernels side by side (ms_wgrad_flush), so a layer need not fill the chip alone; measured on the train step: 512 -> 3.17 ms,
// 256 -> 3.12, 128 -> 3.06, 64 -> 3.15 per G-step (fewer, longer workgroups leave fewer partial slabs to store and reduce)
int g_wgrad16_target_wgs = 128;
extern int g_conv16_dbg;

// LDS pitches (16-byte vectors) of a channel block's pixels.  The transposed reads of one MFMA operand touch FOUR channel blocks
// at once (lanes 0-31 of ds_read_b64_tr_b16: 4 rows x 16 banks each), so the block pitch must be = 16 banks (4 vectors) mod 64
// banks: the natural pitches -- 64 vectors for dy, TH*PCX = 66 for a k3 row -- put the four blocks on the same / overlapping
// banks (SQ_LDS_BANK_CONFLICT was 40 % of SQ_LDS_IDX_ACTIVE on the decoder layer).
constexpr int WG16_DP = 68;   // dy: 64 pixels + 4
__host__ __device__ inline int wg16_xpitch(int thpcx) { return ((thpcx + 11) & ~15) + 4; }   // smallest pitch >= thpcx that is 4 mod 16
constexpr int WG16_DYV = 8 * WG16_DP;    // vectors of the dy part of a stage
constexpr int WG16_NPX = 5;   // register-staged form (upsample-add input): input-row vectors a thread stages per tile at most

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
// one transposed 4-row read at a byte address (constant byte offsets fold into the instruction's offset field)
__device__ __forceinline__ unsigned long long tr_read_b64(const char* ptr) {
  // (whole-value bit cast: element access on a builtin's vector result is miscompiled by this clang)
  return __builtin_bit_cast(unsigned long long, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)ptr));
}
__device__ __forceinline__ u32x4 tr_pair(const char* p0, const char* p1) {
  const unsigned long long lo = tr_read_b64(p0), hi = tr_read_b64(p1);
  u32x4 v;
  v[0] = (unsigned)lo; v[1] = (unsigned)(lo >> 32); v[2] = (unsigned)hi; v[3] = (unsigned)(hi >> 32);
  return v;
}

// The kernel was INSTRUCTION-bound, not memory- or MFMA-bound (SQ counters on the decoder layer: 11 VALU and 11 SALU
// instructions per MFMA, matrix pipe busy 12 % of the wave cycles): every operand read re-derived its LDS address from the tile
// geometry, every tile re-derived eight staging slots (five of them unused for k3 rows) and two scalar divisions.  Now:
//  * LDS operand addresses are per-lane constants formed once (9 registers); per tile they get the buffer base added (9 VALU)
//    and every read uses an immediate offset (k-step, tap, second row group);
//  * staging is LDS-DMA (buffer_load ... lds: no staging registers, no ds_write) from offsets that are tile-invariant per slot:
//    per tile and slot one add + the bounds test; NPXT = slots a thread owns (3 / 5 in registers; 0 = any number, re-derived per
//    tile: only the layers with a handful of pixels per row need more than 5);
//  * the tile coordinates advance incrementally.
// UP2 (x = up2(a) + r formed in registers) keeps global -> registers -> LDS staging with two buffers.
template <typename DT, int TP, bool UP2, int NPXT>
__device__ __forceinline__ void wgrad16_body(const Wgrad16Args& p, const int bid, u32x4* smem) {
  constexpr bool DMA = !UP2;
  constexpr int NREG = UP2 ? WG16_NPX : (NPXT > 0 ? NPXT : 1);
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wm = wid >> 1, wn = wid & 1, h = lane >> 5;
  const int TW = 1 << p.ltw, TH = p.TH, PCX = p.PCX, S = p.S, SV = p.SV;
  const int XP = wg16_xpitch(TH * PCX);        // padded pitch of a channel block's input rows
  const int xv = 8 * XP;                       // input-row vectors per tile (pad vectors are never loaded and never read)
  const int xv64 = (xv + 63) & ~63;            // DMA: a wave skips its 64-vector share of a slab that lies beyond the rows
  const int stage_vecs = DMA ? WG16_DYV + xv64 : WG16_DYV + xv + 1;

  const int vid = xcd_remap(bid, p.gx * p.gy * p.gz);
  const int bx_ = vid % p.gx, by_ = (vid / p.gx) % p.gy, bz_ = vid / (p.gx * p.gy);
  // z = ((group * KH + kh) * ktg + tap group) * splits + split
  const int split = bz_ % p.splits;
  int zz = bz_ / p.splits;
  const int tgi = zz % p.ktg; zz /= p.ktg;
  const int kh = zz % p.KH;
  const int g = zz / p.KH;
  const int kw0 = tgi * TP;
  const int co0 = by_ * 64, ci0 = bx_ * 64;     // channel offsets inside the group (multiples of 8)
  const int cog8 = (p.Cog + 7) >> 3, cig8 = (p.Cig + 7) >> 3;
  const int dy_cb0 = ((g * p.Cog) >> 3) + (co0 >> 3);
  const int x_cb0 = (p.bcast ? 0 : ((g * p.Cig) >> 3)) + (ci0 >> 3);
  const int tile_beg = split * p.tiles_per_split, tile_end = min(p.n_tiles, tile_beg + p.tiles_per_split);
  const int tiles_per_img = p.tiles_y * p.tiles_x;

  const __amdgpu_buffer_rsrc_t rsD = buf_rsrc(p.dyr), rsS = buf_rsrc(p.src), rsS2 = buf_rsrc(UP2 ? p.src2 : p.src);

  // ---- tile-invariant parts of the staging offsets
  // dy: vector e = cb*64 + n, n = ty*TW + tx; element offset inside the tensor relative to the tile's first pixel
  int d_ty[2], d_tx[2], d_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = t + i * 256, cb = e >> 6, n = e & 63;
    d_ty[i] = n >> p.ltw; d_tx[i] = n & (TW - 1);
    if ((co0 >> 3) + cb >= cog8) d_ty[i] = 1 << 28;                        // a channel block beyond the group: never in range
    d_off[i] = cb * p.o_cblk + d_ty[i] * p.o_row + d_tx[i];
  }
  const unsigned mg_xp = 0xFFFFFFFFu / (unsigned)XP + 1u, mg_pcx = 0xFFFFFFFFu / (unsigned)PCX + 1u;   // e < 2^16, small divisors: exact
  // input-row vector e -> (channel block, tile row, column); false for pad vectors and channel blocks beyond the group
  auto x_decode = [&](int e, int& cb, int& ty, int& c) -> bool {
    cb = (int)__umulhi((unsigned)e, mg_xp);
    const int rem = e - cb * XP;
    ty = PCX == 1 ? rem : (int)__umulhi((unsigned)rem, mg_pcx);
    c = rem - ty * PCX;
    return (e < xv) & (rem < TH * PCX) & ((ci0 >> 3) + cb < cig8);
  };
  int x_r[NREG], x_c[NREG], x_off[NREG], x_lds[UP2 ? NREG : 1];
#pragma unroll
  for (int i = 0; i < NREG; ++i) {
    int cb, ty, c;
    const int e = t + i * 256;
    const bool ok = x_decode(e, cb, ty, c);
    x_r[i] = ok ? ty * SV : (1 << 28);                                       // (a slot that does not exist: never in range)
    x_c[i] = c;
    x_off[i] = cb * p.s_cblk + ty * SV * p.s_row + c;
    if (UP2) x_lds[UP2 ? i : 0] = ok ? WG16_DYV + e : WG16_DYV + xv;
  }

  // tile coordinates of the tile to be staged next, advanced incrementally (wave-uniform)
  int s_img = tile_beg / tiles_per_img, s_ty, s_tx;
  {
    const int trem = tile_beg - s_img * tiles_per_img;
    s_ty = trem / p.tiles_x; s_tx = trem - s_ty * p.tiles_x;
  }
  auto advance = [&]() {
    if (++s_tx == p.tiles_x) { s_tx = 0; if (++s_ty == p.tiles_y) { s_ty = 0; ++s_img; } }
  };

  u32x4 rd[DMA ? 1 : 2], rx[DMA ? 1 : NREG], rx2[UP2 ? NREG : 1];
  auto load_tile = [&]() {          // register-staged form
    if constexpr (!DMA) {
      const int oy0 = s_ty * TH, ox0 = s_tx << p.ltw;
      const int dbase = __builtin_amdgcn_readfirstlane(s_img * p.o_img + dy_cb0 * p.o_cblk + oy0 * p.o_row + ox0);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bool ok = (d_ty[i] < p.OUTH - oy0) & (d_tx[i] < p.OUTW - ox0);
        rd[i] = buf_load_v(rsD, ok ? 16u * (unsigned)d_off[i] : BUF_OOB, 16u * (unsigned)dbase);
      }
      const int iy0 = oy0 * SV - p.PH + kh, ix0 = ox0 * S - p.PW + kw0;
      const int sbase = __builtin_amdgcn_readfirstlane(s_img * p.s_img + x_cb0 * p.s_cblk);
      const int toff = iy0 * p.s_row + ix0;
#pragma unroll
      for (int i = 0; i < NREG; ++i) {
        const bool ok = ((unsigned)(iy0 + x_r[i]) < (unsigned)p.SRCH) & ((unsigned)(ix0 + x_c[i]) < (unsigned)p.SRCW);
        const int o = x_off[i] + toff, ix = ix0 + x_c[i];
        // x = nearest_up2(a) + r: a has half the row length, hence half of every stride (all strides are even)
        rx[i] = buf_load_v(rsS, ok ? 16u * (unsigned)(((o - ix) >> 1) + (ix >> 1)) : BUF_OOB, 16u * (unsigned)(sbase >> 1));
        rx2[UP2 ? i : 0] = buf_load_v(rsS2, ok ? 16u * (unsigned)o : BUF_OOB, 16u * (unsigned)sbase);
      }
      advance();
    }
  };
  auto store_tile = [&](int buf) {
    if constexpr (!DMA) {
      u32x4* st = smem + buf * stage_vecs;
#pragma unroll
      for (int i = 0; i < 2; ++i) st[((t + i * 256) >> 6) * WG16_DP + (t & 63)] = rd[i];
#pragma unroll
      for (int i = 0; i < NREG; ++i) {
        float fa[8], fr[8];
        unpack8<DT>(rx[i], fa);
        unpack8<DT>(rx2[UP2 ? i : 0], fr);
#pragma unroll
        for (int j = 0; j < 8; ++j) fa[j] += fr[j];
        st[x_lds[UP2 ? i : 0]] = pack8<DT>(fa);
      }
    }
  };

  f32x16 acc[TP];
#pragma unroll
  for (int q = 0; q < TP; ++q)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;

  // ---- transposed-read lane roles: group of 16 lanes = 16 channels (2 blocks); lane 4q+pp addresses row q, 4 channels pp.
  // Byte offsets inside a stage, formed once: the dy operand of k-step kk sits 256 bytes per kk further (16 pixels), its second
  // row group 64 bytes further; the input operand's two row groups per k-step (bx0, bx1) depend on the tile shape and stride.
  const int li = lane & 15, q4 = li >> 2, pp = li & 3, half16 = (lane >> 4) & 1;
  const int sub8 = (pp & 1) * 8;
  const int a_blk = wm * 4 + half16 * 2 + (pp >> 1);     // dy block inside the 64-channel tile
  const int b_blk = wn * 4 + half16 * 2 + (pp >> 1);     // x block inside the 64-channel tile
  const int a_off = (a_blk * WG16_DP + 8 * h + q4) * 16 + sub8;
  int bx0[4], bx1[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int p0 = 16 * kk + 8 * h + q4, p1 = p0 + 4;           // this lane's rows (pixels) of the two 4-row reads
    bx0[kk] = (WG16_DYV + b_blk * XP + (p0 >> p.ltw) * PCX + (p0 & (TW - 1)) * S) * 16 + sub8;
    bx1[kk] = (WG16_DYV + b_blk * XP + (p1 >> p.ltw) * PCX + (p1 & (TW - 1)) * S) * 16 + sub8;
  }
  const int stage_bytes = stage_vecs * 16;

  // One tile = 4 k-steps of 16 pixels, TP MFMAs each; the operand reads of k-step kk+1 are requested before the MFMAs of k-step kk.
  auto compute_tile = [&](int cur) {
    const char* stb = reinterpret_cast<const char*>(smem) + cur * stage_bytes;
    const char* ap = stb + a_off;
    const char* b0[4];
    const char* b1[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { b0[kk] = stb + bx0[kk]; b1[kk] = stb + bx1[kk]; }
    u32x4 av[2], bv[2][TP];
    av[0] = tr_pair(ap, ap + 64);
#pragma unroll
    for (int q = 0; q < TP; ++q) bv[0][q] = tr_pair(b0[0] + 16 * q, b1[0] + 16 * q);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (kk + 1 < 4) {
        av[(kk + 1) & 1] = tr_pair(ap + 256 * (kk + 1), ap + 256 * (kk + 1) + 64);
#pragma unroll
        for (int q = 0; q < TP; ++q) bv[(kk + 1) & 1][q] = tr_pair(b0[(kk + 1) & 3] + 16 * q, b1[(kk + 1) & 3] + 16 * q);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < TP; ++q) acc[q] = DT::mfma(av[kk & 1], bv[kk & 1][q], acc[q]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  if constexpr (DMA) {
    const int wave = __builtin_amdgcn_readfirstlane(wid);
    auto issue_tile = [&](int buf) {
      const int oy0 = s_ty * TH, ox0 = s_tx << p.ltw;
      u32x4* dst = smem + buf * stage_vecs;
      const int dbase = __builtin_amdgcn_readfirstlane(s_img * p.o_img + dy_cb0 * p.o_cblk + oy0 * p.o_row + ox0);
      const int oh_left = p.OUTH - oy0, ow_left = p.OUTW - ox0;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bool ok = (d_ty[i] < oh_left) & (d_tx[i] < ow_left);
        // (a wave's 64 lanes = the 64 pixels of channel block wave + 4*i: the destination is that block's padded row)
        dma16(rsD, dst + (wave + 4 * i) * WG16_DP, ok ? 16u * (unsigned)d_off[i] : BUF_OOB, 16u * (unsigned)dbase);
      }
      const int iy0 = oy0 * SV - p.PH + kh, ix0 = ox0 * S - p.PW + kw0;
      const int sbase = __builtin_amdgcn_readfirstlane(s_img * p.s_img + x_cb0 * p.s_cblk);
      const int toff = iy0 * p.s_row + ix0;                          // (may be negative: it goes into the per-lane offset)
      if constexpr (NPXT > 0) {
#pragma unroll
        for (int i = 0; i < NPXT; ++i) {
          if (i * 256 + wave * 64 < xv) {                            // (wave-uniform: a slab beyond the rows is not issued)
            const bool ok = ((unsigned)(iy0 + x_r[i]) < (unsigned)p.SRCH) & ((unsigned)(ix0 + x_c[i]) < (unsigned)p.SRCW);
            dma16(rsS, dst + WG16_DYV + wave * 64 + i * 256, ok ? 16u * (unsigned)(x_off[i] + toff) : BUF_OOB, 16u * (unsigned)sbase);
          }
        }
      } else {
        for (int e0 = wave * 64; e0 < xv; e0 += 256) {                // any number of slots, re-derived per tile
          int cb, ty, c;
          const bool okc = x_decode(e0 + lane, cb, ty, c);
          const int iy = iy0 + ty * SV, ix = ix0 + c;
          const bool ok = okc & ((unsigned)iy < (unsigned)p.SRCH) & ((unsigned)ix < (unsigned)p.SRCW);
          dma16(rsS, dst + WG16_DYV + e0, ok ? 16u * (unsigned)(cb * p.s_cblk + iy * p.s_row + ix) : BUF_OOB, 16u * (unsigned)sbase);
        }
      }
      advance();
    };
    const int ntl = (p.dbg & 4) ? 0 : tile_end - tile_beg;
    const int per_tile = 2 + max(0, (xv - wave * 64 + 255) >> 8);   // LDS-DMA instructions THIS wave issues per tile
    const int ahead = p.nstg - 2;                      // tiles that stay in flight behind the one being multiplied
    if (p.nstg == 2) {
      // two buffers (the default): the wait is a literal vmcnt(0) -- the run-time count of the general form below is a 64-way
      // switch INSIDE the loop, around which the register allocator moved all accumulators between the AGPR and VGPR files every
      // tile (48 v_accvgpr moves per 12 MFMAs)
      if (ntl > 0) issue_tile(0);
      for (int i = 0; i < ntl; ++i) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // everyone's part of tile i is in LDS, everyone is done with tile i-1
        asm volatile("" ::: "memory");
        if (i + 1 < ntl) issue_tile((i + 1) & 1);
        if (!(p.dbg & 2)) compute_tile(i & 1);
      }
    } else {
      for (int i = 0; i < p.nstg - 1 && i < ntl; ++i) issue_tile(i);
      for (int i = 0; i < ntl; ++i) {
        wait_vmcnt(i + ahead < ntl ? ahead * per_tile : 0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (i + p.nstg - 1 < ntl) issue_tile((i + p.nstg - 1) % p.nstg);    // refills the buffer of tile i-1
        if (!(p.dbg & 2)) compute_tile(i % p.nstg);
      }
    }
  } else if (tile_beg < tile_end) {
    load_tile();
    store_tile(0);
    __syncthreads();
    for (int tile = tile_beg; tile < tile_end; ++tile) {
      const int cur = (tile - tile_beg) & 1;
      if (tile + 1 < tile_end) load_tile();
      compute_tile(cur);
      if (tile + 1 < tile_end) store_tile(cur ^ 1);
      __syncthreads();
    }
  }

  // ---- store: D[row = co][col = ci] per tap; dw is (groups*Cog, Cig, KH, KW) fp32.  For one (co, ci) the TP taps of this
  // workgroup are adjacent in memory, and a lane holds all of them (same accumulator register of the TP accumulators): one
  // TP-dword store per (lane, register); consecutive lanes = consecutive ci, KH*KW floats apart (1-D k3: a contiguous run)
  const int r = lane & 31;
  float* out = p.out + (size_t)split * p.out_split_stride;
  const int ci = ci0 + wn * 32 + r;
  struct __attribute__((packed, aligned(4))) Taps { float v[TP]; };
  if (p.dbg & 1) return;
  // accumulating form (queued launches that write dw themselves): all 16 read-modify-writes read first, then write
  Taps prev[16];
  if (p.accumulate && kw0 + TP <= p.KW) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int co = min(co0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h, p.Cog - 1);
      prev[e] = *reinterpret_cast<const Taps*>(out + (((size_t)(g * p.Cog + co) * p.Cig + min(ci, p.Cig - 1)) * p.KH + kh) * p.KW + kw0);
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int co = co0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
    if (co < p.Cog && ci < p.Cig) {
      float* dst = out + (((size_t)(g * p.Cog + co) * p.Cig + ci) * p.KH + kh) * p.KW + kw0;
      if (kw0 + TP <= p.KW) {
        Taps tv;
#pragma unroll
        for (int q = 0; q < TP; ++q) tv.v[q] = p.accumulate ? prev[e].v[q] + acc[q][e] : acc[q][e];
        *reinterpret_cast<Taps*>(dst) = tv;
      } else {
#pragma unroll
        for (int q = 0; q < TP; ++q)
          if (kw0 + q < p.KW) dst[q] = p.accumulate ? dst[q] + acc[q][e] : acc[q][e];
      }
    }
  }
}

template <typename DT, int TP, bool UP2, int NPXT>
__global__ __launch_bounds__(256) void wgrad16_kernel(const Wgrad16Args p) {
  prefetch_kernargs<sizeof(Wgrad16Args)>();
  extern __shared__ u32x4 smem[];
  wgrad16_body<DT, TP, UP2, NPXT>(p, (int)blockIdx.x, smem);
}

// many blocks' weight gradients in one launch: a workgroup finds its job in the table of block ranges
template <typename DT, int TP, bool UP2, int NPXT>
__global__ __launch_bounds__(256) void wgrad16_multi_kernel(const Wgrad16Batch b) {
  extern __shared__ u32x4 smem[];
  prefetch_kernargs<128>();                                   // n and the table of block ranges
  int j = 0;
  while (j + 1 < b.n && (int)blockIdx.x >= b.block_end[j]) ++j;
  const int b0 = j ? b.block_end[j - 1] : 0;
  prefetch_kernargs<sizeof(Wgrad16Args)>((int)offsetof(Wgrad16Batch, job) + j * (int)sizeof(Wgrad16Args));
  wgrad16_body<DT, TP, UP2, NPXT>(b.job[j], (int)blockIdx.x - b0, smem);
}

// ---------------------------------------------------------------------------------------------
// Single-input-channel blocks (the first audio-encoder layer: 1 -> 64 channels, 3x3): dw[co][0][kh][kw] is a handful of numbers
// per output channel, reduced over every pixel -- on the MFMA kernel above a 64 x 64 x taps tile whose input-channel side is 63/64
// zeros (2.7 TF, and a quarter of all tile iterations of the backward pass's weight-gradient launch).  Here it is what it is, a
// stream over dy on the vector unit: a workgroup owns one channel block (8 output channels) and one pixel split; a thread takes one
// pixel per step, loads its dy vector and the KH*KW input values around it (channel 0 of the cb8 input: neighbouring lanes
// read neighbouring pixels, served by L1 / L2) and keeps 8 x KH*KW running sums.  Fixed-order reduction: lanes (DPP), the 4
// waves through LDS, the splits by the caller's slab reduction like every other weight gradient.
constexpr int WGC1_MAX_TAPS = 9;

template <typename DT>
__global__ __launch_bounds__(256) void wgrad16_c1_kernel(const Wgrad16Args p) {
  prefetch_kernargs<sizeof(Wgrad16Args)>();
  __shared__ float red[4][8 * WGC1_MAX_TAPS];
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int cb = blockIdx.x % p.gy, split = blockIdx.x / p.gy;          // gy = output channel blocks
  const int TW = 1 << p.ltw, taps = p.KH * p.KW;
  const int tile_beg = split * p.tiles_per_split, tile_end = min(p.n_tiles, tile_beg + p.tiles_per_split);
  const int tiles_per_img = p.tiles_y * p.tiles_x;
  const u32x4* dy = reinterpret_cast<const u32x4*>(p.dyr);
  const unsigned short* xs = reinterpret_cast<const unsigned short*>(p.src);
  float acc[8][WGC1_MAX_TAPS];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int q = 0; q < WGC1_MAX_TAPS; ++q) acc[j][q] = 0.f;
  // 16 tiles of 64 pixels per step (4 per wave): every load of the step is issued before the first use (clamped addresses, masked
  // values) -- one memory round trip per step instead of one per tile
  constexpr int U = 4;
  for (int tile0 = tile_beg; tile0 < tile_end; tile0 += 4 * U) {
    u32x4 gv[U];
    unsigned short raw[U][WGC1_MAX_TAPS];
    bool okv[U][WGC1_MAX_TAPS];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int tile = tile0 + wid * U + u;
      const int tl = min(tile, p.n_tiles - 1);
      const int img = tl / tiles_per_img, trem = tl - img * tiles_per_img;
      const int tyi = trem / p.tiles_x, txi = trem - tyi * p.tiles_x;
      const int oy = tyi * p.TH + (lane >> p.ltw), ox = (txi << p.ltw) + (lane & (TW - 1));
      const bool ok = (tile < tile_end) & (oy < p.OUTH) & (ox < p.OUTW);
      const int oyc = min(oy, p.OUTH - 1), oxc = min(ox, p.OUTW - 1);
      gv[u] = dy[(size_t)img * p.o_img + (size_t)cb * p.o_cblk + (size_t)oyc * p.o_row + oxc];
      if (!ok) gv[u] = u32x4{0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < WGC1_MAX_TAPS; ++q) {
        const int kh = q / p.KW, kw = q - kh * p.KW;
        const int iy = oy * p.SV - p.PH + kh, ix = ox * p.S - p.PW + kw;
        okv[u][q] = ok & (q < taps) & ((unsigned)iy < (unsigned)p.SRCH) & ((unsigned)ix < (unsigned)p.SRCW);
        const int iyc = min(max(iy, 0), p.SRCH - 1), ixc = min(max(ix, 0), p.SRCW - 1);
        raw[u][q] = xs[((size_t)img * p.s_img + (size_t)iyc * p.s_row + ixc) * 8];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float g[8];
      unpack8<DT>(gv[u], g);
#pragma unroll
      for (int q = 0; q < WGC1_MAX_TAPS; ++q) {
        const float xq = okv[u][q] ? DT::lo((unsigned)raw[u][q]) : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j][q] = fmaf(g[j], xq, acc[j][q]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int q = 0; q < WGC1_MAX_TAPS; ++q) {
      const float v = wave_sum(acc[j][q]);
      if (lane == 0) red[wid][j * WGC1_MAX_TAPS + q] = v;
    }
  __syncthreads();
  if (t < 8 * WGC1_MAX_TAPS) {
    const int j = t / WGC1_MAX_TAPS, q = t - j * WGC1_MAX_TAPS;
    const int co = cb * 8 + j;
    if (co < p.Cog && q < taps) {
      const float v = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
      float* out = p.out + (size_t)split * p.out_split_stride + (size_t)co * taps + q;
      *out = p.accumulate ? *out + v : v;
    }
  }
}

// 3x3, stride 1 (the layer this exists for).  A workgroup owns a run of consecutive rows of the flattened (image, row) space
// (one pixel split) x 64 output columns; a lane is one column, the 4 waves take two channel blocks each.  Every wave slides a
// 3 x 3 window of input values down the rows (three new values per row; the four waves read the same input -- L1 hits) and
// loads its two dy vectors per row; U rows per step with all loads of the step issued first.  dy is read exactly once from HBM
// and the input once per workgroup: the generic form above re-read the input per channel block (9x its size beyond L2) and sat
// at 69 % of its wave cycles waiting on memory.
template <typename DT>
__global__ __launch_bounds__(256) void wgrad16_c1_3x3_kernel(const Wgrad16Args p, int rows_per_split, int col_tiles) {
  prefetch_kernargs<sizeof(Wgrad16Args) + 8>();
  __shared__ float red[144];
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int ct = blockIdx.x % col_tiles, split = blockIdx.x / col_tiles;
  const int ox = ct * 64 + lane;
  const int total_rows = p.n_tiles;                               // (reused field: images x output rows)
  const int r0 = split * rows_per_split, r1 = min(total_rows, r0 + rows_per_split);
  const u32x4* dy = reinterpret_cast<const u32x4*>(p.dyr);
  const unsigned short* xs = reinterpret_cast<const unsigned short*>(p.src);
  const bool col_ok = ox < p.OUTW;
  const int oxc = min(ox, p.OUTW - 1);
  const int ncb = p.gy;                                           // output channel blocks
  // the three input columns of this lane: ix = ox - PW + kw; out of range -> 0 (padding)
  bool cok[3];
  int cix[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int ix = ox - p.PW + kw;
    cok[kw] = col_ok & ((unsigned)ix < (unsigned)p.SRCW);
    cix[kw] = min(max(ix, 0), p.SRCW - 1) * 8;
  }
  auto load_row = [&](int img, int iy, float (&v)[3]) {
    const bool rok = (unsigned)iy < (unsigned)p.SRCH;
    const unsigned short* row = xs + ((size_t)img * p.s_img + (size_t)min(max(iy, 0), p.SRCH - 1) * p.s_row) * 8;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const unsigned short raw = row[cix[kw]];
      v[kw] = (rok & cok[kw]) ? DT::lo((unsigned)raw) : 0.f;
    }
  };
  for (int cbp = wid; 2 * cbp < ncb; cbp += 4) {                   // this wave's pairs of channel blocks (one pair for 64 channels)
    const int cb0 = 2 * cbp, cb1 = min(2 * cbp + 1, ncb - 1);
    float acc[2][8][9];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int q = 0; q < 9; ++q) acc[c][j][q] = 0.f;
    constexpr int U = 4;
    for (int rb = r0; rb < r1; rb += U) {
      u32x4 gv[U][2];
      float xin[U + 2][3];
      int img_u[U], oy_u[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int r = min(rb + u, r1 - 1);
        img_u[u] = r / p.OUTH; oy_u[u] = r - img_u[u] * p.OUTH;
        const size_t base = (size_t)img_u[u] * p.o_img + (size_t)oy_u[u] * p.o_row + oxc;
        gv[u][0] = dy[base + (size_t)cb0 * p.o_cblk];
        gv[u][1] = dy[base + (size_t)cb1 * p.o_cblk];
      }
      // rows of one image are consecutive: input rows oy_0 - PH .. + U + 1 serve all U output rows unless an image boundary falls
      // inside the step (then each row loads its own three input rows: once per image)
      const bool same = img_u[U - 1] == img_u[0] && oy_u[U - 1] == oy_u[0] + U - 1;
      if (same) {
#pragma unroll
        for (int k = 0; k < U + 2; ++k) load_row(img_u[0], oy_u[0] - p.PH + k, xin[k]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (rb + u < r1) {
          float w3[3][3];
          if (same) {
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
              for (int kw = 0; kw < 3; ++kw) w3[kh][kw] = xin[u + kh][kw];
          } else {
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) load_row(img_u[u], oy_u[u] - p.PH + kh, w3[kh]);
          }
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            float g[8];
            unpack8<DT>(col_ok ? gv[u][c] : u32x4{0, 0, 0, 0}, g);
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
              for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) acc[c][j][kh * 3 + kw] = fmaf(g[j], w3[kh][kw], acc[c][j][kh * 3 + kw]);
          }
        }
      }
    }
    // lanes -> one value per (channel, tap); the column tiles meet in the slab through the caller's fixed-order reduction: slab
    // index = split * col_tiles + ct
    float* out = p.out + (size_t)(split * col_tiles + ct) * p.out_split_stride;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int q = 0; q < 9; ++q) {
          const float v = wave_sum(acc[c][j][q]);
          const int co = (2 * cbp + c) * 8 + j;
          if (lane == 0 && co < p.Cog && (c == 0 || 2 * cbp + 1 < ncb)) out[(size_t)co * 9 + q] = p.accumulate ? out[(size_t)co * 9 + q] + v : v;
        }
  }
  (void)red;
}

bool wgrad16_c1_ok(const Wgrad16Args& a, bool up2) {
  return !up2 && a.Cig == 1 && a.groups == 1 && a.KH * a.KW <= WGC1_MAX_TAPS;
}

// ---------------------------------------------------------------------------------------------
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

int g_wgrad16_ring = 2;        // LDS-DMA ring depth (ms_debug_set_wgrad16_ring)

Wgrad16Plan plan_wgrad16(int nd, int Cog, int Cig, int groups, int KH, int KW, int SH, int SW, int B, int OH, int OW, bool up2) {
  Wgrad16Plan pl = {};
  const int rows = nd == 1 ? B : OH, imgs = nd == 1 ? 1 : B;
  pl.tp = KW == 1 ? 1 : KW == 3 ? 3 : (KW % 4 == 0) ? 4 : (KW == 2 ? 2 : 0);
  if (!pl.tp) return pl;
  pl.ktg = KW / pl.tp;
  pl.tw = std::min(pow2_at_least(OW), 64);
  pl.th = 64 / pl.tw;
  pl.pcx = (pl.tw - 1) * SW + pl.tp;
  const int xp = wg16_xpitch(pl.th * pl.pcx), xv = 8 * xp;
  pl.npx = cdiv(xv, 256);                              // 256-vector slabs of input rows per tile
  if (up2 && pl.npx > WG16_NPX) return (pl.tp = 0, pl);
  if (xv > 65535) return (pl.tp = 0, pl);
  pl.tiles_y = cdiv(rows, pl.th); pl.tiles_x = cdiv(OW, pl.tw);
  pl.n_tiles = imgs * pl.tiles_y * pl.tiles_x;
  // pixel splits: fill ~2 workgroups per CU, at least 2 tiles per workgroup
  const long base = (long)cdiv(Cog, 64) * cdiv(Cig, 64) * groups * KH * pl.ktg;
  int splits = (int)std::max<long>(1, std::min<long>((g_wgrad16_target_wgs + base - 1) / base, pl.n_tiles / 2));
  splits = std::min(splits, 64);
  pl.tiles_per_split = cdiv(pl.n_tiles, std::max(1, splits));
  pl.splits = cdiv(pl.n_tiles, pl.tiles_per_split);
  if (Cig == 1 && groups == 1 && KH == 3 && KW == 3 && SW == 1 && !up2) {
    // single-input-channel 3x3 block (wgrad16_c1_3x3_kernel: a 33 MB stream over dy at the headline size): slabs = row splits x
    // 64-column tiles, one 4-wave workgroup each -- enough of them (256) to keep the stream in flight; a slab is Cog x 9 floats
    const int col_tiles = cdiv(OW, 64), total_rows = imgs * rows;
    const int row_splits = std::max(1, std::min(256 / col_tiles, total_rows / 8));
    const int rps = cdiv(total_rows, row_splits);
    pl.splits = cdiv(total_rows, rps) * col_tiles;
    pl.tiles_per_split = cdiv(pl.n_tiles, pl.splits);            // (the MFMA kernel is not used for this block)
  }
  // register-staged form (upsample-add input): two buffers.  LDS-DMA form: a ring of g_wgrad16_ring buffers of whole 64-vector
  // wave slabs (17 KB for a k3 row: two buffers leave room for four workgroups per CU)
  const int stage = (WG16_DYV + cdiv(xv, 64) * 64) * 16;
  pl.nstg = up2 ? 2 : std::max(2, std::min(std::min(g_wgrad16_ring, pl.tiles_per_split + 1), 160 * 1024 / stage));
  pl.lds_bytes = up2 ? 2 * (WG16_DYV + xv + 1) * 16 : pl.nstg * stage;
  if (pl.lds_bytes > 160 * 1024) return (pl.tp = 0, pl);
  (void)SH;
  return pl;
}

template <typename DT, int TP, bool UP2, int NPXT>
static int launch_w(const Wgrad16Args& a, int lds, int nwg, hipStream_t s) {
  static bool attr_done = false;
  auto fn = wgrad16_kernel<DT, TP, UP2, NPXT>;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return set_error("wgrad16: cannot raise the dynamic LDS limit");
    attr_done = true;
  }
  hipLaunchKernelGGL(fn, dim3(nwg), dim3(256), lds, s, a);
  return 0;
}

template <typename DT, int TP, bool UP2, int NPXT>
static int launch_wm(const Wgrad16Batch& b, int lds, hipStream_t s) {
  static bool attr_done = false;
  auto fn = wgrad16_multi_kernel<DT, TP, UP2, NPXT>;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return set_error("wgrad16: cannot raise the dynamic LDS limit");
    attr_done = true;
  }
  hipLaunchKernelGGL(fn, dim3(b.block_end[b.n - 1]), dim3(256), lds, s, b);
  return 0;
}

// npxt: slots of input rows a thread keeps in registers (3, 5) or 0 = the per-tile loop (layers with a few pixels per row).
// Queued jobs (wgrad16_flush) all take the 5-slot instance up to 5 slots: a slot beyond a job's rows is a wave-uniform skip, and
// the jobs of a backward pass then share ONE launch per number of taps -- with 3- and 5-slot instances apart, a single
// 192-workgroup job ran alone on a corner of the chip for 55 us.  (A universal kernel holding every instance behind a switch was
// tried: the register allocator gives the cases disjoint accumulator ranges -- 168 VGPR + 128 AGPR, 3.7 KB of scratch.)
inline int wg16_npxt(int npx) { return npx <= 3 ? 3 : npx <= 5 ? 5 : 0; }
inline int wg16_npxt_queued(int npx) { return npx <= 5 ? 5 : 0; }

template <typename DT, int TP>
static int launch_np_multi(const Wgrad16Batch& b, int npxt, int lds, hipStream_t s) {
  if (npxt == 3) return launch_wm<DT, TP, false, 3>(b, lds, s);
  if (npxt == 5) return launch_wm<DT, TP, false, 5>(b, lds, s);
  return launch_wm<DT, TP, false, 0>(b, lds, s);
}
template <typename DT, int TP>
static int launch_np(const Wgrad16Args& a, int npxt, int lds, int nwg, hipStream_t s) {
  if (npxt == 3) return launch_w<DT, TP, false, 3>(a, lds, nwg, s);
  if (npxt == 5) return launch_w<DT, TP, false, 5>(a, lds, nwg, s);
  return launch_w<DT, TP, false, 0>(a, lds, nwg, s);
}

template <typename DT>
static int launch_tp_multi(const Wgrad16Batch& b, int tp, bool up2, int npxt, int lds, hipStream_t s) {
  if (up2) return launch_wm<DT, 3, true, 0>(b, lds, s);
  switch (tp) {
    case 1: return launch_np_multi<DT, 1>(b, npxt, lds, s);
    case 2: return launch_np_multi<DT, 2>(b, npxt, lds, s);
    case 3: return launch_np_multi<DT, 3>(b, npxt, lds, s);
    case 4: return launch_np_multi<DT, 4>(b, npxt, lds, s);
  }
  return set_error("wgrad16: no kernel for %d taps", tp);
}

template <typename DT>
static int launch_tp(const Wgrad16Args& a, const Wgrad16Plan& pl, bool up2, int nwg, hipStream_t s) {
  if (up2) return launch_w<DT, 3, true, 0>(a, pl.lds_bytes, nwg, s);
  const int npxt = wg16_npxt(pl.npx);
  switch (pl.tp) {
    case 1: return launch_np<DT, 1>(a, npxt, pl.lds_bytes, nwg, s);
    case 2: return launch_np<DT, 2>(a, npxt, pl.lds_bytes, nwg, s);
    case 3: return launch_np<DT, 3>(a, npxt, pl.lds_bytes, nwg, s);
    case 4: return launch_np<DT, 4>(a, npxt, pl.lds_bytes, nwg, s);
  }
  return set_error("wgrad16: no kernel for %d taps", pl.tp);
}

// plan fields and grid of a launch; returns the workgroup count (0: error set)
static double finish_wgrad16_args(const Wgrad16Args& a, const Wgrad16Plan& pl, bool up2, Wgrad16Args* out) {
  if (!pl.tp) { set_error("wgrad16: geometry not supported"); return 0; }
  if (up2 && pl.tp != 3) { set_error("wgrad16: the upsample-add input needs a k3 block"); return 0; }
  if ((a.groups > 1) && ((a.Cog & 7) || (!a.bcast && (a.Cig & 7)))) {
    set_error("wgrad16: grouped blocks need channels per group % 8 == 0");
    return 0;
  }
  Wgrad16Args b = a;
  b.ltw = ilog2(pl.tw); b.TH = pl.th; b.PCX = pl.pcx; b.tiles_x = pl.tiles_x; b.tiles_y = pl.tiles_y; b.n_tiles = pl.n_tiles;
  b.tiles_per_split = pl.tiles_per_split; b.splits = pl.splits; b.ktg = pl.ktg; b.nstg = pl.nstg; b.dbg = g_conv16_dbg & 15;
  b.gx = cdiv(a.Cig, 64); b.gy = cdiv(a.Cog, 64); b.gz = a.groups * a.KH * pl.ktg * pl.splits;
  const double nwg = (double)b.gx * b.gy * b.gz;
  if (nwg > 2.0e9) { set_error("wgrad16: grid too large"); return 0; }
  *out = b;
  return nwg;
}

// ---- queued launches (ms_bwd_options.defer_wgrad_launch / ms_wgrad_flush)
struct PendingWgrad16 { int dt, tp, up2, npxt, lds, nwg; double flops, bytes; Wgrad16Args a; };
// process-wide: autograd runs the blocks' backward on its device thread and the end-of-backward callback on the caller's
static std::vector<PendingWgrad16> t_pending;
static std::mutex t_pending_mu;

int queue_wgrad16(int dt, const Wgrad16Args& a, const Wgrad16Plan& pl, bool up2, double flops, double bytes) {
  PendingWgrad16 pw;
  const double nwg = finish_wgrad16_args(a, pl, up2, &pw.a);
  if (nwg <= 0) return -1;
  // a queued kernel that writes dw itself (no pixel split) ADDS to it: by the time it runs, autograd may already have
  // accumulated the parameter's other uses of this step into the same slot (the slot starts the step zeroed)
  if (pl.splits == 1) pw.a.accumulate = 1;
  pw.dt = dt; pw.tp = up2 ? 3 : pl.tp; pw.up2 = up2 ? 1 : 0; pw.npxt = up2 ? 0 : wg16_npxt_queued(pl.npx); pw.lds = pl.lds_bytes; pw.nwg = (int)nwg; pw.flops = flops; pw.bytes = bytes;
  std::lock_guard<std::mutex> lk(t_pending_mu);
  t_pending.push_back(pw);
  return 0;
}

void wgrad16_discard() {
  std::lock_guard<std::mutex> lk(t_pending_mu);
  t_pending.clear();
}

int wgrad16_flush(hipStream_t s) {
  // one launch per kernel instance (element type, taps per row, input form) and per WG16_MAX_JOBS blocks, in queue order
  std::vector<PendingWgrad16> q;
  {
    std::lock_guard<std::mutex> lk(t_pending_mu);
    q.swap(t_pending);
  }
  std::vector<char> done(q.size(), 0);
  for (size_t i = 0; i < q.size(); ++i) {
    if (done[i]) continue;
    Wgrad16Batch b;
    b.n = 0;
    int lds = 0;
    long blocks = 0;
    double flops = 0, bytes = 0;
    auto launch = [&]() -> int {
      if (!b.n) return 0;
      TimingScope ts(s, flops, bytes, "wgrad16_multi_kernel<%s,%d,%d>|conv_wgrad_cb8 multi taps%d up%d jobs%d wgs%ld",
                     q[i].dt == DT_BF16 ? "bf16" : "f16", q[i].tp, q[i].up2, q[i].tp, q[i].up2, b.n, blocks);
      int rc = 0;
      if (!ts.skip()) {
        rc = q[i].dt == DT_BF16 ? launch_tp_multi<BF16>(b, q[i].tp, q[i].up2 != 0, q[i].npxt, lds, s)
                                : launch_tp_multi<F16>(b, q[i].tp, q[i].up2 != 0, q[i].npxt, lds, s);
        if (!rc) rc = check_launch("wgrad16_multi_kernel");
      }
      b.n = 0; lds = 0; blocks = 0; flops = bytes = 0;
      return rc;
    };
    for (size_t k = i; k < q.size(); ++k) {
      if (done[k] || q[k].dt != q[i].dt || q[k].tp != q[i].tp || q[k].up2 != q[i].up2 || q[k].npxt != q[i].npxt) continue;
      if (b.n == WG16_MAX_JOBS || blocks + q[k].nwg > 0x3fffffff) {
        const int rc = launch();
        if (rc) return rc;
      }
      blocks += q[k].nwg;
      b.block_end[b.n] = (int)blocks;
      b.job[b.n] = q[k].a;
      ++b.n;
      lds = std::max(lds, q[k].lds);
      flops += q[k].flops; bytes += q[k].bytes;
      done[k] = 1;
    }
    const int rc = launch();
    if (rc) return rc;
  }
  return 0;
}

int launch_wgrad16(int dt, const Wgrad16Args& a, const Wgrad16Plan& pl, bool up2, double flops, double bytes, hipStream_t s) {
  Wgrad16Args b;
  const double nwg = finish_wgrad16_args(a, pl, up2, &b);
  if (nwg <= 0) return -1;
  if (wgrad16_c1_ok(a, up2)) {
    b.gy = c8_of(a.Cog);
    TimingScope ts(s, flops, bytes, "wgrad16_c1_kernel<%s>|conv_wgrad_cb8 c1 k%dx%d s%d Cog%d tiles%d splits%d", dt == DT_BF16 ? "bf16" : "f16",
                   a.KH, a.KW, a.S, a.Cog, pl.n_tiles, pl.splits);
    if (ts.skip()) return 0;
    const int col_tiles = cdiv(a.OUTW, 64);
    if (a.KH == 3 && a.KW == 3 && a.S == 1 && a.SV == 1 && pl.splits % col_tiles == 0 && pl.splits / col_tiles >= 1) {
      // the plan's `splits` slabs = (row splits) x (column tiles of 64): every slab is written, the caller sums them in order
      const int imgs = pl.n_tiles / (pl.tiles_y * pl.tiles_x), row_splits = pl.splits / col_tiles;
      b.n_tiles = imgs * a.OUTH;                                   // rows of the flattened (image, output row) space
      const int rows_per_split = cdiv(b.n_tiles, std::max(1, std::min(256 / col_tiles, b.n_tiles / 8)));      // (as plan_wgrad16)
      if (cdiv(b.n_tiles, rows_per_split) == row_splits) {
        if (dt == DT_BF16) hipLaunchKernelGGL(wgrad16_c1_3x3_kernel<BF16>, dim3(pl.splits), dim3(256), 0, s, b, rows_per_split, col_tiles);
        else hipLaunchKernelGGL(wgrad16_c1_3x3_kernel<F16>, dim3(pl.splits), dim3(256), 0, s, b, rows_per_split, col_tiles);
        return check_launch("wgrad16_c1_3x3_kernel");
      }
      b.n_tiles = pl.n_tiles;
    }
    if (dt == DT_BF16) hipLaunchKernelGGL(wgrad16_c1_kernel<BF16>, dim3(b.gy * pl.splits), dim3(256), 0, s, b);
    else hipLaunchKernelGGL(wgrad16_c1_kernel<F16>, dim3(b.gy * pl.splits), dim3(256), 0, s, b);
    return check_launch("wgrad16_c1_kernel");
  }
  TimingScope ts(s, flops, bytes, "wgrad16_kernel<%s,%d,%d>|conv_wgrad_cb8 k%dx%d s%d Cog%d Kg%d g%d tiles%d tw%d splits%d",
                 dt == DT_BF16 ? "bf16" : "f16", pl.tp, up2 ? 1 : 0, a.KH, a.KW, a.S, a.Cog, a.Cig * a.KH * a.KW, a.groups, pl.n_tiles,
                 pl.tw, pl.splits);
  if (ts.skip()) return 0;
  const int rc = dt == DT_BF16 ? launch_tp<BF16>(b, pl, up2, (int)nwg, s) : launch_tp<F16>(b, pl, up2, (int)nwg, s);
  if (rc) return rc;
  return check_launch("wgrad16_kernel");
}

}  // namespace ms
