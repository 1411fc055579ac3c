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

__device__ inline u32x4 tr_read2(const u32x4* base, int vec0, int vec1, int sub8) {
  // vecN: index of the cb8 vector (row of the 4x16 block this lane addresses), sub8: 0/8 byte offset inside it
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const char* b = reinterpret_cast<const char*>(base);
  // (whole-value bit casts: element access on a builtin's vector result is miscompiled by this clang)
  const unsigned long long lo = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(b + (size_t)vec0 * 16 + sub8)));
  const unsigned long long hi = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(b + (size_t)vec1 * 16 + sub8)));
  u32x4 v;
  v[0] = (unsigned)lo;
  v[1] = (unsigned)(lo >> 32);
  v[2] = (unsigned)hi;
  v[3] = (unsigned)(hi >> 32);
  return v;
}

// workgroups a layer's launch aims for through pixel splits (ms_debug_set_wgrad16_target).  The trainer launches the layers'
// kernels side by side (ms_wgrad_flush), so a layer need not fill the chip alone; measured on the train step: 512 -> 3.17 ms,
// 256 -> 3.12, 128 -> 3.06, 64 -> 3.15 per G-step (fewer, longer workgroups leave fewer partial slabs to store and reduce)
int g_wgrad16_target_wgs = 128;

constexpr int WG16_NPX = 8;   // input-row vectors a thread stages per tile at most (plan_wgrad16 keeps 8*TH*PCX <= 8*256)

template <typename DT, int TP, bool UP2>
__device__ __forceinline__ void wgrad16_body(const Wgrad16Args& p, const int bid, u32x4* smem) {
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wm = wid >> 1, wn = wid & 1, h = lane >> 5;
  const int TW = 1 << p.ltw, TH = p.TH, PCX = p.PCX, S = p.S, SV = p.SV;
  const int xv = 8 * TH * PCX;                 // input-row vectors per tile
  const int stage_vecs = 512 + xv + 1;

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
  // dy: vector e = cb*64 + n, n = ty*TW + tx
  int d_cb[2], d_ty[2], d_tx[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = t + i * 256;
    d_cb[i] = e >> 6;
    const int n = e & 63;
    d_ty[i] = n >> p.ltw; d_tx[i] = n & (TW - 1);
  }
  int x_cb[WG16_NPX], x_ty[WG16_NPX], x_c[WG16_NPX], x_lds[WG16_NPX];
#pragma unroll
  for (int i = 0; i < WG16_NPX; ++i) {
    const int e = t + i * 256;
    const int cb = e / (TH * PCX), rem = e - cb * TH * PCX, ty = rem / PCX;
    x_cb[i] = cb; x_ty[i] = ty; x_c[i] = rem - ty * PCX;
    x_lds[i] = e < xv ? 512 + e : 512 + xv;
  }

  u32x4 rd[2], rx[WG16_NPX], rx2[UP2 ? WG16_NPX : 1];
  auto load_tile = [&](int tile) {
    const int img = tile / tiles_per_img, trem = tile - img * tiles_per_img;
    const int tyi = trem / p.tiles_x, txi = trem - tyi * p.tiles_x;
    const int oy0 = tyi * TH, ox0 = txi << p.ltw;
    const int dbase = __builtin_amdgcn_readfirstlane(img * p.o_img + dy_cb0 * p.o_cblk);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int oy = oy0 + d_ty[i], ox = ox0 + d_tx[i];
      const bool ok = (oy < p.OUTH) & (ox < p.OUTW) & ((co0 >> 3) + d_cb[i] < cog8);
      rd[i] = buf_load_v(rsD, ok ? 16u * (unsigned)(d_cb[i] * p.o_cblk + oy * p.o_row + ox) : BUF_OOB, 16u * (unsigned)dbase);
    }
    const int iy0 = oy0 * SV - p.PH + kh, ix0 = ox0 * S - p.PW + kw0;
    const int sbase = __builtin_amdgcn_readfirstlane(img * p.s_img + x_cb0 * p.s_cblk);
#pragma unroll
    for (int i = 0; i < WG16_NPX; ++i) {
      const int iy = iy0 + x_ty[i] * SV, ix = ix0 + x_c[i];
      const bool ok = (x_lds[i] < 512 + xv) & ((unsigned)iy < (unsigned)p.SRCH) & ((unsigned)ix < (unsigned)p.SRCW) &
                      ((ci0 >> 3) + x_cb[i] < cig8);
      const int o = x_cb[i] * p.s_cblk + iy * p.s_row;
      if (UP2) {
        rx[i] = buf_load_v(rsS, ok ? 16u * (unsigned)((o >> 1) + (ix >> 1)) : BUF_OOB, 16u * (unsigned)(sbase >> 1));
        rx2[UP2 ? i : 0] = buf_load_v(rsS2, ok ? 16u * (unsigned)(o + ix) : BUF_OOB, 16u * (unsigned)sbase);
      } else {
        rx[i] = buf_load_v(rsS, ok ? 16u * (unsigned)(o + ix) : BUF_OOB, 16u * (unsigned)sbase);
      }
    }
  };
  auto store_tile = [&](int buf) {
    u32x4* st = smem + buf * stage_vecs;
#pragma unroll
    for (int i = 0; i < 2; ++i) st[t + i * 256] = rd[i];
#pragma unroll
    for (int i = 0; i < WG16_NPX; ++i) {
      if (UP2) {
        float fa[8], fr[8];
        unpack8<DT>(rx[i], fa);
        unpack8<DT>(rx2[UP2 ? i : 0], fr);
#pragma unroll
        for (int j = 0; j < 8; ++j) fa[j] += fr[j];
        st[x_lds[i]] = pack8<DT>(fa);
      } else {
        st[x_lds[i]] = rx[i];
      }
    }
  };

  f32x16 acc[TP];
#pragma unroll
  for (int q = 0; q < TP; ++q)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;

  // ---- transposed-read lane roles: group of 16 lanes = 16 channels (2 blocks); lane 4q+pp addresses row q, 4 channels pp
  const int li = lane & 15, q4 = li >> 2, pp = li & 3, half16 = (lane >> 4) & 1;
  const int sub8 = (pp & 1) * 8;
  const int a_blk = wm * 4 + half16 * 2 + (pp >> 1);     // dy block inside the 64-channel tile
  const int b_blk = wn * 4 + half16 * 2 + (pp >> 1);     // x block inside the 64-channel tile
  const int thpcx = TH * PCX;

  auto compute_tile = [&](int cur) {
    const u32x4* st = smem + cur * stage_vecs;
    const u32x4* xs = st + 512;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int p0 = 16 * kk + 8 * h + q4, p1 = p0 + 4;           // this lane's rows (pixels) of the two 4-row reads
      const u32x4 av = tr_read2(st, a_blk * 64 + p0, a_blk * 64 + p1, sub8);
      const int x0 = b_blk * thpcx + (p0 >> p.ltw) * PCX + (p0 & (TW - 1)) * S;
      const int x1 = b_blk * thpcx + (p1 >> p.ltw) * PCX + (p1 & (TW - 1)) * S;
#pragma unroll
      for (int q = 0; q < TP; ++q) {
        const u32x4 bv = tr_read2(xs, x0 + q, x1 + q, sub8);
        acc[q] = DT::mfma(av, bv, acc[q]);
      }
    }
  };

  if (tile_beg < tile_end) {
    load_tile(tile_beg);
    store_tile(0);
    __syncthreads();
    for (int tile = tile_beg; tile < tile_end; ++tile) {
      const int cur = (tile - tile_beg) & 1;
      if (tile + 1 < tile_end) load_tile(tile + 1);
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

template <typename DT, int TP, bool UP2>
__global__ __launch_bounds__(256) void wgrad16_kernel(const Wgrad16Args p) {
  prefetch_kernargs<sizeof(Wgrad16Args)>();
  extern __shared__ u32x4 smem[];
  wgrad16_body<DT, TP, UP2>(p, (int)blockIdx.x, smem);
}

// many blocks' weight gradients in one launch: a workgroup finds its job in the table of block ranges
template <typename DT, int TP, bool UP2>
__global__ __launch_bounds__(256) void wgrad16_multi_kernel(const Wgrad16Batch b) {
  extern __shared__ u32x4 smem[];
  prefetch_kernargs<128>();                                   // n and the table of block ranges
  int j = 0;
  while (j + 1 < b.n && (int)blockIdx.x >= b.block_end[j]) ++j;
  const int b0 = j ? b.block_end[j - 1] : 0;
  prefetch_kernargs<sizeof(Wgrad16Args)>((int)offsetof(Wgrad16Batch, job) + j * (int)sizeof(Wgrad16Args));
  wgrad16_body<DT, TP, UP2>(b.job[j], (int)blockIdx.x - b0, smem);
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

Wgrad16Plan plan_wgrad16(int nd, int Cog, int Cig, int groups, int KH, int KW, int SH, int SW, int B, int OH, int OW) {
  Wgrad16Plan pl = {};
  const int rows = nd == 1 ? B : OH, imgs = nd == 1 ? 1 : B;
  pl.tp = KW == 1 ? 1 : KW == 3 ? 3 : (KW % 4 == 0) ? 4 : (KW == 2 ? 2 : 0);
  if (!pl.tp) return pl;
  pl.ktg = KW / pl.tp;
  pl.tw = std::min(pow2_at_least(OW), 64);
  pl.th = 64 / pl.tw;
  pl.pcx = (pl.tw - 1) * SW + pl.tp;
  if (8 * pl.th * pl.pcx > WG16_NPX * 256) return (pl.tp = 0, pl);
  pl.tiles_y = cdiv(rows, pl.th); pl.tiles_x = cdiv(OW, pl.tw);
  pl.n_tiles = imgs * pl.tiles_y * pl.tiles_x;
  // pixel splits: fill ~2 workgroups per CU, at least 2 tiles per workgroup
  const long base = (long)cdiv(Cog, 64) * cdiv(Cig, 64) * groups * KH * pl.ktg;
  int splits = (int)std::max<long>(1, std::min<long>((g_wgrad16_target_wgs + base - 1) / base, pl.n_tiles / 2));
  splits = std::min(splits, 64);
  pl.tiles_per_split = cdiv(pl.n_tiles, std::max(1, splits));
  pl.splits = cdiv(pl.n_tiles, pl.tiles_per_split);
  pl.lds_bytes = 2 * (512 + 8 * pl.th * pl.pcx + 1) * 16;
  (void)SH;
  return pl;
}

template <typename DT, int TP, bool UP2>
static int launch_w(const Wgrad16Args& a, int lds, int nwg, hipStream_t s) {
  static bool attr_done = false;
  auto fn = wgrad16_kernel<DT, TP, UP2>;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return set_error("wgrad16: cannot raise the dynamic LDS limit");
    attr_done = true;
  }
  hipLaunchKernelGGL(fn, dim3(nwg), dim3(256), lds, s, a);
  return 0;
}

template <typename DT, int TP, bool UP2>
static int launch_wm(const Wgrad16Batch& b, int lds, hipStream_t s) {
  static bool attr_done = false;
  auto fn = wgrad16_multi_kernel<DT, TP, UP2>;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return set_error("wgrad16: cannot raise the dynamic LDS limit");
    attr_done = true;
  }
  hipLaunchKernelGGL(fn, dim3(b.block_end[b.n - 1]), dim3(256), lds, s, b);
  return 0;
}

template <typename DT>
static int launch_tp_multi(const Wgrad16Batch& b, int tp, bool up2, int lds, hipStream_t s) {
  if (up2) return launch_wm<DT, 3, true>(b, lds, s);
  switch (tp) {
    case 1: return launch_wm<DT, 1, false>(b, lds, s);
    case 2: return launch_wm<DT, 2, false>(b, lds, s);
    case 3: return launch_wm<DT, 3, false>(b, lds, s);
    case 4: return launch_wm<DT, 4, false>(b, lds, s);
  }
  return set_error("wgrad16: no kernel for %d taps", tp);
}

template <typename DT>
static int launch_tp(const Wgrad16Args& a, const Wgrad16Plan& pl, bool up2, int nwg, hipStream_t s) {
  if (up2) return launch_w<DT, 3, true>(a, pl.lds_bytes, nwg, s);
  switch (pl.tp) {
    case 1: return launch_w<DT, 1, false>(a, pl.lds_bytes, nwg, s);
    case 2: return launch_w<DT, 2, false>(a, pl.lds_bytes, nwg, s);
    case 3: return launch_w<DT, 3, false>(a, pl.lds_bytes, nwg, s);
    case 4: return launch_w<DT, 4, false>(a, pl.lds_bytes, nwg, s);
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
  b.tiles_per_split = pl.tiles_per_split; b.splits = pl.splits; b.ktg = pl.ktg;
  b.gx = cdiv(a.Cig, 64); b.gy = cdiv(a.Cog, 64); b.gz = a.groups * a.KH * pl.ktg * pl.splits;
  const double nwg = (double)b.gx * b.gy * b.gz;
  if (nwg > 2.0e9) { set_error("wgrad16: grid too large"); return 0; }
  *out = b;
  return nwg;
}

// ---- queued launches (ms_bwd_options.defer_wgrad_launch / ms_wgrad_flush)
struct PendingWgrad16 { int dt, tp, up2, lds, nwg; double flops, bytes; Wgrad16Args a; };
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
  pw.dt = dt; pw.tp = up2 ? 3 : pl.tp; pw.up2 = up2 ? 1 : 0; pw.lds = pl.lds_bytes; pw.nwg = (int)nwg; pw.flops = flops; pw.bytes = bytes;
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
        rc = q[i].dt == DT_BF16 ? launch_tp_multi<BF16>(b, q[i].tp, q[i].up2 != 0, lds, s)
                                : launch_tp_multi<F16>(b, q[i].tp, q[i].up2 != 0, lds, s);
        if (!rc) rc = check_launch("wgrad16_multi_kernel");
      }
      b.n = 0; lds = 0; blocks = 0; flops = bytes = 0;
      return rc;
    };
    for (size_t k = i; k < q.size(); ++k) {
      if (done[k] || q[k].dt != q[i].dt || q[k].tp != q[i].tp || q[k].up2 != q[i].up2) continue;
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
  TimingScope ts(s, flops, bytes, "wgrad16_kernel<%s,%d,%d>|conv_wgrad_cb8 k%dx%d s%d Cog%d Kg%d g%d tiles%d tw%d splits%d",
                 dt == DT_BF16 ? "bf16" : "f16", pl.tp, up2 ? 1 : 0, a.KH, a.KW, a.S, a.Cog, a.Cig * a.KH * a.KW, a.groups, pl.n_tiles,
                 pl.tw, pl.splits);
  if (ts.skip()) return 0;
  const int rc = dt == DT_BF16 ? launch_tp<BF16>(b, pl, up2, (int)nwg, s) : launch_tp<F16>(b, pl, up2, (int)nwg, s);
  if (rc) return rc;
  return check_launch("wgrad16_kernel");
}

}  // namespace ms
