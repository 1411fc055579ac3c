// Clip-resident 1-D conv block, fp32 (exact fp32 matrix products): Conv1d (k3 s1 p1, optionally on nearest_up2(a) + r; or k4 s2 p1)
// [+ BatchNorm1d + LeakyReLU] for the layers of the path whose whole reduction fits a workgroup: ConvNormRelu (layers.py:32-78)
// inside UNet1D (layers.py:80-157), ClusterClassify (layers.py:446-467), PoseStyleEncoder (layers.py:246-289), D.conv1 / D.conv2
// (speech2gesture.py:50-57).  Same recipe as the chained decoder (chain32.hip), one block per launch:
//   * a workgroup owns 32 output channels x 64 (stride 2: 32) output frames = whole clips; the clips' COMPLETE input -- all input
//     channels, frames + halo -- is staged once into LDS as [8-channel group][half][slot][4 channels];
//   * the reduction is split over the four waves by channel groups; every wave's weight rows come from HBM / L2 straight into
//     registers in MFMA operand order (prepared once per optimizer update: ms_fwd_weights_prepare / ms_dgrad_weights_prepare) -- the
//     whole K slice of a wave is in flight at once, there is no weight staging, no barrier and no split-K slab in HBM;
//   * the four partial accumulators meet in LDS; thread (channel, 8-lane frame slice) then finishes the block: bias, the clips'
//     partial batch statistics, ONE in-launch meeting of the workgroups that share the channel tile (BN_TRAIN), normalise,
//     activate, store y_raw / y -- no statistics launch, no normalising launch.
// Replaces, per block: patch conv (intra-workgroup split) or im2col-gather conv + split-K epilogue + bn_finalize_apply.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "kernels.h"
#include "conv16.h"

namespace ms {

typedef unsigned int cl_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int cl_u32x2 __attribute__((ext_vector_type(2)));
constexpr int CL_SPIN_LIMIT = 1 << 21;
constexpr int CL_MAXK8W = 9;              // channel groups of 8 per wave at most (Cin <= 288)

struct Clip32Args {
  const float* x;        // (B, Cin, Ti) -- UP2: a (B, Cin, Ti/2)
  const float* x2;       // UP2: r (B, Cin, Ti)
  const float* wp;       // prepared: [channel tile][wave][k8w][KW][64 lanes][4]
  const float* bias;
  const float* gamma;
  const float* beta;
  float* rm;
  float* rv;
  float* y_raw;          // BN_TRAIN: conv + bias (B, Cout, To), or null
  float* y;              // block output (B, Cout, To)
  float* y2;             // EP_DGRAD_UP2: gradient of r (full resolution); y then is the gradient of a (half resolution)
  float* save;
  float* part;           // [channel tiles][pixel workgroups][32][2]
  int* sync;
  int cnt_base;
  int B, Cin, Cout, To, Ti, ep, k8w, npw, rows_valid, nct;      // Cout: output channels (addressing) = rows_valid; nct = ceil(Cout / 32)
  float slope, eps, momentum;
  // EP_DGRAD_BN: BatchNorm + LeakyReLU backward of the PRODUCER of this data gradient's rows, in the epilogue (y holds its dy_raw)
  const float* pv_y; const float* pv_y_raw; const float* pv_save; const float* pv_gamma;
  float* pv_dgamma; float* pv_dbeta; float* pv_dbias;
  int sg;                // BN_TRAIN: statistics groups along the batch (MS_DT_STAT_PAIR: 2), each npw / sg consecutive pixel workgroups
  int raw_all;           // BN_TRAIN: y_raw for every channel (0: only for channels whose backward cannot take x_hat from y: conv16.h bn_inv_unsafe)
  const float* acc;      // data gradient: (B, Cout, To) added to the result (ms_bwd_options.dx_accum: the input's other consumer's gradient)
  int cx, px;            // placement over the 8 XCDs (workgroup id % 8 = XCD): XCD (xc, xp) of a cx x px grid owns nct / cx channel tiles
                         // x npw / px pixel workgroups, so that what its L2 fetches (weight slices + clip images) is smallest; cx = 0: linear
  unsigned long long* stamps;   // diagnostics (MS_CLIP_DBG=32): [workgroup][8] s_memrealtime stamps (100 MHz)
};
#define CL_STAMP(k) do { if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)

__host__ __device__ constexpr int cl_pslots(int n) { return (n + 3) & ~3; }      // slots per plane
// position of slot s (0 = left halo, 1 + frame, Ti + 1 = right halo) inside a clip's row.  Stride-2 forward: even and odd slots
// apart ([row / 2 even slots][row / 2 odd slots]), so that the lanes of a K-loop read (output frame tt, tap k -> slot 2 tt + k) are
// one slot apart instead of two (two slots apart is a 2-way conflict of every ds_read_b128: 16 lanes cover 32 slots of the 16-slot row).
template <bool SPLIT>
__device__ __forceinline__ int cl_pos(int s, int half) { return SPLIT ? (s & 1) * half + (s >> 1) : s; }

__device__ __forceinline__ float cl_f4e(const float4& v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }

template <int CTRL>
__device__ __forceinline__ double cl_dpp_d(double x) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)u, CTRL, 0xf, 0xf, true);
  const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double cl_sum8_d(double v) {
  v += cl_dpp_d<0xB1>(v);
  v += cl_dpp_d<0x4E>(v);
  v += cl_dpp_d<0x141>(v);
  return v;
}

__device__ __forceinline__ float cl_sum8(float v) {
  v += dpp_rot<0xB1>(v);      // quad_perm [1,0,3,2]
  v += dpp_rot<0x4E>(v);      // quad_perm [2,3,0,1]
  v += dpp_rot<0x141>(v);     // row_half_mirror
  return v;
}

// KW taps, stride S; NB = 32-frame blocks of output per workgroup (2, stride 2: 1); UP2: the input is nearest_up2(a) + r.
// K8W > 0: the channel groups per wave are a compile-time constant (the K loop is ONE basic block: operand reads of a unit are
// issued a unit ahead, no wait sits in front of an MFMA but the one for its own operands) and, where a clip's frames come in
// fours, every wave stages exactly the planes its own K slice reads (16-byte loads, all in flight at once, no workgroup barrier
// between staging and the K loop).  K8W = 0: any geometry (run-time trip counts, scalar staging).
// DG2: data gradient of a k4 s2 p1 block (KW = 4, S = 1, NB = 2): the image holds the Ti = To / 2 frames of dy_raw per clip, output
// frame 2m is taps (1, 3) at frames (m, m - 1), frame 2m + 1 taps (0, 2) at (m + 1, m): one accumulator per parity, every unit
// (channel group, tap) feeds the one its tap belongs to -- no zero-stuffed operand, half the matrix work of a k4 s1 conv.
template <int KW, int S, int NB, bool UP2, int K8W, bool DG2 = false>
__global__ __launch_bounds__(256, 1) void clip32_kernel(const Clip32Args p) {
  prefetch_kernargs<sizeof(Clip32Args)>();
  extern __shared__ float cl_smem[];
  CL_STAMP(0);
  constexpr int NPX = 32 * NB;                       // output frames per workgroup
  constexpr int NWR = (K8W ? K8W : CL_MAXK8W) * KW;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, n0 = lane & 31, h = lane >> 5;
  const int nct = p.nct;
  int ct = blockIdx.x % nct, pw = blockIdx.x / nct;  // channel tile, pixel workgroup
  if (p.cx) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, tpx = nct / p.cx, ppx = p.npw / p.px;
    const int xc = xcd % p.cx, xp = xcd / p.cx, jp = j / tpx;
    ct = xc * tpx + (j - jp * tpx);
    pw = xp * ppx + jp;
  }
  const int To = p.To, Ti = p.Ti;
  const int ncl = NPX / To;                          // clips per workgroup (To <= NPX, a power of two)
  const int b0 = pw * ncl;
  const int row = Ti + 2;                            // slots per clip: halo + Ti + halo
  const int pslots = cl_pslots(ncl * row);           // slots per plane
  constexpr bool SPLIT = S == 2 && !DG2;             // (stride-2 forward: even / odd slots apart, cl_pos)
  const int half = row >> 1;
  const int k8w = K8W ? K8W : p.k8w, nk8 = 4 * k8w;  // channel groups per wave / staged in all (zero beyond Cin)

  // ---- this wave's weights: the whole slice in flight (k8w * KW loads of 16 B per lane).  Where the wave stages its own planes
  // the image loads go first: they are needed first, and the counter of outstanding loads retires in order
  float4 wr[NWR];
  const float4* ws = reinterpret_cast<const float4*>(p.wp) + ((size_t)(ct * 4 + w) * k8w * KW) * 64 + lane;
  const bool own_planes = K8W > 0 && (Ti & 3) == 0;  // (uniform)
  if (!own_planes) {
#pragma unroll
    for (int i = 0; i < NWR; ++i) wr[i] = ws[(size_t)min(i, k8w * KW - 1) * 64];
  }

  // ---- the clips' input image: plane (k8, hh) = channels 8*k8 + 4*hh + 0..3, slot = clip * row + 1 + frame.
  if (own_planes) {
    // wave w stages planes [2*K8W*w, 2*K8W*(w+1)): item = (plane, quad of frames): four 16-byte loads (the plane's four
    // channels) -> four 16-byte LDS stores (the quad's four slots); ceil(K8W / 2) items per lane, every load issued first
    constexpr int NQ = DG2 ? 8 : 16;                 // quads of input frames per plane (64 frames; DG2: 32)
    constexpr int NITEMS = 2 * K8W * NQ;
    constexpr int NI = K8W ? (NITEMS + 63) / 64 : 1;
    const int plw0 = 2 * K8W * w;
    for (int e = lane; e < ncl * 2 * 2 * K8W; e += 64) {         // halos of this wave's planes
      const int pl = plw0 + e / (ncl * 2), r2 = e % (ncl * 2), cl = r2 >> 1, sl = (r2 & 1) ? Ti + 1 : 0;
      *reinterpret_cast<float4*>(cl_smem + ((size_t)pl * pslots + cl * row + sl) * 4) = float4{0.f, 0.f, 0.f, 0.f};
    }
    float4 v[NI][4];
    float2 va[NI][4];
#pragma unroll
    for (int q = 0; q < NI; ++q) {
      const int it = min(lane + 64 * q, NITEMS - 1);
      const int pl = plw0 + it / NQ, pos = 4 * (it % NQ);
      const int cl = pos / Ti, ti = pos - cl * Ti;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = min(4 * pl + j, p.Cin - 1);
        if (UP2) {
          v[q][j] = *reinterpret_cast<const float4*>(p.x2 + ((size_t)(b0 + cl) * p.Cin + c) * Ti + ti);
          va[q][j] = *reinterpret_cast<const float2*>(p.x + ((size_t)(b0 + cl) * p.Cin + c) * (Ti / 2) + ti / 2);
        } else {
          v[q][j] = *reinterpret_cast<const float4*>(p.x + ((size_t)(b0 + cl) * p.Cin + c) * Ti + ti);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NWR; ++i) wr[i] = ws[(size_t)i * 64];
#pragma unroll
    for (int q = 0; q < NI; ++q) {
      const int it = lane + 64 * q;
      if (it < NITEMS) {
        const int pl = plw0 + it / NQ, pos = 4 * (it % NQ);
        const int cl = pos / Ti, ti = pos - cl * Ti;
        // (Neighbouring lanes hold neighbouring quads of frames -- a load instruction reads 256 contiguous bytes of a channel row -- so
        // the 8 lanes of a ds_write_b128 group store 64 bytes apart: a 4-way bank conflict on each of these 16 stores, and all of the
        // kernel's SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.51.  Measured in round 6 and left alone: two conflict-free forms
        // (neighbouring lanes = neighbouring planes; each lane pair starting one frame on, components rotated by selects) brought
        // the ratio to 0.12 and the block from 19.4 to 21.1 / 22.0 us -- the LDS is busy for 5 % of the launch, and the loads' shape
        // or 32 selects per item cost more than the conflicts.  DESIGN.md 4e.)
        float o[4][4];                                 // [frame][channel]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool ok = 4 * pl + j < p.Cin;
          float4 u = v[q][j];
          if (UP2) { u.x += va[q][j].x; u.y += va[q][j].x; u.z += va[q][j].y; u.w += va[q][j].y; }
          o[0][j] = ok ? u.x : 0.f; o[1][j] = ok ? u.y : 0.f; o[2][j] = ok ? u.z : 0.f; o[3][j] = ok ? u.w : 0.f;
        }
        float* dst = cl_smem + ((size_t)pl * pslots + cl * row) * 4;
#pragma unroll
        for (int f = 0; f < 4; ++f) *reinterpret_cast<float4*>(dst + 4 * cl_pos<SPLIT>(1 + ti + f, half)) = float4{o[f][0], o[f][1], o[f][2], o[f][3]};
      }
    }
  } else {
    // Positions (clip, frame) are spread over the lanes, 4 coalesced loads + one 16-byte LDS store per position and plane.
    const int npos = ncl * Ti;                       // input frames of this workgroup (64 at every depth)
    for (int e = t; e < ncl * 2; e += 256) {         // halos
      const int cl = e >> 1, sl = (e & 1) ? Ti + 1 : 0;
      for (int pl = 0; pl < nk8 * 2; ++pl) *reinterpret_cast<float4*>(cl_smem + ((size_t)pl * pslots + cl * row + sl) * 4) = float4{0.f, 0.f, 0.f, 0.f};
    }
    const int pos = t % npos, pl0 = t / npos, pstep = 256 / npos;      // npos divides 256 (npos = 64)
    const int cl = pos / Ti, ti = pos - cl * Ti;
    const size_t xoff = (size_t)(b0 + cl) * p.Cin * (UP2 ? Ti / 2 : Ti) + (UP2 ? ti / 2 : ti);
    const size_t x2off = (size_t)(b0 + cl) * p.Cin * Ti + ti;
    const int cstride = UP2 ? Ti / 2 : Ti;
    for (int plb = pl0; plb < nk8 * 2; plb += 8 * pstep) {
      float v[8][4];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int pl = plb + i * pstep;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = min(4 * pl + j, p.Cin - 1);
          float val = p.x[xoff + (size_t)c * cstride];
          if (UP2) val += p.x2[x2off + (size_t)c * Ti];
          v[i][j] = val;
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int pl = plb + i * pstep;
        if (pl < nk8 * 2) {
          float4 o;
          o.x = 4 * pl < p.Cin ? v[i][0] : 0.f;
          o.y = 4 * pl + 1 < p.Cin ? v[i][1] : 0.f;
          o.z = 4 * pl + 2 < p.Cin ? v[i][2] : 0.f;
          o.w = 4 * pl + 3 < p.Cin ? v[i][3] : 0.f;
          *reinterpret_cast<float4*>(cl_smem + ((size_t)pl * pslots + cl * row + cl_pos<SPLIT>(1 + ti, half)) * 4) = o;
        }
      }
    }
  }
  // per-channel parameters of the finishing thread (channel = t >> 3), fetched now
  const int chl = t >> 3, pq = t & 7;
  const int cgl = ct * 32 + chl;
  const bool rowok = cgl < p.rows_valid;
  const int cgc = min(cgl, p.rows_valid - 1);
  const float bias_c = p.bias ? p.bias[cgc] : 0.f;
  float gam = 1.f, bet = 0.f, rmo = 0.f, rvo = 1.f;
  if (p.ep == EP_RAW_STATS || p.ep == EP_BN_EVAL) { gam = p.gamma[cgc]; bet = p.beta[cgc]; rmo = p.rm[cgc]; rvo = p.rv[cgc]; }
  // (data gradient) what is added to this thread's frames, requested ahead of the K loop
  constexpr int FPT_ = NPX / 8;
  float av[UP2 ? 1 : FPT_];
  if constexpr (!UP2) {
#pragma unroll
    for (int k = 0; k < FPT_; ++k) av[k] = 0.f;
    if (p.acc && rowok) {
      const int f0a = FPT_ * pq;
      if (To >= FPT_) {
        const int cl = f0a / To, tt = f0a - cl * To;
        const float* sp = p.acc + ((size_t)(b0 + cl) * p.Cout + cgl) * To + tt;
#pragma unroll
        for (int k = 0; k < FPT_; k += 4) {
          const float4 u = *reinterpret_cast<const float4*>(sp + k);
          av[k] = u.x; av[k + 1] = u.y; av[k + 2] = u.z; av[k + 3] = u.w;
        }
      } else {
#pragma unroll
        for (int k = 0; k < FPT_; ++k) {
          const int f = f0a + k, cl = f / To, tt = f - cl * To;
          av[k] = p.acc[((size_t)(b0 + cl) * p.Cout + cgl) * To + tt];
        }
      }
    }
  }
  CL_STAMP(1);
  if (own_planes) {
    // the wave reads what it stored itself: its LDS stores complete in order ahead of its reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
  CL_STAMP(2);

  // ---- K loop: this wave's channel groups x taps, no barrier
  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[nb][q] = 0.f;
  int bbase[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    if (DG2) {                                       // lane n0 = input frame m of the workgroup's 32: both parities read around it
      const int cl = n0 / Ti, tt = n0 - cl * Ti;
      bbase[nb] = cl * row + tt;
    } else {
      const int n = 32 * nb + n0, cl = n / To, tt = n - cl * To;
      bbase[nb] = cl * row + (SPLIT ? tt : tt * S);
    }
  }
  {
    const float* img = cl_smem + ((size_t)(w * k8w) * 2 + h) * pslots * 4;
    if constexpr (K8W > 0 && DG2) {
      constexpr int NU = K8W * 4;                    // units (channel group, tap k): slot offset (4 - k) >> 1, parity (k + 1) & 1
      float4 bf[2];
      bf[0] = *reinterpret_cast<const float4*>(img + ((size_t)bbase[0] + 2) * 4);
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        if (u + 1 < NU) {
          const int i1 = (u + 1) / 4, k1 = (u + 1) % 4;
          bf[(u + 1) & 1] = *reinterpret_cast<const float4*>(img + ((size_t)i1 * 2 * pslots + bbase[0] + ((4 - k1) >> 1)) * 4);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int par = ((u % 4) + 1) & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[par] = __builtin_amdgcn_mfma_f32_32x32x2f32(cl_f4e(wr[u], j), cl_f4e(bf[u & 1], j), acc[par], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (K8W > 0) {
      constexpr int NU = K8W * KW;                   // units (channel group, tap)
      float4 bf[2][NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) bf[0][nb] = *reinterpret_cast<const float4*>(img + ((size_t)bbase[nb]) * 4);
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        if (u + 1 < NU) {
          const int i1 = (u + 1) / KW, tap1 = (u + 1) % KW;
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            bf[(u + 1) & 1][nb] = *reinterpret_cast<const float4*>(img + ((size_t)i1 * 2 * pslots + bbase[nb] + cl_pos<SPLIT>(tap1, half)) * 4);
        }
        __builtin_amdgcn_sched_barrier(0);           // keep the next unit's LDS reads ahead of these MFMAs
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(cl_f4e(wr[u], j), cl_f4e(bf[u & 1][nb], j), acc[nb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < CL_MAXK8W; ++i) {
        if (i < k8w) {
#pragma unroll
          for (int tap = 0; tap < KW; ++tap) {
            if constexpr (DG2) {
              const float4 b1 = *reinterpret_cast<const float4*>(img + ((size_t)i * 2 * pslots + bbase[0] + ((4 - tap) >> 1)) * 4);
#pragma unroll
              for (int j = 0; j < 4; ++j)
                acc[(tap + 1) & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cl_f4e(wr[i * KW + tap], j), cl_f4e(b1, j), acc[(tap + 1) & 1], 0, 0, 0);
              continue;
            }
            float4 bf[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bf[nb] = *reinterpret_cast<const float4*>(img + ((size_t)i * 2 * pslots + bbase[nb] + cl_pos<SPLIT>(tap, half)) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int nb = 0; nb < NB; ++nb)
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(cl_f4e(wr[i * KW + tap], j), cl_f4e(bf[nb], j), acc[nb], 0, 0, 0);
          }
        }
      }
    }
  }
  CL_STAMP(3);
  __syncthreads();                                   // every wave is done with the image: it becomes the exchange buffer
  CL_STAMP(4);

  // ---- the four partial tiles meet in LDS: red[wave][channel 32][frame NPX + 4]
  constexpr int RP = NPX + 4;
  float* red = cl_smem;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int q = 0; q < 16; ++q) red[(w * 32 + 8 * (q >> 2) + 4 * h + (q & 3)) * RP + (DG2 ? 2 * n0 + nb : 32 * nb + n0)] = acc[nb][q];
  __syncthreads();

  // ---- thread (channel, slice of NPX / 8 frames): sum of the waves in wave order + bias
  constexpr int FPT = NPX / 8;                       // frames per thread: 8 (stride 2: 4)
  float v[FPT];
  {
    const float* rp = red + chl * RP + FPT * pq;
#pragma unroll
    for (int k = 0; k < FPT; k += 4) {
      float4 s = *reinterpret_cast<const float4*>(rp + k);
#pragma unroll
      for (int ww = 1; ww < 4; ++ww) {
        const float4 u = *reinterpret_cast<const float4*>(rp + ww * 32 * RP + k);
        s.x += u.x; s.y += u.y; s.z += u.z; s.w += u.w;
      }
      v[k] = s.x + bias_c; v[k + 1] = s.y + bias_c; v[k + 2] = s.z + bias_c; v[k + 3] = s.w + bias_c;
    }
    if constexpr (!UP2) {
#pragma unroll
      for (int k = 0; k < FPT; ++k) v[k] += av[k];
    }
  }
  // frame f = FPT * pq + k of the workgroup -> (clip, frame inside the clip); To >= FPT keeps a thread inside one clip
  const int f0 = FPT * pq;
  auto store_frames = [&](float* dst, const float (&val)[FPT]) {
    if (To >= FPT) {
      const int cl = f0 / To, tt = f0 - cl * To;
      float* d = dst + ((size_t)(b0 + cl) * p.Cout + cgl) * To + tt;
#pragma unroll
      for (int k = 0; k < FPT; k += 4) *reinterpret_cast<float4*>(d + k) = float4{val[k], val[k + 1], val[k + 2], val[k + 3]};
    } else {
#pragma unroll
      for (int k = 0; k < FPT; ++k) {
        const int f = f0 + k, cl = f / To, tt = f - cl * To;
        dst[((size_t)(b0 + cl) * p.Cout + cgl) * To + tt] = val[k];
      }
    }
  };

  float sc = 1.f, sh = 0.f;
  if (p.ep == EP_RAW_STATS) {
    // this workgroup's partial statistics of the channel: (mean, M2) over its NPX frames
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < FPT; ++k) s += v[k];
    const float mean_w = cl_sum8(s) * (1.0f / NPX);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < FPT; ++k) { const float d = v[k] - mean_w; q = fmaf(d, d, q); }
    const float m2_w = cl_sum8(q);
    double mean = (double)mean_w, m2 = (double)m2_w;
    // statistics group of this workgroup (MS_DT_STAT_PAIR: the batch holds two passes of the module side by side): npg consecutive
    // pixel workgroups, first one pw0
    const int npg = p.npw / p.sg, sgi = pw / npg, pw0 = sgi * npg;
    const double n = (double)npg * NPX;
    const __amdgpu_buffer_rsrc_t rsPart = buf_rsrc(p.part);
    // (mean, M2) of the statistics group that starts at pixel workgroup `first`: the channel's partials, every 8th one per lane (up
    // to 4 in flight), in fp64: sum of means, sum of M2, then the spread of the means about the mean (equal counts) -- lane sums
    // in lane order, then the 8 lanes by a fixed tree
    auto group_stats = [&](int first, double& gmean, double& gm2) {
      float2 pv[4];
      const unsigned base = 8u * (unsigned)((ct * p.npw + first) * 32 + chl);
      double ms = 0.0, qs = 0.0, dv = 0.0;
      for (int i0 = pq; i0 < npg; i0 += 32) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = min(i0 + 8 * u, npg - 1);
          pv[u] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rsPart, (int)(base + 8u * (unsigned)(i * 32)), 0, 16));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (i0 + 8 * u < npg) { ms += (double)pv[u].x; qs += (double)pv[u].y; }
      }
      gmean = cl_sum8_d(ms) / (double)npg;
      for (int i0 = pq; i0 < npg; i0 += 32) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = min(i0 + 8 * u, npg - 1);
          pv[u] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rsPart, (int)(base + 8u * (unsigned)(i * 32)), 0, 16));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (i0 + 8 * u < npg) { const double dl = (double)pv[u].x - gmean; dv += dl * dl; }
      }
      gm2 = cl_sum8_d(qs) + (double)NPX * cl_sum8_d(dv);
    };
    bool expired = false;
    if (p.npw > 1) {
      if (pq == 0)
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(cl_u32x2, float2{mean_w, m2_w}), rsPart,
                                              (int)(8u * (unsigned)((ct * p.npw + pw) * 32 + chl)), 0, 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      CL_STAMP(5);
      int* lflag = reinterpret_cast<int*>(cl_smem + 4 * 32 * RP);
      if (t == 0) {
        int* counter = p.sync + p.cnt_base + 32 * ct;
        const int prior = __hip_atomic_load(p.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sticky error word: chain32.hip, chain_meet
        const unsigned old = (unsigned)__hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (old / (unsigned)p.npw + 1u) * (unsigned)p.npw;
        int spins = 0, bad = 0;
        while ((int)((unsigned)__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > CL_SPIN_LIMIT) { __hip_atomic_store(p.sync, 20, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); bad = 1; break; }
        }
        lflag[0] = bad | (prior != 0);
      }
      __syncthreads();
      CL_STAMP(6);
      group_stats(pw0, mean, m2);
      expired = lflag[0] != 0;
      if (expired) { mean = __builtin_nan(""); }
    }
    const float var = (float)(m2 / n), fmean = (float)mean;
    const float invstd = 1.0f / sqrtf(var + p.eps);
    sc = gam * invstd;
    sh = bet - fmean * sc;
    if (pw == pw0 && pq == 0 && rowok && fmean == fmean && p.save) {
      float* sv = p.save + (size_t)sgi * 4 * p.Cout;
      sv[cgl] = fmean; sv[p.Cout + cgl] = invstd; sv[2 * p.Cout + cgl] = sc; sv[3 * p.Cout + cgl] = sh;
    }
    if (pw == (p.sg - 1) * npg && !expired) {
      // the running statistics move once per statistics group, in group order (two forward passes of the module, gan.py:120,126);
      // the first workgroup of the LAST group applies all of them: the earlier groups' moments from their partials, then its own
      float rmc = rmo, rvc = rvo;
      for (int gi = 0; gi + 1 < p.sg; ++gi) {
        double gmean, gm2;
        group_stats(gi * npg, gmean, gm2);
        if (pq == 0 && rowok) {
          const float ub = n > 1.0 ? (float)(gm2 / (n - 1.0)) : (float)(gm2 / n);
          running_stats_update(&p.rm[cgl], &p.rv[cgl], rmc, rvc, p.momentum, (float)gmean, ub);
          rmc = p.rm[cgl]; rvc = p.rv[cgl];
        }
      }
      if (pq == 0 && rowok && fmean == fmean) {
        const float unbiased = n > 1.0 ? (float)(m2 / (n - 1.0)) : var;
        running_stats_update(&p.rm[cgl], &p.rv[cgl], rmc, rvc, p.momentum, fmean, unbiased);
      }
    }
    if (p.y_raw && rowok && (p.raw_all || bn_inv_unsafe(fmean, invstd, sc, sh, p.slope))) store_frames(p.y_raw, v);
  } else if (p.ep == EP_BN_EVAL) {
    sc = gam * (1.0f / sqrtf(rvo + p.eps));
    sh = bet - rmo * sc;
  } else if (p.ep == EP_DGRAD_BN) {
    // v = gradient w.r.t. the OUTPUT y_p of the block that produced this block's input (channel cgl, FPT frames).  Its BatchNorm +
    // LeakyReLU backward (elementwise.hip: bn_bwd_fused_kernel, same arithmetic) needs the channel's sums of dz and dz * x_hat over
    // the whole batch: the workgroups of the channel tile meet once, exactly like the forward statistics above.
    const int C = p.Cout;
    const float mean = p.pv_save[cgc], invstd = p.pv_save[C + cgc], scp = p.pv_save[2 * C + cgc], shp = p.pv_save[3 * C + cgc];
    const float gm = p.pv_gamma[cgc];
    const bool from_y = !bn_inv_unsafe(mean, invstd, scp, shp, p.slope);      // (per channel: the 8 threads of a channel agree)
    const float* src = from_y ? p.pv_y : p.pv_y_raw;
    float yv[FPT];
    if (To >= FPT) {
      const int cl = f0 / To, tt = f0 - cl * To;
      const float* sp = src + ((size_t)(b0 + cl) * C + cgc) * To + tt;
#pragma unroll
      for (int k = 0; k < FPT; k += 4) {
        const float4 u = *reinterpret_cast<const float4*>(sp + k);
        yv[k] = u.x; yv[k + 1] = u.y; yv[k + 2] = u.z; yv[k + 3] = u.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < FPT; ++k) {
        const int f = f0 + k, cl = f / To, tt = f - cl * To;
        yv[k] = src[((size_t)(b0 + cl) * C + cgc) * To + tt];
      }
    }
    const float beta_c = fmaf(mean, scp, shp), inv_sl = 1.0f / p.slope, inv_g = invstd / scp;     // (from_y: slope, scale are not tiny)
    float dz[FPT], xh[FPT];
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int k = 0; k < FPT; ++k) {
      bool pos;
      float xv;
      if (from_y) {
        pos = yv[k] > 0.f;
        xv = ((pos ? yv[k] : yv[k] * inv_sl) - beta_c) * inv_g;
      } else {
        pos = fmaf(yv[k], scp, shp) > 0.f;
        xv = (yv[k] - mean) * invstd;
      }
      dz[k] = v[k] * (pos ? 1.f : p.slope);
      xh[k] = xv;
      s1 += dz[k];
      s2 = fmaf(dz[k], xv, s2);
      s3 += xv;
    }
    s1 = cl_sum8(s1); s2 = cl_sum8(s2); s3 = cl_sum8(s3);
    double S1 = (double)s1, S2 = (double)s2, S3 = (double)s3;
    if (p.npw > 1) {
      const __amdgpu_buffer_rsrc_t rsPart = buf_rsrc(p.part);
      if (pq == 0)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(cl_u32x4, float4{s1, s2, s3, 0.f}), rsPart,
                                               (int)(16u * (unsigned)((ct * p.npw + pw) * 32 + chl)), 0, 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      int* lflag = reinterpret_cast<int*>(cl_smem + 4 * 32 * RP);
      if (t == 0) {
        int* counter = p.sync + p.cnt_base + 32 * ct;
        const int prior = __hip_atomic_load(p.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sticky error word: chain32.hip, chain_meet
        const unsigned old = (unsigned)__hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (old / (unsigned)p.npw + 1u) * (unsigned)p.npw;
        int spins = 0, bad = 0;
        while ((int)((unsigned)__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > CL_SPIN_LIMIT) { __hip_atomic_store(p.sync, 21, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); bad = 1; break; }
        }
        lflag[0] = bad | (prior != 0);
      }
      __syncthreads();
      // the channel's partials of every pixel workgroup, every 8th one per lane (4 in flight), summed in fp64 in a fixed order
      const unsigned base = 16u * (unsigned)(ct * p.npw * 32 + chl);
      double a1 = 0.0, a2 = 0.0, a3 = 0.0;
      float4 pv4[4];
      for (int i0 = pq; i0 < p.npw; i0 += 32) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = min(i0 + 8 * u, p.npw - 1);
          pv4[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsPart, (int)(base + 16u * (unsigned)(i * 32)), 0, 16));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (i0 + 8 * u < p.npw) { a1 += (double)pv4[u].x; a2 += (double)pv4[u].y; a3 += (double)pv4[u].z; }
      }
      S1 = cl_sum8_d(a1); S2 = cl_sum8_d(a2); S3 = cl_sum8_d(a3);
      if (lflag[0]) S1 = __builtin_nan("");
    }
    const double n = (double)p.npw * NPX;
    const float m1 = (float)(S1 / n), m2 = (float)(S2 / n), gi = gm * invstd;
#pragma unroll
    for (int k = 0; k < FPT; ++k) v[k] = gi * (dz[k] - m1 - xh[k] * m2);
    if (pw == 0 && pq == 0 && rowok) {
      if (p.pv_dgamma) { p.pv_dgamma[cgl] = (float)S2; p.pv_dbeta[cgl] = (float)S1; }
      if (p.pv_dbias) p.pv_dbias[cgl] = -gi * m2 * (float)S3;        // = the sum of dy_raw over the batch (zero but for rounding)
    }
    if (rowok) store_frames(p.y, v);
    return;
  }
  if (!rowok) return;
  if (p.ep == EP_DGRAD_UP2) {
    // data gradient of an upsample-add input: y2 = gradient of r (full resolution), y = gradient of a = sums of frame pairs
    store_frames(p.y2, v);
    float hv[FPT / 2];
#pragma unroll
    for (int k = 0; k < FPT / 2; ++k) hv[k] = v[2 * k] + v[2 * k + 1];
    const int Th = To / 2, fh = f0 / 2;
    if (Th >= FPT / 2) {
      const int cl = fh / Th, tt = fh - cl * Th;
      float* d = p.y + ((size_t)(b0 + cl) * p.Cout + cgl) * Th + tt;
#pragma unroll
      for (int k = 0; k < FPT / 2; ++k) d[k] = hv[k];
    } else {
#pragma unroll
      for (int k = 0; k < FPT / 2; ++k) {
        const int f = fh + k, cl = f / Th, tt = f - cl * Th;
        p.y[((size_t)(b0 + cl) * p.Cout + cgl) * Th + tt] = hv[k];
      }
    }
    return;
  }
  if (p.ep != EP_BARE) {
#pragma unroll
    for (int k = 0; k < FPT; ++k) v[k] = lrelu(p.ep == EP_LRELU ? v[k] : fmaf(v[k], sc, sh), p.slope);
  }
  store_frames(p.y, v);
  CL_STAMP(7);
}

// ---------------------------------------------------------------------------------------------
// weights in stream order: [channel tile][wave][k8 local][tap][lane][4] = row 32*ct + (lane & 31), channels
// 8*(wave*k8w + k8 local) + 4*(lane >> 5) + 0..3.  transposed = 0: w (Cout, Cin, KW), row = output channel;
// transposed = 1 (data gradient of a k3 s1 conv): row = input channel ci, reduction over co, taps reversed: w[co][ci][KW-1-tap]
struct ClipPrepJob {
  const float* w;
  float* out;
  int rows, red, KW, k8w, transposed, w_rows, w_cols, block_end;
};
enum { CLIP_PREP_MAX = 40 };
struct ClipPrepBatch { int n; ClipPrepJob job[CLIP_PREP_MAX]; };

// One workgroup per (job, channel tile, wave): the weights it needs are staged through LDS as they lie in the weight tensor
// (contiguous row segments, coalesced) and leave in stream order (1 KB per (channel group, tap)).
constexpr int CLP_LDS = 32 * 289;
__global__ __launch_bounds__(256) void clip32_prep_kernel(const ClipPrepBatch pb) {
  __shared__ float lds[CLP_LDS];
  int j = 0;
  while (j + 1 < pb.n && (int)blockIdx.x >= pb.job[j].block_end) ++j;
  const ClipPrepJob& jb = pb.job[j];
  const int blk = blockIdx.x - (j ? pb.job[j - 1].block_end : 0);
  const int wv = blk & 3, ct = blk >> 2, t = threadIdx.x;
  const int KW = jb.KW, k8w = jb.k8w;
  const int c_lo = wv * k8w * 8, nc = min(k8w * 8, max(0, jb.red - c_lo));       // reduction channels of this wave
  const int r_lo = 32 * ct, nr = min(32, jb.rows - r_lo);                          // rows of this tile
  int pitch;
  // (every load of a batch is issued before the first LDS store: one at a time this kernel was a chain of 32 / 24 dependent round
  // trips per thread -- 17 us per launch for 31 MB)
  if (!jb.transposed) {
    // source rows = output rows: w[(r_lo + rr) * w_cols + c_lo ..][tap], nc * KW contiguous floats each
    const int seg = nc * KW;
    pitch = (k8w * 8 * KW) | 1;
    for (int k = t; k < seg; k += 256) {
      const float* src = jb.w + ((size_t)r_lo * jb.w_cols + c_lo) * KW + k;
      const size_t rstride = (size_t)jb.w_cols * KW;
      float v[32];
#pragma unroll
      for (int rr = 0; rr < 32; ++rr) v[rr] = src[(size_t)min(rr, nr - 1) * rstride];
#pragma unroll
      for (int rr = 0; rr < 32; ++rr)
        if (rr < nr) lds[rr * pitch + k] = v[rr];
    }
  } else {
    // source rows = reduction channels co: w[(c_lo + cc) * w_cols + r_lo ..][tap], nr * KW contiguous floats each
    const int seg = nr * KW;
    pitch = (32 * KW) | 1;
    const int n = nc * seg;
    for (int e0 = t; e0 < n; e0 += 8 * 256) {
      float v[8];
      int dst[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = min(e0 + 256 * u, n - 1);
        const int cc = e / seg, k = e - cc * seg;
        dst[u] = cc * pitch + k;
        v[u] = jb.w[((size_t)(c_lo + cc) * jb.w_cols + r_lo) * KW + k];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (e0 + 256 * u < n) lds[dst[u]] = v[u];
    }
  }
  __syncthreads();
  float4* dst = reinterpret_cast<float4*>(jb.out) + ((size_t)(ct * 4 + wv) * k8w * KW) * 64;
  const int lane = t & 63, rr = lane & 31, h = lane >> 5;
  for (int u = t >> 6; u < k8w * KW; u += 4) {
    const int k8l = u / KW, tap = u - k8l * KW;
    float vv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cl = 8 * k8l + 4 * h + q;
      float val = 0.f;
      if (rr < nr && cl < nc) val = jb.transposed ? lds[cl * pitch + rr * KW + (jb.transposed == 1 ? KW - 1 - tap : tap)] : lds[rr * pitch + cl * KW + tap];
      vv[q] = val;
    }
    dst[(size_t)u * 64 + lane] = float4{vv[0], vv[1], vv[2], vv[3]};
  }
}

// ---------------------------------------------------------------------------------------------
int g_clip32 = 1;       // ms_debug_set_clip32 / MS_CLIP32=0: the clip-resident kernels off (ablations)

static int clip_k8w(int red) { return (cdiv(red, 8) + 3) / 4; }

// rows: output rows of the launch (forward: Cout; data gradient: Cin); red: reduction channels (forward: Cin; dgrad: Cout)
static bool clip_geom_ok(int B, int rows, int red, int To, int KW, int S, int groups, int nb) {
  if (!g_clip32 || groups != 1) return false;
  if (!((KW == 3 && S == 1) || (KW == 4 && S == 2))) return false;
  if (To < 1 || To > 32 * nb || (To & (To - 1))) return false;
  if ((B * To) % (32 * nb)) return false;
  if (red > 8 * 4 * CL_MAXK8W || rows < 1) return false;
  return true;
}

bool clip32_fwd_ok(const ms_conv_desc* d) {
  if ((d->dtype & 0xff) != 0) return false;
  if (!(d->H == 1 && d->KH == 1 && d->PW == 1 && d->in_mode != MS_IN_BCAST)) return false;
  const int nb = d->SW == 2 ? 1 : 2;
  if (d->in_mode == MS_IN_UP2ADD && !(d->KW == 3 && d->SW == 1)) return false;
  if (d->OW * d->SW != d->W) return false;
  return clip_geom_ok(d->B, d->Cout, d->Cin, d->OW, d->KW, d->SW, d->groups, nb);
}

// data gradient of a k3 s1 p1 block as a k3 s1 conv of dy_raw with the transposed, tap-reversed weights
// ... and of a k4 s2 p1 block as two 2-tap convs, one per output parity, in the same launch (DG2)
bool clip32_dgrad_ok(const ms_conv_desc* d) {
  if ((d->dtype & 0xff) != 0) return false;
  if (!(d->H == 1 && d->KH == 1 && d->PW == 1 && d->in_mode != MS_IN_BCAST)) return false;
  if (d->KW == 4 && d->SW == 2) return d->in_mode == 0 && 2 * d->OW == d->W && clip_geom_ok(d->B, d->Cin, d->Cout, d->W, 4, 2, d->groups, 2);
  if (!(d->KW == 3 && d->SW == 1)) return false;
  return clip_geom_ok(d->B, d->Cin, d->Cout, d->W, 3, 1, d->groups, 2);
}

size_t clip32_weight_floats(int rows, int red, int KW) { return (size_t)cdiv(rows, 32) * 4 * clip_k8w(red) * KW * 256; }

size_t clip32_fwd_weight_bytes(const ms_conv_desc* d) { return clip32_fwd_ok(d) ? clip32_weight_floats(d->Cout, d->Cin, d->KW) * 4 : 0; }
size_t clip32_dgrad_weight_floats(const ms_conv_desc* d) { return clip32_dgrad_ok(d) ? clip32_weight_floats(d->Cin, d->Cout, d->KW) : 0; }

static ClipPrepBatch g_prep;
static int g_prep_blocks = 0;
int clip32_prep_queue(const float* w, float* out, int rows, int red, int KW, int transposed, int w_cols, hipStream_t s) {
  if (g_prep.n == CLIP_PREP_MAX) { const int rc = clip32_prep_flush(s); if (rc) return rc; }
  ClipPrepJob jb = {w, out, rows, red, KW, clip_k8w(red), transposed, 0, w_cols, 0};
  g_prep_blocks += cdiv(rows, 32) * 4;
  jb.block_end = g_prep_blocks;
  g_prep.job[g_prep.n++] = jb;
  return 0;
}
void clip32_prep_discard() { g_prep.n = 0; g_prep_blocks = 0; }
int clip32_prep_flush(hipStream_t s) {
  if (!g_prep.n) return 0;
  TimingScope ts(s, 0, 0, "clip32_prep_kernel|clip_prep jobs%d", g_prep.n);
  const int blocks = g_prep_blocks;
  ClipPrepBatch pb = g_prep;
  g_prep.n = 0; g_prep_blocks = 0;
  if (ts.skip()) return 0;
  hipLaunchKernelGGL(clip32_prep_kernel, dim3(blocks), dim3(256), 0, s, pb);
  return check_launch("clip32_prep_kernel");
}

size_t clip32_part_bytes(int rows, int npw) { return align_up((size_t)cdiv(rows, 32) * npw * 32 * 2 * sizeof(float), 256); }
int clip32_sync_words(int rows) { return 32 * (cdiv(rows, 32) + 1); }

template <int KW, int S, int NB, bool UP2, int K8W, bool DG2 = false>
static int clip32_launch_k(const Clip32Args& a, int nwg, int lds_bytes, hipStream_t s) {
  auto fn = clip32_kernel<KW, S, NB, UP2, K8W, DG2>;
  static unsigned long long attr_done = 0;          // (per device)
  if (first_time_on_device(attr_done)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return set_error("clip32: cannot raise the dynamic LDS limit");
    done_on_device(attr_done);
  }
  hipLaunchKernelGGL(fn, dim3(nwg), dim3(256), lds_bytes, s, a);
  return check_launch("clip32_kernel");
}
// the channel-group counts of the path's layers (64, 104 / 128, 256, 266 channels in) have instances of their own
template <int KW, int S, int NB, bool UP2, bool DG2 = false>
static int clip32_launch_t(const Clip32Args& a, int nwg, int lds_bytes, hipStream_t s) {
  switch (a.k8w) {
    case 2: return clip32_launch_k<KW, S, NB, UP2, 2, DG2>(a, nwg, lds_bytes, s);
    case 4: return clip32_launch_k<KW, S, NB, UP2, 4, DG2>(a, nwg, lds_bytes, s);
    case 8: return clip32_launch_k<KW, S, NB, UP2, 8, DG2>(a, nwg, lds_bytes, s);
    case 9: return clip32_launch_k<KW, S, NB, UP2, 9, DG2>(a, nwg, lds_bytes, s);
    default: return clip32_launch_k<KW, S, NB, UP2, 0, DG2>(a, nwg, lds_bytes, s);
  }
}

// ep: EP_BARE / EP_LRELU / EP_BN_EVAL / EP_RAW_STATS (= BN_TRAIN, everything in this launch) / EP_DGRAD_UP2
int clip32_launch(Clip32Args a, int KW, int S, bool up2, const char* what, hipStream_t s, bool dg2 = false) {
  const int nb = (S == 2 && !dg2) ? 1 : 2, npx = 32 * nb;
  a.k8w = clip_k8w(a.Cin);
  a.npw = a.B * a.To / npx;
  if (a.sg < 1) a.sg = 1;
  if (a.npw % a.sg) return -2;           // (a statistics group is whole pixel workgroups; else the per-layer kernels: ms_stat_pair_ok)
  const int nct = cdiv(a.Cout, 32);
  a.nct = nct;
  const int ncl = npx / a.To, pslots = cl_pslots(ncl * (a.Ti + 2));
  const int img = 4 * a.k8w * 2 * pslots * 16, red = (4 * 32 * (npx + 4) + 4) * 4;
  const int lds = std::max(img, red);
  if (lds > 160 * 1024) return set_error("clip32: image of %d bytes", lds);
  const int nwg = nct * a.npw;
  // Placement over the XCDs (workgroups go to XCD id % 8, each XCD has its own L2): of the cx x px grids that divide the launch, the
  // one whose XCD fetches the fewest bytes -- its share of the weight slices plus its share of the clip images.  The linear order
  // (ct = id % nct) gives every XCD ONE channel tile and ALL images: 3.7x the algorithmic bytes at the fabric for a 256 -> 256
  // block of 32 clips (profiles/r05_sq_backward.json); 4 tiles x 8 clips per XCD is 1.9x.  MS_CLIP_XCD=0: linear (A/B runs).
  static int xcd_on = -1;
  if (xcd_on < 0) { const char* e = getenv("MS_CLIP_XCD"); xcd_on = e ? atoi(e) : 1; }
  a.cx = a.px = 0;
  if (xcd_on && nwg % 8 == 0) {
    const double wct = 32.0 * a.Cin * KW, imgb = (double)a.Cin * ncl * a.Ti * (up2 ? 1.5 : 1.0);
    double best = nct % 8 == 0 ? wct * (nct / 8) + imgb * a.npw : 1e30;      // (linear order with nct % 8 == 0 is the 8 x 1 grid)
    for (int cx = 4; cx >= 1; cx >>= 1) {
      const int px = 8 / cx;
      if (nct % cx || a.npw % px) continue;
      const double cost = wct * (nct / cx) + imgb * (a.npw / px);
      if (cost < best) { best = cost; a.cx = cx; a.px = px; }
    }
  }
  int cus = 256;
  cus = current_device_cus();
  if (!cus) cus = 256;
  // (one workgroup per CU is what every instance of the kernel is sure to get: LDS would allow two of the smaller images, the
  // register file of the upsample-add instance does not)
  if ((a.ep == EP_RAW_STATS || a.ep == EP_DGRAD_BN) && a.npw > 1 && nwg > cus) return -2;      // caller falls back to the per-layer kernels
  const double flops = 2.0 * a.rows_valid * a.Cin * (dg2 ? 2 : KW) * (double)a.B * a.To;
  const double bytes = 4.0 * ((double)a.rows_valid * a.Cin * KW + (double)a.B * a.Cin * a.Ti + (double)a.B * a.rows_valid * a.To);
  TimingScope ts(s, flops, bytes, "clip32_kernel<%d,%d,%d,%d>|conv_%s_clip k1x%d s%d rows%d red%d T%d B%d ep%d", KW, S, nb, up2 ? 1 : 0, what, KW,
                 S, a.rows_valid, a.Cin, a.To, a.B, a.ep);
  if (ts.skip()) return 0;
  static int dbg = -1;
  if (dbg < 0) { const char* e = getenv("MS_CLIP_DBG"); dbg = e ? atoi(e) : 0; }
  static unsigned long long* g_stamps = nullptr;
  if (dbg & 32) {                           // diagnostics only: stamps of every workgroup, printed after a synchronisation
    if (!g_stamps && hipMalloc(&g_stamps, 4096 * 8 * 8) != hipSuccess) g_stamps = nullptr;
    if (g_stamps) (void)hipMemsetAsync(g_stamps, 0, 4096 * 8 * 8, s);
    a.stamps = nwg <= 4096 ? g_stamps : nullptr;
  }
  int rc;
  if (dg2) rc = clip32_launch_t<4, 1, 2, false, true>(a, nwg, lds, s);
  else if (KW == 3 && S == 1) rc = up2 ? clip32_launch_t<3, 1, 2, true>(a, nwg, lds, s) : clip32_launch_t<3, 1, 2, false>(a, nwg, lds, s);
  else rc = clip32_launch_t<4, 2, 1, false>(a, nwg, lds, s);
  if (a.stamps) {
    std::vector<unsigned long long> h((size_t)nwg * 8);
    if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(h.data(), g_stamps, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
      unsigned long long t0 = ~0ull;
      for (int i = 0; i < nwg; ++i) t0 = std::min(t0, h[(size_t)i * 8]);
      fprintf(stderr, "clip32 %s k%d s%d rows%d red%d ep%d: stamps (us after the first workgroup's entry; min / median / max over %d workgroups)\n", what, KW, S,
              a.rows_valid, a.Cin, a.ep, nwg);
      for (int k = 0; k < 8; ++k) {
        std::vector<double> v;
        for (int i = 0; i < nwg; ++i) if (h[(size_t)i * 8 + k]) v.push_back((double)(h[(size_t)i * 8 + k] - t0) * 0.01);
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        fprintf(stderr, "  stamp %d: %7.2f %7.2f %7.2f\n", k, v.front(), v[v.size() / 2], v.back());
      }
    }
  }
  return rc;
}

int clip32_block_fwd(const ms_conv_desc* d, const float* x, const float* x2, const float* wp, const float* bias, const float* gamma,
                     const float* beta, float* rm, float* rv, float* y_raw, float* y, float* save, float* part, int* sync,
                     int sync_words, hipStream_t s) {
  Clip32Args a = {};
  a.x = x; a.x2 = x2; a.wp = wp; a.bias = bias; a.gamma = gamma; a.beta = beta; a.rm = rm; a.rv = rv;
  a.y_raw = d->mode == MS_BN_TRAIN ? y_raw : nullptr; a.y = y; a.save = save; a.part = part; a.sync = sync; a.cnt_base = 32;
  a.B = d->B; a.Cin = d->Cin; a.Cout = d->Cout; a.rows_valid = d->Cout; a.To = d->OW; a.Ti = d->W;
  a.ep = d->mode == MS_BARE ? EP_BARE : d->mode == MS_LRELU ? EP_LRELU : d->mode == MS_BN_EVAL ? EP_BN_EVAL : EP_RAW_STATS;
  a.slope = d->slope; a.eps = d->eps; a.momentum = d->momentum;
  if (a.ep == EP_RAW_STATS && (!sync || sync_words < 32 + clip32_sync_words(d->Cout) || !part)) return -2;
  // the one-launch BatchNorm backward (bn_bwd_fused*) reads the block's output where the map inverts; larger layers' two-pass
  // backward reads y_raw everywhere
  a.sg = a.ep == EP_RAW_STATS ? sg_of(d) : 1;
  a.raw_all = (long)d->B / a.sg * d->OW > BN_BWD32_FUSED_MAX ? 1 : 0;
  return clip32_launch(a, d->KW, d->SW, d->in_mode == MS_IN_UP2ADD, "fwd", s);
}

// the fused form needs the plain-input data gradient (not the upsample-add one) and every workgroup of the launch resident at once
bool clip32_dgrad_bn_ok(const ms_conv_desc* d) {
  if (!clip32_dgrad_ok(d) || d->in_mode != MS_IN_PLAIN) return false;
  const int cus = current_device_cus();
  return cus > 0 && cdiv(d->Cin, 32) * (d->B * d->W / 64) <= cus;
}
size_t clip32_dgrad_bn_part_bytes(const ms_conv_desc* d) {
  return clip32_dgrad_ok(d) && d->in_mode == MS_IN_PLAIN ? align_up((size_t)cdiv(d->Cin, 32) * (d->B * d->W / 64) * 32 * 4 * sizeof(float), 256) : 0;
}

int clip32_block_dgrad(const ms_conv_desc* d, const float* g, const float* wp, float* dx, float* dx2, hipStream_t s,
                       const Clip32PrevBN* pv, float* part, int* sync, int sync_words, const float* accum) {
  Clip32Args a = {};
  const bool up2 = d->in_mode == MS_IN_UP2ADD;
  if (accum && up2) return set_error("clip32 data gradient: dx_accum with an upsample-add input");
  a.acc = accum;
  a.x = g; a.wp = wp;
  a.y = dx; a.y2 = dx2;
  a.B = d->B; a.Cin = d->Cout; a.Cout = d->Cin; a.rows_valid = d->Cin; a.To = d->W; a.Ti = d->W;
  a.ep = up2 ? EP_DGRAD_UP2 : EP_BARE;
  if (pv) {
    if (up2 || !part || !sync || sync_words < 32 + clip32_sync_words(d->Cin) || !pv->y || !pv->y_raw || !pv->save || !pv->gamma)
      return set_error("clip32 data gradient with the producer's BatchNorm backward: missing buffer");
    a.ep = EP_DGRAD_BN;
    a.pv_y = pv->y; a.pv_y_raw = pv->y_raw; a.pv_save = pv->save; a.pv_gamma = pv->gamma;
    a.pv_dgamma = pv->dgamma; a.pv_dbeta = pv->dbeta; a.pv_dbias = pv->dbias;
    a.slope = pv->slope; a.part = part; a.sync = sync; a.cnt_base = 32;
  }
  if (d->KW == 4) {                                  // k4 s2 block: the image is dy_raw at half the output's resolution
    a.Ti = d->OW;
    return clip32_launch(a, 4, 2, false, "dgrad", s, true);
  }
  return clip32_launch(a, 3, 1, false, "dgrad", s);
}

}  // namespace ms

