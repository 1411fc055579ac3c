// Chained pose decoder, 16-bit arithmetic (bf16 / fp16 operands on v_mfma_f32_32x32x16_*, fp32 accumulate, fp32 BatchNorm
// statistics): decoder.0-3 + logits + softmax mixture (JL:69-83,106-115,186-194) in ONE launch.  Same decomposition as chain32.hip:
// a workgroup owns one clip (64 frames) of one sub-generator through all four blocks, the activations stay in LDS as cb8 vectors
// (8 channels of a frame = 16 bytes = one lane's MFMA operand), only BatchNorm's per-clip partial statistics and the mixture's
// per-group terms cross workgroups.
//
// At this size the layer is bound by what a CU can take in, not by the matrix pipe (6.4 GFLOP over 256 CUs = 2.6 us per block;
// 393 KB of weights per CU and block): the weights never touch LDS -- every wave streams the rows of ITS 64 output channels from
// L2 straight into registers in MFMA operand order (ms_decoder_chain_prepare), through a ring of 24 k-step units (192 registers,
// half a block's reduction = ~50 KB per wave in flight), with no barrier inside a block's K loop.  The reduction order is
// (k-step of 16 channels, tap); the activation operand of a unit is the resident image shifted by the tap.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "conv16_kernel.h"

namespace ms {

constexpr int C16_T = 64, C16_C = 256, C16_NL = 4;
constexpr int C16_PITCH = 68;                        // vectors per image plane: halo + 64 frames + halo + 2
constexpr int C16_CB0 = 34;                          // channel blocks of block 0's input (257..272 channels)
constexpr int C16_IMG0 = C16_CB0 * C16_PITCH, C16_IMG1 = 32 * C16_PITCH;       // vectors
constexpr int C16_SPL = 516;                         // floats per scratch plane [64 frames][8 channels] + 4 (bank offset)
constexpr int C16_SCR = 32 * C16_SPL;                // floats
constexpr int C16_RING = 24;                         // units (k-step, tap) in the register ring
constexpr int C16_UNITS_X = 3, C16_UNITS_L = 48, C16_UNITS_LOGIT = 8, C16_UNITS_SLACK = 16;
constexpr int C16_UNITS = C16_UNITS_X + C16_NL * C16_UNITS_L + C16_UNITS_LOGIT + C16_UNITS_SLACK;   // 219
constexpr size_t C16_WAVE_STREAM = (size_t)C16_UNITS * 128;     // 16-byte vectors per (group, wave)
constexpr int C16_PPAD = 128;
constexpr int C16_SPIN_LIMIT = 1 << 21;
constexpr int C16_MAXM = 32;                           // groups (the softmax of a frame is formed in registers)
constexpr int C16_LDS_BYTES = (C16_IMG0 + C16_IMG1) * 16 + C16_SCR * 4 + (3 * C16_NL * 256 + 2 * 256 + 64) * 4 + 40 * 4;

struct Chain16Args {
  const u32x4* x;             // cb8 (B, 34, 64)
  const u32x4* wp;            // prepared weight streams [M][4][C16_WAVE_STREAM]
  const float* bias[C16_NL];
  const float* gamma[C16_NL];
  const float* beta[C16_NL];
  float* rm[C16_NL];
  float* rv[C16_NL];
  u32x4* y_raw[C16_NL];       // cb8 (B, M*32, 64) or null
  u32x4* y[C16_NL];           // cb8 or null
  float* save[C16_NL];
  const float* bias_l;
  float* z;                   // (B, M*P, 64) fp32 or null
  const float* score;
  float* soft;
  float* out;
  float* part;                // [NL][M][B][256][2]
  float* mixpart;             // [B][M][128][64]
  int* sync;
  int cnt_base;
  int B, M, P, train, raw_all;
  float slope, eps, momentum;
  int dbg;                    // timing ablations (MS_CHAIN_DBG): bit 0 no weight refills, bit 1 no MFMAs, bit 2 no epilogue
  unsigned long long* stamps; // diagnostics (MS_CHAIN_DBG bit 5): [workgroup][16] s_memrealtime stamps (100 MHz)
};

__device__ __forceinline__ bool chain16_meet(int* counter, int members, int* err_word, int code) {
  const int prior = __hip_atomic_load(err_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sticky error word: chain32.hip, chain_meet
  const unsigned old = (unsigned)__hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned target = (old / (unsigned)members + 1u) * (unsigned)members;
  int spins = 0;
  while ((int)((unsigned)__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
    __builtin_amdgcn_s_sleep(2);
    if (++spins > C16_SPIN_LIMIT) {
      __hip_atomic_store(err_word, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(err_word + 1, (int)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
  }
  return prior == 0;
}

typedef float f32x4n __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store_v(u32x4* dst, u32x4 v) { __builtin_nontemporal_store(v, dst); }
__device__ __forceinline__ void nt_store_f4(float* dst, float4 v) {
  __builtin_nontemporal_store(f32x4n{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4n*>(dst));
}

// sum over the 8 consecutive lanes that share lane >> 3 (every lane gets the total)
__device__ __forceinline__ float sum8(float v) {
  v += dpp_mov<0xB1>(v);      // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);      // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);     // row_half_mirror: the other quad of the 8
  return v;
}

#define C16_STAMP(k) do { if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)

template <typename DT>
__global__ __launch_bounds__(256, 1) void chain16_kernel(const Chain16Args p) {
  prefetch_kernargs<sizeof(Chain16Args)>();
  C16_STAMP(0);
  extern __shared__ u32x4 smem16[];
  u32x4* imgA = smem16;
  u32x4* imgB = imgA + C16_IMG0;
  float* scr = reinterpret_cast<float*>(imgB + C16_IMG1);
  // per-block parameter tables, filled once at the start (no global load sits between two blocks' K loops: a wait for it would
  // also wait for every weight load issued before it).  train: bias | gamma | beta; eval: bias | scale | shift
  float* tb0 = scr + C16_SCR;               // [NL][256]
  float* tb1 = tb0 + C16_NL * 256;
  float* tb2 = tb1 + C16_NL * 256;
  float* psc = tb2 + C16_NL * 256;          // [256] scale of the current block (train: after the meeting)
  float* psh = psc + 256;
  float* sg = psh + 256;
  int* rawflag = reinterpret_cast<int*>(sg + 64);     // [32] per channel block: keep y_raw
  int* lflag = rawflag + 32;

  const int t = threadIdx.x, lane = t & 63, w = t >> 6, n0 = lane & 31, h = lane >> 5;
  const int g = blockIdx.x % p.M, b = blockIdx.x / p.M;
  const int C = p.M * C16_C, C8 = p.M * 32;

  // ---- this wave's weight stream: unit = (k-step, tap) = [row block 0: 64 lanes x 16 B][row block 1]
  const u32x4* ws = p.wp + ((size_t)g * 4 + w) * C16_WAVE_STREAM + lane;
  u32x4 rx[C16_UNITS_X][2], ra[C16_RING][2];
  // (their first loads go out behind the prologue's own loads, below: loads return in order, and the input image must not wait
  // for 50 KB of weights per wave)
  // ---- LDS: halos, the clip's input image, this group's softmax weights
  if (t == 0) lflag[0] = 0;
  for (int e = t; e < (C16_CB0 + 32) * 4; e += 256) {
    const int plane = e >> 2, s = e & 3;
    u32x4* pl = plane < C16_CB0 ? imgA + plane * C16_PITCH : imgB + (plane - C16_CB0) * C16_PITCH;
    pl[s == 0 ? 0 : 64 + s] = u32x4{0u, 0u, 0u, 0u};
  }
  {
    // every global load of the prologue goes out before the first LDS store (each wait would otherwise be a round trip of its own)
    constexpr int NX = (C16_CB0 * C16_T + 255) / 256;
    u32x4 xv[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int e = min(t + 256 * i, C16_CB0 * C16_T - 1);
      xv[i] = p.x[(size_t)b * C16_CB0 * C16_T + e];
    }
    float q0[C16_NL], q1[C16_NL], q2[C16_NL], q3[C16_NL], q4[C16_NL];
    const int cgp = g * C16_C + t;
#pragma unroll
    for (int l = 0; l < C16_NL; ++l) {
      q0[l] = p.bias[l] ? p.bias[l][cgp] : 0.f;
      q1[l] = p.gamma[l][cgp];
      q2[l] = p.beta[l][cgp];
      q3[l] = p.train ? 0.f : p.rm[l][cgp];
      q4[l] = p.train ? 1.f : p.rv[l][cgp];
    }
    float sv[C16_MAXM];
    {
      const float* sp = p.score + (size_t)b * p.M * C16_T + (t & 63);
#pragma unroll
      for (int m = 0; m < C16_MAXM; ++m) sv[m] = sp[(size_t)min(m, p.M - 1) * C16_T];
    }
    // the weight ring's first fill, behind the loads above
#pragma unroll
  for (int j = 0; j < C16_UNITS_X; ++j)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) rx[j][mb] = ws[(size_t)j * 128 + mb * 64];
#pragma unroll
  for (int j = 0; j < C16_RING; ++j)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) ra[j][mb] = ws[(size_t)(C16_UNITS_X + j) * 128 + mb * 64];

#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int e = t + 256 * i;
      if (e < C16_CB0 * C16_T) imgA[(e >> 6) * C16_PITCH + 1 + (e & 63)] = xv[i];
    }
#pragma unroll
    for (int l = 0; l < C16_NL; ++l) {
      tb0[l * 256 + t] = q0[l];
      if (p.train) { tb1[l * 256 + t] = q1[l]; tb2[l * 256 + t] = q2[l]; }
      else {
        const float sc = q1[l] * (1.0f / sqrtf(q4[l] + p.eps));
        tb1[l * 256 + t] = sc;
        tb2[l * 256 + t] = q2[l] - q3[l] * sc;
      }
    }
    if (t < C16_T) {
      // softmax over the M cluster scores of frame t (JL:186-187); group 0's workgroup of the clip writes the monitor tensor
      float mx = sv[0];
#pragma unroll
      for (int m = 1; m < C16_MAXM; ++m) mx = m < p.M ? fmaxf(mx, sv[m]) : mx;
      float den = 0.f, mine = 0.f;
#pragma unroll
      for (int m = 0; m < C16_MAXM; ++m) {
        sv[m] = m < p.M ? __expf(sv[m] - mx) : 0.f;
        den += sv[m];
        mine = m == g ? sv[m] : mine;
      }
      sg[t] = mine / den;
      if (g == 0 && p.soft) {
#pragma unroll
        for (int m = 0; m < C16_MAXM; ++m)
          if (m < p.M) p.soft[((size_t)b * C16_T + t) * p.M + m] = sv[m] / den;
      }
    }
  }
  // the leader of a group (clip 0) updates the running statistics: their old values, fetched now
  float rm_old[C16_NL], rv_old[C16_NL];
#pragma unroll
  for (int l = 0; l < C16_NL; ++l) {
    const bool lead = p.train && b == 0;
    rm_old[l] = lead ? p.rm[l][g * C16_C + t] : 0.f;
    rv_old[l] = lead ? p.rv[l][g * C16_C + t] : 0.f;
  }

  f32x16 acc[2][2];
  size_t pos = C16_UNITS_X;                 // stream position of ring slot 0's current unit
  u32x4* bin = imgA;
  u32x4* bout = imgB;
  const __amdgpu_buffer_rsrc_t rsPart = buf_rsrc(p.part);

  // 24 units = k-steps ks0 .. ks0+7 x 3 taps against the resident image; every slot is refilled with the unit 24 positions on
  // (the loop body is kept free of branches and identical for every half, the stream's last one included -- its refills read the
  // slack units behind the logits: with a differing tail the compiler's wait counts at the loop joins turn conservative and the
  // ring no longer runs ahead; measured 5.2 -> 10 us per block)
  auto run_half = [&](const u32x4* img, int ks0) {
    // activation fragments three units ahead of their MFMAs (a unit is 4 MFMAs = 128 cycles: about one LDS round trip)
    u32x4 bf[4][2];
    auto fetch_b = [&](int j, u32x4 (&dst)[2]) {
      const int ks = j / 3, tap = j - 3 * ks;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) dst[nb] = img[(2 * (ks0 + ks) + h) * C16_PITCH + 32 * nb + n0 + tap];
    };
    fetch_b(0, bf[0]);
    fetch_b(1, bf[1]);
    fetch_b(2, bf[2]);
#pragma unroll
    for (int j = 0; j < C16_RING; ++j) {
      if (j + 3 < C16_RING) fetch_b(j + 3, bf[(j + 3) & 3]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = DT::mfma(ra[j][mb], bf[j & 3][nb], acc[mb][nb]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) ra[j][mb] = ws[(pos + C16_RING + j) * 128 + mb * 64];
    }
    pos += C16_RING;
  };

  C16_STAMP(1);
  for (int l = 0; l < C16_NL; ++l) {
    const float* pb = tb0 + l * 256;
    if (!p.train) { psc = tb1 + l * 256; psh = tb2 + l * 256; }
    if (l == 0) __syncthreads();
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[mb][nb][q] = 0.f;

    if (l == 0) {
      // block 0's 17th k-step (the style channels 256..271): its three units came first in the stream
#pragma unroll
      for (int tap = 0; tap < 3; ++tap) {
        u32x4 bfx[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) bfx[nb] = bin[(32 + h) * C16_PITCH + 32 * nb + n0 + tap];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = DT::mfma(rx[tap][mb], bfx[nb], acc[mb][nb]);
      }
    }
    run_half(bin, 0);
    run_half(bin, 8);
    __syncthreads();                        // pb / psc / psh visible; the scratch is free
    C16_STAMP(2 + 2 * l);
    if (p.dbg & 4) { u32x4* tmp0 = bin; bin = bout; bout = tmp0; if (l == 0) bout = imgA; continue; }

    // ---- epilogue: conv + bias (fp32) through the scratch [channel block][frame][8]; thread (cb, pq) then owns frames 8*k + pq
    {
      const int cb0 = 8 * w;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int c0 = 64 * w + 32 * mb + 8 * rq + 4 * h;
          const float4 bs = *reinterpret_cast<const float4*>(pb + c0);
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            const float4 v = {acc[mb][nb][4 * rq] + bs.x, acc[mb][nb][4 * rq + 1] + bs.y, acc[mb][nb][4 * rq + 2] + bs.z,
                              acc[mb][nb][4 * rq + 3] + bs.w};
            *reinterpret_cast<float4*>(scr + (cb0 + 4 * mb + rq) * C16_SPL + (32 * nb + n0) * 8 + 4 * h) = v;
          }
        }
    }
    __syncthreads();
    const int cb = t >> 3, pq = t & 7;
    float v[8][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float4 lo = *reinterpret_cast<const float4*>(scr + cb * C16_SPL + (8 * k + pq) * 8);
      const float4 hi = *reinterpret_cast<const float4*>(scr + cb * C16_SPL + (8 * k + pq) * 8 + 4);
      v[k][0] = lo.x; v[k][1] = lo.y; v[k][2] = lo.z; v[k][3] = lo.w;
      v[k][4] = hi.x; v[k][5] = hi.y; v[k][6] = hi.z; v[k][7] = hi.w;
    }
    if (p.train) {
      float mean[8], m2[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[k][j];
        mean[j] = sum8(s) * (1.0f / C16_T);
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float d = v[k][j] - mean[j]; q = fmaf(d, d, q); }
        m2[j] = sum8(q);
      }
      if (pq == 0) {
        const unsigned off = 8u * (unsigned)((((l * p.M + g) * p.B + b) * C16_C) + 8 * cb);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, float4{mean[2 * i], m2[2 * i], mean[2 * i + 1], m2[2 * i + 1]}),
                                                 rsPart, (int)(off + 16u * (unsigned)i), 0, 16);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (l == 0) C16_STAMP(13);
      if (t == 0) {
        if (!chain16_meet(p.sync + p.cnt_base + 32 * (l * p.M + g), p.B, p.sync, 1 + l)) lflag[0] = 1;
      }
      __syncthreads();
      if (l == 0) C16_STAMP(14);
      {
        // (an opaque zero: what is derived from it is computed HERE in every block instead of being hoisted out of the block loop
        // and kept alive -- spilled to scratch, as it turned out -- across the K loops, where the register file is full)
        int opq = 0;
        asm volatile("" : "+s"(opq));
        const int Bq = p.B + opq;
        const unsigned base = 8u * (unsigned)(((l * p.M + g) * Bq) * C16_C + t);
        // equal counts (64 frames per clip): mean = average of the clips' means, M2 = sum of their M2 + 64 * sum (mean_i - mean)^2
        // -- two passes over the partials in clip order, fp64, no division in the loops (Chan's update costs two fp64 divisions
        // per clip: 3 us of the meeting)
        double msum = 0.0, m2c = 0.0, dev = 0.0;
        float2 pv[32];
        const int nb32 = min(Bq, 32);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          const int bb2 = min(i, Bq - 1);
          pv[i] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rsPart, (int)(base + 8u * (unsigned)(bb2 * C16_C)), 0, 16));
        }
#pragma unroll
        for (int i = 0; i < 32; ++i)
          if (i < nb32) { msum += (double)pv[i].x; m2c += (double)pv[i].y; }
        for (int bb2 = 32; bb2 < Bq; ++bb2) {              // (B > 32: the rest one by one)
          const float2 q = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rsPart, (int)(base + 8u * (unsigned)(bb2 * C16_C)), 0, 16));
          msum += (double)q.x; m2c += (double)q.y;
        }
        const double n = (double)Bq * C16_T, mean_c = msum / (double)Bq;
#pragma unroll
        for (int i = 0; i < 32; ++i)
          if (i < nb32) { const double dl = (double)pv[i].x - mean_c; dev += dl * dl; }
        for (int bb2 = 32; bb2 < Bq; ++bb2) {
          const float2 q = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rsPart, (int)(base + 8u * (unsigned)(bb2 * C16_C)), 0, 16));
          const double dl = (double)q.x - mean_c; dev += dl * dl;
        }
        m2c += (double)C16_T * dev;
        const int cg = g * C16_C + t + opq;
        const float var = (float)(m2c / n), fmean = (float)mean_c;
        const float invstd = 1.0f / sqrtf(var + p.eps);
        float sc = tb1[l * 256 + t] * invstd;
        float sh = tb2[l * 256 + t] - fmean * sc;
        const bool unsafe_c = bn_inv_unsafe(fmean, invstd, sc, sh, p.slope);
        if (lflag[0]) { sc = __builtin_nanf(""); sh = sc; }
        else if (b == 0) {
          if (p.save[l]) { float* sv = p.save[l]; sv[cg] = fmean; sv[C + cg] = invstd; sv[2 * C + cg] = sc; sv[3 * C + cg] = sh; }
          const float unbiased = n > 1.0 ? (float)(m2c / (n - 1.0)) : var;
          running_stats_update(&p.rm[l][cg], &p.rv[l][cg], rm_old[l], rv_old[l], p.momentum, fmean, unbiased);
        }
        psc[t] = sc;
        psh[t] = sh;
        const unsigned long long unsafe = __ballot(unsafe_c);
        if (!(t & 7)) rawflag[t >> 3] = (int)((unsafe >> (t & 63)) & 0xffull) != 0;
      }
      __syncthreads();
      if (l == 0) C16_STAMP(15);
    }
    {
      // normalise + activate: 8 frames x 8 channels per thread -> cb8 vectors: the next block's input image and HBM
      const float4 sc0 = *reinterpret_cast<const float4*>(psc + 8 * cb), sc1 = *reinterpret_cast<const float4*>(psc + 8 * cb + 4);
      const float4 sh0 = *reinterpret_cast<const float4*>(psh + 8 * cb), sh1 = *reinterpret_cast<const float4*>(psh + 8 * cb + 4);
      const float scv[8] = {sc0.x, sc0.y, sc0.z, sc0.w, sc1.x, sc1.y, sc1.z, sc1.w};
      const float shv[8] = {sh0.x, sh0.y, sh0.z, sh0.w, sh1.x, sh1.y, sh1.z, sh1.w};
      const bool keep_raw = p.train && p.y_raw[l] && (p.raw_all || rawflag[cb]);
      const size_t gvec = ((size_t)b * C8 + (size_t)g * 32 + cb) * C16_T;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int px = 8 * k + pq;
        float yv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) yv[j] = lrelu(fmaf(v[k][j], scv[j], shv[j]), p.slope);
        const u32x4 yvec = pack8<DT>(yv);
        bout[cb * C16_PITCH + 1 + px] = yvec;
        if (p.y[l]) nt_store_v(p.y[l] + gvec + px, yvec);
        if (keep_raw) nt_store_v(p.y_raw[l] + gvec + px, pack8<DT>(v[k]));
      }
    }
    __syncthreads();
    C16_STAMP(3 + 2 * l);
    u32x4* tmp = bin; bin = bout; bout = tmp;
    if (l == 0) bout = imgA;
  }

  // ---- logits (1x1; this wave's 32 of the P rows) + the group's mixture term; the 8 logits units sit in ring slots 0..7
  if (!(p.dbg & 16)) {
    f32x16 za[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int q = 0; q < 16; ++q) za[nb][q] = 0.f;
#pragma unroll
    for (int u = 0; u < C16_UNITS_LOGIT; ++u)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int ks = 2 * u + e;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const u32x4 bv = bin[(2 * ks + h) * C16_PITCH + 1 + 32 * nb + n0];
          za[nb] = DT::mfma(ra[u][e], bv, za[nb]);
        }
      }
    C16_STAMP(10);
    float* mt = scr;                        // [128 rows][64 frames] fp32
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int prow = 32 * w + 8 * (q >> 2) + 4 * h + (q & 3);
      const float bl = prow < p.P ? p.bias_l[g * p.P + prow] : 0.f;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) mt[prow * C16_T + 32 * nb + n0] = za[nb][q] + bl;
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsMix = buf_rsrc(p.mixpart);
    {
      const unsigned mbase = 4u * (unsigned)(((b * p.M + g) * C16_PPAD) * C16_T);
      for (int e = t; e < p.P * (C16_T / 4); e += 256) {
        float4 v4 = *reinterpret_cast<const float4*>(mt + 4 * e);
        if (p.z) nt_store_f4(p.z + ((size_t)b * p.M * p.P + (size_t)g * p.P) * C16_T + 4 * e, v4);
        const float4 sw = *reinterpret_cast<const float4*>(sg + 4 * (e & 15));
        v4 = float4{v4.x * sw.x, v4.y * sw.y, v4.z * sw.z, v4.w * sw.w};
        if (lflag[0]) v4 = float4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v4), rsMix, (int)(mbase + 16u * (unsigned)e), 0, 16);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0 && !(p.dbg & 8)) {
      if (!chain16_meet(p.sync + p.cnt_base + 32 * (C16_NL * p.M + b), p.M, p.sync, 9)) lflag[0] = 1;
    }
    __syncthreads();
    C16_STAMP(11);
    const int t0 = (g * C16_T / p.M) & ~3, t1 = g + 1 == p.M ? C16_T : (((g + 1) * C16_T / p.M) & ~3);
    const int nq = (t1 - t0) >> 2;
    for (int e = t; e < nq * p.P; e += 256) {
      const int tq = e / p.P, pp = e - tq * p.P;
      float4 s = {0.f, 0.f, 0.f, 0.f};
      for (int m0 = 0; m0 < p.M; m0 += 8) {              // 8 groups' terms in flight, added in group order
        float4 v4[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const unsigned off = 4u * (unsigned)((((b * p.M + min(m0 + i, p.M - 1)) * C16_PPAD) + pp) * C16_T + t0 + 4 * tq);
          v4[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsMix, (int)off, 0, 16));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (m0 + i < p.M) { s.x += v4[i].x; s.y += v4[i].y; s.z += v4[i].z; s.w += v4[i].w; }
      }
      if (lflag[0]) s = float4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
      float* o = p.out + ((size_t)b * C16_T + t0 + 4 * tq) * p.P + pp;
      o[0] = s.x; o[p.P] = s.y; o[2 * p.P] = s.z; o[3 * p.P] = s.w;
    }
  }
  C16_STAMP(12);
}

// ---------------------------------------------------------------------------------------------
// weight streams: fp32 master weights -> 16-bit vectors in stream order.  Conv blocks: one workgroup per (block, group, wave,
// row block, channel half) stages its 32 rows x 128 (144) channels x 3 taps through LDS (contiguous row segments in, whole
// 1 KB units out); the logits rows are gathered directly (8 consecutive channels of a row are contiguous).
struct Chain16PrepArgs {
  const float* w[C16_NL];
  const float* wl;
  u32x4* out;
  int M, P, cin0;
};
constexpr int C16P_ROWP = 433;

template <typename DT>
__global__ __launch_bounds__(256) void chain16_prep_kernel(const Chain16PrepArgs p) {
  __shared__ float lds[32 * C16P_ROWP];
  const int t = threadIdx.x;
  const int n_conv = C16_NL * p.M * 4 * 2 * 2;
  if ((int)blockIdx.x < n_conv) {
    int id = blockIdx.x;
    const int kh = id & 1; id >>= 1;
    const int mb = id & 1; id >>= 1;
    const int wv = id & 3; id >>= 2;
    const int g = id % p.M, l = id / p.M;
    const int cin = l == 0 ? p.cin0 : C16_C;
    const int c_lo = 128 * kh, c_hi = kh ? (l == 0 ? 8 * C16_CB0 : C16_C) : 128;
    const int nfl = (min(c_hi, cin) - c_lo) * 3;
    const int row0 = g * C16_C + 64 * wv + 32 * mb;
    {
      const float* src = p.w[l] + ((size_t)row0 * cin + c_lo) * 3;
      for (int k = t; k < nfl; k += 256) {
#pragma unroll
        for (int r0 = 0; r0 < 32; r0 += 8) {
          float tmp[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) tmp[i] = src[(size_t)(r0 + i) * cin * 3 + k];
#pragma unroll
          for (int i = 0; i < 8; ++i) lds[(r0 + i) * C16P_ROWP + k] = tmp[i];
        }
      }
    }
    __syncthreads();
    const int nks = (c_hi - c_lo) / 16;                  // 8 (9 for block 0's upper half)
    u32x4* dst = p.out + ((size_t)g * 4 + wv) * C16_WAVE_STREAM;
    const int lane = t & 63, rr = lane & 31, h = lane >> 5;
    for (int qi = t >> 6; qi < nks * 3; qi += 4) {
      const int ksl = qi / 3, tap = qi - 3 * ksl;
      const int ks = c_lo / 16 + ksl;
      const int cl = 16 * ksl + 8 * h;
      float vv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) vv[j] = (c_lo + cl + j < cin) ? lds[rr * C16P_ROWP + (cl + j) * 3 + tap] : 0.f;
      const int unit = ks < 16 ? C16_UNITS_X + C16_UNITS_L * l + 3 * ks + tap : tap;       // (k-step 16 exists in block 0 only)
      dst[((size_t)unit * 2 + mb) * 64 + lane] = pack8<DT>(vv);
    }
    return;
  }
  const size_t gid = (size_t)(blockIdx.x - n_conv) * 256 + t;
  if (gid >= (size_t)p.M * 4 * C16_UNITS_LOGIT * 128) return;
  const int lane = (int)(gid & 63), e = (int)((gid >> 6) & 1), u = (int)((gid >> 7) % C16_UNITS_LOGIT), gw = (int)(gid / (C16_UNITS_LOGIT * 128));
  const int g = gw >> 2, wv = gw & 3, h = lane >> 5, prow = 32 * wv + (lane & 31), ks = 2 * u + e;
  float vv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (prow < p.P) {
    const float* src = p.wl + ((size_t)(g * p.P + prow)) * C16_C + 16 * ks + 8 * h;
#pragma unroll
    for (int j = 0; j < 8; ++j) vv[j] = src[j];
  }
  p.out[(size_t)gw * C16_WAVE_STREAM + ((size_t)(C16_UNITS_X + C16_NL * C16_UNITS_L + u) * 2 + e) * 64 + lane] = pack8<DT>(vv);
}

static int chain16_shape_ok(const ms_chain_desc* d) {
  return d && d->T == C16_T && d->C == C16_C && d->n_blocks == C16_NL && d->cin0 > C16_C && d->cin0 <= 8 * C16_CB0 && d->P >= 1 &&
         d->P <= C16_PPAD && d->B >= 1 && d->M >= 1 && d->M <= C16_MAXM && (d->mode == MS_BN_TRAIN || d->mode == MS_BN_EVAL) &&
         (d->dtype == MS_BF16 || d->dtype == MS_F16);
}

size_t chain16_prepared_bytes(const ms_chain_desc* d) { return (size_t)d->M * 4 * C16_WAVE_STREAM * 16; }

size_t chain16_workspace(const ms_chain_desc* d) {
  return align_up((size_t)C16_NL * d->M * d->B * C16_C * 2 * sizeof(float), 256) +
         align_up((size_t)d->B * d->M * C16_PPAD * C16_T * sizeof(float), 256) + 256;
}

int chain16_sync_words(const ms_chain_desc* d) { return 32 * (C16_NL * d->M + d->B + 1); }

int chain16_supported(const ms_chain_desc* d) {
  if (!chain16_shape_ok(d)) return 0;
  const int cus = current_device_cus();          // (per device, like the raised LDS limit: chain32_supported)
  static unsigned long long lds_done = 0;
  if (!cus) return 0;
  if (first_time_on_device(lds_done)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(chain16_kernel<BF16>), hipFuncAttributeMaxDynamicSharedMemorySize, C16_LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(chain16_kernel<F16>), hipFuncAttributeMaxDynamicSharedMemorySize, C16_LDS_BYTES) != hipSuccess) return 0;
    done_on_device(lds_done);
  }
  return d->B * d->M <= cus;
}

int chain16_prepare(const ms_chain_desc* d, const float* const* w, const float* wl, void* prepared, hipStream_t s) {
  if (!chain16_shape_ok(d)) return set_error("ms_decoder_chain_prepare: unsupported shape");
  Chain16PrepArgs a = {};
  for (int l = 0; l < C16_NL; ++l) a.w[l] = w[l];
  a.wl = wl; a.out = (u32x4*)prepared; a.M = d->M; a.P = d->P; a.cin0 = d->cin0;
  TimingScope ts(s, 0, (double)chain16_prepared_bytes(d), "chain16_prep_kernel|chain_prep %s M%d", d->dtype == MS_BF16 ? "bf16" : "f16", d->M);
  if (ts.skip()) return 0;
  const int n_conv = C16_NL * d->M * 4 * 2 * 2, n_log = (d->M * 4 * C16_UNITS_LOGIT * 128 + 255) / 256;
  if (d->dtype == MS_BF16) hipLaunchKernelGGL(chain16_prep_kernel<BF16>, dim3(n_conv + n_log), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(chain16_prep_kernel<F16>, dim3(n_conv + n_log), dim3(256), 0, s, a);
  return check_launch("chain16_prep_kernel");
}

int chain16_fwd(const ms_chain_desc* d, const ms_chain_tensors* tn, void* workspace, size_t workspace_bytes, hipStream_t s) {
  if (!chain16_supported(d)) return set_error("ms_decoder_chain_fwd: shape / device not supported (ms_decoder_chain_supported)");
  if (!tn || !tn->x || !tn->score || !tn->out || !tn->prepared || !tn->sync || !tn->w_logits)
    return set_error("ms_decoder_chain_fwd: null tensor");
  if (workspace_bytes < chain16_workspace(d) || !workspace) return set_error("ms_decoder_chain_fwd: workspace too small");
  if (tn->sync_words < d->sync_first_word + chain16_sync_words(d)) return set_error("ms_decoder_chain_fwd: sync buffer too small");
  Chain16Args a = {};
  a.x = (const u32x4*)tn->x; a.wp = (const u32x4*)tn->prepared;
  for (int l = 0; l < C16_NL; ++l) {
    if (!tn->gamma[l] || !tn->beta[l] || !tn->running_mean[l] || !tn->running_var[l]) return set_error("ms_decoder_chain_fwd: BN tensors missing");
    a.bias[l] = tn->bias[l]; a.gamma[l] = tn->gamma[l]; a.beta[l] = tn->beta[l]; a.rm[l] = tn->running_mean[l]; a.rv[l] = tn->running_var[l];
    a.y_raw[l] = (u32x4*)tn->y_raw[l]; a.y[l] = (u32x4*)tn->y[l]; a.save[l] = tn->save[l];
  }
  a.bias_l = tn->bias_logits; a.z = tn->z; a.score = tn->score; a.soft = tn->soft; a.out = tn->out;
  a.part = (float*)workspace;
  a.mixpart = (float*)((char*)workspace + align_up((size_t)C16_NL * d->M * d->B * C16_C * 2 * sizeof(float), 256));
  a.sync = tn->sync; a.cnt_base = d->sync_first_word + 32;
  a.B = d->B; a.M = d->M; a.P = d->P; a.train = d->mode == MS_BN_TRAIN;
  a.raw_all = (d->keep_all_raw || (long)d->B * C16_T > BN_BWD16_FUSED_MAX) ? 1 : 0;      // (its two-pass backward reads y_raw, not y)
  a.slope = d->slope; a.eps = d->eps; a.momentum = d->momentum;
  { static int dbg = -1; if (dbg < 0) { const char* e = getenv("MS_CHAIN_DBG"); dbg = e ? atoi(e) : 0; } a.dbg = dbg; }
  const double bt = (double)d->B * C16_T;
  const double flops = 2.0 * bt * d->M * (C16_C * 3.0 * (d->cin0 + 3.0 * C16_C) + (double)d->P * C16_C);
  const double bytes = 81.8e6 * (bt / 2048.0) * (d->M / 8.0);          // the unit's algorithmic bytes in 16 bits
  TimingScope ts(s, flops, bytes, "chain16_kernel<%s>|decoder_chain_fwd %s M%d B%d P%d cin%d %s", d->dtype == MS_BF16 ? "bf16" : "f16",
                 d->dtype == MS_BF16 ? "bf16" : "f16", d->M, d->B, d->P, d->cin0, a.train ? "train" : "eval");
  if (ts.skip()) return 0;
  static unsigned long long* g_stamps = nullptr;
  if (a.dbg & 32) {                         // diagnostics only: stamps of every workgroup, printed after a synchronisation
    if (!g_stamps && hipMalloc(&g_stamps, 4096 * 16 * 8) != hipSuccess) g_stamps = nullptr;
    if (g_stamps) (void)hipMemsetAsync(g_stamps, 0, 4096 * 16 * 8, s);
    a.stamps = g_stamps;
  }
  if (d->dtype == MS_BF16) hipLaunchKernelGGL(chain16_kernel<BF16>, dim3(d->B * d->M), dim3(256), C16_LDS_BYTES, s, a);
  else hipLaunchKernelGGL(chain16_kernel<F16>, dim3(d->B * d->M), dim3(256), C16_LDS_BYTES, s, a);
  if (a.stamps) {
    const int nwg = d->B * d->M;
    std::vector<unsigned long long> h((size_t)nwg * 16);
    if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(h.data(), g_stamps, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
      unsigned long long t0 = ~0ull;
      for (int i = 0; i < nwg; ++i) t0 = std::min(t0, h[(size_t)i * 16]);
      fprintf(stderr, "chain16 stamps (us after the first workgroup's entry; min / median / max over %d workgroups):\n", nwg);
      for (int k = 0; k < 16; ++k) {
        std::vector<double> v;
        for (int i = 0; i < nwg; ++i) if (h[(size_t)i * 16 + k]) v.push_back((double)(h[(size_t)i * 16 + k] - t0) * 0.01);
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        fprintf(stderr, "  stamp %2d: %7.2f %7.2f %7.2f\n", k, v.front(), v[v.size() / 2], v.back());
      }
    }
  }
  return check_launch("chain16_kernel");
}

}  // namespace ms
