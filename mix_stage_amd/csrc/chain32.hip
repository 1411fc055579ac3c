// Chained pose decoder, fp32 (exact fp32 matrix products): decoder.0-3 (grouped Conv1d k3 + BatchNorm1d + LeakyReLU,
// JL:69-77,190-192) + the grouped 1x1 `logits` conv (JL:83,193) + the softmax mixture of the M sub-generators
// (JL:106-115,186-187,194) in ONE launch.
//
// Decomposition (clip-stationary): a workgroup owns ONE clip (T = 64 frames) of ONE sub-generator (group) through all four
// blocks.  A k3 / pad 1 conv never looks across a clip boundary, so the block's input -- all 256 (272) channels x 64 frames of
// that clip, 68 KB -- stays in LDS from layer to layer; only the batch statistics of BatchNorm cross workgroups (one meeting of
// the B workgroups of a group per layer, 2 floats per channel each).  B x M workgroups = 256 at the headline size: one per CU.
//
// K loop without a barrier: the four waves own 64 output channels each (2 x 2 accumulators of 32 x 32), so the weight operand
// of a wave is private to it -- it streams from HBM / L2 STRAIGHT INTO REGISTERS in MFMA operand order (prepared once per
// optimizer update: ms_decoder_chain_prepare; a lane's 16-byte load = its operand of four consecutive MFMAs), two blocks of
// 6 k-groups (= 5 us of MFMA work) ahead of their use; the activation operand is the resident LDS image
// [8-channel group][half][pixel 68][4 channels], one ds_read_b128 per four MFMAs and pixel block.  No LDS staging of weights,
// no s_barrier, no split reduction: the loop is v_mfma_f32_32x32x2_f32 back to back on one wave per SIMD.
//
// Inter-workgroup hand-offs (BatchNorm partials of a layer; the mixture's per-group terms) follow MI355X_MICROARCH.md,
// "Valid forms", table row 1: sc1 stores of the payload, every storing wave's s_waitcnt vmcnt(0), workgroup barrier, ONE
// agent-scope atomic add by one lane, sc1 poll by that lane, workgroup barrier, sc1 loads.  Counters are monotonic (a launch adds
// exactly `members` to each): no reset, no epoch argument, graph replays need nothing from the host.  Forward progress needs every
// workgroup of the launch resident at once (one per CU: checked by the launcher); the spin is bounded -- on expiry word 0 of the
// sync buffer is raised and the outputs are NaN.
#include <algorithm>

#include "kernels.h"
#include "conv16.h"

namespace ms {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
// 16-byte non-temporal store (HIP's float4 is a struct: the builtin wants a native vector)
__device__ __forceinline__ void nt_store4(float* dst, float a, float b, float c, float d) {
  __builtin_nontemporal_store(f32x4_t{a, b, c, d}, reinterpret_cast<f32x4_t*>(dst));
}

constexpr int CH_T = 64;                  // frames per clip = pixels per workgroup
constexpr int CH_C = 256;                 // channels per group of every chained block
constexpr int CH_NL = 4;                  // chained conv blocks
constexpr int CH_PITCH = 68;              // pixels per LDS plane: halo + 64 + halo + 2 (68 * 4 words = 16 mod 64 banks)
constexpr int CH_PLANE = CH_PITCH * 4;    // floats per plane [pixel][4 channels]
constexpr int CH_K8_0 = 34;               // 8-channel groups of the first block's input (257..272 channels)
constexpr int CH_BUF0 = CH_K8_0 * 2 * CH_PLANE;
constexpr int CH_BUF1 = 32 * 2 * CH_PLANE;
constexpr int CH_BLK = 6;                 // k-groups per stream block: 2 channel groups x 3 taps
constexpr int CH_NBLK0 = 17, CH_NBLK = 16;                                  // stream blocks of block 0 / blocks 1-3
constexpr int CH_CONV_BLOCKS = CH_NBLK0 + 3 * CH_NBLK;                      // 65
constexpr int CH_LOGIT_Q = 32;            // k-groups of the 1x1 logits conv (256 channels / 8)
constexpr size_t CH_WAVE_STREAM = (size_t)CH_CONV_BLOCKS * CH_BLK * 2 * 256 + (size_t)CH_LOGIT_Q * 256 + 2 * CH_BLK * 2 * 256;   // floats per (group, wave), + 2 blocks of slack for the ring's read-ahead
constexpr int CH_SPIN_LIMIT = 1 << 21;
constexpr int CH_PPAD = 128;              // rows of the logits tile (P <= 128)
constexpr int CH_MAXM = 32;               // groups (the softmax of a frame is formed in registers)
constexpr int CH_LDS_FLOATS = CH_BUF0 + CH_BUF1 + 3 * CH_NL * 256 + 2 * 256 + 64 + 8 + 2 * 256;

struct Chain32Args {
  const float* x;             // (B, cin0, 64)
  const float* wp;            // prepared weight streams [M][4][CH_WAVE_STREAM]
  const float* bias[CH_NL];
  const float* gamma[CH_NL];
  const float* beta[CH_NL];
  float* rm[CH_NL];
  float* rv[CH_NL];
  float* y_raw[CH_NL];        // (B, M*256, 64) or null
  float* y[CH_NL];            // (B, M*256, 64) or null
  float* save[CH_NL];         // 4 * M*256 or null
  const float* bias_l;        // (M*P)
  float* z;                   // (B, M*P, 64) or null
  const float* score;         // (B, M, 64)
  float* soft;                // (B, 64, M)
  float* out;                 // (B, 64, P)
  float* part;                // [NL][M][B][256][2]
  float* mixpart;             // [B][M][128][64]
  int* sync;                  // word 0: error flag
  int cnt_base;               // first counter word: [NL][M] layer meetings, then [B] clip meetings (each on a line of its own: stride 32)
  int B, M, P, cin0, train;
  int raw_all;                // train: y_raw for every channel (0: only where the backward pass cannot take x_hat from y: bn_inv_unsafe)
  float slope, eps, momentum;
};

__device__ __forceinline__ float f4e(const float4& v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }

__device__ __forceinline__ float4 ld_global_f4(const float4* p) { return *p; }

// one lane arrives for the workgroup and waits for `members` arrivals of this launch; returns false when the bound expired
__device__ __forceinline__ bool chain_meet(int* counter, int members, int* err_word, int code) {
  // a raised error word is STICKY: an expired launch may leave these counters out of step, so every later launch on them gives up
  // too (NaN outputs, which the optimizer refuses: adam_prep_seg_kernel) until the host has cleared the buffer.  The load travels
  // with the arrival: no extra round trip.
  const int prior = __hip_atomic_load(err_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned old = (unsigned)__hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned target = (old / (unsigned)members + 1u) * (unsigned)members;
  int spins = 0;
  while ((int)((unsigned)__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
    __builtin_amdgcn_s_sleep(2);
    if (++spins > CH_SPIN_LIMIT) {
      // word 0: which meeting gave up (diagnostic: 1 + block index, 9 = the mixture's clip meeting); word 1: the workgroup; word 2: what it saw
      __hip_atomic_store(err_word, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(err_word + 1, (int)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(err_word + 2, __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
  }
  return prior == 0;
}

__global__ __launch_bounds__(256, 1) void chain32_kernel(const Chain32Args p) {
  prefetch_kernargs<sizeof(Chain32Args)>();
  extern __shared__ float smem[];
  float* bufA = smem;
  float* bufB = bufA + CH_BUF0;
  // per-block parameter tables, filled once at the start (no global load sits between two blocks' K loops: a wait for it would
  // also wait for every weight load issued before it).  train: bias | gamma | beta; eval: bias | scale | shift
  float* tb0 = bufB + CH_BUF1;              // [NL][256]
  float* tb1 = tb0 + CH_NL * 256;
  float* tb2 = tb1 + CH_NL * 256;
  float* psc = tb2 + CH_NL * 256;           // [256] scale of the current block (train: after the meeting)
  float* psh = psc + 256;                   // [256] shift
  float* sg = psh + 256;                    // [64] this group's softmax weight per frame
  int* lflag = reinterpret_cast<int*>(sg + 64);   // [0]: a meeting expired
  float* pmn = sg + 64 + 8;                 // [256] batch mean of the current block (train) ...
  float* pin = pmn + 256;                   // [256] ... and 1 / std: what the y_raw test reads

  const int t = threadIdx.x, lane = t & 63, w = t >> 6, r = lane & 31, h = lane >> 5;
  const int g = blockIdx.x % p.M, b = blockIdx.x / p.M;
  const int C = p.M * CH_C;

  // ---- this wave's weight stream: 16 bytes per lane and (k-group, row block); two stream blocks ahead of their use
  const float4* ws = reinterpret_cast<const float4*>(p.wp + ((size_t)g * 4 + w) * CH_WAVE_STREAM) + lane;
  float4 ra0[CH_BLK][2], ra1[CH_BLK][2];
  // (their first loads go out behind the prologue's own loads, below: loads return in order, and the input image must not wait
  // for 50 KB of weights per wave)
  // ---- LDS: zero the halos (pixel slots 0 and 65..67 of every plane), stage the clip's input, this group's softmax weights
  if (t == 0) lflag[0] = 0;
  for (int e = t; e < (CH_K8_0 + 32) * 2 * 4; e += 256) {
    const int plane = e >> 2, s = e & 3;
    float* pl = (plane < CH_K8_0 * 2 ? bufA + plane * CH_PLANE : bufB + (plane - CH_K8_0 * 2) * CH_PLANE);
    *reinterpret_cast<float4*>(pl + (s == 0 ? 0 : 64 + s) * 4) = float4{0.f, 0.f, 0.f, 0.f};
  }
  {
    // every global load of the prologue goes out before the first LDS store (each wait would otherwise be a round trip of its own).
    // Input image: plane (k8, hh) = channels 8*k8 + 4*hh + 0..3; a wave takes every 4th plane, a lane one frame: coalesced row
    // loads, one 16-byte LDS store per plane
    constexpr int NPL = (CH_K8_0 * 2 + 3) / 4;                  // planes per wave
    const float* xb = p.x + (size_t)b * p.cin0 * CH_T + lane;
    float xv[NPL][4];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int pl = w + 4 * i;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = 4 * pl + j;
        xv[i][j] = xb[(size_t)min(c, p.cin0 - 1) * CH_T];
      }
    }
    float q0[CH_NL], q1[CH_NL], q2[CH_NL], q3[CH_NL], q4[CH_NL];
    const int cgp = g * CH_C + t;
#pragma unroll
    for (int l = 0; l < CH_NL; ++l) {
      q0[l] = p.bias[l] ? p.bias[l][cgp] : 0.f;
      q1[l] = p.gamma[l][cgp];
      q2[l] = p.beta[l][cgp];
      q3[l] = p.train ? 0.f : p.rm[l][cgp];
      q4[l] = p.train ? 1.f : p.rv[l][cgp];
    }
    float sv[CH_MAXM];
    {
      const float* sp = p.score + (size_t)b * p.M * CH_T + lane;
#pragma unroll
      for (int m = 0; m < CH_MAXM; ++m) sv[m] = sp[(size_t)min(m, p.M - 1) * CH_T];
    }
    // the weight ring's first fill, behind the loads above
#pragma unroll
  for (int u = 0; u < CH_BLK; ++u)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      ra0[u][mb] = ld_global_f4(ws + ((0 * CH_BLK + u) * 2 + mb) * 64);
      ra1[u][mb] = ld_global_f4(ws + ((1 * CH_BLK + u) * 2 + mb) * 64);
    }

#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int pl = w + 4 * i;
      if (pl < CH_K8_0 * 2) {
        float4 v;
        v.x = 4 * pl < p.cin0 ? xv[i][0] : 0.f;
        v.y = 4 * pl + 1 < p.cin0 ? xv[i][1] : 0.f;
        v.z = 4 * pl + 2 < p.cin0 ? xv[i][2] : 0.f;
        v.w = 4 * pl + 3 < p.cin0 ? xv[i][3] : 0.f;
        *reinterpret_cast<float4*>(bufA + pl * CH_PLANE + (1 + lane) * 4) = v;
      }
    }
#pragma unroll
    for (int l = 0; l < CH_NL; ++l) {
      tb0[l * 256 + t] = q0[l];
      if (p.train) { tb1[l * 256 + t] = q1[l]; tb2[l * 256 + t] = q2[l]; }
      else {
        const float sc = q1[l] * (1.0f / sqrtf(q4[l] + p.eps));
        tb1[l * 256 + t] = sc;
        tb2[l * 256 + t] = q2[l] - q3[l] * sc;
      }
    }
    if (t < CH_T) {
      // softmax over the M cluster scores of frame t (JL:186-187); group 0's workgroup of the clip writes the monitor tensor
      float mx = sv[0];
#pragma unroll
      for (int m = 1; m < CH_MAXM; ++m) mx = m < p.M ? fmaxf(mx, sv[m]) : mx;
      float den = 0.f, mine = 0.f;
#pragma unroll
      for (int m = 0; m < CH_MAXM; ++m) {
        sv[m] = m < p.M ? __expf(sv[m] - mx) : 0.f;
        den += sv[m];
        mine = m == g ? sv[m] : mine;
      }
      sg[t] = mine / den;
      if (g == 0 && p.soft) {
#pragma unroll
        for (int m = 0; m < CH_MAXM; ++m)
          if (m < p.M) p.soft[((size_t)b * CH_T + t) * p.M + m] = sv[m] / den;
      }
    }
  }
  // the leader of a group (clip 0) updates the running statistics: their old values, fetched now
  float rm_old[CH_NL], rv_old[CH_NL];
#pragma unroll
  for (int l = 0; l < CH_NL; ++l) {
    const bool lead = p.train && b == 0;
    rm_old[l] = lead ? p.rm[l][g * CH_C + t] : 0.f;
    rv_old[l] = lead ? p.rv[l][g * CH_C + t] : 0.f;
  }

  f32x16 acc[2][2];
  size_t blk = 0;                          // next stream block to be computed
  float* bin = bufA;
  float* bout = bufB;
  const int n0 = r;                        // this lane's pixel inside a 32-pixel block
  const __amdgpu_buffer_rsrc_t rsPart = buf_rsrc(p.part);

  // one stream block (2 channel groups x 3 taps) against the resident input image; afterwards its ring half is refilled with
  // stream block `blk + 2`
  auto run_block = [&](float4 (&ra)[CH_BLK][2], const float* bb, size_t refill) {
    float4 bf[2][2];
    auto fetch_b = [&](int u, float4 (&dst)[2]) {
      const int k8s = u / 3, tap = u - 3 * k8s;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
        dst[nb] = *reinterpret_cast<const float4*>(bb + (k8s * 2 + h) * CH_PLANE + (32 * nb + n0 + tap) * 4);
    };
    fetch_b(0, bf[0]);
#pragma unroll
    for (int u = 0; u < CH_BLK; ++u) {
      if (u + 1 < CH_BLK) fetch_b(u + 1, bf[(u + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4e(ra[u][mb], j), f4e(bf[u & 1][nb], j), acc[mb][nb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) ra[u][mb] = ld_global_f4(ws + ((refill * CH_BLK + u) * 2 + mb) * 64);
    }
  };

  for (int l = 0; l < CH_NL; ++l) {
    const float* pb = tb0 + l * 256;
    if (!p.train) { psc = tb1 + l * 256; psh = tb2 + l * 256; }
    if (l == 0) __syncthreads();           // the input image is complete
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[mb][nb][q] = 0.f;

    // ---- K loop: no barrier, no LDS traffic but the activation operand
    const float* bb = bin;
    if (l == 0) {                          // block 0 has 17 stream blocks: the odd one first
      run_block(ra0, bb, blk + 2);
      ++blk; bb += 4 * CH_PLANE;
    }
    for (int d = 0; d < CH_NBLK / 2; ++d) {
      run_block(ra1, bb, blk + 2);
      run_block(ra0, bb + 4 * CH_PLANE, blk + 3);
      blk += 2; bb += 8 * CH_PLANE;
    }
    __syncthreads();                       // pb / psc / psh of this block are visible

    // ---- epilogue.  conv + bias goes into the output image; from there every thread takes 4 channels x 16 frames
    // (plane pl = t >> 2, frames 16*pq ..): the clip's statistics, y_raw and y leave as whole 16-byte vectors along the frames
    // (a quad of lanes = one 256-byte channel row), and the activated values overwrite the image in place.
    const int wc = 64 * w;                 // this wave's first channel inside the group
    float* yr = p.y_raw[l];
    float* yo = p.y[l];
    const size_t gbase = ((size_t)b * C + (size_t)g * CH_C) * CH_T;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int c0 = wc + 32 * mb + 8 * rq + 4 * h;
        const float4 bs = *reinterpret_cast<const float4*>(pb + c0);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const float4 v = {acc[mb][nb][4 * rq] + bs.x, acc[mb][nb][4 * rq + 1] + bs.y, acc[mb][nb][4 * rq + 2] + bs.z,
                            acc[mb][nb][4 * rq + 3] + bs.w};
          *reinterpret_cast<float4*>(bout + ((c0 >> 3) * 2 + h) * CH_PLANE + (1 + 32 * nb + n0) * 4) = v;
        }
      }
    __syncthreads();
    // thread (plane pl, quarter pq) owns frames 16*k + 4*pq + f (k, f = 0..3) of the plane's 4 channels: a quad of lanes then
    // writes 64 contiguous bytes of a channel row per store instruction (whole 64-byte sectors), 16 rows per wave instruction
    const int pl = t >> 2, pq = t & 3;
    float* img = bout + pl * CH_PLANE + (1 + 4 * pq) * 4;
    float4 v[16];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int f = 0; f < 4; ++f) v[4 * k + f] = *reinterpret_cast<const float4*>(img + (16 * k + f) * 4);
    // rows of channel 4*pl + j.  Non-temporal: nothing reads these tensors before the backward pass, and the weight streams
    // should keep the L2s.
    auto store_rows = [&](float* dst_base) {
      float* dst = dst_base + gbase + (size_t)(4 * pl) * CH_T + 4 * pq;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        nt_store4(dst + 16 * k, v[4 * k].x, v[4 * k + 1].x, v[4 * k + 2].x, v[4 * k + 3].x);
        nt_store4(dst + CH_T + 16 * k, v[4 * k].y, v[4 * k + 1].y, v[4 * k + 2].y, v[4 * k + 3].y);
        nt_store4(dst + 2 * CH_T + 16 * k, v[4 * k].z, v[4 * k + 1].z, v[4 * k + 2].z, v[4 * k + 3].z);
        nt_store4(dst + 3 * CH_T + 16 * k, v[4 * k].w, v[4 * k + 1].w, v[4 * k + 2].w, v[4 * k + 3].w);
      }
    };
    auto store_row = [&](float* dst_base, int j) {           // one channel (4*pl + j) of the plane
      float* dst = dst_base + gbase + (size_t)(4 * pl + j) * CH_T + 4 * pq;
#pragma unroll
      for (int k = 0; k < 4; ++k) nt_store4(dst + 16 * k, f4e(v[4 * k], j), f4e(v[4 * k + 1], j), f4e(v[4 * k + 2], j), f4e(v[4 * k + 3], j));
    };
    if (p.train) {
      // statistics of this clip: two passes over registers (mean, then M2 about it), the 4 lanes of a plane combined by DPP
      {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[0] += v[i].x; s[1] += v[i].y; s[2] += v[i].z; s[3] += v[i].w; }
        float mean[4], m2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s[j] += lane_xor1(s[j]);
          s[j] += lane_xor2(s[j]);
          mean[j] = s[j] * (1.0f / CH_T);
          m2[j] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float d0 = v[i].x - mean[0], d1 = v[i].y - mean[1], d2 = v[i].z - mean[2], d3 = v[i].w - mean[3];
          m2[0] = fmaf(d0, d0, m2[0]); m2[1] = fmaf(d1, d1, m2[1]); m2[2] = fmaf(d2, d2, m2[2]); m2[3] = fmaf(d3, d3, m2[3]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          m2[j] += lane_xor1(m2[j]);
          m2[j] += lane_xor2(m2[j]);
        }
        if (pq == 0) {
          // channels 4*pl .. 4*pl+3 of this group: (mean, M2) pairs, 32 contiguous bytes, written through (sc1)
          const unsigned off = 8u * (unsigned)((((l * p.M + g) * p.B + b) * CH_C) + 4 * pl);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, float4{mean[0], m2[0], mean[1], m2[1]}), rsPart, (int)off, 0, 16);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, float4{mean[2], m2[2], mean[3], m2[3]}), rsPart, (int)(off + 16u), 0, 16);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave: its partials have left
      __syncthreads();
      // conv + bias for the backward pass.  Layers whose backward is the one-launch form read the block's OUTPUT instead wherever
      // the BatchNorm + LeakyReLU map inverts safely (elementwise.hip: bn_bwd_fused*): only the other channels' rows are kept
      // (below, once the statistics are known).  Larger batches keep every row: stored now, draining while the group meets.
      if (yr && p.raw_all && t != 0) store_rows(yr);
      if (t == 0) {
        if (!chain_meet(p.sync + p.cnt_base + 32 * (l * p.M + g), p.B, p.sync, 1 + l)) lflag[0] = 1;
        if (yr && p.raw_all) store_rows(yr);
      }
      __syncthreads();
      {
        // thread = channel: the B clips' partials in clip order (Chan, fp64): every workgroup of the group computes the same bits
        const unsigned base = 8u * (unsigned)(((l * p.M + g) * p.B) * CH_C + t);
        // equal counts (64 frames per clip): mean = average of the clips' means, M2 = sum of their M2 + 64 * sum (mean_i - mean)^2
        // -- two passes over the partials in clip order, fp64, no division in the loops
        double msum = 0.0, m2 = 0.0, dev = 0.0;
        float2 pv[32];
        const int nb32 = min(p.B, 32);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          const int bb2 = min(i, p.B - 1);
          pv[i] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rsPart, (int)(base + 8u * (unsigned)(bb2 * CH_C)), 0, 16));
        }
#pragma unroll
        for (int i = 0; i < 32; ++i)
          if (i < nb32) { msum += (double)pv[i].x; m2 += (double)pv[i].y; }
        for (int bb2 = 32; bb2 < p.B; ++bb2) {
          const float2 q = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rsPart, (int)(base + 8u * (unsigned)(bb2 * CH_C)), 0, 16));
          msum += (double)q.x; m2 += (double)q.y;
        }
        const double n = (double)p.B * CH_T, mean = msum / (double)p.B;
#pragma unroll
        for (int i = 0; i < 32; ++i)
          if (i < nb32) { const double dl = (double)pv[i].x - mean; dev += dl * dl; }
        for (int bb2 = 32; bb2 < p.B; ++bb2) {
          const float2 q = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rsPart, (int)(base + 8u * (unsigned)(bb2 * CH_C)), 0, 16));
          const double dl = (double)q.x - mean; dev += dl * dl;
        }
        m2 += (double)CH_T * dev;
        const int cg = g * CH_C + t;
        const float var = (float)(m2 / n), fmean = (float)mean;
        const float invstd = 1.0f / sqrtf(var + p.eps);
        float sc = tb1[l * 256 + t] * invstd;
        float sh = tb2[l * 256 + t] - fmean * sc;
        if (lflag[0]) { sc = __builtin_nanf(""); sh = sc; }
        else if (b == 0) {
          if (p.save[l]) { float* sv = p.save[l]; sv[cg] = fmean; sv[C + cg] = invstd; sv[2 * C + cg] = sc; sv[3 * C + cg] = sh; }
          const float unbiased = n > 1.0 ? (float)(m2 / (n - 1.0)) : var;
          running_stats_update(&p.rm[l][cg], &p.rv[l][cg], rm_old[l], rv_old[l], p.momentum, fmean, unbiased);
        }
        psc[t] = sc;
        psh[t] = sh;
        pmn[t] = fmean;
        pin[t] = invstd;
      }
      __syncthreads();
      if (yr && !p.raw_all) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int cj = 4 * pl + j;
          if (bn_inv_unsafe(pmn[cj], pin[cj], psc[cj], psh[cj], p.slope)) store_row(yr, j);
        }
      }
    }
    // normalise + activate the 4 x 16 values in registers: into the image in place (the next block's input) and to HBM
    {
      const float4 sc = *reinterpret_cast<const float4*>(psc + 4 * pl);
      const float4 sh = *reinterpret_cast<const float4*>(psh + 4 * pl);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        v[i].x = lrelu(fmaf(v[i].x, sc.x, sh.x), p.slope);
        v[i].y = lrelu(fmaf(v[i].y, sc.y, sh.y), p.slope);
        v[i].z = lrelu(fmaf(v[i].z, sc.z, sh.z), p.slope);
        v[i].w = lrelu(fmaf(v[i].w, sc.w, sh.w), p.slope);
        *reinterpret_cast<float4*>(img + (16 * (i >> 2) + (i & 3)) * 4) = v[i];
      }
      if (yo) store_rows(yo);
    }
    __syncthreads();                       // the next block's input image is complete; pb / psc / psh may be rewritten
    float* tmp = bin; bin = bout; bout = tmp;
    if (l == 0) bout = bufA;               // (bufA holds 34 channel groups, bufB 32: from block 1 on either serves)
  }

  // ---- logits (1x1, P rows of this group) + this group's term of the mixture
  {
    const float4* wl = ws + (size_t)CH_CONV_BLOCKS * CH_BLK * 2 * 64;
    f32x16 za[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int q = 0; q < 16; ++q) za[nb][q] = 0.f;
    float4 wv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) wv[i] = ld_global_f4(wl + i * 64);
    for (int q0 = 0; q0 < CH_LOGIT_Q; q0 += 8) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float4 bfr[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          bfr[nb] = *reinterpret_cast<const float4*>(bin + ((q0 + i) * 2 + h) * CH_PLANE + (1 + 32 * nb + n0) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) za[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4e(wv[i], j), f4e(bfr[nb], j), za[nb], 0, 0, 0);
        wv[i] = ld_global_f4(wl + (size_t)(min(q0 + 8 + i, CH_LOGIT_Q - 1)) * 64);
      }
    }
    // rows p = 32*w + 8*(q>>2) + 4*h + (q&3); z = acc + bias goes through LDS (bout is free) so that z and this group's mixture
    // term s_g[t] * z leave as whole 16-byte vectors along the frames
    float* mt = bout;                      // [128 rows][64 frames]
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int prow = 32 * w + 8 * (q >> 2) + 4 * h + (q & 3);
      const float bl = prow < p.P ? p.bias_l[g * p.P + prow] : 0.f;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) mt[prow * CH_T + 32 * nb + n0] = za[nb][q] + bl;
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsMix = buf_rsrc(p.mixpart);
    {
      const unsigned mbase = 4u * (unsigned)(((b * p.M + g) * CH_PPAD) * CH_T);
      for (int e = t; e < p.P * (CH_T / 4); e += 256) {
        float4 v = *reinterpret_cast<const float4*>(mt + 4 * e);
        if (p.z) nt_store4(p.z + ((size_t)b * p.M * p.P + (size_t)g * p.P) * CH_T + 4 * e, v.x, v.y, v.z, v.w);
        const float4 sw = *reinterpret_cast<const float4*>(sg + 4 * (e & 15));
        v = float4{v.x * sw.x, v.y * sw.y, v.z * sw.z, v.w * sw.w};
        if (lflag[0]) v = float4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rsMix, (int)(mbase + 16u * (unsigned)e), 0, 16);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
      if (!chain_meet(p.sync + p.cnt_base + 32 * (CH_NL * p.M + b), p.M, p.sync, 9)) lflag[0] = 1;
    }
    __syncthreads();
    // this workgroup sums frames [t0, t1) of its clip over the M groups, in group order (JL:106-115)
    const int t0 = (g * CH_T / p.M) & ~3, t1 = g + 1 == p.M ? CH_T : (((g + 1) * CH_T / p.M) & ~3);
    const int nq = (t1 - t0) >> 2;
    for (int e = t; e < nq * p.P; e += 256) {
      const int tq = e / p.P, pp = e - tq * p.P;
      float4 s = {0.f, 0.f, 0.f, 0.f};
      for (int m0 = 0; m0 < p.M; m0 += 8) {              // 8 groups' terms in flight, added in group order
        float4 v4[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const unsigned off = 4u * (unsigned)((((b * p.M + min(m0 + i, p.M - 1)) * CH_PPAD) + pp) * CH_T + t0 + 4 * tq);
          v4[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsMix, (int)off, 0, 16));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (m0 + i < p.M) { s.x += v4[i].x; s.y += v4[i].y; s.z += v4[i].z; s.w += v4[i].w; }
      }
      if (lflag[0]) s = float4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
      float* o = p.out + ((size_t)b * CH_T + t0 + 4 * tq) * p.P + pp;
      o[0] = s.x; o[p.P] = s.y; o[2 * p.P] = s.z; o[3 * p.P] = s.w;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// One grouped k3 s1 p1 conv, 256 -> 256 channels per group, T = 64, bias-free, with the same recipe (a workgroup = one clip of one
// group, all 256 output rows; input image resident in LDS; weights streamed into registers; no barrier in the K loop): the DATA
// GRADIENT of decoder.1-3 (what autograd derives for layers.py:78 / JL:69-77: dx = conv(dy_raw, w transposed, taps reversed)).
// Independent workgroups: no meeting, any batch size.
constexpr int GD_NBLK = 16;
constexpr size_t GD_WAVE_STREAM = (size_t)(GD_NBLK + 2) * CH_BLK * 2 * 256;      // floats per (group, wave): 16 blocks + 2 of slack
struct Gconv32Args {
  const float* x;      // (B, M*256, 64)
  const float* wp;     // [M][4][GD_WAVE_STREAM]
  float* y;            // (B, M*256, 64)
  int B, M;
};

__global__ __launch_bounds__(256, 1) void gconv32_kernel(const Gconv32Args p) {
  extern __shared__ float smem[];
  float* img = smem;                         // [32 groups][2][68][4]
  float* oimg = img + CH_BUF1;               // the output tile in the same layout
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, n0 = lane & 31, h = lane >> 5;
  const int g = blockIdx.x % p.M, b = blockIdx.x / p.M;
  const int C = p.M * CH_C;
  const size_t gbase = ((size_t)b * C + (size_t)g * CH_C) * CH_T;
  // input image: every load first, then the weight ring's first fill behind them
  constexpr int NPL = 16;                    // planes per wave (64 planes, a wave takes every 4th)
  float xv[NPL][4];
  {
    const float* xb = p.x + gbase + lane;
#pragma unroll
    for (int i = 0; i < NPL; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) xv[i][j] = xb[(size_t)(4 * (w + 4 * i) + j) * CH_T];
  }
  const float4* ws = reinterpret_cast<const float4*>(p.wp + ((size_t)g * 4 + w) * GD_WAVE_STREAM) + lane;
  float4 ra0[CH_BLK][2], ra1[CH_BLK][2];
#pragma unroll
  for (int u = 0; u < CH_BLK; ++u)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      ra0[u][mb] = ld_global_f4(ws + ((0 * CH_BLK + u) * 2 + mb) * 64);
      ra1[u][mb] = ld_global_f4(ws + ((1 * CH_BLK + u) * 2 + mb) * 64);
    }
  for (int e = t; e < 64 * 4; e += 256) {
    const int plane = e >> 2, sl = e & 3;
    *reinterpret_cast<float4*>(img + plane * CH_PLANE + (sl == 0 ? 0 : 64 + sl) * 4) = float4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int i = 0; i < NPL; ++i)
    *reinterpret_cast<float4*>(img + (w + 4 * i) * CH_PLANE + (1 + lane) * 4) = float4{xv[i][0], xv[i][1], xv[i][2], xv[i][3]};
  __syncthreads();

  f32x16 acc[2][2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[mb][nb][q] = 0.f;
  auto run_block = [&](float4 (&ra)[CH_BLK][2], const float* bb, size_t refill) {
    float4 bf[2][2];
    auto fetch_b = [&](int u, float4 (&dst)[2]) {
      const int k8s = u / 3, tap = u - 3 * k8s;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
        dst[nb] = *reinterpret_cast<const float4*>(bb + (k8s * 2 + h) * CH_PLANE + (32 * nb + n0 + tap) * 4);
    };
    fetch_b(0, bf[0]);
#pragma unroll
    for (int u = 0; u < CH_BLK; ++u) {
      if (u + 1 < CH_BLK) fetch_b(u + 1, bf[(u + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4e(ra[u][mb], j), f4e(bf[u & 1][nb], j), acc[mb][nb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) ra[u][mb] = ld_global_f4(ws + ((refill * CH_BLK + u) * 2 + mb) * 64);
    }
  };
  {
    const float* bb = img;
    size_t blk = 0;
    for (int d = 0; d < GD_NBLK / 2; ++d) {
      run_block(ra0, bb, blk + 2);
      run_block(ra1, bb + 4 * CH_PLANE, blk + 3);
      blk += 2; bb += 8 * CH_PLANE;
    }
  }
  // output through LDS: whole 16-byte vectors along the frames (thread = 4 channels x 16 frames, as in the chained kernel)
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int c0 = 64 * w + 32 * mb + 8 * rq + 4 * h;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
        *reinterpret_cast<float4*>(oimg + ((c0 >> 3) * 2 + h) * CH_PLANE + (1 + 32 * nb + n0) * 4) =
            float4{acc[mb][nb][4 * rq], acc[mb][nb][4 * rq + 1], acc[mb][nb][4 * rq + 2], acc[mb][nb][4 * rq + 3]};
    }
  __syncthreads();
  const int pl = t >> 2, pq = t & 3;
  const float* src = oimg + pl * CH_PLANE + (1 + 4 * pq) * 4;
  float* dst = p.y + gbase + (size_t)(4 * pl) * CH_T + 4 * pq;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float4 v[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) v[f] = *reinterpret_cast<const float4*>(src + (16 * k + f) * 4);
    *reinterpret_cast<float4*>(dst + 16 * k) = float4{v[0].x, v[1].x, v[2].x, v[3].x};
    *reinterpret_cast<float4*>(dst + CH_T + 16 * k) = float4{v[0].y, v[1].y, v[2].y, v[3].y};
    *reinterpret_cast<float4*>(dst + 2 * CH_T + 16 * k) = float4{v[0].z, v[1].z, v[2].z, v[3].z};
    *reinterpret_cast<float4*>(dst + 3 * CH_T + 16 * k) = float4{v[0].w, v[1].w, v[2].w, v[3].w};
  }
}

// transposed stream: rows = input channels ci of the group, reduction over (co, tap) with the taps reversed:
// A(ci, (co, tap)) = w[g*256 + co][ci][2 - tap].  One workgroup per (group, wave, row block, half of the co range): per co the 32
// rows x 3 taps are 96 contiguous floats of the weight tensor.
enum { GD_PREP_MAX = 8 };
struct GdPrepBatch { int n; const float* w[GD_PREP_MAX]; float* out[GD_PREP_MAX]; };
__global__ __launch_bounds__(256) void gconv32_prep_kernel(const GdPrepBatch pb, int M) {
  __shared__ float lds[128 * 97];
  const float* w = pb.w[blockIdx.y];
  float* out = pb.out[blockIdx.y];
  const int t = threadIdx.x;
  int id = blockIdx.x;
  const int kh = id & 1; id >>= 1;
  const int mb = id & 1; id >>= 1;
  const int wv = id & 3; id >>= 2;
  const int g = id;
  const int ci0 = 64 * wv + 32 * mb;
  for (int e = t; e < 128 * 96; e += 256) {
    const int col = e / 96, k = e - col * 96;
    lds[col * 97 + k] = w[((size_t)(g * CH_C + 128 * kh + col) * CH_C + ci0) * 3 + k];
  }
  __syncthreads();
  float4* dst = reinterpret_cast<float4*>(out) + ((size_t)g * 4 + wv) * (GD_WAVE_STREAM / 4);
  const int lane = t & 63, rr = lane & 31, h = lane >> 5;
  for (int qi = t >> 6; qi < 16 * 3; qi += 4) {
    const int k8l = qi / 3, tap = qi - 3 * k8l, k8 = 16 * kh + k8l;
    float vv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) vv[j] = lds[(8 * k8l + 4 * h + j) * 97 + rr * 3 + (2 - tap)];
    dst[((size_t)((k8 >> 1) * CH_BLK + (k8 & 1) * 3 + tap) * 2 + mb) * 64 + lane] = float4{vv[0], vv[1], vv[2], vv[3]};
  }
}

bool gdgrad32_ok(const ms_conv_desc* d) {
  return (d->dtype & 0xff) == 0 && d->H == 1 && d->KH == 1 && d->KW == 3 && d->SW == 1 && d->PW == 1 && d->W == CH_T && d->groups > 1 &&
         d->Cin == CH_C && d->Cout == CH_C && d->in_mode == MS_IN_PLAIN;
}
size_t gdgrad32_weight_floats(const ms_conv_desc* d) { return gdgrad32_ok(d) ? (size_t)d->groups * 4 * GD_WAVE_STREAM : 0; }
// queued: the blocks of one ms_dgrad_weights_prepare call (equal group counts) share a launch
static GdPrepBatch g_gd_prep;
static int g_gd_prep_groups = 0;
void gdgrad32_prep_discard() { g_gd_prep.n = 0; }
int gdgrad32_prep_flush(hipStream_t s) {
  if (!g_gd_prep.n) return 0;
  const GdPrepBatch pb = g_gd_prep;
  const int groups = g_gd_prep_groups;
  g_gd_prep.n = 0;
  TimingScope ts(s, 0, 0, "gconv32_prep_kernel|gdgrad_prep g%d jobs%d", groups, pb.n);
  if (ts.skip()) return 0;
  hipLaunchKernelGGL(gconv32_prep_kernel, dim3(groups * 4 * 2 * 2, pb.n), dim3(256), 0, s, pb, groups);
  return check_launch("gconv32_prep_kernel");
}
int gdgrad32_prepare(const ms_conv_desc* d, const float* w, float* out, hipStream_t s) {
  if (g_gd_prep.n && (g_gd_prep.n == GD_PREP_MAX || g_gd_prep_groups != d->groups)) { const int rc = gdgrad32_prep_flush(s); if (rc) return rc; }
  g_gd_prep_groups = d->groups;
  g_gd_prep.w[g_gd_prep.n] = w;
  g_gd_prep.out[g_gd_prep.n] = out;
  ++g_gd_prep.n;
  return 0;
}
int gdgrad32_launch(const ms_conv_desc* d, const float* g, const float* wp, float* dx, hipStream_t s) {
  static unsigned long long attr_done = 0;
  const int lds = 2 * CH_BUF1 * (int)sizeof(float);
  if (first_time_on_device(attr_done)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gconv32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return set_error("gconv32: cannot raise the dynamic LDS limit");
    done_on_device(attr_done);
  }
  Gconv32Args a = {g, wp, dx, d->B, d->groups};
  const double flops = 2.0 * CH_C * CH_C * 3.0 * d->B * CH_T * d->groups;
  const double bytes = 4.0 * (2.0 * d->B * d->groups * CH_C * CH_T + (double)d->groups * CH_C * CH_C * 3);
  TimingScope ts(s, flops, bytes, "gconv32_kernel|conv_dgrad_gclip k1x3 s1 g%d B%d", d->groups, d->B);
  if (ts.skip()) return 0;
  hipLaunchKernelGGL(gconv32_kernel, dim3(d->B * d->groups), dim3(256), lds, s, a);
  return check_launch("gconv32_kernel");
}

// ---------------------------------------------------------------------------------------------
// weight streams.  One thread per 16-byte vector: [group][wave][stream position][row block][lane] = 4 consecutive reduction
// channels 8*k8 + 4*(lane>>5) + 0..3 at one tap, of output row 64*wave + 32*mb + (lane & 31)
struct ChainPrepArgs {
  const float* w[CH_NL];     // (M*256, cin_l, 3)
  const float* wl;           // (M*P, 256, 1)
  float* out;
  int M, P, cin0;
};

// Conv blocks: one workgroup per (block l, group, wave, row block mb, channel half): its 32 weight rows x 128 (144) channels x 3
// taps are read as they lie in the weight tensor (contiguous row segments, coalesced) into LDS and leave in stream order (1 KB per
// k-group).  The logits rows (tiny) are gathered directly.
constexpr int CHP_ROWP = 433;            // LDS row pitch (odd: the 32 rows of a k-group read hit 32 banks)
__global__ __launch_bounds__(256) void chain32_prep_kernel(const ChainPrepArgs p) {
  __shared__ float lds[32 * CHP_ROWP];
  const int t = threadIdx.x;
  const int n_conv = CH_NL * p.M * 4 * 2 * 2;
  const size_t per_wave = CH_WAVE_STREAM / 4;                 // float4 per (group, wave)
  float4* out4 = reinterpret_cast<float4*>(p.out);
  if ((int)blockIdx.x < n_conv) {
    int id = blockIdx.x;
    const int kh = id & 1; id >>= 1;
    const int mb = id & 1; id >>= 1;
    const int wv = id & 3; id >>= 2;
    const int g = id % p.M, l = id / p.M;
    const int cin = l == 0 ? p.cin0 : CH_C;
    const int c_lo = 128 * kh, c_hi = kh ? (l == 0 ? 8 * CH_K8_0 : CH_C) : 128;      // channel range (padded to whole groups)
    const int nfl = (min(c_hi, cin) - c_lo) * 3;                                     // floats per row actually present
    const int row0 = g * CH_C + 64 * wv + 32 * mb;
    {
      // 8 rows in flight per thread and pass, consecutive lanes = consecutive floats of a row (no division in the loop)
      const float* src = p.w[l] + ((size_t)row0 * cin + c_lo) * 3;
      for (int k = t; k < nfl; k += 256) {
#pragma unroll
        for (int r0 = 0; r0 < 32; r0 += 8) {
          float tmp[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) tmp[i] = src[(size_t)(r0 + i) * cin * 3 + k];
#pragma unroll
          for (int i = 0; i < 8; ++i) lds[(r0 + i) * CHP_ROWP + k] = tmp[i];
        }
      }
    }
    __syncthreads();
    const int nk8 = (c_hi - c_lo) / 8;                                               // 16 (18 for block 0's upper half)
    const int blk0 = l == 0 ? 0 : CH_NBLK0 + (l - 1) * CH_NBLK;
    float4* dst = out4 + ((size_t)g * 4 + wv) * per_wave;
    const int lane = t & 63, rr = lane & 31, h = lane >> 5;
    for (int qi = t >> 6; qi < nk8 * 3; qi += 4) {
      const int k8l = qi / 3, tap = qi - 3 * k8l;
      const int k8 = c_lo / 8 + k8l;
      const int cl = 8 * k8l + 4 * h;                                                // channel inside the staged range
      float vv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) vv[j] = (c_lo + cl + j < cin) ? lds[rr * CHP_ROWP + (cl + j) * 3 + tap] : 0.f;
      const int blkn = blk0 + (k8 >> 1), u = (k8 & 1) * 3 + tap;
      dst[((size_t)(blkn * CH_BLK + u) * 2 + mb) * 64 + lane] = float4{vv[0], vv[1], vv[2], vv[3]};
    }
    return;
  }
  // logits: [group][wave][k-group q][lane] behind the conv stream
  const size_t gid = (size_t)(blockIdx.x - n_conv) * 256 + t;
  if (gid >= (size_t)p.M * 4 * CH_LOGIT_Q * 64) return;
  const int lane = (int)(gid & 63), q = (int)((gid >> 6) % CH_LOGIT_Q), gw = (int)(gid / (CH_LOGIT_Q * 64));
  const int g = gw >> 2, wv = gw & 3, h = lane >> 5, prow = 32 * wv + (lane & 31);
  float4 v = {0.f, 0.f, 0.f, 0.f};
  if (prow < p.P) {
    const float* src = p.wl + ((size_t)(g * p.P + prow)) * CH_C + 8 * q + 4 * h;
    v = float4{src[0], src[1], src[2], src[3]};
  }
  out4[(size_t)gw * per_wave + (size_t)CH_CONV_BLOCKS * CH_BLK * 2 * 64 + (size_t)q * 64 + lane] = v;
}

static int chain32_shape_ok(const ms_chain_desc* d) {
  return d && d->T == CH_T && d->C == CH_C && d->n_blocks == CH_NL && d->cin0 > CH_C && d->cin0 <= 8 * CH_K8_0 && d->P >= 1 &&
         d->P <= CH_PPAD && d->B >= 1 && d->M >= 1 && d->M <= CH_MAXM && (d->mode == MS_BN_TRAIN || d->mode == MS_BN_EVAL) && d->dtype == MS_F32;
}

size_t chain32_prepared_bytes(const ms_chain_desc* d) { return (size_t)d->M * 4 * CH_WAVE_STREAM * sizeof(float); }

size_t chain32_workspace(const ms_chain_desc* d) {
  return align_up((size_t)CH_NL * d->M * d->B * CH_C * 2 * sizeof(float), 256) +
         align_up((size_t)d->B * d->M * CH_PPAD * CH_T * sizeof(float), 256) + 256;
}

int chain32_sync_words(const ms_chain_desc* d) { return 32 * (CH_NL * d->M + d->B + 1); }

int chain32_supported(const ms_chain_desc* d) {
  if (!chain32_shape_ok(d)) return 0;
  // (per device: the CU count and the raised LDS limit belong to the device that is current now)
  const int cus = current_device_cus();
  static unsigned long long lds_done = 0;
  if (!cus) return 0;
  if (first_time_on_device(lds_done)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(chain32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            CH_LDS_FLOATS * (int)sizeof(float)) != hipSuccess) return 0;
    done_on_device(lds_done);
  }
  // train mode: the workgroups of a group meet inside the launch -- all of them must be resident at once (one per CU: 146 KB LDS)
  if (d->mode == MS_BN_TRAIN && d->B * d->M > cus) return 0;
  // (eval: the mixture's clip meeting has the same requirement)
  if (d->B * d->M > cus) return 0;
  return 1;
}

int chain32_prepare(const ms_chain_desc* d, const float* const* w, const float* wl, void* prepared, hipStream_t s) {
  if (!chain32_shape_ok(d)) return set_error("ms_decoder_chain_prepare: unsupported shape");
  ChainPrepArgs a = {};
  for (int l = 0; l < CH_NL; ++l) a.w[l] = w[l];
  a.wl = wl; a.out = (float*)prepared; a.M = d->M; a.P = d->P; a.cin0 = d->cin0;
  const size_t vecs = (size_t)d->M * 4 * (CH_WAVE_STREAM / 4);
  TimingScope ts(s, 0, 16.0 * vecs, "chain32_prep_kernel|chain_prep f32 M%d", d->M);
  if (ts.skip()) return 0;
  const int n_conv = CH_NL * d->M * 4 * 2 * 2, n_log = (d->M * 4 * CH_LOGIT_Q * 64 + 255) / 256;
  hipLaunchKernelGGL(chain32_prep_kernel, dim3(n_conv + n_log), dim3(256), 0, s, a);
  return check_launch("chain32_prep_kernel");
}

int chain32_fwd(const ms_chain_desc* d, const ms_chain_tensors* tn, void* workspace, size_t workspace_bytes, hipStream_t s) {
  if (!chain32_supported(d)) return set_error("ms_decoder_chain_fwd: shape / device not supported (ms_decoder_chain_supported)");
  if (!tn || !tn->x || !tn->score || !tn->out || !tn->prepared || !tn->sync || !tn->w_logits)
    return set_error("ms_decoder_chain_fwd: null tensor");
  if (workspace_bytes < chain32_workspace(d) || !workspace) return set_error("ms_decoder_chain_fwd: workspace too small");
  if (tn->sync_words < d->sync_first_word + chain32_sync_words(d)) return set_error("ms_decoder_chain_fwd: sync buffer too small");
  Chain32Args a = {};
  a.x = (const float*)tn->x; a.wp = (const float*)tn->prepared;
  for (int l = 0; l < CH_NL; ++l) {
    if (!tn->gamma[l] || !tn->beta[l] || !tn->running_mean[l] || !tn->running_var[l]) return set_error("ms_decoder_chain_fwd: BN tensors missing");
    a.bias[l] = tn->bias[l]; a.gamma[l] = tn->gamma[l]; a.beta[l] = tn->beta[l]; a.rm[l] = tn->running_mean[l]; a.rv[l] = tn->running_var[l];
    a.y_raw[l] = (float*)tn->y_raw[l]; a.y[l] = (float*)tn->y[l]; a.save[l] = tn->save[l];
  }
  a.bias_l = tn->bias_logits; a.z = tn->z; a.score = tn->score; a.soft = tn->soft; a.out = tn->out;
  a.part = (float*)workspace;
  a.mixpart = (float*)((char*)workspace + align_up((size_t)CH_NL * d->M * d->B * CH_C * 2 * sizeof(float), 256));
  a.sync = tn->sync; a.cnt_base = d->sync_first_word + 32;
  a.B = d->B; a.M = d->M; a.P = d->P; a.cin0 = d->cin0; a.train = d->mode == MS_BN_TRAIN;
  a.raw_all = (d->keep_all_raw || (long)d->B * CH_T > BN_BWD32_FUSED_MAX) ? 1 : 0;      // (its two-pass backward reads y_raw, not y)
  a.slope = d->slope; a.eps = d->eps; a.momentum = d->momentum;
  const double bt = (double)d->B * CH_T;
  const double flops = 2.0 * bt * d->M * (CH_C * 3.0 * (d->cin0 + 3.0 * CH_C) + (double)d->P * CH_C);
  // algorithmic bytes of the unit (163.6 MB in fp32 at B*T = 2048, M = 8; linear in both)
  const double bytes = 163.6e6 * (bt / 2048.0) * (d->M / 8.0);
  TimingScope ts(s, flops, bytes, "chain32_kernel|decoder_chain_fwd f32 M%d B%d P%d cin%d %s", d->M, d->B, d->P, d->cin0, a.train ? "train" : "eval");
  if (ts.skip()) return 0;
  // (meeting counters are monotonic: a launch adds exactly `members` to each one it uses.  Launches of DIFFERENT (B, M) must not
  // share counter words -- the caller gives every shape its own words: include/mixstage.h)
  hipLaunchKernelGGL(chain32_kernel, dim3(d->B * d->M), dim3(256), CH_LDS_FLOATS * sizeof(float), s, a);
  return check_launch("chain32_kernel");
}

}  // namespace ms

using namespace ms;
static inline bool chain_is16(const ms_chain_desc* d) { return d->dtype == MS_BF16 || d->dtype == MS_F16; }
extern "C" {
int ms_decoder_chain_supported(const ms_chain_desc* d) { return !d ? 0 : chain_is16(d) ? chain16_supported(d) : chain32_supported(d); }
size_t ms_decoder_chain_prepared_bytes(const ms_chain_desc* d) {
  return !d ? 0 : chain_is16(d) ? chain16_prepared_bytes(d) : d->dtype == MS_F32 ? chain32_prepared_bytes(d) : 0;
}
size_t ms_decoder_chain_workspace(const ms_chain_desc* d) { return !d ? 256 : chain_is16(d) ? chain16_workspace(d) : chain32_workspace(d); }
int ms_decoder_chain_sync_words(const ms_chain_desc* d) { return !d ? 0 : chain_is16(d) ? chain16_sync_words(d) : chain32_sync_words(d); }
int ms_decoder_chain_prepare(const ms_chain_desc* d, const float* const* w, const float* w_logits, void* prepared, void* stream) {
  if (!d || !w || !w_logits || !prepared) return set_error("ms_decoder_chain_prepare: null argument");
  for (int l = 0; l < CH_NL; ++l) if (!w[l]) return set_error("ms_decoder_chain_prepare: null weight");
  return chain_is16(d) ? chain16_prepare(d, w, w_logits, prepared, (hipStream_t)stream) : chain32_prepare(d, w, w_logits, prepared, (hipStream_t)stream);
}
int ms_decoder_chain_fwd(const ms_chain_desc* d, const ms_chain_tensors* t, void* workspace, size_t workspace_bytes, void* stream) {
  if (!d || !t) return set_error("ms_decoder_chain_fwd: null argument");
  return chain_is16(d) ? chain16_fwd(d, t, workspace, workspace_bytes, (hipStream_t)stream) : chain32_fwd(d, t, workspace, workspace_bytes, (hipStream_t)stream);
}
}
