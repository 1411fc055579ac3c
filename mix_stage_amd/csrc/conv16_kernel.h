// conv16_kernel<DT, KW, WM, WN, UP2>: forward conv / data gradient on v_mfma_f32_32x32x16_{bf16,f16}, fp32 accumulate.
//
// A workgroup (4 waves, 2 x 2) owns BM = 64*WM output channels x BN = 64*WN output pixels (a TH x TW block of one image;
// 1-D convs use the batch axis as the row axis).  The K loop runs over stages (channel chunk of CK8 blocks, kernel row kh):
// a stage holds the weight slice [KW][CK8*8 channels][BM rows] and the raw input rows [CK8][TH][(TW-1)*S + KW] in LDS, both
// as 16-byte vectors of 8 consecutive channels -- what one lane feeds to one MFMA:
//   A (weights):      lane (r, h) reads slot [kw][ks][h][row r]                    (32 lanes = 512 contiguous bytes)
//   B (activations):  lane (r, h) reads block 2*ks+h, pixel (ty, tx*S + kw)        (32 lanes = consecutive pixels)
// so the inner loop is ds_read_b128 + MFMA only.  Staging is global -> registers -> LDS with raw buffer loads (padding and
// ragged edges by an out-of-range offset that the hardware answers with zeros); the next stage's loads are in flight while
// the current one computes.  Weights arrive pre-arranged per (m tile, stage) by prep16_kernel, so their staging is a linear copy.
#pragma once
#include <type_traits>

#include "conv16.h"

namespace ms {

struct BF16 {
  typedef __bf16 v8 __attribute__((ext_vector_type(8)));
  typedef __bf16 v2 __attribute__((ext_vector_type(2)));
  static constexpr const char* name = "bf16";
  __device__ static inline f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
  }
  __device__ static inline unsigned pack2(float lo, float hi) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f2){lo, hi}, v2));
  }
  __device__ static inline float lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
  __device__ static inline float hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
};
struct F16 {
  typedef _Float16 v8 __attribute__((ext_vector_type(8)));
  typedef _Float16 v2 __attribute__((ext_vector_type(2)));
  static constexpr const char* name = "f16";
  __device__ static inline f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
  }
  __device__ static inline unsigned pack2(float lo, float hi) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f2){lo, hi}, v2));
  }
  __device__ static inline float lo(unsigned u) {
    return (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu));
  }
  __device__ static inline float hi(unsigned u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16)); }
};

template <typename DT>
__device__ inline void unpack8(u32x4 v, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { f[2 * i] = DT::lo(v[i]); f[2 * i + 1] = DT::hi(v[i]); }
}
template <typename DT>
__device__ inline u32x4 pack8(const float (&f)[8]) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = DT::pack2(f[2 * i], f[2 * i + 1]);
  return v;
}

__device__ inline u32x4 buf_load_v(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

constexpr int conv16_ck8(int KW) { return KW == 8 ? 2 : 4; }
constexpr int CONV16_NP = 6;      // patch vectors a thread stages per stage at most (plan_conv16 keeps CK8*TH*PC <= 6*256)
constexpr int CONV16_MAX_RING = 8; // LDS-DMA ring depth limit (buffers; one less is in flight)

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate)
__device__ inline void wait_vmcnt(int n) {
#define MS_VM(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
  switch (n) {
    MS_VM(0) MS_VM(1) MS_VM(2) MS_VM(3) MS_VM(4) MS_VM(5) MS_VM(6) MS_VM(7) MS_VM(8) MS_VM(9) MS_VM(10) MS_VM(11) MS_VM(12)
    MS_VM(13) MS_VM(14) MS_VM(15) MS_VM(16) MS_VM(17) MS_VM(18) MS_VM(19) MS_VM(20) MS_VM(21) MS_VM(22) MS_VM(23) MS_VM(24)
    MS_VM(25) MS_VM(26) MS_VM(27) MS_VM(28) MS_VM(29) MS_VM(30) MS_VM(31) MS_VM(32) MS_VM(33) MS_VM(34) MS_VM(35) MS_VM(36)
    MS_VM(37) MS_VM(38) MS_VM(39) MS_VM(40) MS_VM(41) MS_VM(42) MS_VM(43) MS_VM(44) MS_VM(45) MS_VM(46) MS_VM(47) MS_VM(48)
    MS_VM(49) MS_VM(50) MS_VM(51) MS_VM(52) MS_VM(53) MS_VM(54) MS_VM(55) MS_VM(56) MS_VM(57) MS_VM(58) MS_VM(59) MS_VM(60)
    MS_VM(61) MS_VM(62) MS_VM(63)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef MS_VM
}

// the same with a compile-time count (a literal in the instruction: no branching)
template <int N>
__device__ __forceinline__ void wait_vmcnt_lit() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// 16 bytes per lane from a buffer straight into LDS (buffer_load_dwordx4 ... lds): the destination is the wave-uniform
// base `lds` + lane*16 -- the staging images below are laid out in exactly that order -- and out-of-range lanes store zeros
__device__ inline void dma16(__amdgpu_buffer_rsrc_t r, u32x4* lds, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, (int)soff, 0, 0);
}

struct Tile16 { int g, m0, wm, wn, r, h, lane, t, img, oy0, ox0, OUTH, OUTW, o_ry, o_rx, bx, by; };

// Sum each of v[0..15] over the 32 lanes that share lane>>5: at every step a lane hands half of its values to its partner and
// keeps (and accumulates) the other half, so 16 values cost 8+4+2+1+1 exchanges in 5 dependent steps instead of 80.
// Afterwards lane r holds in v[0] the total of value index (r >> 1) & 15.  All exchanges are VALU lane operations
// (v_permlane16_swap between the two 16-lane rows of a half, DPP inside a row): __shfl_xor goes through ds_bpermute and
// measured 14 us for the 64 exchanges of a 128 x 128 tile.
template <int CTRL>
__device__ inline float dpp_mov(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
__device__ inline void reduce16_over_half(float (&v)[16], int r) {
  // rows 0/1 (2/3) of the wave = the two 16-lane halves of lane half h = 0 (1): swap odd rows of x with even rows of y
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const unsigned x = __builtin_bit_cast(unsigned, v[k]), y = __builtin_bit_cast(unsigned, v[k + 8]);
    const unsigned long long sw = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_permlane16_swap(x, y, false, false));
    // even rows now hold (value k of the even row, value k of the odd row), odd rows the same for value k + 8
    v[k] = __builtin_bit_cast(float, (unsigned)sw) + __builtin_bit_cast(float, (unsigned)(sw >> 32));
  }
  {   // partner lane ^ 8: a rotation by 8 inside the row
    const bool up = (r & 8) != 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float send = up ? v[k] : v[k + 4], keep = up ? v[k + 4] : v[k];
      v[k] = keep + dpp_mov<0x128>(send);                  // row_ror:8
    }
  }
  {   // partner lane ^ 4: row_half_mirror (i -> 7 - i) followed by the quad reversal (i -> i ^ 3)
    const bool up = (r & 4) != 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float send = up ? v[k] : v[k + 2], keep = up ? v[k + 2] : v[k];
      v[k] = keep + dpp_mov<0x1B>(dpp_mov<0x141>(send));   // quad_perm:[3,2,1,0] of row_half_mirror
    }
  }
  {   // partner lane ^ 2
    const bool up = (r & 2) != 0;
    const float send = up ? v[0] : v[1], keep = up ? v[1] : v[0];
    v[0] = keep + dpp_mov<0x4E>(send);                     // quad_perm:[2,3,0,1]
  }
  v[0] += dpp_mov<0xB1>(v[0]);                             // quad_perm:[1,0,3,2]
}

template <typename DT, int WM, int WN, int NWN, int EP, bool OUTF32>
__device__ __forceinline__ void conv16_epilogue(const Conv16Args& p, const f32x16 (&acc)[WM][WN], const Tile16& tl, u32x4* smem,
                                                const float (&bias_pre)[WM][16]) {
  constexpr int BM = 64 * WM;
  const int TW = 1 << p.ltw;
  const int ctot = p.groups * p.Mg;
  float* red = reinterpret_cast<float*>(smem);     // EP_RAW_STATS: [NWN pixel waves][BM][2]
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const int mrow0 = tl.m0 + (tl.wm * WM + i) * 32 + 4 * tl.h;
    float bsv[16], scv[EP == EP_BN_EVAL ? 16 : 1], shv[EP == EP_BN_EVAL ? 16 : 1];
    if (WM == 2) {                                     // 128-row tiles: this row block's 16 values now, in one round trip
#pragma unroll
      for (int q = 0; q < 16; ++q)
        bsv[q] = p.bias ? p.bias[tl.g * p.Mg + min(mrow0 + (q & 3) + 8 * (q >> 2), p.Mg - 1)] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int m = mrow0 + (q & 3) + 8 * (q >> 2);
      const int chn = tl.g * p.Mg + min(m, p.Mg - 1);
      bsv[q] = m < p.Mg ? (WM == 2 ? bsv[q] : bias_pre[i][q]) : 0.f;       // WM == 1: fetched before the K loop (conv16_kernel)
      if (EP == EP_BN_EVAL) {
        const float sc = p.bn_g[chn] * (1.0f / sqrtf(p.bn_v[chn] + p.eps));
        scv[q] = sc; shv[q] = p.bn_b[chn] - p.bn_m[chn] * sc;
      }
    }
    float s1[16], s2[16];
    if (EP == EP_RAW_STATS) {
#pragma unroll
      for (int q = 0; q < 16; ++q) { s1[q] = 0.f; s2[q] = 0.f; }
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int n = (tl.wn * WN + j) * 32 + tl.r;
      const int oy = tl.oy0 + (n >> p.ltw), ox = tl.ox0 + (n & (TW - 1));
      const bool cval = (oy < tl.OUTH) & (ox < tl.OUTW);
      // (plain array, not an ext_vector: this clang miscompiles constant-index writes followed by reads on a local f32x16)
      float c[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        float v = acc[i][j][q] + bsv[q];
        if (EP == EP_RAW_STATS) {
          const float vm = cval ? v : 0.f;             // (rows beyond Mg: zero weights and zero bias -> v == 0)
          s1[q] += vm;
          s2[q] = fmaf(vm, vm, s2[q]);
        }
        if (EP == EP_BN_EVAL) v = lrelu(fmaf(v, scv[EP == EP_BN_EVAL ? q : 0], shv[EP == EP_BN_EVAL ? q : 0]), p.slope);
        if (EP == EP_LRELU) v = lrelu(v, p.slope);
        if (EP == EP_BN_EVAL) v = (mrow0 + (q & 3) + 8 * (q >> 2) < p.Mg) ? v : 0.f;     // shift of a pad row
        c[q] = v;
      }
      if (OUTF32) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int m = mrow0 + (q & 3) + 8 * (q >> 2);
          if (m < p.Mg && cval)
            p.out_f32[(size_t)tl.img * p.of_img + (size_t)(tl.g * p.Mg + m) * p.of_chan + (size_t)oy * p.of_row + ox] = c[q];
        }
      } else {
        // 8 consecutive channels of a pixel sit in two lanes (l, l+32): v_permlane32_swap pairs register groups so that the
        // lower lane ends up with blocks 0,1 and the upper lane with blocks 2,3 of this 32-row tile, one 16-byte store each
        float vec[2][8];
#pragma unroll
        for (int pr = 0; pr < 2; ++pr)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned x = __builtin_bit_cast(unsigned, c[4 * pr + e]), y = __builtin_bit_cast(unsigned, c[4 * (pr + 2) + e]);
            const unsigned long long sw = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_permlane32_swap(x, y, false, false));
            vec[pr][e] = __builtin_bit_cast(float, (unsigned)sw);
            vec[pr][4 + e] = __builtin_bit_cast(float, (unsigned)(sw >> 32));
          }
        const int cb_tile = (tl.g * p.Mg + tl.m0 + (tl.wm * WM + i) * 32) >> 3;
        const int cb_end = (tl.g * p.Mg + p.Mg + 7) >> 3;
        const size_t obase = (size_t)tl.img * p.o_img + (size_t)(oy * p.o_sh + tl.o_ry) * p.o_row + (size_t)(ox * p.o_sw + tl.o_rx);
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const int cb = cb_tile + pr + 2 * tl.h;
          if (EP == EP_DGRAD_UP2) {
            // 1-D stride-1 data gradient of an upsample-add input: out2 = grad of the residual (full resolution),
            // out = grad of the half-resolution tensor = sum over the pair of columns (adjacent lanes)
            float pair[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) pair[e] = vec[pr][e] + lane_xor1(vec[pr][e]);
            if (cval && cb < cb_end) {
              reinterpret_cast<u32x4*>(p.out2)[obase + (size_t)cb * p.o_cblk] = pack8<DT>(vec[pr]);
              if (!(tl.lane & 1)) {
                const size_t hb = (size_t)tl.img * (p.o_img >> 1) + (size_t)oy * (p.o_row >> 1) + (size_t)(ox >> 1);
                reinterpret_cast<u32x4*>(p.out)[hb + (size_t)cb * (p.o_cblk >> 1)] = pack8<DT>(pair);
              }
            }
          } else if (cval && cb < cb_end && !(p.dbg & 1)) {
            reinterpret_cast<u32x4*>(p.out)[obase + (size_t)cb * p.o_cblk] = pack8<DT>(vec[pr]);
          }
        }
      }
    }
    if (EP == EP_RAW_STATS) {
      // per-channel (sum, sum of squares) over the 32 pixels of a lane half; the two pixel waves meet in LDS below
      if (!(p.dbg & 2)) {
        reduce16_over_half(s1, tl.r);
        reduce16_over_half(s2, tl.r);
      }
      if (!(tl.r & 1)) {
        const int q = (tl.r >> 1) & 15;
        const int ml = (tl.wm * WM + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * tl.h;
        red[(tl.wn * BM + ml) * 2] = s1[0];
        red[(tl.wn * BM + ml) * 2 + 1] = s2[0];
      }
    }
  }
  if (EP == EP_RAW_STATS) {
    // fixed order over the pixel waves; stored as (sum, M2 about the tile mean) like the fp32 kernels
    __syncthreads();
    const int th_v = min(p.TH, tl.OUTH - tl.oy0), tw_v = min(TW, tl.OUTW - tl.ox0);
    if (tl.t < BM && tl.m0 + tl.t < p.Mg) {
      float sa = 0.f, sb = 0.f;
#pragma unroll
      for (int w = 0; w < NWN; ++w) { sa += red[(w * BM + tl.t) * 2]; sb += red[(w * BM + tl.t) * 2 + 1]; }
      float* stp = p.stats + ((size_t)tl.bx * ctot + tl.g * p.Mg + tl.m0 + tl.t) * 2;
      stp[0] = sa;
      stp[1] = fmaxf(sb - sa * sa / (float)(th_v * tw_v), 0.f);
    }
    if (tl.t == 0 && tl.by == 0 && tl.g == 0) p.counts[tl.bx] = (float)(th_v * tw_v);
  }
}

// One 32-row x 32-pixel accumulator block -> cb8 vectors in HBM.  8 consecutive channels of a pixel sit in two lanes (l, l+32):
// v_permlane32_swap pairs register groups so that the lower lane ends up with blocks 0,1 and the upper lane with blocks 2,3 of
// the 32-row tile, one 16-byte store each.
template <typename DT>
__device__ __forceinline__ void store_block16(const Conv16Args& p, const Tile16& tl, void* out, int row_block, int oy, int ox, bool cval,
                                              const float (&c)[16], int pr_mask = 3) {
  float vec[2][8];
#pragma unroll
  for (int pr = 0; pr < 2; ++pr)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned x = __builtin_bit_cast(unsigned, c[4 * pr + e]), y = __builtin_bit_cast(unsigned, c[4 * (pr + 2) + e]);
      const unsigned long long sw = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_permlane32_swap(x, y, false, false));
      vec[pr][e] = __builtin_bit_cast(float, (unsigned)sw);
      vec[pr][4 + e] = __builtin_bit_cast(float, (unsigned)(sw >> 32));
    }
  const int cb_tile = (tl.g * p.Mg + tl.m0 + row_block * 32) >> 3;
  const int cb_end = (tl.g * p.Mg + p.Mg + 7) >> 3;
  const size_t obase = (size_t)tl.img * p.o_img + (size_t)(oy * p.o_sh + tl.o_ry) * p.o_row + (size_t)(ox * p.o_sw + tl.o_rx);
#pragma unroll
  for (int pr = 0; pr < 2; ++pr) {
    const int cb = cb_tile + pr + 2 * tl.h;
    if (cval && cb < cb_end && ((pr_mask >> pr) & 1)) reinterpret_cast<u32x4*>(out)[obase + (size_t)cb * p.o_cblk] = pack8<DT>(vec[pr]);
  }
}

// ---- EP_BN_FUSED: train-mode BatchNorm + LeakyReLU applied INSIDE the conv launch (layers.py:77-78 as one HBM pass).
// The batch statistics of a channel need every pixel tile of that channel tile: the gx workgroups (g, by, bx = 0..gx-1) form a
// GROUP that meets once.  Each keeps its accumulators in registers, publishes its per-channel partial (sum, sum of
// squares, count) with write-through (sc1) 16-byte stores, arrives on the group's counter, waits until all gx have arrived
// (one lane polls with sc1 loads), reads the gx partials back (sc1 loads: served by L2 / the fabric, never by this CU's L1)
// and sums them in tile order in fp64 -- every workgroup of the group computes bit-identical statistics --
// then normalises, activates and stores y from registers.  The backward pass takes x_hat and the activation mask from y
// itself (conv16.h: bn_inv_unsafe); y_raw is stored only for channel blocks where that inversion is unsafe.  One HBM pass:
// x and w in, y out -- no y_raw write, no re-read, no second launch.
// Hand-off form: MI355X_MICROARCH.md "Valid forms", table row 3 (sc1 payload stores, every storing wave's vmcnt(0), workgroup
// barrier, one agent-scope atomic add per workgroup, sc1 poll, workgroup barrier, sc1 loads).
// Forward progress: the launcher uses this epilogue only when the WHOLE grid is co-resident (conv16_coresident), so every
// member of a group is running whatever the dispatch order; members of a group have consecutive logical ids, so under the
// in-order dispatch the hardware performs at most one group per XCD is ever partially dispatched.  The spin is bounded: on
// expiry the kernel raises word 0 of the sync buffer (ms_set_bn_sync_buffer) and finishes with whatever statistics it has.
constexpr int BNF_SPIN_LIMIT = 1 << 21;
constexpr int BNF_SYNC_STRIDE = BNF_SYNC_WORDS_PER_GROUP;      // int32 words per group in the sync buffer (arrive, depart on a 128-byte line of their own)

template <typename DT, int WM, int WN, int NWN>
__device__ __forceinline__ void conv16_epilogue_bnfused(const Conv16Args& p, const f32x16 (&acc)[WM][WN], const Tile16& tl, u32x4* smem,
                                                        const float (&bias_pre)[WM][16]) {
  constexpr int BM = 64 * WM, NT = 128 * NWN, P = NT / BM;
  const int TW = 1 << p.ltw;
  float* red = reinterpret_cast<float*>(smem);                         // [NWN pixel waves][BM][4] = (sum, sum of squares about the pivot, pivot, -)
  double* dred = reinterpret_cast<double*>(red + NWN * BM * 4);        // [P][BM][3]
  float* scsh = reinterpret_cast<float*>(dred + P * BM * 3);           // [BM][2]
  int* rawflag = reinterpret_cast<int*>(scsh + BM * 2);                // [BM / 8]
  int* wcnt = rawflag + BM / 8;                                        // [NWN] valid pixels of each pixel wave; [NWN]: spin expired
  const int grp = tl.g * p.gy + tl.by;
  int* arrive = p.bn_sync + (size_t)(1 + grp) * BNF_SYNC_STRIDE;
  int* depart = arrive + 1;

  // ---- A. conv + bias stays in the accumulators; per-channel sums of this tile ABOUT THE CHANNEL'S BIAS -- the part of the mean
  // the kernel knows beforehand: sum acc and sum acc^2 of the bias-free accumulators, the tile's partial is published as
  // (mean = bias + sum / n, M2 = sum acc^2 - (sum acc)^2 / n, n) and the tiles are combined in fp64.  A channel whose mean is
  // large BECAUSE OF ITS BIAS keeps its variance (fp32 sum (acc + b)^2 - (sum (acc + b))^2 / n would lose it); what remains is
  // the conv output's own mean against its spread, O(1) behind a normalised input.  (A per-tile pivot taken from the data --
  // readlane + a third reduction -- was measured: +1.4 us per launch, 50 us per G-step.)
  // (the accumulators are only ever READ element-wise: this clang miscompiles constant-index element writes into a local
  // f32x16.  The bias is added at each use; 128-row tiles re-fetch it -- an L2 hit -- instead of holding 32 more
  // registers across the gathering)
  auto bias_of = [&](int i, float (&bsv)[16]) {
    const int mrow0 = tl.m0 + (tl.wm * WM + i) * 32 + 4 * tl.h;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int m = mrow0 + (q & 3) + 8 * (q >> 2);
      const float b = WM == 2 ? (p.bias ? p.bias[tl.g * p.Mg + min(m, p.Mg - 1)] : 0.f) : bias_pre[i][q];
      bsv[q] = m < p.Mg ? b : 0.f;
    }
  };
  // this thread's channel of the tile (t < BM): its bias, requested now, used after the reduction
  const float bias_t = (p.bias && tl.t < BM) ? p.bias[tl.g * p.Mg + min(tl.m0 + tl.t, p.Mg - 1)] : 0.f;
  {
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int n = (tl.wn * WN + j) * 32 + tl.r;
      const int oy = tl.oy0 + (n >> p.ltw), ox = tl.ox0 + (n & (TW - 1));
      cnt += __popc((unsigned)__ballot((oy < tl.OUTH) & (ox < tl.OUTW)));      // lanes 0-31 = the 32 pixels of block j
    }
    if (tl.lane == 0 && tl.wm == 0) wcnt[tl.wn] = cnt;
    if (tl.t == 0) wcnt[NWN] = 0;
  }
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    float s1[16], s2[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) { s1[q] = 0.f; s2[q] = 0.f; }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int n = (tl.wn * WN + j) * 32 + tl.r;
      const int oy = tl.oy0 + (n >> p.ltw), ox = tl.ox0 + (n & (TW - 1));
      const bool cval = (oy < tl.OUTH) & (ox < tl.OUTW);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float dv = cval ? acc[i][j][q] : 0.f;
        s1[q] += dv;
        s2[q] = fmaf(dv, dv, s2[q]);
      }
    }
    reduce16_over_half(s1, tl.r);
    reduce16_over_half(s2, tl.r);
    if (!(tl.r & 1)) {
      const int q = (tl.r >> 1) & 15;
      const int ml = (tl.wm * WM + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * tl.h;
      *reinterpret_cast<float2*>(red + (tl.wn * BM + ml) * 4) = float2{s1[0], s2[0]};
    }
  }
  __syncthreads();

  // ---- B. publish this tile's partial, arrive
  const int th_v = min(p.TH, tl.OUTH - tl.oy0), tw_v = min(TW, tl.OUTW - tl.ox0);
  const __amdgpu_buffer_rsrc_t rsP = buf_rsrc(p.bn_part);
  const bool chan_ok = tl.t < BM && tl.m0 + tl.t < p.Mg;
  const int chn = tl.g * p.Mg + min(tl.m0 + (tl.t & (BM - 1)), p.Mg - 1);
  float gam = 0.f, bet = 0.f, rmean = 0.f, rvar = 0.f;
  if (tl.t < BM) {
    // the pixel waves of this tile in fixed order (Chan): (count, mean, M2 about the mean)
    float cn = 0.f, cmean = 0.f, cm2 = 0.f;
#pragma unroll
    for (int w = 0; w < NWN; ++w) {
      const float2 rv = *reinterpret_cast<const float2*>(red + (w * BM + tl.t) * 4);
      const float nw = (float)wcnt[w];
      if (nw > 0.f) {
        const float mw = bias_t + rv.x / nw, m2w = fmaxf(rv.y - rv.x * rv.x / nw, 0.f);
        const float tot = cn + nw, dl = mw - cmean;
        cmean += dl * (nw / tot);
        cm2 += m2w + dl * dl * (cn * nw / tot);
        cn = tot;
      }
    }
    (void)th_v; (void)tw_v;
    const float4 part = {cmean, cm2, chan_ok ? cn : 0.f, 0.f};
    // one wave instruction = 64 lanes x 16 B = eight whole 128-byte lines, written through to the fabric
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, part), rsP,
                                           (int)(16u * (unsigned)((grp * p.gx + tl.bx) * BM + tl.t)), 0, 16 /* sc1 */);
    gam = p.bn_g[chn]; bet = p.bn_b[chn];
    if (tl.bx == 0) { rmean = p.bn_m[chn]; rvar = p.bn_v[chn]; }
  }
  // (p.dbg bits 4..7: timing ablations of this epilogue -- no y_raw store / no wait / no partial loads / no publish; results are
  // then meaningless.  ms_debug_set_conv16_ring)
  if (!(p.dbg & 128)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every wave: its partial has left before the workgroup signals
  __syncthreads();
  if (tl.t == 0 && !(p.dbg & 128)) __hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tl.t == 0 && !(p.dbg & (32 | 128))) {
    // (a raised error word is sticky until the host clears the buffer: chain32.hip, chain_meet)
    if (__hip_atomic_load(p.bn_sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) wcnt[NWN] = 1;
    int spins = 0;
    while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < p.gx) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > BNF_SPIN_LIMIT) { __hip_atomic_store(p.bn_sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); wcnt[NWN] = 1; break; }
    }
  }
  __syncthreads();

  // ---- C. the group's partials, summed in fp64 in tile order (thread = channel c, every P-th tile; then the P shares in order)
  {
    const int c = tl.t & (BM - 1), share = tl.t / BM;
    const unsigned base = 16u * (unsigned)(grp * p.gx * BM + c);
    // fp64 moments of the group's tiles in tile order: N = sum n_i, S = sum n_i mean_i, Q = sum (M2_i + n_i mean_i^2); then
    // mean = S / N, M2 = Q - S mean.  (The subtraction is harmless in fp64 -- a relative 1e-16 (mean / sigma)^2 -- and the loop holds
    // no division: Chan's pairwise update costs two fp64 divisions per tile, ~3 us of every launch at 16-32 tiles per thread.)
    double n = 0.0, sm = 0.0, sq = 0.0;
    for (int k0 = share; k0 < ((p.dbg & 64) ? 0 : p.gx); k0 += 4 * P) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = min(k0 + u * P, p.gx - 1);
        v[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsP, (int)(base + 16u * (unsigned)(k * BM)), 0, 16 /* sc1 */));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (k0 + u * P < p.gx) {
          const double nw = (double)v[u].z, mw = (double)v[u].x;
          n += nw; sm += nw * mw; sq += (double)v[u].y + nw * mw * mw;
        }
    }
    dred[(share * BM + c) * 3] = n; dred[(share * BM + c) * 3 + 1] = sm; dred[(share * BM + c) * 3 + 2] = sq;
  }
  __syncthreads();
  if (tl.t == 0 && !(p.dbg & 128)) {            // every load of the group's partials by this workgroup has returned: depart; the last one re-arms
    const int old = __hip_atomic_fetch_add(depart, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == p.gx - 1) {
      __hip_atomic_store(arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(depart, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (tl.t < BM) {
    double n = dred[tl.t * 3], sm = dred[tl.t * 3 + 1], sq = dred[tl.t * 3 + 2];
#pragma unroll
    for (int s = 1; s < P; ++s) { n += dred[(s * BM + tl.t) * 3]; sm += dred[(s * BM + tl.t) * 3 + 1]; sq += dred[(s * BM + tl.t) * 3 + 2]; }
    const double mean = n > 0.0 ? sm / n : 0.0, m2 = fmax(sq - sm * mean, 0.0);
    float sc = 0.f, shf = 0.f;
    bool unsafe_c = false;
    const bool expired = wcnt[NWN] != 0;
    if (expired) {
      // this workgroup gave up waiting for its group (word 0 of the sync buffer is raised): statistics are incomplete.  Poison the
      // output instead of normalising with them, and leave the running statistics and the saved vector alone.
      sc = __builtin_nanf(""); shf = sc;
    } else if (chan_ok && n > 0.0) {
      const float var = (float)(m2 / n);
      const float invstd = 1.0f / sqrtf(var + p.eps);
      const float fmean = (float)mean;
      sc = gam * invstd;
      shf = bet - fmean * sc;
      unsafe_c = bn_inv_unsafe(fmean, invstd, sc, shf, p.slope);
      if (tl.bx == 0) {
        const int ctot = p.groups * p.Mg;
        p.save[chn] = fmean; p.save[ctot + chn] = invstd; p.save[2 * ctot + chn] = sc; p.save[3 * ctot + chn] = shf;
        const float unbiased = n > 1.0 ? (float)(m2 / (n - 1.0)) : var;
        running_stats_update(&p.bn_m[chn], &p.bn_v[chn], rmean, rvar, p.momentum, fmean, unbiased);
      }
    }
    scsh[tl.t * 2] = sc; scsh[tl.t * 2 + 1] = shf;
    // 8-channel blocks whose backward cannot take x_hat from y (conv16.h: bn_inv_unsafe): their y_raw is kept.  t < BM are
    // whole waves: the ballot holds the 64 channels of this wave, bit (t & 63)
    const unsigned long long unsafe = __ballot(unsafe_c);
    if (!(tl.t & 7)) rawflag[tl.t >> 3] = (int)((unsafe >> (tl.t & 63)) & 0xffull) != 0;
  }
  __syncthreads();

  // ---- D. normalise + activate from registers, store y
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    float scv[16], shv[16], bsv[16];
    bias_of(i, bsv);
    // this lane's two channel blocks of row block i: pr + 2h (store_block16)
    const int rb = (tl.wm * WM + i) * 4 + 2 * tl.h;
    const int raw_mask = (p.dbg & 16) ? 0 : p.raw_all ? 3 : (rawflag[rb] ? 1 : 0) | (rawflag[rb + 1] ? 2 : 0);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int ml = (tl.wm * WM + i) * 32 + 4 * tl.h + (q & 3) + 8 * (q >> 2);
      const float2 ss = *reinterpret_cast<const float2*>(scsh + ml * 2);
      scv[q] = ss.x; shv[q] = fmaf(bsv[q], ss.x, ss.y);          // (acc + b) * sc + sh = acc * sc + (b * sc + sh)
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int n = (tl.wn * WN + j) * 32 + tl.r;
      const int oy = tl.oy0 + (n >> p.ltw), ox = tl.ox0 + (n & (TW - 1));
      float c[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) c[q] = lrelu(fmaf(acc[i][j][q], scv[q], shv[q]), p.slope);
      store_block16<DT>(p, tl, p.out, tl.wm * WM + i, oy, ox, (oy < tl.OUTH) & (ox < tl.OUTW), c);
      // (rare: a channel block whose BatchNorm map does not invert safely.  Wave-uniform condition: assembling a vector takes
      // lanes l and l + 32 together (v_permlane32_swap), whichever of them owns the flagged block)
      if (p.out_raw && __ballot(raw_mask != 0) != 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) c[q] = acc[i][j][q] + bsv[q];
        store_block16<DT>(p, tl, p.out_raw, tl.wm * WM + i, oy, ox, (oy < tl.OUTH) & (ox < tl.OUTW), c, raw_mask);
      }
    }
  }
}

// DMA = true: the stages are filled by LDS-DMA loads into a ring of p.nstg buffers -- no staging registers, no ds_write, and
// the loads of the next nstg-2 stages stay in flight behind the current stage's MFMAs (counted vmcnt, one raw barrier per
// stage).  DMA = false: global -> registers -> LDS, two buffers (needed where the input is formed on the way: UP2).
// NWN: waves along the pixels (2 waves along the channels): 2 = 4 waves, 4 = 8 waves (two per SIMD: one wave's staging and
// epilogue arithmetic runs beside the other's MFMAs).  Tile = 64*WM channels x 32*WN*NWN pixels.
template <typename DT, int KW, int WM, int WN, bool UP2, bool DMA, int NWN = 2, int CKX = 1>
__global__ __launch_bounds__(128 * NWN) void conv16_kernel(const Conv16Args p) {
  prefetch_kernargs<sizeof(Conv16Args)>();
  // CKX = 2: stages of twice the channels (half as many barriers / waits / DMA issues per reduction) for the layers whose
  // workgroups run alone on their CU -- their K loop is a latency chain, not a throughput problem (plan_conv16)
  constexpr int CK8 = conv16_ck8(KW) * CKX, KS = CK8 / 2;
  constexpr int BM = 64 * WM;
  constexpr int NT = 128 * NWN;
  constexpr int NAV = KW * CK8 * BM, NA = (NAV + NT - 1) / NT;
  static_assert(NAV % NT == 0, "the weight stage is copied in whole slabs");
  constexpr int NP = CONV16_NP;
  static_assert(!(UP2 && DMA), "the upsample-add input is formed in registers");
  extern __shared__ u32x4 smem[];

  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wm = wid / NWN, wn = wid % NWN, r = lane & 31, h = lane >> 5;
  const int TW = 1 << p.ltw, TH = p.TH, PC = p.PC, S = p.S, SV = p.SV, KH = p.KH;
  const int thpc = TH * PC, pv = CK8 * thpc;
  const int npd = (pv + NT - 1) / NT;            // DMA: NT-vector slabs of input rows per stage
  // register path: + one dummy vector (out-of-range staging stores land there); DMA path: whole slabs
  const int stage_vecs = DMA ? NAV + npd * NT : NAV + pv + 1;

  // logical block id: channel tile fastest, then pixel tile, then (class, group); one contiguous range per XCD
  const int vid = xcd_remap(blockIdx.x, p.gx * p.gy * p.gz);
  // (EP_BN_FUSED: pixel tile fastest -- the gx workgroups that meet for a channel tile's statistics have consecutive ids)
  const bool px_first = p.ep == EP_BN_FUSED;
  const int by_ = px_first ? (vid / p.gx) % p.gy : vid % p.gy, bx_ = px_first ? vid % p.gx : (vid / p.gy) % p.gx,
            bz_ = vid / (p.gy * p.gx);
  const int cls = p.ncls > 1 ? bz_ / p.groups : 0, g = bz_ - cls * p.groups, m0 = by_ * BM;
  const int PHc = p.ncls > 1 ? p.cls_PH[cls] : p.PH, PWc = p.ncls > 1 ? p.cls_PW[cls] : p.PW;
  const int OUTHc = p.ncls > 1 ? p.cls_OUTH[cls] : p.OUTH, OUTWc = p.ncls > 1 ? p.cls_OUTW[cls] : p.OUTW;
  const int o_ryc = p.ncls > 1 ? p.cls_ry[cls] : p.o_ry, o_rxc = p.ncls > 1 ? p.cls_rx[cls] : p.o_rx;
  const int tiles_per_img = p.tiles_y * p.tiles_x;
  const int img = bx_ / tiles_per_img;
  const int trem = bx_ - img * tiles_per_img;
  const int tyi = trem / p.tiles_x, txi = trem - tyi * p.tiles_x;
  const int oy0 = tyi * TH, ox0 = txi << p.ltw;
  const int iy0 = oy0 * SV - PHc, ix0 = ox0 * S - PWc;
  const int cbase8 = p.bcast ? 0 : g * p.Kc8g;

  const __amdgpu_buffer_rsrc_t rsA = buf_rsrc(p.A), rsS = buf_rsrc(p.src), rsS2 = buf_rsrc(UP2 ? p.src2 : p.src);
  const unsigned a_wg = (unsigned)cls * p.a_cls_stride + (unsigned)g * p.a_group_stride + (unsigned)by_ * p.a_mt_stride;
  const int nstages = p.nchunks * KH;
  const int nst_run = (p.dbg & 4) ? 0 : nstages;
  // the weight slab of stage 0 needs nothing but the tile indices: its LDS-DMA goes out BEFORE the ~300 vector instructions of
  // per-thread offset arithmetic below (the kernel start is a latency chain; this puts the first memory round trip under it)
  if constexpr (DMA) {
    if (nst_run > 0) {
      u32x4* dst0 = smem + __builtin_amdgcn_readfirstlane(wid) * 64;
      const unsigned sa0 = __builtin_amdgcn_readfirstlane(16u * a_wg);
#pragma unroll
      for (int i = 0; i < NA; ++i) dma16(rsA, dst0 + i * NT, 16u * (unsigned)(t + i * NT), sa0);
    }
  }

  // the epilogue's bias values, requested now (clamped addresses, no branches): after the K loop they were 16 dependent
  // load -> wait rounds.  They are older than every staging load below, so the counted waits of the ring cover them.
  // (128-row tiles fetch them at the start of the epilogue instead, in one round trip: 32 more registers across the K loop
  // would put the kernel over 256 and cost it its second wave per SIMD)
  float bias_pre[WM][16];
  if (WM == 1 && p.bias) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int m = m0 + (wm * WM + i) * 32 + 4 * h + (q & 3) + 8 * (q >> 2);
        bias_pre[i][q] = p.bias[g * p.Mg + min(m, p.Mg - 1)];
      }
  } else {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) bias_pre[i][q] = 0.f;
  }

  // ---- stage-invariant staging offsets of the input rows (e / thpc and rem / PC by multiply-high with a reciprocal formed
  // once: e < 2^16 and the divisors are small, for which floor(2^32 / d) + 1 is exact)
  int poff[NP], prow[NP], plds[NP], pcb[NP], pix[UP2 ? NP : 1];
  bool pcol[NP];
  const unsigned mg_thpc = 0xFFFFFFFFu / (unsigned)thpc + 1u, mg_pc = 0xFFFFFFFFu / (unsigned)PC + 1u;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    if (DMA && i >= npd) continue;                 // the LDS-DMA path stages whole slabs: only npd of the NP slots exist
    const int e = t + i * NT;
    const int cb = thpc == 1 ? e : (int)__umulhi((unsigned)e, mg_thpc), rem = e - cb * thpc;
    const int ty = PC == 1 ? rem : (int)__umulhi((unsigned)rem, mg_pc), c = rem - ty * PC;
    const int iy = iy0 + ty * SV, ix = ix0 + c;
    pcol[i] = (e < pv) & ((unsigned)ix < (unsigned)p.SRCW);
    prow[i] = iy;
    pcb[i] = cb;
    poff[i] = cb * p.s_cblk + iy * p.s_row + ix;
    if (UP2) pix[i] = ix;
    plds[i] = e < pv ? NAV + e : NAV + pv;
  }
  u32x4 ra[NA], rb[NP], rb2[UP2 ? NP : 1];
  auto load_stage = [&](int st) {
    const int ch = st / KH, kh = st - ch * KH;
    const unsigned sa = __builtin_amdgcn_readfirstlane(16u * (a_wg + (unsigned)st * NAV));
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      ra[i] = buf_load_v(rsA, 16u * (unsigned)(t + i * NT), sa);
    }
    const int cb0 = ch * CK8;
    const int sbase = __builtin_amdgcn_readfirstlane(img * p.s_img + (cbase8 + cb0) * p.s_cblk);
    const int khrow = kh * p.s_row;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const bool ok = pcol[i] & ((unsigned)(prow[i] + kh) < (unsigned)p.SRCH) & (cb0 + pcb[i] < p.Kc8g);
      if (UP2) {
        // x = nearest_up2(a) + r: a has half the row length, hence half of every stride (all strides are even)
        const int o = poff[i] + khrow, c = pix[UP2 ? i : 0];
        rb[i] = buf_load_v(rsS, ok ? 16u * (unsigned)(((o - c) >> 1) + (c >> 1)) : BUF_OOB, 16u * (unsigned)(sbase >> 1));
        rb2[UP2 ? i : 0] = buf_load_v(rsS2, ok ? 16u * (unsigned)o : BUF_OOB, 16u * (unsigned)sbase);
      } else {
        rb[i] = buf_load_v(rsS, ok ? 16u * (unsigned)(poff[i] + khrow) : BUF_OOB, 16u * (unsigned)sbase);
      }
    }
  };
  auto store_stage = [&](int buf) {
    u32x4* st = smem + buf * stage_vecs;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      st[t + i * NT] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      if (UP2) {                       // the sum is formed here, after the stage's MFMAs: the loads stay in flight meanwhile
        float fa[8], fr[8];
        unpack8<DT>(rb[i], fa);
        unpack8<DT>(rb2[UP2 ? i : 0], fr);
#pragma unroll
        for (int j = 0; j < 8; ++j) fa[j] += fr[j];
        st[plds[i]] = pack8<DT>(fa);
      } else {
        st[plds[i]] = rb[i];
      }
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  // ---- per-lane operand bases
  const int a_base = h * BM + wm * 32 * WM + r;
  int b_base[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int n = (wn * WN + j) * 32 + r;
    b_base[j] = NAV + h * thpc + (n >> p.ltw) * PC + (n & (TW - 1)) * S;
  }

  // One stage = KW*KS k-steps of WM*WN MFMAs.  The operand fragments of k-step s+2 are requested before the MFMAs of k-step s
  // (three register sets, order pinned): with one wave per SIMD nothing else hides the LDS latency, and left to the
  // scheduler the stage was a chain of ~14 read -> wait -> MFMA rounds (0.95 us per stage against 0.32 us of MFMA issue).
  // Operand addresses: one add per operand row group and stage (buffer base + per-lane constant), every read of the stage then
  // uses an immediate offset (weight slot / tap).  Before, each of the stage's ds_reads re-formed its address (18-23 v_add per 12
  // MFMAs on the decoder layer; one wave issues a VALU instruction every ~4 cycles, an MFMA occupies its pipe for 32).
  auto compute_stage = [&](const u32x4* st) {
    constexpr int NSTEP = KW * KS;
    const u32x4* ap = st + a_base;
    const u32x4* bp[WN][KS];
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) bp[j][ks] = st + b_base[j] + 2 * ks * thpc;
    u32x4 av[3][WM], bv[3][WN];
    auto fetch = [&](int s, u32x4 (&a)[WM], u32x4 (&b)[WN]) {
      const int kw = s / KS, ks = s - kw * KS;
#pragma unroll
      for (int i = 0; i < WM; ++i) a[i] = ap[((kw * KS + ks) * 2) * BM + i * 32];
#pragma unroll
      for (int j = 0; j < WN; ++j) b[j] = bp[j][ks][kw];
    };
    fetch(0, av[0], bv[0]);
    if (NSTEP > 1) fetch(1, av[1], bv[1]);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      if (s + 2 < NSTEP) fetch(s + 2, av[(s + 2) % 3], bv[(s + 2) % 3]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = DT::mfma(av[s % 3][i], bv[s % 3][j], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  if constexpr (DMA) {
    const int wave = __builtin_amdgcn_readfirstlane(wid);
    // the stage to be issued next: (channel chunk, kernel row), its ring buffer and its weight slab, all advanced incrementally
    // (st / KH, st % nstg as run-time scalar divisions were ~100 SALU instructions per stage)
    int is_ch = 0, is_kh = 0, is_buf = 0;
    unsigned is_sa = __builtin_amdgcn_readfirstlane(16u * a_wg);
    bool first = true;
    const unsigned t16 = 16u * (unsigned)t;
    auto issue_stage = [&]() {
      u32x4* dst = smem + is_buf * stage_vecs + wave * 64;
      if (!first) {                                    // (stage 0's weight slab left at the top of the kernel)
#pragma unroll
        for (int i = 0; i < NA; ++i) dma16(rsA, dst + i * NT, t16, is_sa + 16u * (unsigned)(i * NT));
      }
      first = false;
      const int cb0 = is_ch * CK8;
      const int sbase = __builtin_amdgcn_readfirstlane(img * p.s_img + (cbase8 + cb0) * p.s_cblk);
      const int khrow = is_kh * p.s_row;
      const int cb_left = p.Kc8g - cb0;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        if (i < npd) {
          const bool ok = pcol[i] & ((unsigned)(prow[i] + is_kh) < (unsigned)p.SRCH) & (pcb[i] < cb_left);
          dma16(rsS, dst + NAV + i * NT, ok ? 16u * (unsigned)(poff[i] + khrow) : BUF_OOB, 16u * (unsigned)sbase);
        }
      }
      if (++is_kh == KH) { is_kh = 0; ++is_ch; }
      if (++is_buf == p.nstg) is_buf = 0;
      is_sa += 16u * (unsigned)NAV;
    };
    const int per_stage = NA + npd;                    // LDS-DMA instructions a wave issues per stage
    const int ahead = p.nstg - 2;                      // stages that stay in flight behind the one being computed
    const int wcount = ahead * per_stage;
    // WC >= 0: the steady-state wait is the literal vmcnt(WC).  The run-time form (WC < 0) is a 64-way switch inside the loop;
    // around it the register allocator moved every accumulator between the AGPR and the VGPR file each stage (32 v_accvgpr
    // moves per 12 MFMAs on the decoder layer).  The counts the path's layers produce get a loop of their own.
    auto k_loop = [&](auto wc) {
      constexpr int WC = decltype(wc)::value;
      int issued = 0;
      for (; issued < p.nstg - 1 && issued < nst_run; ++issued) issue_stage();
      const u32x4* cur = smem;
      int cur_i = 0;
      for (int st = 0; st < nst_run; ++st) {
        // stage st has landed once at most `ahead` younger stages are outstanding (in the tail fewer are: drain)
        if (st + ahead < nst_run) {
          if constexpr (WC >= 0) wait_vmcnt_lit<(WC >= 0 ? WC : 0)>(); else wait_vmcnt(wcount);
        } else {
          wait_vmcnt_lit<0>();
        }
        __builtin_amdgcn_s_barrier();                  // everyone's part of stage st is in LDS, everyone is done with stage st-1
        asm volatile("" ::: "memory");
        if (issued < nst_run) { issue_stage(); ++issued; }    // refills the buffer of stage st-1
        compute_stage(cur);
        cur += stage_vecs;
        if (++cur_i == p.nstg) { cur_i = 0; cur = smem; }
      }
    };
    switch (wcount) {
#define MS_KLOOP(N) case N: k_loop(std::integral_constant<int, N>{}); break;
      MS_KLOOP(0) MS_KLOOP(6) MS_KLOOP(8) MS_KLOOP(10) MS_KLOOP(11) MS_KLOOP(13) MS_KLOOP(18) MS_KLOOP(20) MS_KLOOP(24)
#undef MS_KLOOP
      default: k_loop(std::integral_constant<int, -1>{}); break;
    }
  } else {
    load_stage(0);
    store_stage(0);
    __syncthreads();
    for (int st = 0; st < nstages; ++st) {
      const int cur = st & 1;
      if (st + 1 < nstages) load_stage(st + 1);
      compute_stage(smem + cur * stage_vecs);
      if (st + 1 < nstages) store_stage(cur ^ 1);
      __syncthreads();
    }
  }

  // ---------------- epilogue ----------------
  const Tile16 tl = {g, m0, wm, wn, r, h, lane, t, img, oy0, ox0, OUTHc, OUTWc, o_ryc, o_rxc, bx_, by_};
  __syncthreads();                                   // the staging buffers become the statistics scratch
  // one specialised instance per epilogue kind (wave-uniform switch): no per-element branching on the kind
  if (p.out_f32) {
    switch (p.ep) {
      case EP_BARE: conv16_epilogue<DT, WM, WN, NWN, EP_BARE, true>(p, acc, tl, smem, bias_pre); break;
      case EP_LRELU: conv16_epilogue<DT, WM, WN, NWN, EP_LRELU, true>(p, acc, tl, smem, bias_pre); break;
      default: conv16_epilogue<DT, WM, WN, NWN, EP_BN_EVAL, true>(p, acc, tl, smem, bias_pre); break;
    }
  } else {
    switch (p.ep) {
      case EP_BARE: conv16_epilogue<DT, WM, WN, NWN, EP_BARE, false>(p, acc, tl, smem, bias_pre); break;
      case EP_LRELU: conv16_epilogue<DT, WM, WN, NWN, EP_LRELU, false>(p, acc, tl, smem, bias_pre); break;
      case EP_BN_EVAL: conv16_epilogue<DT, WM, WN, NWN, EP_BN_EVAL, false>(p, acc, tl, smem, bias_pre); break;
      case EP_RAW_STATS: conv16_epilogue<DT, WM, WN, NWN, EP_RAW_STATS, false>(p, acc, tl, smem, bias_pre); break;
      case EP_BN_FUSED: conv16_epilogue_bnfused<DT, WM, WN, NWN>(p, acc, tl, smem, bias_pre); break;
      default: conv16_epilogue<DT, WM, WN, NWN, EP_DGRAD_UP2, false>(p, acc, tl, smem, bias_pre); break;
    }
  }
}

}  // namespace ms
