// 16-bit (bf16 / fp16) arithmetic mode of libmixstage_hip.so: shared declarations.
//
// Tensor layout "cb8": activations live in HBM as [B][C8][H][W][8] 16-bit elements, C8 = ceil(C/8) channel blocks; one
// 16-byte vector = 8 consecutive channels of one pixel, pad channels are zero.  The time / frequency axis stays the
// coalesced axis (a wave reads 64 consecutive pixels = 1 KiB), and a vector is exactly what one lane feeds to
// v_mfma_f32_32x32x16_{bf16,f16} (8 consecutive k = 8 consecutive channels at one tap), so the matrix operands go
// HBM -> LDS -> VGPR as 16-byte moves with no transposition: every tap or stride offset is a whole number of vectors.
#pragma once
#include "kernels.h"

namespace ms {

enum { DT_F32 = 0, DT_BF16 = 1, DT_F16 = 2 };
inline int dt_of(const ms_conv_desc* d) { return d->dtype & 0xff; }
inline bool out_f32_of(const ms_conv_desc* d) { return (d->dtype & MS_DT_OUT_F32) != 0; }
// statistics groups of a BN_TRAIN block along the batch axis (MS_DT_STAT_PAIR: the two passes of gan.py:120,126 side by side)
inline int sg_of(const ms_conv_desc* d) { return (d->dtype & MS_DT_STAT_PAIR) ? 2 : 1; }
// 16-bit BN_TRAIN blocks with MS_DT_STAT_PAIR (api16.hip)
bool stat_pair16_ok(const ms_conv_desc* d);
inline bool bn_folded_of(const ms_conv_desc* d) { return d->mode == MS_BN_EVAL && (d->dtype & MS_DT_BN_FOLDED) != 0; }
inline int c8_of(int c) { return (c + 7) >> 3; }

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

// ---- forward / data-gradient conv (conv16_kernel.h).  All strides of cb8 tensors count 16-byte vectors.
struct Conv16Args {
  const void* A;        // prepared weights: [cls][group][m tile][chunk][kh] stages of KW*CK8*BM vectors, slot order [kw][ks][h][row]
  const void* src;      // cb8 (UP2: the half-resolution tensor a)
  const void* src2;     // UP2: residual r (full resolution); strides below describe r / the plain input
  void* out;            // cb8
  void* out2;           // EP_DGRAD_UP2: gradient of the residual (cb8)
  float* out_f32;       // != NULL: plain (B, C, OH, OW) fp32 output instead of `out`
  const float* bias;
  const float* bn_g;
  const float* bn_b;
  float* bn_m;          // running statistics: read by EP_BN_EVAL, updated by EP_BN_FUSED
  float* bn_v;
  float* stats;         // EP_RAW_STATS: [n_tiles][ctot][2] = (sum, M2 about the tile mean)
  float* counts;        // EP_RAW_STATS: [n_tiles]
  // EP_BN_FUSED (train-mode BatchNorm inside the launch, conv16_kernel.h)
  void* out_raw;        // cb8 y_raw = conv + bias for the backward pass (NULL: not kept)
  float* bn_part;       // [groups*gy][gx][BM] x (sum, sum of squares, count, 0): the tiles' partial statistics
  int* bn_sync;         // word 0: error flag; per group (g*gy + by) BNF_SYNC_STRIDE words from word BNF_SYNC_STRIDE on: arrive, depart
  float* save;          // mean | invstd | scale | shift
  float momentum;
  int raw_all;          // EP_BN_FUSED: keep y_raw for every channel block (the block's backward will not read y)
  int Mg, groups, Kc8g, bcast, ep;     // rows per group, groups, source channel blocks per group
  int KH, S, SV;                       // kernel rows, column stride, row stride (1 for 1-D convs: rows are batch items)
  int SRCH, SRCW, s_img, s_cblk, s_row;
  int OUTH, OUTW, o_img, o_cblk, o_row, o_sh, o_sw, o_ry, o_rx;
  int of_img, of_chan, of_row;         // out_f32 strides (elements)
  int PH, PW;
  int ltw, TH, PC, nchunks, tiles_x, tiles_y, gx, gy, gz;
  int nstg;                            // LDS-DMA path: buffers in the ring (2..CONV16_MAX_RING)
  int dbg;                             // tuning experiments only (ms_debug_set_conv16_ring): bit 0 no stores, 1 no statistics reduce, 2 no K loop
  int ncls, cls_PH[4], cls_PW[4], cls_OUTH[4], cls_OUTW[4], cls_ry[4], cls_rx[4];
  unsigned a_mt_stride, a_group_stride, a_cls_stride;   // vectors
  float slope, eps;
  int is_dgrad;
};

struct Conv16Plan {
  int ok, wm, wn, nwn, tw, th, tiles_y, tiles_x, n_tiles, ck8, nchunks, pc, lds_bytes, dma, nstg;
};
Conv16Plan plan_conv16(int nd, int Mg, int groups, int Kc, int KH, int KW, int SH, int SW, int B, int OH, int OW, int zmul,
                       bool up2);
// bytes of the prepared A operand for (rows per group Mg, groups, input channels Kc, taps) under plan pl, ncls classes
size_t conv16_weight_bytes(const Conv16Plan& pl, int Mg, int groups, int Kc, int KH, int KW, int ncls);
int launch_conv16(int dt, const Conv16Args& a, const Conv16Plan& pl, int KW, bool up2, double flops, double bytes,
                  hipStream_t s);
bool conv16_coresident(int dt, const Conv16Plan& pl, int KW, bool up2, int nwg);
extern int g_bn_fused;
extern int g_bn_fused_min_wgs;
extern int* g_bn_sync;
extern int g_bn_sync_n;
constexpr int BNF_SYNC_WORDS_PER_GROUP = 32;
constexpr int BN_BWD16_FUSED_MAX = 2048;     // pixels per channel up to which BatchNorm backward is one launch (launch_bn_bwd16)

// ---- BN_TRAIN blocks: what the backward pass reads.  BatchNorm + LeakyReLU backward needs x_hat and the sign of z per element.
// Both follow from the block's OUTPUT y = lrelu(z), z = x_hat * gamma + beta, when the map is invertible and well conditioned:
//   z = y > 0 ? y : y / slope;   x_hat = (z - beta) / gamma.
// Then the raw conv output y_raw need not be kept at all (the in-launch BatchNorm writes y only: one HBM pass, no extra store).
// The inversion amplifies y's rounding by (|x_hat| + |beta / gamma|); a channel is UNSAFE -- y_raw is kept and read for its
// 8-channel block -- when gamma is tiny, beta dominates gamma, or the activation is not invertible (ReLU).  Forward and
// backward evaluate this predicate on the SAME floats (the block's save vector: mean | invstd | scale | shift) with the same
// operations (explicit fma, single multiplies: nothing the compiler could contract differently in the two kernels).
__host__ __device__ inline bool bn_inv_unsafe(float mean, float invstd, float scale, float shift, float slope) {
  const float beta = fmaf(mean, scale, shift);
  return !(slope >= 1e-3f) || !(fabsf(scale) >= 1e-3f * invstd) || (fabsf(beta) * invstd > 4.0f * fabsf(scale));
}

// ---- weight preparation: fp32 master weights -> 16-bit A-operand stages (once per optimizer update)
struct Prep16Job {          // (narrow fields: 72 jobs fit the 4 KB of kernel arguments -- a generator's ~90 operands take 2 launches)
  const float* w;        // (groups*Cog, Cig, KH, KW) fp32
  void* out;
  const float* scale;    // optional per-output-channel factor (eval BatchNorm folded into the weights), forward only
  short groups, Cog, Cig, BM, nchunks, n_mt;
  unsigned char dgrad;   // 0: forward rows = output channels; 1: data gradient rows = input channels, taps reversed, classes
  unsigned char KH, KW, SH, SW, PH, PW, bcast, CK8, dt;
  int block_end;
};
enum { PREP16_BATCH_MAX = 72 };
static_assert(sizeof(Prep16Job) * PREP16_BATCH_MAX + 8 <= 4096, "Prep16Batch exceeds the kernel-argument segment");
struct Prep16Batch { int n; Prep16Job job[PREP16_BATCH_MAX]; };
int launch_prep16_multi(Prep16Batch& pb, hipStream_t s);

// ---- weight gradient (wgrad16.hip)
struct Wgrad16Args {
  const void* dyr;      // cb8 (B, groups*Cog, OH, OW)
  const void* src;      // cb8 x (UP2: a)
  const void* src2;     // UP2: r
  float* out;           // [splits][groups*Cog][Cig][KH][KW] fp32
  int Cog, Cig, groups, bcast, KH, KW, S, SV, PH, PW;
  int SRCH, SRCW, s_img, s_cblk, s_row;
  int OUTH, OUTW, o_img, o_cblk, o_row;
  int ltw, TH, PCX, tiles_x, tiles_y, n_tiles, tiles_per_split, splits;
  int ktg;              // tap groups per kernel row
  int nstg;             // LDS-DMA form: buffers in the ring
  int dbg;              // timing ablations (ms_debug_set_conv16_ring flag bits 4..6): no stores / no MFMAs / no tile loop
  int gx, gy, gz;
  int accumulate;       // splits == 1 only: out += result instead of out = result (queued launches)
  size_t out_split_stride;
};
struct Wgrad16Plan { int tp, tw, th, tiles_y, tiles_x, n_tiles, splits, tiles_per_split, ktg, pcx, lds_bytes, nstg, npx; };
Wgrad16Plan plan_wgrad16(int nd, int Cog, int Cig, int groups, int KH, int KW, int SH, int SW, int B, int OH, int OW, bool up2);
int launch_wgrad16(int dt, const Wgrad16Args& a, const Wgrad16Plan& pl, bool up2, double flops, double bytes, hipStream_t s);
// the same launch queued (process-wide queue) for wgrad16_flush: many blocks' kernels side by side in one multi-block launch
constexpr int WG16_MAX_JOBS = 20;    // 20 x sizeof(Wgrad16Args) + the table of block ranges stays below the 4 KB of kernel arguments
struct Wgrad16Batch {
  int n;
  int block_end[WG16_MAX_JOBS];      // exclusive prefix sums of the jobs' workgroup counts
  Wgrad16Args job[WG16_MAX_JOBS];
};
int queue_wgrad16(int dt, const Wgrad16Args& a, const Wgrad16Plan& pl, bool up2, double flops, double bytes);
bool wgrad16_c1_ok(const Wgrad16Args& a, bool up2);      // single-input-channel block: its own (vector-unit) kernel, launched at once
int wgrad16_flush(hipStream_t s);
void wgrad16_discard();

// ---- elementwise (elementwise16.hip)
// single-input-channel 3x3 block on the vector unit (conv16_c1.hip)
bool conv16_c1_ok(const ms_conv_desc* d);
int conv16_c1_tiles(const ms_conv_desc* d);
int launch_conv16_c1(int dt, const void* x, const float* w, const float* wscale, const float* bias, void* out, const float* bn_g, const float* bn_b,
                     const float* bn_m, const float* bn_v, float* stats, float* counts, int B, int H, int W, int ep, float slope,
                     float eps, hipStream_t s);
int launch_bn_apply16(int dt, const void* y_raw, void* y, float* y_f32, const float* save, int B, int C, int HW, float slope,
                      hipStream_t s);
int bwd16_chunks(int B, int C8, int HW, int* b_per_chunk);
int launch_bn_finalize_apply16(int dt, const float* stats, const float* counts, int n_tiles, int N, const float* gamma,
                               const float* beta, float* rm, float* rv, float* save, float eps, float momentum, const void* y_raw,
                               void* y, float* y_f32, int B, int C, int HW, float slope, hipStream_t s, int sg = 1);
// dy: cb8, or plain fp32 (B,C,HW) when dy_f32 != NULL
// y: the block's cb8 output, or NULL: x_hat and the activation mask always from y_raw (fp32-output blocks, callers without y).
// Only the one-launch form (B*HW <= BN_BWD16_FUSED_MAX pixels) reads y; blocks with more pixels keep y_raw whole (block_fwd16)
// and their two-pass backward reads it.
int launch_bn_bwd16(int dt, const void* dy, const float* dy_f32, const void* y_raw, const void* y, const float* save, const float* gamma,
                    float* partial, void* dyr, float* colpart, float* dbias, float* dgamma, float* dbeta, int B, int C, int HW,
                    float slope, int* bias_done, hipStream_t s, int sg = 1);
// mode 1: dyr = dy * lrelu'(y); mode 0: dyr = dy (written only when dy arrives as fp32); colsum partials always
int launch_act_bwd16(int dt, const void* dy, const float* dy_f32, const void* y, void* dyr, float* colpart, int B, int C, int HW,
                     int mode, float slope, hipStream_t s);
int launch_colsum16(const float* colpart, float* out, int C, int nchunk, hipStream_t s);
int launch_bn_fold(const float* bias, const float* g, const float* b, const float* m, const float* v, float* scale,
                   float* bias_out, int C, float eps, hipStream_t s);


// ---- block orchestration (api16.hip), reached from the ms_conv_block_* entry points when desc.dtype != MS_F32
size_t block_fwd16_workspace(const ms_conv_desc* d);
size_t block_bwd16_workspace(const ms_conv_desc* d);
int block_fwd16(const ms_conv_desc* d, const void* x, const void* x2, const float* w, const float* bias, const float* gamma,
                const float* beta, float* running_mean, float* running_var, void* y_raw, void* y, float* save, void* workspace,
                size_t workspace_bytes, hipStream_t s, const void* w_prepared, int32_t* bn_sync = nullptr, int bn_sync_words = 0);
int block_bwd16(const ms_conv_desc* d, const void* x, const void* x2, const float* w, const float* gamma, const void* y_raw,
                const void* y, const float* save, const void* dy, void* dyr, void* dx, void* dx2, float* dw, float* dbias,
                float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, hipStream_t s, const void* wt_prepared,
                float* wgrad_partials, int defer_wgrad_launch);
size_t weights16_bytes(const ms_conv_desc* d, int which);
int wgrad16_splits(const ms_conv_desc* d);

}  // namespace ms
