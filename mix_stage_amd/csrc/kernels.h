// Internal launcher interface shared by the translation units of libmixstage_hip.so.
#pragma once
#include "common.h"

namespace ms {

enum { EP_BARE = 0, EP_LRELU = 1, EP_BN_EVAL = 2, EP_RAW_STATS = 3, EP_DGRAD = 4, EP_DGRAD_UP2 = 5, EP_BN_FUSED = 6, EP_DGRAD_BN = 7 };

struct GatherArgs {
  const float* A;     // [groups][Mg][Kg]
  const float* src;   // gathered tensor
  const float* src2;  // UP2 forward: residual
  float* out;         // [B][groups*Mg][OUTH][OUTW]
  float* out2;        // UP2 dgrad: grad of the residual
  const float* bias;
  const float* bn_g;
  const float* bn_b;
  const float* bn_m;
  const float* bn_v;
  float* stats;  // EP_RAW_STATS: [n_tiles][groups*Mg][2] = (sum, M2 about the tile mean)
  int Mg, Kg, groups, Kc, src_ctotal, SRCH, SRCW, OUTH, OUTW, Npix;
  int KH, KW, SH, SW, PH, PW;
  int bcast, a_vec, ep;
  int batch;     // transposed (data gradient) only: B
  int splitk, k_per_split;   // split-K: blockIdx.z also enumerates K slices; raw partial tiles go to `part`
  float* part;               // [splitk][same layout as out]
  size_t part_stride;
  float slope, eps;
};

struct WgradArgs {
  const float* dyr;   // [B][groups*Cog][OH][OW]
  const float* src;   // x (UP2: half-resolution a)
  const float* src2;  // UP2: residual r
  float* out;         // [splits][groups*Cog][Kg]
  int Cog, Cig, Kg, groups, src_ctotal, H, W, OH, OW, Npix;
  int KH, KW, SH, SW, PH, PW;
  int bcast, splits, r_per_split;
  int accumulate;     // splits == 1: add to `out` instead of overwriting it (queued launches: wgrad_gather_flush)
};
constexpr int WG_MAX_JOBS = 24;     // (24 x 120 B of kernel arguments)
struct WgradBatch {
  int n;
  int block_end[WG_MAX_JOBS];
  WgradArgs job[WG_MAX_JOBS];
};

// RAII pair of HIP events around a launch (no-op unless ms_timing_enable(1)); timing.hip
struct TimingScope {
  TimingScope(hipStream_t s, double flops, double bytes, const char* fmt, ...);
  ~TimingScope();
  bool skip() const { return skip_; }   // ms_debug_set_skip matched this launch's label: the launcher returns without launching
  int idx_;
  hipStream_t s_;
  bool skip_;
};

// ---- patch-staged forward conv (conv_patch.hip)
struct PatchArgs {
  const float* A;      // [groups][Mg][Kg]
  const float* src;    // UP2: the half-resolution tensor a
  const float* src2;   // UP2: residual r (full resolution); strides below describe r / the plain input
  float* out;
  const float* bias;
  const float* bn_g;
  const float* bn_b;
  const float* bn_m;
  const float* bn_v;
  float* stats;        // EP_RAW_STATS: [n_tiles][ctot][2]
  float* counts;       // EP_RAW_STATS: [n_tiles] valid pixels per tile
  int Mg, Kg, groups, Kc, bcast, a_vec, ep;   // a_vec: 0 scalar A loads, 1 16-B loads, 2 A is the conv weight in place
  int SRCH, SRCW, s_img, s_chan, s_row;   // source image rows/cols and element strides
  int OUTH, OUTW, o_img, o_chan, o_row;
  int o_sh, o_sw, o_ry, o_rx;   // output scatter: (oy*o_sh + o_ry, ox*o_sw + o_rx); data gradient of a strided conv
  float* out2;                  // EP_DGRAD_UP2: gradient of the residual
  int is_dgrad;                 // label only
  int gx, gy, gz;               // logical grid (pixel tiles, channel tiles, groups*splitk); launched 1-D, XCD-remapped
  int splitk, chunks_per_split; // split-K over workgroups: raw partial tiles go to part[ks] (output layout)
  float* part;
  size_t part_stride;
  int PH, PW, tiles_x, tiles_y;
  float slope, eps;
  size_t src_elems, a_elems;    // extents of src (UP2: of src2) and A, for the 32-bit buffer offsets
  // data gradient of a strided conv: all SH*SW output-parity classes in ONE launch (blockIdx also enumerates the class);
  // per class: padding, output extent, scatter phase; weights of class c start at A + c*cls_a_stride
  int ncls, cls_PH[4], cls_PW[4], cls_OUTH[4], cls_OUTW[4], cls_ry[4], cls_rx[4];
  unsigned cls_a_stride;
  int cls_fast;       // the parity class is the fastest index of the logical workgroup id (conv_tile / conv_patch: shared dy window in L2)
  // bf16x6 kernels (conv_patch6.hip): pre-split weight planes [3][rows][a_row_elems] bf16; cls_a_stride counts ROWS there
  const unsigned short* Aplanes;
  unsigned plane_stride;       // elements per plane
  int a_row_elems;
};
struct PatchPlan { int ok, tm, tw, tiles_y, tiles_x, n_tiles, splitk, chunks_per_split, tn, wm, ksi, p6, tile; };   // wm: 32-row wave tiles per tile; ksi: intra-workgroup K split; tile: lean kernel (conv_tile.hip), 1 = 128 x 128, 2 = 64 x 256
PatchPlan plan_patch(int nd, int Mg, int groups, int Kc, int KH, int KW, int SH, int SW, int B, int OH, int OW, int zmul = 1, int in_w = 0);   // in_w: row length of the conv's input (0: unknown -> never the lean kernel)
extern int g_conv_tile;
bool conv_tile_shape_ok(int KH, int KW, int S);
int launch_tile(const PatchArgs& a, const PatchPlan& pl, int KH, int KW, int S, double flops, double bytes, hipStream_t s);
int patch_chunk_channels(int KH, int KW);
bool patch_dgrad_direct_ok(const float* w, int Cin_g, int KH, int KW, int SH, int SW, bool up2_or_bcast);
int launch_patch(const PatchArgs& a, const PatchPlan& pl, int KH, int KW, int S, bool up2, double flops, double bytes,
                 hipStream_t s);

// single-input-channel 3x3 conv on the VALU (conv_c1.hip)
bool conv_c1_ok(int groups, int Cin, int Cout, int KH, int KW, int SH, int SW, int PH, int PW, int H, int in_plain);
int conv_c1_tiles(int B, int H, int W);
// ... and its weight gradient: wgrad_c1_splits slabs of 576 partial sums (the caller reduces them like any split weight gradient)
bool wgrad_c1_ok(int groups, int Cin, int Cout, int KH, int KW, int SH, int SW, int PH, int PW, int H, int W, int in_plain);
int wgrad_c1_splits(int B, int H);
int launch_wgrad_c1(const float* dyr, const float* x, float* part, int B, int H, int W, hipStream_t s);
int launch_conv_c1(const float* x, const float* w, const float* bias, float* out, const float* bn_g, const float* bn_b,
                   const float* bn_m, const float* bn_v, float* stats, float* counts, int B, int H, int W, int ep, float slope,
                   float eps, hipStream_t s);
extern int g_patch_intra;
extern int g_precision;        // 0: exact-fp32 MFMA kernels; 1: bf16x6 split-operand kernels where they exist (conv_patch6.hip)
bool patch6_supported(int KH, int KW, int S);
int patch6_row_elems(int Kc, int KH, int KW);
int launch_split_weights(const float* w, unsigned short* planes, int rows, int Kc, int KH, int KW, hipStream_t s);
struct SplitJob {
  const float* w;            // [rows][Kc][KHW] fp32
  unsigned short* planes;    // [3][rows][row_elems] bf16
  int rows, Kc, KHW, CK, row_elems, block_end;
};
enum { SPLIT_BATCH_MAX = 64 };
struct SplitBatch { int n; SplitJob job[SPLIT_BATCH_MAX]; };
int launch_split_weights_multi(SplitBatch& sb, hipStream_t s);
int launch_patch6(const PatchArgs& a, const PatchPlan& pl, int KH, int KW, int S, bool up2, double flops, double bytes,
                  hipStream_t s);
extern int g_patch_min_wgs;   // test knob (ms_debug_set_patch_min_workgroups): 0 forces the patch kernels

// ---- patch-staged weight gradient (wgrad_patch.hip)
struct WgradPatchArgs {
  const float* dyr;    // [.., groups*Cog, OH, OW]
  const float* src;    // x (UP2: the half-resolution tensor a)
  const float* src2;   // UP2: residual r
  float* out;          // [splits][groups*Cog][Kg]
  int Cog, Cig, Kg, groups, bcast;
  int SRCH, SRCW, s_img, s_chan, s_row, PH, PW;
  int OUTH, OUTW, o_img, o_chan, o_row;
  int tiles_x, tiles_y, n_tiles, tiles_per_split, splits;
  int gx, gy, gz;      // logical grid (column tiles, channel tiles, groups*splits); launched 1-D, XCD-remapped
  int accumulate;      // splits == 1 only: out += result instead of out = result (queued launches, ms_wgrad_flush)
  int n_steps, steps_per_split;   // lean kernel (wgrad_wave_body): the reduction counted in 16-pixel runs of one output row
  int wave_kind;                  // ... its instance: stride * 2 + (tile == 64 x 256) ...
  int KH, KW, w4, nx4, xrp, xcp, xbuf;   // ... and the shape's geometry: taps, 16-byte slots per window row / in all, LDS pitches of the x window (row, channel), its size
  int* counters;       // per (group, channel tile, column tile) arrival counters: the last split sums the slabs in-launch
  float* final_out;    // dw, written by the last arriver
};
struct WgradPatchPlan { int ok, tw, tiles_y, tiles_x, n_tiles, splits, tiles_per_split, p6, wave; };   // wave: lean kernel's tile (0: the 64 x 64 kernel, 1: 128 x 128, 2: 64 x 256)
// queued launches: many blocks' weight gradients side by side in one multi-block launch per kernel instance
constexpr int WGP_MAX_JOBS = 48;    // (48 x 192 B of kernel arguments: every fp32 layer of the G-step in one launch)
struct WgradPatchBatch {
  int n;
  int block_end[WGP_MAX_JOBS];
  WgradPatchArgs job[WGP_MAX_JOBS];
};
int queue_wgrad_patch(const WgradPatchArgs& a, const WgradPatchPlan& pl, int KH, int KW, int S, bool up2, double flops, double bytes);
int wgrad_patch_flush(hipStream_t s);
void wgrad_patch_discard();
bool wgrad6_supported(int KH, int KW, int S);
int launch_wgrad_patch6(const WgradPatchArgs& a, const WgradPatchPlan& pl, int KH, int KW, int S, bool up2, double flops,
                        double bytes, hipStream_t s);
WgradPatchPlan plan_wgrad_patch(int nd, int Cog, int Kg, int groups, int KH, int KW, int SH, int SW, int B, int OH, int OW, int W = 0,
                                bool up2 = false);   // W: input row length (0: unknown -> the 64 x 64 kernel)
int launch_wgrad_patch(const WgradPatchArgs& a, const WgradPatchPlan& pl, int KH, int KW, int S, bool up2, double flops,
                       double bytes, hipStream_t s);
int launch_reduce_splits(const float* part, float* out, int n, int splits, hipStream_t s);

// ---- clip-resident 1-D conv blocks, fp32 (clip32.hip)
struct Clip32Args;
bool clip32_fwd_ok(const ms_conv_desc* d);
bool clip32_dgrad_ok(const ms_conv_desc* d);
size_t clip32_fwd_weight_bytes(const ms_conv_desc* d);
size_t clip32_dgrad_weight_floats(const ms_conv_desc* d);
size_t clip32_weight_floats(int rows, int red, int KW);
int clip32_prep_queue(const float* w, float* out, int rows, int red, int KW, int transposed, int w_cols, hipStream_t s);
int clip32_prep_flush(hipStream_t s);
void clip32_prep_discard();
void gdgrad32_prep_discard();
// the weight-preparation entry points queue jobs in process-wide batches and launch them at their end: an early error return must not leave
// queued jobs behind for the next, unrelated call to launch on its own stream against buffers that may be gone
struct PrepQueueGuard { ~PrepQueueGuard() { clip32_prep_discard(); gdgrad32_prep_discard(); } };
size_t clip32_part_bytes(int rows, int npw);
int clip32_sync_words(int rows);
// forward of block d (ep by mode) / data gradient of a k3 s1 block; -2: not resident at once, use the other kernels
int clip32_block_fwd(const ms_conv_desc* d, const float* x, const float* x2, const float* wp, const float* bias, const float* gamma,
                     const float* beta, float* rm, float* rv, float* y_raw, float* y, float* save, float* part, int* sync,
                     int sync_words, hipStream_t s);
// BatchNorm + LeakyReLU backward of the block that PRODUCED the input of block d (its output y / y_raw, save, gamma; gradient slots),
// carried out in the epilogue of d's data-gradient launch (EP_DGRAD_BN): dx then holds dy_raw of the producer
struct Clip32PrevBN { const float* y; const float* y_raw; const float* save; const float* gamma; float* dgamma; float* dbeta; float* dbias; float slope; };
size_t clip32_dgrad_bn_part_bytes(const ms_conv_desc* d);
bool clip32_dgrad_bn_ok(const ms_conv_desc* d);
int clip32_block_dgrad(const ms_conv_desc* d, const float* g, const float* wp, float* dx, float* dx2, hipStream_t s,
                       const Clip32PrevBN* pv = nullptr, float* part = nullptr, int* sync = nullptr, int sync_words = 0,
                       const float* accum = nullptr);

// data gradient of the grouped decoder blocks (chain32.hip: gconv32_kernel)
bool gdgrad32_ok(const ms_conv_desc* d);
size_t gdgrad32_weight_floats(const ms_conv_desc* d);
int gdgrad32_prepare(const ms_conv_desc* d, const float* w, float* out, hipStream_t s);      // queued ...
int gdgrad32_prep_flush(hipStream_t s);                                                       // ... until this
int gdgrad32_launch(const ms_conv_desc* d, const float* g, const float* wp, float* dx, hipStream_t s);

// ---- chained pose decoder (chain32.hip: fp32; chain16.hip: bf16 / fp16), behind ms_decoder_chain_*
int chain32_supported(const ms_chain_desc* d);
size_t chain32_prepared_bytes(const ms_chain_desc* d);
size_t chain32_workspace(const ms_chain_desc* d);
int chain32_sync_words(const ms_chain_desc* d);
int chain32_prepare(const ms_chain_desc* d, const float* const* w, const float* wl, void* prepared, hipStream_t s);
int chain32_fwd(const ms_chain_desc* d, const ms_chain_tensors* tn, void* workspace, size_t workspace_bytes, hipStream_t s);
int chain16_supported(const ms_chain_desc* d);
size_t chain16_prepared_bytes(const ms_chain_desc* d);
size_t chain16_workspace(const ms_chain_desc* d);
int chain16_sync_words(const ms_chain_desc* d);
int chain16_prepare(const ms_chain_desc* d, const float* const* w, const float* wl, void* prepared, hipStream_t s);
int chain16_fwd(const ms_chain_desc* d, const ms_chain_tensors* tn, void* workspace, size_t workspace_bytes, hipStream_t s);

struct GatherPlan { int tm, tn, splitk, k_per_split, n_tiles; };
// tile shape + split-K factor for an (Mg x npix) output per z-slice (z = groups * parity classes), reduction Kg
GatherPlan plan_gather(int Mg, int npix, int zcount, int Kg);
int launch_gather(GatherArgs a, bool transposed, bool up2, const GatherPlan& plan, hipStream_t s);
// sums the split-K partial tiles and applies the block epilogue; one workgroup per output channel
int launch_splitk_fwd_epilogue(const float* part, int splitk, size_t part_stride, const float* bias, const float* gamma,
                               const float* beta, float* rm, float* rv, float* y_raw, float* y, float* save, int B, int C,
                               int HW, int ep, float slope, float eps, float momentum, hipStream_t s, int sg = 1);
int launch_splitk_dgrad_epilogue(const float* part, int splitk, size_t part_stride, float* dx, float* dx2, size_t n, int W,
                                 int up2, hipStream_t s);
int wgrad_splits(int Cog, int Kg, int groups, int Npix);
int launch_wgrad(WgradArgs a, bool up2, float* dw, float* partial_ws, bool defer_reduce, hipStream_t s, bool queue = false);
int wgrad_gather_flush(hipStream_t s);
void wgrad_gather_discard();
struct ReduceJob {
  const float* part;   // [splits][n]
  float* out;          // [n], accumulated into
  int n, splits, wave, block_end;
};
enum { REDUCE_BATCH_MAX = 96 };   // 96 x 32 B of kernel arguments
struct ReduceBatch { int n; ReduceJob job[REDUCE_BATCH_MAX]; };
int launch_reduce_splits_multi(ReduceBatch& rb, hipStream_t s);
struct TransposeJob {
  const float* w;
  float* wt;
  int groups, Cog, Cig, KH, KW, SH, SW, PH, PW, flip;
  int block_end;     // multi launch: one past this job's last workgroup
};
enum { TRANSPOSE_BATCH_MAX = 48 };   // 48 x 64 B of kernel arguments
struct TransposeBatch { int n; TransposeJob job[TRANSPOSE_BATCH_MAX]; };
int launch_transpose_weight_multi(TransposeBatch& tb, hipStream_t s);
int launch_transpose_weight(const float* w, float* wt, int groups, int Cog, int Cig, int KH, int KW, int SH, int SW,
                            int PH, int PW, int flip, hipStream_t s);
size_t dgrad_weight_elems(int groups, int Cog, int Cig, int KH, int KW, int SH, int SW);
int launch_bn_finalize(const float* stats, const float* counts, int n_tiles, int tile_n, int N, int C, const float* gamma,
                       const float* beta, float* rm, float* rv, float* save, float eps, float momentum, hipStream_t s);
int launch_bn_apply(const float* y_raw, float* y, const float* save, int C, int HW, size_t total, float slope, hipStream_t s);
int launch_bn_finalize_apply(const float* stats, const float* counts, int n_tiles, int tile_n, int N, int C, const float* gamma,
                             const float* beta, float* rm, float* rv, float* save, float eps, float momentum, const float* y_raw,
                             float* y, int B, int HW, float slope, hipStream_t s);
int bwd_chunks(int B, int C, int* b_per_chunk);
// values per channel (B * OH * OW) up to which BatchNorm backward is ONE launch that can take x_hat from the block's output y
// (elementwise.hip: bn_bwd_fused*): the in-launch forward forms keep y_raw only where that inversion is unsafe
constexpr int BN_BWD32_FUSED_MAX = 256 * 16;
int launch_bn_bwd(const float* dy, const float* y_raw, const float* y, const float* save, const float* gamma, float* partial, float* dyr,
                  float* colpart, float* dbias, float* dgamma, float* dbeta, int B, int C, int HW, float slope, int* fused,
                  hipStream_t s, int sg = 1);
int launch_act_bwd(const float* dy, const float* y, float* dyr, float* colpart, float* dbias, int B, int C, int HW, int mode, float slope, int* fused,
                   hipStream_t s);
int launch_colsum_finalize(const float* colpart, float* out, int B, int C, hipStream_t s);

}  // namespace ms
