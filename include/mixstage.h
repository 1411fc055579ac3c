/* mixstage.h -- C-ABI of libmixstage_hip.so: the MI355X (gfx950) kernels under the
 * Mix-StAGE audio->pose GAN path.
 *
 * The reference (chahuja/mix-stage) has no FFI layer: its path is pure Python on torch.nn
 * (SURVEY.md section 8b).  The entry points below are therefore what a binding of the reference's
 * nn.Module calls would bind; each cites the reference lines it replaces
 * (paths relative to /root/reference/src/model).
 *
 * Conventions
 *   - plain C: device pointers (HBM), sizes, one POD descriptor; no torch types.
 *   - all tensors are dense row-major fp32 in the reference's (B, C, T) / (B, C, H, W) layout:
 *     the time (or frequency) axis is contiguous.  Blocks whose descriptor says dtype = MS_BF16 / MS_F16 take their
 *     activation tensors (x, x2, y_raw, y, dy, dyr, dx, dx2) as 16-bit "cb8" buffers through the same pointer
 *     parameters (see ms_dtype); parameters, statistics and parameter gradients stay fp32.
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued, never synchronised;
 *     nothing is allocated inside (callers pass outputs and workspace) -> graph-capturable.
 *   - return 0 on success, negative on error; ms_last_error() gives the text (thread-local).
 */
#ifndef MIXSTAGE_H_
#define MIXSTAGE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3: ms_bwd_options grew (prev_*, bn_sync, dy_is_dyr), MS_DT_STAT_PAIR (save holds two vectors), ms_lp_mean_pair_*, Adam state words 2 / 3 */
#define MS_ABI_VERSION 4

/* epilogue of a conv block */
enum ms_block_mode {
  MS_BARE = 0,     /* conv + bias                        (nn.Conv1d: JL:83 logits, layers.py:459, S2G:63) */
  MS_LRELU = 1,    /* conv + bias + LeakyReLU            (S2G:50-51 D.conv1)                              */
  MS_BN_TRAIN = 2, /* conv + bias + BatchNorm(batch stats, running update) + LeakyReLU  (layers.py:78)    */
  MS_BN_EVAL = 3   /* conv + bias + BatchNorm(running stats) + LeakyReLU                (layers.py:78)    */
};

/* how the block reads its input */
enum ms_input_mode {
  MS_IN_PLAIN = 0,
  MS_IN_BCAST = 1, /* every group reads the same Cin/groups... channels: replaces torch.cat([x]*M,1) JL:190 */
  MS_IN_UP2ADD = 2 /* 1-D only: x = nearest_up2(a) + r, replaces layers.py:151 upconv(x)+residual        */
};

/* Arithmetic / storage type of a conv block (ms_conv_desc.dtype, low byte).
 *   MS_F32   fp32 tensors in the reference's (B, C, T) / (B, C, H, W) layout, exact fp32 matrix products (the parity path).
 *   MS_BF16  / MS_F16: 16-bit operands on v_mfma_f32_32x32x16_{bf16,f16}, fp32 accumulation, fp32 BatchNorm statistics,
 *            fp32 master weights, biases, BN parameters and weight gradients.  Activations and their gradients live in HBM
 *            in the channel-blocked layout "cb8": [B][ceil(C/8)][H][W][8] 16-bit elements (one 16-byte vector = 8
 *            consecutive channels of one pixel, pad channels zero; grouped blocks need channels-per-group % 8 == 0).
 *            What trainer.py:138 selects by casting the model (`.double()` there; the reference runs any dtype its modules
 *            are cast to) -- BASELINE configs[1], [3] (bf16) and [4] (fp16, eval).
 * MS_DT_OUT_F32 (flag): the block's OUTPUT y is a plain fp32 (B, C, OH, OW) tensor (score-producing blocks whose consumers
 *            are the fp32 loss kernels); dy of its backward then is plain fp32 too.  y_raw stays cb8.
 * MS_DT_BN_FOLDED (flag, BN_EVAL blocks): inference form -- the eval BatchNorm is folded into the prepared forward weights
 *            (w * gamma/sqrt(var+eps)) and bias; valid while the running statistics do not change (sampling).  Without it the
 *            epilogue applies the running statistics as they are at launch time. */
enum ms_dtype { MS_F32 = 0, MS_BF16 = 1, MS_F16 = 2 };
#define MS_DT_OUT_F32 0x100
#define MS_DT_BN_FOLDED 0x200
/* MS_DT_STAT_PAIR (flag, every arithmetic mode): the batch holds TWO passes of the module side by side -- clips [0, B/2) and
 * [B/2, B) -- as gan.py:120,126 runs the discriminator on the fake and then on the real poses.  BN_TRAIN blocks take their batch
 * statistics per half (save holds 2 x 4*C floats: the first half's vector, then the second half's), update the running statistics
 * twice in that order, and their backward reduces per half; blocks without BatchNorm just see a batch of B.  Gradients of the
 * weights are the sum over both halves, which is what two separate backward passes accumulate.  B must be even.
 * ms_stat_pair_ok(d) says whether block d's kernels implement the flag. */
#define MS_DT_STAT_PAIR 0x400

/* Geometry of one conv block (1-D convs use H = KH = SH = 1, PH = 0). */
typedef struct ms_conv_desc {
  int32_t B;          /* batch                                                   */
  int32_t Cin;        /* input channels PER GROUP                                */
  int32_t H, W;       /* input spatial size (H = 1 for Conv1d; W = time)         */
  int32_t Cout;       /* output channels PER GROUP                               */
  int32_t groups;
  int32_t KH, KW, SH, SW, PH, PW;
  int32_t OH, OW;     /* output spatial size                                     */
  int32_t mode;       /* ms_block_mode                                           */
  int32_t in_mode;    /* ms_input_mode                                           */
  float slope;        /* LeakyReLU negative slope (0.2 on the path, 0 = ReLU)    */
  float eps;          /* BatchNorm eps                                           */
  float momentum;     /* BatchNorm momentum                                      */
  int32_t dtype;      /* ms_dtype | flags; 0 = fp32                              */
} ms_conv_desc;

const char* ms_last_error(void);
int ms_abi_version(void);
/* 1 when forward and backward of block d (d->dtype carrying MS_DT_STAT_PAIR) are implemented for its geometry; 0: run the two
 * passes one after the other. */
int ms_stat_pair_ok(const ms_conv_desc* d);

/* Bytes of scratch the forward / backward of this block needs. */
size_t ms_conv_block_fwd_workspace(const ms_conv_desc* d);
size_t ms_conv_block_bwd_workspace(const ms_conv_desc* d);

/* ConvNormRelu.forward (layers.py:77-78), nn.Conv1d.forward, D.conv1 (S2G:50-51,68).
 *   x        in_mode PLAIN: (B, groups*Cin, H, W); BCAST: (B, Cin, H, W); UP2ADD: a = (B, groups*Cin, W/2)
 *   x2       UP2ADD only: residual r (B, groups*Cin, W); else NULL
 *   w        (groups*Cout, Cin, KH, KW)   bias (groups*Cout)
 *   gamma,beta,running_mean,running_var   (groups*Cout)   BN modes only; running_* updated in BN_TRAIN
 *   y_raw    BN_TRAIN only: conv+bias output kept for backward, same shape as y
 *   y        (B, groups*Cout, OH, OW) block output
 *   save     BN_TRAIN only: 4*groups*Cout floats = mean | invstd | scale | shift
 * Alignment: every tensor starts on a 16-byte boundary (what any allocator of device memory returns; torch's are 256-byte aligned).
 * The kernels load 16 bytes per lane wherever a row allows it; the 2-D layers with at least 256 output tiles have no other form, and a
 * call that hands them an odd pointer (a sliced view with an element offset) returns an error that names it instead of computing.
 */
int ms_conv_block_fwd(const ms_conv_desc* d, const float* x, const float* x2, const float* w,
                      const float* bias, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, float* y_raw, float* y, float* save, void* workspace,
                      size_t workspace_bytes, void* stream);

/* Full form of the forward.  w_planes: this block's buffer from ms_fwd_weights_prepare (bf16x6 mode: the weights split into
 * three bf16 planes); NULL = split them per call into the scratch. */
typedef struct ms_fwd_options {
  const void* w_planes;
  /* 16-bit BN_TRAIN blocks: zero-initialised int32 words through which the workgroups of the launch meet for the batch statistics
   * (BatchNorm inside the conv launch).  Give every (device, stream) that runs such blocks concurrently a buffer of its own;
   * NULL = the process-wide buffer of ms_set_bn_sync_buffer.  Word 0 is raised when a workgroup gave up waiting (the block's
   * output is then NaN, the running statistics are left alone).
   * fp32 1-D BN_TRAIN blocks (MS_F32): the same words for the clip-resident one-launch form -- owned by ONE (device, stream, block
   * shape): the counters are monotonic and launches of different shapes must not share them.  Blocks of at most 4096 values per
   * channel that ran in this form have y_raw written only for the channels whose BatchNorm + LeakyReLU map does not invert safely
   * from y (tiny gamma, beta dominating gamma, ReLU): ms_conv_block_bwd must then be handed y (it reads y_raw for exactly those
   * channels; handing y is always allowed). */
  int32_t* bn_sync;
  int32_t bn_sync_words;
} ms_fwd_options;
int ms_conv_block_fwd_ex(const ms_conv_desc* d, const float* x, const float* x2, const float* w,
                         const float* bias, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, float* y_raw, float* y, float* save, void* workspace,
                         size_t workspace_bytes, void* stream, const ms_fwd_options* opt);
/* bf16x6 mode (ms_set_precision): bytes of the block's split weight planes (0: the block runs the fp32 kernels), and their
 * batched construction for n blocks in one launch -- once per optimizer update, like ms_dgrad_weights_prepare. */
size_t ms_fwd_weights_bytes(const ms_conv_desc* d);
int ms_fwd_weights_prepare(int n, const ms_conv_desc* descs, const float* const* w, void* const* planes, void* stream);

/* Backward of the same block (what autograd derives for layers.py:78 in the reference).
 *   dy       grad wrt y.
 *   y_raw/save as produced by the forward (BN_TRAIN); y (LRELU mode mask); BN_EVAL uses running stats.
 *   dyr      scratch, same shape as y: grad wrt the raw conv output (returned for inspection)
 *   dx       grad wrt x  (UP2ADD: grad wrt a, (B, groups*Cin, W/2)); NULL = skip
 *   dx2      UP2ADD: grad wrt r; else NULL
 *   dw,dbias,dgamma,dbeta   NULL = skip (dw/dbias together, dgamma/dbeta together)
 */
int ms_conv_block_bwd(const ms_conv_desc* d, const float* x, const float* x2, const float* w,
                      const float* gamma, const float* running_mean, const float* running_var,
                      const float* y_raw, const float* y, const float* save, const float* dy, float* dyr,
                      float* dx, float* dx2, float* dw, float* dbias, float* dgamma, float* dbeta,
                      void* workspace, size_t workspace_bytes, void* stream);

/* Full form of the backward.  All fields optional (zero = the behaviour of ms_conv_block_bwd). */
typedef struct ms_bwd_options {
  void* side_stream;            /* experimental (mixstage_aux.h): weight gradient on this stream; NULL */
  void* side_workspace;
  size_t side_workspace_bytes;
  const float* wt_prepared;     /* this block's buffer from ms_dgrad_weights_prepare (built for the same descriptor and
                                 * the current w): the backward then skips its per-call transposed copy of w */
  float* wgrad_partials;        /* ms_wgrad_partials_elems(d) floats: the pixel-split partial weight gradients are left
                                 * here and dw is NOT written; the caller adds them into dw later, for many blocks in one
                                 * launch, with ms_wgrad_reduce_multi.  Ignored by blocks whose dw needs no split. */
  int defer_wgrad_launch;       /* the block's weight-gradient kernel is not launched but queued; ms_wgrad_flush launches
                                 * every queued block in a few multi-block launches (the small layers' weight gradients
                                 * are latency-bound one-wave kernels: side by side they cost one such latency).  x, x2,
                                 * dy / dyr and dw / wgrad_partials must stay valid until the flush; a queued kernel that
                                 * writes dw itself ADDS to it (dw holds zeros or the step's other contributions).  Blocks
                                 * whose kernel cannot be queued (bf16x6, im2col-gather path) launch at once as before. */
  /* --- fusing the BatchNorm + LeakyReLU backward of the block that PRODUCED this block's input into this block's data-gradient
   * launch (layers.py:77-78 under loss.backward(), trainer.py:1139; fp32 1-D blocks that ms_dgrad_fuses_prev_bn() accepts).
   * prev_*: the producer's output y, y_raw, save vector, gamma and gradient slots (prev_dgamma / prev_dbeta / prev_dbias may be
   * NULL).  dx then receives the gradient w.r.t. the producer's CONV output (its dy_raw) instead of the gradient w.r.t. y: the
   * caller runs the producer's own backward with dy_is_dyr = 1 on it.  bn_sync: zeroed int32 counters as in ms_fwd_options. */
  const float* prev_y;
  const float* prev_y_raw;
  const float* prev_save;
  const float* prev_gamma;
  float* prev_dgamma;
  float* prev_dbeta;
  float* prev_dbias;
  float prev_slope;
  int32_t* bn_sync;
  int32_t bn_sync_words;
  int dy_is_dyr;                /* this block's `dy` already is the gradient w.r.t. its conv output (a consumer's fused launch made it):
                                 * no BatchNorm / activation backward here, dgamma / dbeta / dbias are not written, dyr is not used */
  const float* dx_accum;        /* (ABI 4) same shape as dx: ADDED to the data gradient inside its launch -- the gradient that another
                                 * consumer of this block's input has already produced (the UNet's residual uses every down-path
                                 * output twice, layers.py:139-151: as the next block's input and as the up path's residual), so
                                 * that dx leaves as the complete gradient of x and no separate accumulation launch runs.  Only for
                                 * blocks ms_dgrad_takes_accum() accepts; composes with prev_*: the sum is what the producer's
                                 * BatchNorm backward sees. */
} ms_bwd_options;
/* 1 when block d's data-gradient launch can add ms_bwd_options.dx_accum itself (fp32 clip-resident 1-D blocks, plain input). */
int ms_dgrad_takes_accum(const ms_conv_desc* d);
/* 1 when block d's data-gradient launch can carry the BatchNorm backward of the producer of its input (see ms_bwd_options.prev_*). */
int ms_dgrad_fuses_prev_bn(const ms_conv_desc* d);
int ms_conv_block_bwd_ex(const ms_conv_desc* d, const float* x, const float* x2, const float* w,
                         const float* gamma, const float* running_mean, const float* running_var,
                         const float* y_raw, const float* y, const float* save, const float* dy, float* dyr,
                         float* dx, float* dx2, float* dw, float* dbias, float* dgamma, float* dbeta,
                         void* workspace, size_t workspace_bytes, void* stream, const ms_bwd_options* opt);

/* 16-bit modes: the matrix operands of a block's forward (which = 0; eval BatchNorm folded in: w*scale, and the folded
 * scale | bias behind them) and data gradient (which = 1), converted from the fp32 master weights into the kernels' staging
 * order.  A trainer builds them once per optimizer update, for many blocks in one launch, and passes them back through
 * ms_fwd_options.w_planes / ms_bwd_options.wt_prepared; without them every call converts into its scratch. */
typedef struct ms_prep16_item {
  const ms_conv_desc* desc;
  const float* w;
  const float* bias;            /* BN_EVAL folding only (may be NULL) */
  const float* gamma;
  const float* beta;
  const float* running_mean;
  const float* running_var;
  void* fwd;                    /* ms_weights16_bytes(desc, 0) bytes, or NULL: skip */
  void* dgrad;                  /* ms_weights16_bytes(desc, 1) bytes, or NULL: skip */
} ms_prep16_item;
size_t ms_weights16_bytes(const ms_conv_desc* d, int which);
int ms_weights16_prepare(int n, const ms_prep16_item* items, void* stream);

/* 16-bit modes, BN_TRAIN blocks: BatchNorm INSIDE the conv launch.  layers.py:77-78 is one expression, relu(norm(conv(x))); with
 * this buffer registered, a block whose conv grid is resident on the device all at once computes it in ONE launch and one HBM
 * pass: the workgroups that share a channel tile exchange their partial batch statistics through `zeroed_words` (arrival
 * counters) and the block's scratch, and normalise + activate their accumulators from registers; there is no separate
 * normalising launch.  The backward pass recovers the pre-activation from y (z = y > 0 ? y : y / slope, x_hat = (z - beta) /
 * gamma), so y_raw is written only for the 8-channel blocks that hold a channel where that inversion is ill-conditioned (tiny
 * gamma relative to beta or to 1, slope 0); `save` carries what both passes derive that decision from, and ms_conv_block_bwd
 * must be given y as well as y_raw for these blocks.  Layers of more than 2048 pixels per channel keep y_raw whole.  Larger grids
 * keep the two-launch form (same outputs).
 *   zeroed_words  persistent, ZERO-INITIALISED int32 device buffer of n words (65536 recommended; 32 per channel tile and
 *                 group of the largest block + 32), owned by the caller, serving ONE device and ONE stream at a time (blocks
 *                 launched concurrently on two streams must not share it); the kernels restore the zeros.  Word 0 is an error
 *                 flag: non-zero after a workgroup gave up waiting for its peers (the caller copies word 0 back
 *                 when it wants to check).
 *   NULL / 0 unregisters: every BN_TRAIN block runs conv + statistics, then the normalising launch. */
int ms_set_bn_sync_buffer(int32_t* zeroed_words, int n);

/* cb8 <-> fp32 at the boundaries of the 16-bit path (dtype = MS_BF16 / MS_F16):
 *   plain: fp32 (B, C, HW) channel-major, the layout of the fp32 kernels;  btc: fp32 (B, T, C) time-major, the layout of the
 *   reference's pose tensors (layers.py:229,280).  velocity: v[t] = x[t] - x[t-1], v[0] = 0 (gan.py:47-52) fused into the
 *   conversion; velocity_bwd: its adjoint fused into the conversion back. */
int ms_cb8_from_plain(int dtype, const float* x, void* y, int B, int C, int HW, void* stream);
int ms_cb8_to_plain(int dtype, const void* x, float* y, int B, int C, int HW, void* stream);
int ms_cb8_from_btc(int dtype, const float* x, void* y, int B, int T, int C, int velocity, void* stream);
int ms_cb8_to_btc(int dtype, const void* x, float* y, int B, int T, int C, int velocity_bwd, void* stream);

/* Data-gradient weights.  The data gradient of a block multiplies by w transposed (split into stride-parity classes for
 * strided convs); ms_conv_block_bwd builds that copy in its scratch on every call.  It depends on w and the descriptor
 * only, so a trainer builds it once per optimizer update -- for many blocks in ONE launch -- and passes it back through
 * ms_bwd_options.wt_prepared (autograd has no counterpart: torch's conv backward re-derives its layouts per call).
 *   ms_dgrad_weights_elems   floats the block's copy needs; 0 = its data gradient reads w in place, nothing to prepare
 *   ms_dgrad_weights_prepare descs[n], w[n], wt[n] (wt[i] may be NULL where elems == 0) */
size_t ms_dgrad_weights_elems(const ms_conv_desc* d, const float* w);
/* Arithmetic of the patch-staged conv kernels.  0 (default): exact fp32 products on v_mfma_f32_32x32x2_f32.
 * 1 ("bf16x6"): both fp32 operands split exactly into three bf16 parts, 6 of the 9 partial products on
 * v_mfma_f32_32x32x16_bf16, fp32 accumulation -- the dropped products are <= 2^-24 relative (one fp32 rounding error), measured
 * error against fp64 equals the fp32 kernels'; 16/6 of the fp32 matrix rate.  Returns the previous mode. */
int ms_set_precision(int mode);
int ms_get_precision(void);
int ms_tuning_epoch(void);   /* bumped whenever the dispatch changes (precision mode, tuning aids): prepared weights built
                              * under another value are stale */
int ms_dgrad_weights_prepare(int n, const ms_conv_desc* descs, const float* const* w, float* const* wt, void* stream);

/* Deferred weight-gradient reduction.  Small layers split the pixel reduction of dw over workgroups and sum the partial
 * slabs in a second kernel; a trainer can leave the slabs (ms_bwd_options.wgrad_partials) and sum all blocks' slabs in ONE
 * launch at the end of the backward pass.  dw[i][e] += sum_k partials[i][k][e], k ascending (same order as the per-block
 * kernel), so dw must hold zeros or the step's other contributions.
 *   ms_wgrad_partials_elems   floats of slab space for block d (0: dw is written directly), *splits = slab count */
size_t ms_wgrad_partials_elems(const ms_conv_desc* d, int* splits);
/* Planner hint: on = 1 when the caller queues the weight gradients of a backward pass (ms_bwd_options.defer_wgrad_launch) and
 * launches them side by side with ms_wgrad_flush -- a layer then need not fill the chip alone, its workgroups take at least
 * `tiles_per_workgroup` (<= 0: keep, default 16) 64-pixel tiles of the pixel reduction: fewer, longer workgroups and fewer partial
 * slabs.  Changes ms_wgrad_partials_elems / workspace sizes (ms_tuning_epoch moves): set it before sizing buffers.  Returns the
 * previous `on`.  (Reference: the weight gradients of loss.backward(), trainer.py:1139, have no consumer before the optimizer step.) */
int ms_set_wgrad_batched(int on, int tiles_per_workgroup);
/* Launches the weight gradients queued by ms_bwd_options.defer_wgrad_launch on `stream`; call it
 * before ms_wgrad_reduce_multi.  ms_wgrad_discard drops the queue (after a failed backward pass).  Returns 0 or an error. */
int ms_wgrad_flush(void* stream);
int ms_wgrad_discard(void);
int ms_wgrad_reduce_multi(int n, const float* const* partials, float* const* dw, const int* elems, const int* splits,
                          void* stream);

/* Cross-rank ("global") BatchNorm for data-parallel training: the reference normalises over the whole batch on one device
 * (layers.py:65-70); with the batch sharded over ranks, a block = conv (MS_BARE block) + the pieces below around two small
 * exchanges the host performs (all-gather of the statistics, all-reduce of the backward sums):
 *   ms_bn_stats        stats[C][2] = per channel (sum, M2 about the local mean) of y_raw (B, C, HW)
 *   ms_bn_train_apply  stats_all[world][C][2] (all-gathered, n_local = B*HW values each) -> save = mean|invstd|scale|shift of
 *                      the GLOBAL batch, running statistics update (global count), y = lrelu(y_raw*scale + shift)
 *   ms_bn_bwd_sums     sums[C][2] = this rank's (sum dz, sum dz*xhat), dz = dy*lrelu'(z): also its dbeta / dgamma
 *   ms_bn_bwd_apply    dyr = gamma*invstd*(dz - S1/N - xhat*S2/N) with the all-reduced sums S and the global count N */
int ms_bn_stats(const float* y_raw, float* stats, int B, int C, int HW, void* stream);
int ms_bn_train_apply(const float* stats_all, int world, int n_local, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, const float* y_raw, float* y, float* save, int B, int C, int HW, float eps,
                      float momentum, float slope, void* stream);
size_t ms_bn_bwd_workspace(int B, int C);
int ms_bn_bwd_sums(const float* dy, const float* y_raw, const float* save, float* sums, int B, int C, int HW, float slope,
                   void* workspace, size_t workspace_bytes, void* stream);
int ms_bn_bwd_apply(const float* dy, const float* y_raw, const float* save, const float* gamma, const float* sums_global,
                    double n_global, float* dyr, int B, int C, int HW, float slope, void* stream);

/* AudioEncoder resize (layers.py:197): bilinear to (T,1), align_corners=False, == 1-D lerp in time
 * of frequency column F/2.   x (B,C,Tin,F) -> y (B,C,Tout). */
int ms_lerp_time_fwd(const float* x, float* y, int B, int C, int Tin, int F, int Tout, void* stream);
int ms_lerp_time_bwd(const float* dy, float* dx, int B, int C, int Tin, int F, int Tout, void* stream);

/* Mixture of the M sub-generators (JL:106-115,186-187,194) fused with the softmax of the cluster
 * scores:  soft = softmax_m(score[b,:,t]);  out[b,t,f] = sum_m soft[b,t,m] * z[b, m*P+f, t].
 *   z (B, M*P, T)   score (B, M, T)   soft (B, T, M)   out (B, T, P) */
int ms_softmax_mix_fwd(const float* z, const float* score, float* soft, float* out, int B, int M, int P,
                       int T, void* stream);
/* dout (B,T,P), dsoft_extra (B,T,M) or NULL (extra grad flowing into soft, e.g. none on the path)
 * -> dz (B,M*P,T), dscore (B,M,T) (overwritten) */
int ms_softmax_mix_bwd(const float* z, const float* soft, const float* dout, float* dz, float* dscore,
                       int B, int M, int P, int T, void* stream);

/* The pose decoder as ONE launch: decoder.0-3 (grouped Conv1d k3 + BatchNorm1d + LeakyReLU, JL:69-77,190-192, layers.py:77-78) +
 * the grouped 1x1 `logits` conv (JL:83,193) + softmax over the cluster scores and the mixture of the M sub-generators
 * (JL:106-115,186-187,194).  A workgroup carries ONE clip of ONE sub-generator through all blocks with the activations resident
 * in LDS (a k3 / pad 1 conv never looks across a clip); only BatchNorm's batch statistics (train mode) and the mixture's
 * per-group terms cross workgroups, inside the launch.  The (B, M*P, T) logits tensor of JL:193 is never needed by the forward
 * pass (z is an optional output for callers whose backward pass reads it).
 *   geometry      T = 64 frames, C = 256 channels per group and block, n_blocks = 4, block 0 reads cin0 in (256, 272] channels
 *                 shared by all groups (the broadcast of JL:190), P <= 128; B*M <= compute units of the device (all workgroups
 *                 resident at once: ms_decoder_chain_supported says whether this device / shape qualifies -- else run the blocks
 *                 one by one with ms_conv_block_fwd).
 *   mode          MS_BN_TRAIN: batch statistics over all B clips, running statistics updated, `save` written; MS_BN_EVAL: running
 *                 statistics as they are.
 *   dtype         MS_F32: x (B, cin0, T), y_raw / y (B, M*C, T) fp32.  MS_BF16 / MS_F16: x, y_raw, y are cb8 tensors
 *                 ((B, ceil(cin0/8), T, 8) and (B, M*C/8, T, 8)), 16-bit operands with fp32 accumulation and fp32 statistics;
 *                 y_raw is written only where the backward pass reads it (keep_all_raw); z, soft, out stay fp32.
 *   x             block 0's input.  score (B, M, T): the cluster scores.  w[l] (M*C, cin_l, 3), w_logits (M*P, C, 1): the
 *                 modules' own weights -- read by ms_decoder_chain_prepare only; the launch streams `prepared`.
 *   y_raw, y, save  per block, or NULL: conv + bias, block output, mean|invstd|scale|shift (what ms_conv_block_bwd reads).
 *   z             (B, M*P, T) or NULL.  soft (B, T, M) or NULL: softmax of the scores.  out (B, T, P): the mixture.
 *   prepared      ms_decoder_chain_prepared_bytes bytes filled by ms_decoder_chain_prepare (once per optimizer update).
 *   sync          zero-initialised int32 words owned by one (device, stream) AND one (B, M): ms_decoder_chain_sync_words of them
 *                 from sync_first_word on.  The meeting counters are monotonic (every launch adds the same amount to each word
 *                 it uses, nothing is reset, a replayed graph needs nothing from the host), which holds only among launches of
 *                 one shape.  Word 0 of the buffer is raised when a workgroup gave up waiting (outputs NaN; zero the buffer
 *                 before the next launch). */
#define MS_CHAIN_MAX_BLOCKS 4
typedef struct ms_chain_desc {
  int32_t B, M, T, cin0, C, P, n_blocks;
  int32_t mode;             /* MS_BN_TRAIN | MS_BN_EVAL */
  int32_t dtype;            /* ms_dtype */
  int32_t sync_first_word;  /* first word of `sync` this launch may use (>= 32: word 0 is the error flag) */
  int32_t keep_all_raw;     /* BN_TRAIN: y_raw for every channel / channel block (0: only where ms_conv_block_bwd, handed y, will read it --
                             * blocks whose BatchNorm + LeakyReLU map does not invert safely from y, see ms_set_bn_sync_buffer) */
  float slope, eps, momentum;
} ms_chain_desc;
typedef struct ms_chain_tensors {
  const void* x;
  const float* score;
  const float* w[MS_CHAIN_MAX_BLOCKS];
  const float* bias[MS_CHAIN_MAX_BLOCKS];
  const float* gamma[MS_CHAIN_MAX_BLOCKS];
  const float* beta[MS_CHAIN_MAX_BLOCKS];
  float* running_mean[MS_CHAIN_MAX_BLOCKS];
  float* running_var[MS_CHAIN_MAX_BLOCKS];
  void* y_raw[MS_CHAIN_MAX_BLOCKS];
  void* y[MS_CHAIN_MAX_BLOCKS];
  float* save[MS_CHAIN_MAX_BLOCKS];
  const float* w_logits;
  const float* bias_logits;
  float* z;
  float* soft;
  float* out;
  const void* prepared;
  int32_t* sync;
  int32_t sync_words;
} ms_chain_tensors;
int ms_decoder_chain_supported(const ms_chain_desc* d);
size_t ms_decoder_chain_prepared_bytes(const ms_chain_desc* d);
size_t ms_decoder_chain_workspace(const ms_chain_desc* d);
int ms_decoder_chain_sync_words(const ms_chain_desc* d);
int ms_decoder_chain_prepare(const ms_chain_desc* d, const float* const* w, const float* w_logits, void* prepared, void* stream);
int ms_decoder_chain_fwd(const ms_chain_desc* d, const ms_chain_tensors* t, void* workspace, size_t workspace_bytes, void* stream);

/* Pre-step in front of the path ("next" row N1; src/data/transform.py, src/model/trainer.py:1290-1308), on device:
 * ms_kmeans_labels: KMeans.predict (transform.py:352-410) on RemoveJoints(pose): the feature blocks of KMeans.get_feats
 *   selected by `feats` (bit 0 pose (PK columns), bit 1 velocity (PK), bit 2 speed = |(vx, vy)| per kept joint (PK/2)), in
 *   that order -- src/jobs/mix-stage.py trains with all three, argsUtils' default is pose|velocity -- fp64 squared distance
 *   to `centers` (M, D), first-minimum argmin -> labels (B,T) int64.  keep[PK] = surviving columns (x block, then y block).
 * ms_znorm_select: ZNorm.znorm (transform.py:221-226) then RemoveJoints (transform.py:481-507):
 *   y[r][d] = (x[r][keep[d]] - mean[keep[d]]) * inv_std[keep[d]] in fp64 -> fp32; keep == NULL: all P columns. */
int ms_kmeans_labels(const float* pose, const int32_t* keep, const double* centers, int64_t* labels, int B, int T, int P,
                     int PK, int M, int feats, void* stream);
int ms_znorm_select(const float* x, const int32_t* keep, const double* mean, const double* inv_std, float* y, size_t rows,
                    int P, int PK, void* stream);

/* Step metrics ("next" row N3; src/evaluation/metrics.py:94-131,247-303 as called from trainer.py:865-915), on device:
 * per clip b, out[b][0] = sum |y - gt| over the kept columns (L1 numerator), out[b][1] = the same on first differences in
 * time (VelL1 numerator), out[b][2 + a*J + j] = number of time steps where joint j is within alphas[a] * max(h,w) of the
 * ground truth (PCK hits; poses de-normalised with std/mean, root joint at (0,0), removed joints take the gt values).
 *   ycap (B,T,PK) prediction on the kept columns; gt (B,T,P) normalised full pose; keep[PK]; slot_of[P] (column -> index
 *   in the kept layout or -1); mean,stdv[P] fp64; out (B, 2 + n_alpha*P/2) fp64. */
int ms_step_metrics(const float* ycap, const float* gt, const int32_t* keep, const int32_t* slot_of, const double* mean,
                    const double* stdv, const float* alphas, int n_alpha, double* out, int B, int T, int P, int PK,
                    void* stream);
/* N3, evaluation accumulators: FID sufficient statistics and the W1 histograms of one batch ADDED to running device buffers
 * (reference: evaluation/metrics.py:374-394 FID.__call__, :476-520 W1.__call__, called after every step on CPU copies,
 * model/trainer.py:887,896).  ycap (B,T,PK) normalised prediction in the kept (xy, joint) order, gt (B,T,P) normalised ground
 * truth, keep[PK] columns of gt, mean / stdv [P] (W1 works on de-normalised poses).
 *   fid_sums [2][PK], fid_gram [2][PK][PK] fp64: [0] prediction, [1] ground truth; rows = (b, t), summed in row order
 *   w1_hist [2][2][nbins] u64: [prediction | ground truth][speed | acceleration], edges k * bin_width, k = 0..nbins
 * Any of the two groups may be NULL. */
int ms_eval_accumulate(const float* ycap, const float* gt, const int32_t* keep, const double* mean, const double* stdv,
                       double* fid_sums, double* fid_gram, unsigned long long* w1_hist, int B, int T, int P, int PK,
                       double bin_width, int nbins, void* stream);

/* content || style-embedding concat in channel-major layout (replaces EmbLin 'emb' lookup + torch.cat + transposes,
 * JL:175-180, layers.py:659-663): out (B, C+D, T) = [x (B,C,T) ; E[ids[b,t]] (D)].  ids is addressed as
 * ids[b*ids_stride_b + t*ids_stride_t] (stride_t = 0: one id per clip).  bwd: dx (B,C,T) and/or dE (S,D), NULL = skip. */
int ms_concat_style_fwd(const float* x, const float* emb, const int64_t* ids, int ids_stride_b, int ids_stride_t, float* out,
                        int B, int C, int D, int T, void* stream);
int ms_concat_style_bwd(const float* dout, const int64_t* ids, int ids_stride_b, int ids_stride_t, float* dx, float* demb,
                        int B, int C, int D, int T, int S, void* stream);

/* Cross entropy with mean reduction (JL:159,184,203): score addressed as
 * score[n_outer*stride_outer + c*stride_c + n_inner*stride_inner], rows = n_outer*n_inner.
 * loss[0] = mean_rows( logsumexp - score[target] ).  dscore (same addressing) = gscale[0] *
 * (softmax - onehot)/rows, ADDED to dscore when accumulate != 0. */
int ms_cross_entropy_fwd(const float* score, const int64_t* target, float* loss, float* row_scratch,
                         int n_outer, int n_inner, int C, int stride_outer, int stride_c,
                         int stride_inner, void* stream);
int ms_cross_entropy_bwd(const float* score, const int64_t* target, const float* gscale, float* dscore,
                         int n_outer, int n_inner, int C, int stride_outer, int stride_c,
                         int stride_inner, int accumulate, void* stream);

/* Loss weights folded into the loss kernels (`lambda * criterion(...)`, gan.py:64-75 with the lambda_D / lambda_gan of gan.py:103;
 * `lambda_id * cross_entropy(...)`, joint_late_cluster_soft_style.py:159,184,203): loss_out = loss * scale [* scale_dev[0]] and the
 * gradient uses gscale[0] * scale [* scale_dev[0]] -- the same fp32 products torch forms for `w * loss`, so results are bit-identical
 * to scaling outside.  scale_dev (optional) is ONE float on the device, read when the kernel runs: a weight schedule that moves between
 * replays of a captured step.  ls == NULL: weight 1. */
typedef struct ms_loss_scale {
  float scale;
  const float* scale_dev;
} ms_loss_scale;
int ms_cross_entropy_fwd_ex(const float* score, const int64_t* target, float* loss, int n_outer, int n_inner, int C,
                            int stride_outer, int stride_c, int stride_inner, void* stream, const ms_loss_scale* ls);
int ms_cross_entropy_bwd_ex(const float* score, const int64_t* target, const float* gscale, float* dscore, int n_outer,
                            int n_inner, int C, int stride_outer, int stride_c, int stride_inner, int accumulate, void* stream,
                            const ms_loss_scale* ls);

/* GAN.get_velocity (gan.py:47-52) fused with the (B,T,P)->(B,P,T) transpose D.forward does (S2G:67):
 * v[b,p,0] = 0, v[b,p,t] = x[b,t,p] - x[b,t-1,p]. */
int ms_velocity_fwd(const float* x, float* v, int B, int T, int P, void* stream);
int ms_velocity_bwd(const float* dv, float* dx, int B, int T, int P, void* stream);

/* (B,T,C) <-> (B,C,T) transposes (layers.py:229,280, JL:151,180). */
int ms_transpose_btc(const float* x, float* y, int B, int T, int C, void* stream); /* (B,T,C)->(B,C,T) */
int ms_transpose_bct(const float* x, float* y, int B, int C, int T, void* stream); /* (B,C,T)->(B,T,C) */

/* mean |a-b| (or target constant) -> loss[0]  (gan.py:64-75 with criterion L1Loss), and its backward
 * da = gscale[0] * sign(a-b)/n.  b may be NULL with `target` used instead. */
int ms_l1_mean_fwd(const float* a, const float* b, float target, float* loss, float* partials, size_t n,
                   void* stream);
int ms_l1_mean_bwd(const float* a, const float* b, float target, const float* gscale, float* da, size_t n,
                   void* stream);
/* The same for criterion MSELoss, the GAN constructor's default (gan.py:21,40): mean (a-b)^2, da = gscale[0] * 2(a-b)/n. */
int ms_l2_mean_fwd(const float* a, const float* b, float target, float* loss, float* partials, size_t n, void* stream);
int ms_l2_mean_bwd(const float* a, const float* b, float target, const float* gscale, float* da, size_t n, void* stream);

/* squared = 0: L1Loss, 1: MSELoss; with the loss weight folded in (ms_loss_scale above) */
int ms_lp_mean_fwd_ex(int squared, const float* a, const float* b, float target, float* loss, float* partials, size_t n,
                      void* stream, const ms_loss_scale* ls);
int ms_lp_mean_bwd_ex(int squared, const float* a, const float* b, float target, const float* gscale, float* da, size_t n,
                      void* stream, const ms_loss_scale* ls);

/* Two criterion terms of ONE tensor in one launch each way: halves [0, n) and [n, 2n) of `a` against the constants targets[0] /
 * targets[1] with the weights ls[0] / ls[1] (NULL: 1) -> loss[0], loss[1] (gan.py:121,127: the D-step's fake and real terms on the
 * paired discriminator pass, MS_DT_STAT_PAIR).  1 <= n <= 2048 per half; the same bits as two ms_lp_mean_*_ex calls.  Backward:
 * da (2n) from the two incoming gradients (a NULL one leaves zeros in its half). */
int ms_lp_mean_pair_fwd(int squared, const float* a, const float* targets, float* loss, size_t n, void* stream, const ms_loss_scale* ls);
int ms_lp_mean_pair_bwd(int squared, const float* a, const float* targets, const float* gscale0, const float* gscale1, float* da, size_t n,
                        void* stream, const ms_loss_scale* ls);

/* n device-to-device copies (any sizes, any alignment) in one launch: the batch tensors of a step into the buffers a captured
 * step reads (the reference hands `batch` to the model directly, trainer.py:1077-1110; a HIP graph needs fixed addresses). */
int ms_copy_multi(int n, const void* const* src, void* const* dst, const size_t* bytes, void* stream);

/* n <= 8 host floats -> device, by value through the kernel arguments (no asynchronous read of host memory): the GAN loss
 * weights of a step (gan.py:103, lambda_scheduler.step()) into the 2-float tensor the captured loss kernels read. */
int ms_write_floats(float* dst, const float* host_values, int n, void* stream);

/* Trainer step tail (trainer.py:1138-1146): global L2 norm of a flat gradient buffer, then
 * clip_grad_norm_(., max_norm) folded into a fused Adam step (torch.optim.Adam defaults).
 *   norm_out[0] = ||g||_2 ; coef = min(1, max_norm/(norm+1e-6)) applied to g inside ms_adam_step. */
int ms_sqnorm(const float* g, size_t n, float* norm_out, float* partials, void* stream);
/* step_state: 4 int32 words on the device, word 0 = step count (starts at 0), words 1..3 scratch.
 * A NON-FINITE gradient norm (NaN / Inf in any gradient: e.g. behind a launch whose in-launch meeting timed out) skips the
 * update: p, m, v keep their values, the step count still advances.  ms_adam_step_segmented reports it: word 2 = 1 while the
 * LAST step was skipped, word 3 = number of skipped steps so far. */
int ms_adam_step(float* p, const float* g, float* m, float* v, size_t n, const float* norm, float max_norm,
                 float lr, float beta1, float beta2, float eps, int32_t* step_state, void* stream);
/* Same with torch.optim.Adam's PER-PARAMETER step counts: the flat buffer is a sequence of segments (one per parameter,
 * 64-element aligned); seg_of_chunk[i/64] = segment of element i; seg_first_step[s] = the global step (1-based) at
 * which segment s first received a gradient, -1 = never (such parameters are skipped, as torch does for grad None);
 * seg_scratch = 2*n_seg floats of device scratch. */
int ms_adam_step_segmented(float* p, const float* g, float* m, float* v, size_t n, const float* norm, float max_norm,
                           float lr, float beta1, float beta2, float eps, int32_t* step_state,
                           const int32_t* seg_of_chunk, const int32_t* seg_first_step, float* seg_scratch, int n_seg,
                           void* stream);
size_t ms_reduce_partials_count(size_t n); /* floats needed in `partials` of ms_sqnorm / ms_l1_mean_fwd */

#ifdef __cplusplus
}
#endif
#endif /* MIXSTAGE_H_ */
