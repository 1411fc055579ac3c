/* mixstage_aux.h -- measurement, test and tuning aids of libmixstage_hip.so.
 *
 * NOT part of the drop-in boundary (include/mixstage.h): nothing a binding of the reference's modules needs.  These entry
 * points exist for bench.py (per-launch HIP-event timing), the test-suite (dispatch knobs that force a kernel at small
 * sizes, the MFMA fragment self-test) and for two scheduling experiments that were measured slower and are off
 * (in-launch split reductions, weight gradients on a side stream).  The debug setters change
 * process-global dispatch state: do not call them from product code.
 */
#ifndef MIXSTAGE_AUX_H_
#define MIXSTAGE_AUX_H_

#include "mixstage.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Optional: a persistent, ZERO-INITIALISED int32 buffer (n >= 4096 words recommended: 65536) for in-launch split-K
 * reductions: with it, the workgroup that arrives last on a tile sums the K slices inside the conv launch (fixed order,
 * bitwise reproducible) instead of a separate reduce kernel.  The buffer belongs to the caller, must stay zero between
 * launches (the kernels restore it) and serves one device.  NULL disables the in-launch form. */
int ms_set_counter_buffer(int32_t* zeroed_counters, int n);

/* Same, with the weight gradient (dw) enqueued on `side_stream` (forked from `stream` once dy_raw exists; NOT joined
 * here: the caller makes `stream` wait for `side_stream` before it reads dw, e.g. before the optimizer step).  The side
 * stream gets its own scratch of ms_conv_block_bwd_workspace(d) bytes.  side_stream == NULL: identical to
 * ms_conv_block_bwd.  x, x2, dy and dyr must stay valid until the streams are joined. */
int ms_conv_block_bwd_overlap(const ms_conv_desc* d, const float* x, const float* x2, const float* w,
                              const float* gamma, const float* running_mean, const float* running_var,
                              const float* y_raw, const float* y, const float* save, const float* dy, float* dyr,
                              float* dx, float* dx2, float* dw, float* dbias, float* dgamma, float* dbeta,
                              void* workspace, size_t workspace_bytes, void* stream, void* side_stream,
                              void* side_workspace, size_t side_workspace_bytes);

/* Measurement aid (bench.py): when enabled, every conv / BN launch is bracketed by HIP events on its stream.
 * ms_timing_report writes "label\tcount\ttotal_ms\tflops_per_launch\tbytes_per_launch\n" lines and returns the
 * bytes needed.  Not for use during graph capture. */
int ms_timing_enable(int on);
size_t ms_timing_report(char* buf, size_t cap);

/* Test aid: the patch-staged conv kernel is used when a launch has at least this many workgroups (default 32);
 * tests set 0 to exercise it at small sizes.  Returns the previous value. */
int ms_debug_set_patch_min_workgroups(int n);
/* Tuning aid for the patch-staged conv kernel: intra_split < 0 switches the intra-workgroup K split of small 1-D k3
 * layers off (0 = planner's choice), and a forced split-K factor over workgroups (0 = planner's choice). */
int ms_debug_set_patch_tuning(int intra_split, int force_splitk);

/* Tuning aid for the 16-bit conv kernel: force the workgroup tile to 64*wm output channels x 64*wn pixels (0, 0: planner). */
int ms_debug_set_conv16_tile(int wm, int wn);
/* ... and its LDS-DMA ring depth (2..4, 0 = planner) / the 8-wave form of the 128 x 128 tile (measured: no gain). */
int ms_debug_set_conv16_ring(int nstg, int wide8);
/* ... and the number of workgroups a 16-bit weight-gradient launch aims for when it splits the pixel reduction (default 128);
 * returns the previous value.  Scratch and slab sizes follow: set it before any step is captured. */
int ms_debug_set_wgrad16_target(int workgroups);
int ms_debug_set_wgrad16_ring(int buffers);           /* LDS-DMA ring depth of the 16-bit weight gradient (2..8, default 2) */
int ms_debug_set_wgrad_target(int workgroups);      /* the same for the fp32 patch-staged weight gradient (default 768) */
/* Test / ablation aid: 0 = the fp32 weight gradient never takes the wave-pipelined kernel (wgrad_wave_multi_kernel: no workgroup
 * barrier in the reduction loop), 1 = it does where the layer qualifies (default).  Returns the previous value. */
int ms_debug_set_wgrad_wave(int on);
/* ... and 0 = the fp32 2-D convs never take the lean 128 x 128-tile kernel (conv_tile_kernel).  Returns the previous value. */
int ms_debug_set_conv_tile(int on);
/* Test / ablation aid: 0 = 16-bit BN_TRAIN blocks never take the in-launch BatchNorm form (ms_set_bn_sync_buffer), 1 = they do
 * when eligible (default).  Returns the previous value. */
int ms_debug_set_bn_fused(int on);
/* ... and the smallest conv grid (workgroups) that takes it; returns the previous value. */
int ms_debug_set_bn_fused_min_workgroups(int n);
/* Timing ablations only: launches whose timing label contains one of the ';'-separated substrings are dropped (results are
 * then meaningless); NULL or "" restores normal operation.  Returns the number of patterns. */
int ms_debug_set_skip(const char* patterns);

/* Ablations: 0 = the clip-resident 1-D conv kernels (clip32.hip: conv + BatchNorm + LeakyReLU of a UNet / classifier / style-encoder /
 * discriminator block in one launch) are not used; returns the previous value. */
int ms_debug_set_clip32(int on);

/* Self-test kernel: C(32x32) = A(32xK) * B(Kx32) through the fp32 MFMA path (checks fragment maps). */
int ms_selftest_mfma(const float* A, const float* B, float* C, int K, void* stream);
/* On-box peaks for bench.py's roofline: kind 0 / 1 = a bare fp32 / bf16 MFMA loop (operands in registers, `iters` rounds of 16 MFMAs per
 * wave, one wave per SIMD on every CU; a = 4 bytes of device scratch), kind 2 = a 16-byte-per-lane copy of `iters` bytes from a to b.
 * *work = FLOP (kinds 0, 1) or bytes read + written (kind 2) of the launch; time it with events on `stream`. */
int ms_probe_peak(int kind, long iters, void* a, void* b, double* work, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MIXSTAGE_AUX_H_ */
