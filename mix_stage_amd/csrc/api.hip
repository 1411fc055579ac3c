// Block-level C-ABI entry points: one conv block (conv [+BN] [+LeakyReLU]) forward / backward.
// Orchestrates the kernels of conv_igemm.hip and elementwise.hip on the caller's stream.
#include <algorithm>

#include "conv16.h"

namespace ms {
extern int g_clip32;

static int validate(const ms_conv_desc* d, const char* who) {
  if (!d) return set_error("%s: null descriptor", who);
  if (d->B < 1 || d->Cin < 1 || d->Cout < 1 || d->groups < 1 || d->H < 1 || d->W < 1 || d->KH < 1 || d->KW < 1 ||
      d->SH < 1 || d->SW < 1 || d->PH < 0 || d->PW < 0)
    return set_error("%s: bad geometry", who);
  const int oh = (d->H + 2 * d->PH - d->KH) / d->SH + 1, ow = (d->W + 2 * d->PW - d->KW) / d->SW + 1;
  if (d->H + 2 * d->PH < d->KH || d->W + 2 * d->PW < d->KW || oh != d->OH || ow != d->OW)
    return set_error("%s: output size (%d,%d) does not match geometry (expected %d,%d)", who, d->OH, d->OW, oh, ow);
  if (d->mode < MS_BARE || d->mode > MS_BN_EVAL) return set_error("%s: bad mode %d", who, d->mode);
  if (d->in_mode < MS_IN_PLAIN || d->in_mode > MS_IN_UP2ADD) return set_error("%s: bad in_mode %d", who, d->in_mode);
  if (d->in_mode == MS_IN_UP2ADD && (d->H != 1 || d->KH != 1 || (d->W & 1)))
    return set_error("%s: UP2ADD needs a 1-D block with even W", who);
  if (dt_of(d) > DT_F16 || (d->dtype & ~(0xff | MS_DT_OUT_F32 | MS_DT_BN_FOLDED | MS_DT_STAT_PAIR))) return set_error("%s: bad dtype 0x%x", who, d->dtype);
  if (dt_of(d) == DT_F32 && (d->dtype & ~(0xff | MS_DT_STAT_PAIR))) return set_error("%s: MS_DT_OUT_F32 is a flag of the 16-bit modes", who);
  if ((d->dtype & MS_DT_STAT_PAIR) && (d->B & 1)) return set_error("%s: MS_DT_STAT_PAIR needs an even batch", who);
  const double out_elems = (double)d->B * d->groups * d->Cout * d->OH * d->OW;
  const double in_elems = (double)d->B * d->groups * d->Cin * d->H * d->W;
  // the staging loads address a tensor with 32-bit BYTE offsets against one buffer descriptor: 2 GiB per tensor
  const double lim = dt_of(d) == DT_F32 ? 536870912.0 : 1073741824.0;      // 2^29 fp32 / 2^30 16-bit elements
  if (out_elems >= lim || in_elems >= lim) return set_error("%s: tensor of 2 GiB or more (%.0f elements)", who, std::max(out_elems, in_elems));
  return 0;
}

static inline bool wgrad_c1_of(const ms_conv_desc* d) {
  return dt_of(d) == DT_F32 && g_precision == 0 &&
         wgrad_c1_ok(d->groups, d->Cin, d->Cout, d->KH, d->KW, d->SH, d->SW, d->PH, d->PW, d->H, d->W, d->in_mode == MS_IN_PLAIN);
}

static int wgrad_total_splits(const ms_conv_desc* d) {
  if (wgrad_c1_of(d)) return wgrad_c1_splits(d->B, d->H);
  const int npix = d->B * d->OH * d->OW;
  const bool one_d = d->H == 1 && d->KH == 1;
  const WgradPatchPlan wp = plan_wgrad_patch(one_d ? 1 : 2, d->Cout, d->Cin * d->KH * d->KW, d->groups, d->KH, d->KW, d->SH,
                                             d->SW, d->B, d->OH, d->OW, d->W, d->in_mode == MS_IN_UP2ADD);
  return std::max(wp.ok ? wp.splits : 1, wgrad_splits(d->Cout, d->Cin * d->KH * d->KW, d->groups, npix));
}

static GatherPlan dgrad_plan(const ms_conv_desc* d) {
  const bool bcast = d->in_mode == MS_IN_BCAST;
  const int tg = bcast ? 1 : d->groups, tcog = bcast ? d->groups * d->Cout : d->Cout;
  const int ncls = d->SH * d->SW;
  const int npix_cls = d->B * cdiv(d->H, d->SH) * cdiv(d->W, d->SW);
  return plan_gather(d->Cin, npix_cls, tg * ncls, tcog * cdiv(d->KH, d->SH) * cdiv(d->KW, d->SW));
}

static PatchPlan fwd_patch_plan(const ms_conv_desc* d) {
  const int nd = (d->H == 1 && d->KH == 1) ? 1 : 2;
  return plan_patch(nd, d->Cout, d->groups, d->Cin, d->KH, d->KW, d->SH, d->SW, d->B, d->OH, d->OW, 1, d->W);
}

static inline int ctot_of(const ms_conv_desc* d) { return d->groups * d->Cout; }

// MS_DT_STAT_PAIR, fp32 forward: the block ends in the register-resident split-K epilogue (one workgroup per channel walks the two
// statistics groups) when the clip-resident launch does not take it -- the other BatchNorm finishes have no grouped form
static bool stat_pair_fwd_epilogue_ok(const ms_conv_desc* d) {
  if (d->mode != MS_BN_TRAIN) return true;
  const long n_grp = (long)d->B / 2 * d->OH * d->OW;
  if (n_grp > 4096) return false;
  if (conv_c1_ok(d->groups, d->Cin, d->Cout, d->KH, d->KW, d->SH, d->SW, d->PH, d->PW, d->H, d->in_mode == MS_IN_PLAIN)) return false;
  const PatchPlan pp = fwd_patch_plan(d);
  if (pp.ok) return pp.splitk > 1 || pp.ksi > 1;
  return plan_gather(d->Cout, d->B * d->OH * d->OW, d->groups, d->Cin * d->KH * d->KW).splitk > 1;
}
static inline size_t wsize_of(const ms_conv_desc* d) { return (size_t)d->groups * d->Cout * d->Cin * d->KH * d->KW; }

}  // namespace ms

using namespace ms;

extern "C" {

size_t ms_conv_block_fwd_workspace(const ms_conv_desc* d) {
  if (!d) return 256;
  if (dt_of(d) != DT_F32) return block_fwd16_workspace(d);
  const int npix = d->B * d->OH * d->OW;
  const GatherPlan pl = plan_gather(d->Cout, npix, d->groups, d->Cin * d->KH * d->KW);
  const PatchPlan pp = fwd_patch_plan(d);
  size_t bytes = 256;
  if (d->mode == MS_BN_TRAIN) bytes = std::max(bytes, (size_t)pl.n_tiles * ctot_of(d) * 2 * sizeof(float));
  if (pp.ok && d->mode == MS_BN_TRAIN)
    bytes = std::max(bytes, align_up((size_t)pp.n_tiles * ctot_of(d) * 2 * sizeof(float), 256) + (size_t)pp.n_tiles * sizeof(float));
  if (pp.ok && (pp.splitk > 1 || pp.ksi > 1)) bytes = std::max(bytes, (size_t)pp.splitk * npix * ctot_of(d) * sizeof(float));
  if (pl.splitk > 1) bytes = std::max(bytes, (size_t)pl.splitk * npix * ctot_of(d) * sizeof(float));
  if (conv_c1_ok(d->groups, d->Cin, d->Cout, d->KH, d->KW, d->SH, d->SW, d->PH, d->PW, d->H, d->in_mode == MS_IN_PLAIN)) {
    const int nt = conv_c1_tiles(d->B, d->H, d->W);
    bytes = std::max(bytes, align_up((size_t)nt * ctot_of(d) * 2 * sizeof(float), 256) + (size_t)nt * sizeof(float));
  }
  bytes = align_up(bytes, 256) + 256;
  if (pp.ok && pp.p6) bytes += align_up((size_t)3 * ctot_of(d) * patch6_row_elems(d->Cin, d->KH, d->KW) * 2, 256);   // split weights
  if (g_precision == 0 && clip32_fwd_ok(d))      // clip-resident kernel: statistics partials | weight stream (when not prepared)
    bytes = std::max(bytes, clip32_part_bytes(d->Cout, d->B * d->OW / 32) + align_up(clip32_fwd_weight_bytes(d), 256) + 256);
  return bytes;
}

size_t ms_conv_block_bwd_workspace(const ms_conv_desc* d) {
  if (!d) return 256;
  if (dt_of(d) != DT_F32) return block_bwd16_workspace(d);
  int bpc;
  const int nchunk = bwd_chunks(d->B, ctot_of(d), &bpc);
  const int npix = d->B * d->OH * d->OW;
  const int splits = wgrad_total_splits(d);
  size_t bytes = 0;
  bytes += clip32_dgrad_bn_part_bytes(d);                                  // partials of a fused producer-BatchNorm backward (ms_bwd_options.prev_*)
  bytes += align_up((size_t)ctot_of(d) * nchunk * 2 * sizeof(float), 256);  // bn partials
  bytes += align_up((size_t)ctot_of(d) * nchunk * sizeof(float), 256);      // colsum partials
  bytes += align_up(std::max(dgrad_weight_elems(d->groups, d->Cout, d->Cin, d->KH, d->KW, d->SH, d->SW), std::max(clip32_dgrad_weight_floats(d), gdgrad32_weight_floats(d))) * sizeof(float), 256);
  bytes += align_up(wsize_of(d) * sizeof(float) * (splits > 1 ? splits : 0), 256);
  {
    const bool bc = d->in_mode == MS_IN_BCAST;
    const int tg2 = bc ? 1 : d->groups, tcog2 = bc ? d->groups * d->Cout : d->Cout;
    const bool one_d2 = d->H == 1 && d->KH == 1;
    const PatchPlan pq = plan_patch(one_d2 ? 1 : 2, d->Cin, tg2, tcog2, cdiv(d->KH, d->SH), cdiv(d->KW, d->SW), 1, 1, d->B,
                                    cdiv(d->H, d->SH), cdiv(d->W, d->SW), d->SH * d->SW, d->OW);
    if (pq.ok && pq.splitk > 1) bytes += align_up((size_t)pq.splitk * d->B * tg2 * d->Cin * d->H * d->W * sizeof(float), 256);
    if (pq.ok && pq.p6)   // bf16x6 data gradient: split planes of the transposed weights
      bytes += align_up((size_t)3 * d->SH * d->SW * tg2 * d->Cin * patch6_row_elems(tcog2, cdiv(d->KH, d->SH), cdiv(d->KW, d->SW)) * 2, 256);
  }
  const GatherPlan pl = dgrad_plan(d);
  if (pl.splitk > 1)
    bytes += align_up((size_t)pl.splitk * d->B * (d->in_mode == MS_IN_BCAST ? 1 : d->groups) * d->Cin * d->H * d->W * sizeof(float), 256);
  return bytes + 256;
}

int ms_conv_block_fwd(const ms_conv_desc* d, const float* x, const float* x2, const float* w, const float* bias,
                      const float* gamma, const float* beta, float* running_mean, float* running_var, float* y_raw, float* y,
                      float* save, void* workspace, size_t workspace_bytes, void* stream) {
  return ms_conv_block_fwd_ex(d, x, x2, w, bias, gamma, beta, running_mean, running_var, y_raw, y, save, workspace,
                              workspace_bytes, stream, nullptr);
}

int ms_conv_block_fwd_ex(const ms_conv_desc* d, const float* x, const float* x2, const float* w, const float* bias,
                         const float* gamma, const float* beta, float* running_mean, float* running_var, float* y_raw,
                         float* y, float* save, void* workspace, size_t workspace_bytes, void* stream,
                         const ms_fwd_options* opt) {
  const unsigned short* w_planes = opt ? (const unsigned short*)opt->w_planes : nullptr;
  int rc = validate(d, "ms_conv_block_fwd");
  if (rc) return rc;
  if (!x || !w || !y) return set_error("ms_conv_block_fwd: null tensor");
  const bool bn = d->mode == MS_BN_TRAIN || d->mode == MS_BN_EVAL;
  if (bn && (!gamma || !beta || !running_mean || !running_var)) return set_error("ms_conv_block_fwd: BN tensors missing");
  if (d->mode == MS_BN_TRAIN && (!y_raw || !save || !workspace)) return set_error("ms_conv_block_fwd: y_raw/save/workspace missing");
  if (d->in_mode == MS_IN_UP2ADD && !x2) return set_error("ms_conv_block_fwd: UP2ADD needs x2");
  if (dt_of(d) != DT_F32)     // 16-bit modes: the tensor pointers are cb8 buffers (include/mixstage.h, ms_dtype)
    return block_fwd16(d, x, x2, w, bias, gamma, beta, running_mean, running_var, y_raw, y, save, workspace, workspace_bytes,
                       (hipStream_t)stream, w_planes, opt ? opt->bn_sync : nullptr, opt ? opt->bn_sync_words : 0);
  if (workspace_bytes < ms_conv_block_fwd_workspace(d)) return set_error("ms_conv_block_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int C = ctot_of(d), npix = d->B * d->OH * d->OW, hw = d->OH * d->OW;

  if (g_precision == 0 && clip32_fwd_ok(d)) {
    // 1-D blocks whose whole reduction fits a workgroup (clip32.hip): conv, statistics, meeting, normalisation in ONE launch
    char* wsp = (char*)workspace;
    float* part = (float*)wsp; wsp += clip32_part_bytes(d->Cout, d->B * d->OW / 32);
    const float* wp = (const float*)w_planes;
    if (!wp) {
      rc = clip32_prep_queue(w, (float*)wsp, d->Cout, d->Cin, d->KW, 0, d->Cin, s);
      if (!rc) rc = clip32_prep_flush(s);
      if (rc) return rc;
      wp = (const float*)wsp;
    }
    rc = clip32_block_fwd(d, x, x2, wp, bias, gamma, beta, running_mean, running_var, y_raw, y, save, part, opt ? opt->bn_sync : nullptr,
                          opt ? opt->bn_sync_words : 0, s);
    if (rc != -2) return rc;          // (-2: BN_TRAIN without counters, or a grid that is not resident at once: the kernels below)
    rc = 0;
  }

  GatherArgs a = {};
  a.A = w; a.src = x; a.src2 = x2;
  a.out = d->mode == MS_BN_TRAIN ? y_raw : y;
  a.bias = bias; a.bn_g = gamma; a.bn_b = beta; a.bn_m = running_mean; a.bn_v = running_var;
  a.stats = (float*)workspace;
  a.Mg = d->Cout; a.Kg = d->Cin * d->KH * d->KW; a.groups = d->groups; a.Kc = d->Cin;
  a.bcast = d->in_mode == MS_IN_BCAST;
  a.src_ctotal = a.bcast ? d->Cin : d->groups * d->Cin;
  a.SRCH = d->H; a.SRCW = d->W; a.OUTH = d->OH; a.OUTW = d->OW; a.Npix = npix;
  a.KH = d->KH; a.KW = d->KW; a.SH = d->SH; a.SW = d->SW; a.PH = d->PH; a.PW = d->PW;
  a.a_vec = (a.Kg % 4 == 0) && (((uintptr_t)w & 15) == 0);
  a.ep = d->mode == MS_BARE ? EP_BARE : d->mode == MS_LRELU ? EP_LRELU : d->mode == MS_BN_EVAL ? EP_BN_EVAL : EP_RAW_STATS;
  a.slope = d->slope; a.eps = d->eps;
  const int sg = d->mode == MS_BN_TRAIN ? sg_of(d) : 1;
  if (sg > 1 && !stat_pair_fwd_epilogue_ok(d))
    return set_error("ms_conv_block_fwd: MS_DT_STAT_PAIR is not implemented for this block's kernels (ms_stat_pair_ok)");
  if (conv_c1_ok(d->groups, d->Cin, d->Cout, d->KH, d->KW, d->SH, d->SW, d->PH, d->PW, d->H, d->in_mode == MS_IN_PLAIN)) {
    // one input channel (the AudioEncoder's first block): VALU kernel bound by the output write
    const int nt = conv_c1_tiles(d->B, d->H, d->W);
    float* stats = (float*)workspace;
    float* counts = (float*)((char*)workspace + align_up((size_t)nt * C * 2 * sizeof(float), 256));
    rc = launch_conv_c1(x, w, bias, a.out, gamma, beta, running_mean, running_var, stats, counts, d->B, d->H, d->W, a.ep, d->slope,
                        d->eps, s);
    if (rc) return rc;
    if (d->mode == MS_BN_TRAIN) {
      rc = launch_bn_finalize_apply(stats, counts, nt, 0, npix, C, gamma, beta, running_mean, running_var, save, d->eps, d->momentum,
                                    y_raw, y, d->B, hw, d->slope, s);
    }
    return rc;
  }
  const PatchPlan pp = fwd_patch_plan(d);
  if (pp.ok) {
    // rows >= 16 wide: patch-staged kernel (raw input patch in LDS, no im2col address math in the K loop)
    const bool one_d = d->H == 1 && d->KH == 1;
    PatchArgs q = {};
    q.A = w; q.src = x; q.src2 = x2; q.out = a.out;
    q.bias = bias; q.bn_g = gamma; q.bn_b = beta; q.bn_m = running_mean; q.bn_v = running_var;
    q.stats = (float*)workspace;
    q.counts = (float*)((char*)workspace + align_up((size_t)pp.n_tiles * C * 2 * sizeof(float), 256));
    q.Mg = d->Cout; q.Kg = a.Kg; q.groups = d->groups; q.Kc = d->Cin; q.bcast = a.bcast; q.a_vec = a.a_vec; q.ep = a.ep;
    const int cin_tot = a.src_ctotal;
    if (one_d) {            // the batch axis is the row axis of one image
      q.SRCH = d->B; q.SRCW = d->W; q.s_img = 0; q.s_chan = d->W; q.s_row = cin_tot * d->W;
      q.OUTH = d->B; q.OUTW = d->OW; q.o_img = 0; q.o_chan = d->OW; q.o_row = C * d->OW;
      q.PH = 0;
    } else {
      q.SRCH = d->H; q.SRCW = d->W; q.s_img = cin_tot * d->H * d->W; q.s_chan = d->H * d->W; q.s_row = d->W;
      q.OUTH = d->OH; q.OUTW = d->OW; q.o_img = C * hw; q.o_chan = hw; q.o_row = d->OW;
      q.PH = d->PH;
    }
    q.PW = d->PW; q.tiles_x = pp.tiles_x; q.tiles_y = pp.tiles_y; q.slope = d->slope; q.eps = d->eps;
    q.o_sh = 1; q.o_sw = 1; q.o_ry = 0; q.o_rx = 0;
    q.splitk = pp.splitk; q.chunks_per_split = pp.chunks_per_split;
    q.src_elems = (size_t)d->B * cin_tot * d->H * d->W; q.a_elems = (size_t)C * a.Kg;
    // BN_TRAIN with the intra-workgroup split: raw tile out, then the split-K epilogue (1 slice) does bias + batch
    // statistics + normalisation in ONE launch instead of finalize + apply
    const bool raw_out = pp.splitk > 1 || (pp.ksi > 1 && d->mode == MS_BN_TRAIN);
    if (raw_out) { q.part = (float*)workspace; q.part_stride = (size_t)npix * C; }
    const double flops = 2.0 * d->Cout * a.Kg * (double)npix * d->groups;
    const double bytes = 4.0 * ((double)C * a.Kg + (double)d->B * cin_tot * d->H * d->W + (double)npix * C);
    if (pp.p6) {
      // bf16x6: three bf16 planes of the weights -- the trainer's (ms_fwd_options.w_planes) or built here behind the other scratch
      const int re = patch6_row_elems(d->Cin, d->KH, d->KW);
      const unsigned short* planes = w_planes;
      if (!planes) {
        const size_t planes_bytes = align_up((size_t)3 * C * re * 2, 256);
        unsigned short* mine = (unsigned short*)((char*)workspace + (ms_conv_block_fwd_workspace(d) - planes_bytes));
        rc = launch_split_weights(w, mine, C, d->Cin, d->KH, d->KW, s);
        if (rc) return rc;
        planes = mine;
      }
      q.Aplanes = planes; q.plane_stride = (unsigned)((size_t)C * re); q.a_row_elems = re;
      rc = launch_patch6(q, pp, d->KH, d->KW, d->SW, d->in_mode == MS_IN_UP2ADD, flops, bytes, s);
    } else {
      rc = launch_patch(q, pp, d->KH, d->KW, d->SW, d->in_mode == MS_IN_UP2ADD, flops, bytes, s);
    }
    if (rc) return rc;
    if (raw_out)
      return launch_splitk_fwd_epilogue(q.part, pp.splitk, q.part_stride, bias, gamma, beta, running_mean, running_var, y_raw,
                                        y, save, d->B, C, hw, a.ep, d->slope, d->eps, d->momentum, s, sg);
    if (d->mode == MS_BN_TRAIN) {
      rc = launch_bn_finalize_apply(q.stats, q.counts, pp.n_tiles, 0, npix, C, gamma, beta, running_mean, running_var, save, d->eps,
                                    d->momentum, y_raw, y, d->B, hw, d->slope, s);
    }
    return rc;
  }
  const GatherPlan pl = plan_gather(d->Cout, npix, d->groups, a.Kg);
  if (pl.splitk > 1) {
    a.part = (float*)workspace;
    a.part_stride = (size_t)npix * C;
  }
  rc = launch_gather(a, false, d->in_mode == MS_IN_UP2ADD, pl, s);
  if (rc) return rc;
  if (pl.splitk > 1) {
    // few output pixels: K was sliced over workgroups; one launch sums the slices and finishes the block
    return launch_splitk_fwd_epilogue(a.part, pl.splitk, a.part_stride, bias, gamma, beta, running_mean, running_var, y_raw, y,
                                      save, d->B, C, hw, a.ep, d->slope, d->eps, d->momentum, s, sg);
  }
  if (d->mode == MS_BN_TRAIN) {
    rc = launch_bn_finalize_apply(a.stats, nullptr, pl.n_tiles, 64 * pl.tn, npix, C, gamma, beta, running_mean, running_var, save,
                                  d->eps, d->momentum, y_raw, y, d->B, hw, d->slope, s);
  }
  return rc;
}

// The data-gradient weights of block d: none (the patch kernel reads w in place), or the transposed / parity-class-split
// copy, with the taps reversed when the patch-staged kernels will consume it.  Shared by the backward and the prepare
// entry points so that both take the same decision.
struct DgradWeights { int need, flip, tg, tcog, p6; size_t elems; };
static DgradWeights dgrad_weights_of(const ms_conv_desc* d, const float* w) {
  const bool bcast = d->in_mode == MS_IN_BCAST;
  DgradWeights r;
  if (g_precision == 0 && clip32_dgrad_ok(d)) {     // clip-resident kernel: the transposed, tap-reversed weight stream
    r.need = 2; r.flip = 1; r.tg = 1; r.tcog = d->Cout; r.p6 = 0; r.elems = clip32_dgrad_weight_floats(d);
    return r;
  }
  if (g_precision == 0 && g_clip32 && gdgrad32_ok(d)) {   // grouped decoder blocks: the clip-stationary data-gradient kernel's streams
    r.need = 3; r.flip = 1; r.tg = d->groups; r.tcog = d->Cout; r.p6 = 0; r.elems = gdgrad32_weight_floats(d);
    return r;
  }
  r.tg = bcast ? 1 : d->groups;
  r.tcog = bcast ? d->groups * d->Cout : d->Cout;
  const int jh = cdiv(d->KH, d->SH), jw = cdiv(d->KW, d->SW);
  const bool one_d = d->H == 1 && d->KH == 1;
  const PatchPlan pp0 = plan_patch(one_d ? 1 : 2, d->Cin, r.tg, r.tcog, jh, jw, 1, 1, d->B, cdiv(d->H, d->SH), cdiv(d->W, d->SW),
                                   d->SH * d->SW, d->OW);
  const bool direct = pp0.ok && !pp0.p6 && patch_dgrad_direct_ok(w, d->Cin, d->KH, d->KW, d->SH, d->SW, bcast);
  r.need = direct ? 0 : 1;
  r.flip = pp0.ok ? 1 : 0;
  r.p6 = (pp0.ok && pp0.p6) ? 1 : 0;
  r.elems = dgrad_weight_elems(d->groups, d->Cout, d->Cin, d->KH, d->KW, d->SH, d->SW);
  return r;
}

// Pixel splits of the block's weight gradient as the backward will run it (1: dw is written directly).
static int wgrad_splits_used(const ms_conv_desc* d) {
  if (wgrad_c1_of(d)) return wgrad_c1_splits(d->B, d->H);
  const bool one_d = d->H == 1 && d->KH == 1;
  const WgradPatchPlan wp = plan_wgrad_patch(one_d ? 1 : 2, d->Cout, d->Cin * d->KH * d->KW, d->groups, d->KH, d->KW, d->SH,
                                             d->SW, d->B, d->OH, d->OW, d->W, d->in_mode == MS_IN_UP2ADD);
  return wp.ok ? wp.splits : wgrad_splits(d->Cout, d->Cin * d->KH * d->KW, d->groups, d->B * d->OH * d->OW);
}

size_t ms_wgrad_partials_elems(const ms_conv_desc* d, int* splits) {
  if (validate(d, "ms_wgrad_partials_elems")) return 0;
  const int sp = dt_of(d) != DT_F32 ? wgrad16_splits(d) : wgrad_splits_used(d);
  if (splits) *splits = sp;
  return sp > 1 ? (size_t)sp * wsize_of(d) : 0;
}

int ms_wgrad_reduce_multi(int n, const float* const* partials, float* const* dw, const int* elems, const int* splits,
                          void* stream) {
  if (n < 0 || (n && (!partials || !dw || !elems || !splits))) return set_error("ms_wgrad_reduce_multi: null argument");
  ReduceBatch rb;
  rb.n = 0;
  for (int i = 0; i < n; ++i) {
    if (!partials[i] || !dw[i] || elems[i] <= 0 || splits[i] < 1) return set_error("ms_wgrad_reduce_multi: bad job %d", i);
    ReduceJob jb = {partials[i], dw[i], elems[i], splits[i], 0, 0};
    rb.job[rb.n++] = jb;
    if (rb.n == REDUCE_BATCH_MAX) {
      const int rc = launch_reduce_splits_multi(rb, (hipStream_t)stream);
      if (rc) return rc;
      rb.n = 0;
    }
  }
  return rb.n ? launch_reduce_splits_multi(rb, (hipStream_t)stream) : 0;
}

// bf16x6 data gradient: bytes of the three planes of the transposed class slabs (0: the fp32 kernels run)
static size_t dgrad_planes_bytes(const ms_conv_desc* d, const DgradWeights& dw) {
  if (!dw.p6) return 0;
  const int jh = cdiv(d->KH, d->SH), jw = cdiv(d->KW, d->SW);
  return (size_t)3 * d->SH * d->SW * dw.tg * d->Cin * patch6_row_elems(dw.tcog, jh, jw) * 2;
}

size_t ms_dgrad_weights_elems(const ms_conv_desc* d, const float* w) {
  if (validate(d, "ms_dgrad_weights_elems")) return 0;
  if (dt_of(d) != DT_F32) return 0;          // 16-bit modes: ms_weights16_bytes / ms_weights16_prepare
  const DgradWeights dw = dgrad_weights_of(d, w);
  if (!dw.need) return 0;
  return align_up(dw.elems, 64) + (dgrad_planes_bytes(d, dw) + 3) / 4;      // fp32 copy | bf16 planes (bf16x6 mode)
}

size_t ms_fwd_weights_bytes(const ms_conv_desc* d) {
  if (validate(d, "ms_fwd_weights_bytes")) return 0;
  if (dt_of(d) != DT_F32) return 0;          // 16-bit modes: ms_weights16_bytes / ms_weights16_prepare
  if (g_precision == 0 && clip32_fwd_ok(d)) return clip32_fwd_weight_bytes(d);       // clip-resident kernel: its weight stream
  const PatchPlan pp = fwd_patch_plan(d);
  return (pp.ok && pp.p6) ? (size_t)3 * ctot_of(d) * patch6_row_elems(d->Cin, d->KH, d->KW) * 2 : 0;
}

int ms_fwd_weights_prepare(int n, const ms_conv_desc* descs, const float* const* w, void* const* planes, void* stream) {
  if (n < 0 || (n && (!descs || !w || !planes))) return set_error("ms_fwd_weights_prepare: null argument");
  PrepQueueGuard queue_guard;          // (after the flush below the queues are empty: the guard only matters on error paths)
  SplitBatch sb;
  sb.n = 0;
  for (int i = 0; i < n; ++i) {
    const ms_conv_desc* d = descs + i;
    int rc = validate(d, "ms_fwd_weights_prepare");
    if (rc) return rc;
    if (!ms_fwd_weights_bytes(d)) continue;
    if (!planes[i]) return set_error("ms_fwd_weights_prepare: block %d needs a buffer of ms_fwd_weights_bytes bytes", i);
    if (g_precision == 0 && clip32_fwd_ok(d)) {
      rc = clip32_prep_queue(w[i], (float*)planes[i], d->Cout, d->Cin, d->KW, 0, d->Cin, (hipStream_t)stream);
      if (rc) return rc;
      continue;
    }
    SplitJob jb = {w[i], (unsigned short*)planes[i], ctot_of(d), d->Cin, d->KH * d->KW, 0, 0, 0};
    sb.job[sb.n++] = jb;
    if (sb.n == SPLIT_BATCH_MAX) {
      rc = launch_split_weights_multi(sb, (hipStream_t)stream);
      if (rc) return rc;
      sb.n = 0;
    }
  }
  { const int rcq = clip32_prep_flush((hipStream_t)stream); if (rcq) return rcq; }
  return sb.n ? launch_split_weights_multi(sb, (hipStream_t)stream) : 0;
}

int ms_dgrad_weights_prepare(int n, const ms_conv_desc* descs, const float* const* w, float* const* wt, void* stream) {
  if (n < 0 || (n && (!descs || !w || !wt))) return set_error("ms_dgrad_weights_prepare: null argument");
  PrepQueueGuard queue_guard;
  TransposeBatch tb;
  tb.n = 0;
  for (int i = 0; i < n; ++i) {
    const ms_conv_desc* d = descs + i;
    int rc = validate(d, "ms_dgrad_weights_prepare");
    if (rc) return rc;
    const DgradWeights dw = dgrad_weights_of(d, w[i]);
    if (!dw.need) continue;
    if (!wt[i]) return set_error("ms_dgrad_weights_prepare: block %d needs a buffer of ms_dgrad_weights_elems floats", i);
    if (dw.need == 2) {
      rc = clip32_prep_queue(w[i], wt[i], d->Cin, d->Cout, d->KW, d->KW == 4 ? 2 : 1, d->Cin, (hipStream_t)stream);
      if (rc) return rc;
      continue;
    }
    if (dw.need == 3) {
      rc = gdgrad32_prepare(d, w[i], wt[i], (hipStream_t)stream);
      if (rc) return rc;
      continue;
    }
    TransposeJob jb = {w[i], wt[i], dw.tg, dw.tcog, d->Cin, d->KH, d->KW, d->SH, d->SW, d->PH, d->PW, dw.flip, 0};
    tb.job[tb.n++] = jb;
    if (tb.n == TRANSPOSE_BATCH_MAX) {
      rc = launch_transpose_weight_multi(tb, (hipStream_t)stream);
      if (rc) return rc;
      tb.n = 0;
    }
  }
  if (tb.n) {
    const int rc = launch_transpose_weight_multi(tb, (hipStream_t)stream);
    if (rc) return rc;
  }
  { int rcq = clip32_prep_flush((hipStream_t)stream); if (!rcq) rcq = gdgrad32_prep_flush((hipStream_t)stream); if (rcq) return rcq; }
  // bf16x6 mode: the planes of the fp32 copies just built, behind them
  SplitBatch sb;
  sb.n = 0;
  for (int i = 0; i < n; ++i) {
    const ms_conv_desc* d = descs + i;
    const DgradWeights dw = dgrad_weights_of(d, w[i]);
    if (!dw.need || !dw.p6) continue;
    SplitJob jb = {wt[i], (unsigned short*)(wt[i] + align_up(dw.elems, 64)), d->SH * d->SW * dw.tg * d->Cin, dw.tcog,
                   cdiv(d->KH, d->SH) * cdiv(d->KW, d->SW), 0, 0, 0};
    sb.job[sb.n++] = jb;
    if (sb.n == SPLIT_BATCH_MAX) {
      const int rc = launch_split_weights_multi(sb, (hipStream_t)stream);
      if (rc) return rc;
      sb.n = 0;
    }
  }
  return sb.n ? launch_split_weights_multi(sb, (hipStream_t)stream) : 0;
}

int ms_stat_pair_ok(const ms_conv_desc* d) {
  if (!d || (d->B & 1) || validate(d, "ms_stat_pair_ok")) return 0;
  if (d->mode != MS_BN_TRAIN) return 1;                 // no batch statistics: a batch of B
  if (dt_of(d) != DT_F32) return stat_pair16_ok(d) ? 1 : 0;
  if (g_precision != 0) return 0;
  // backward: the one-launch BatchNorm backward walks the groups
  if ((long)d->B / 2 * d->OH * d->OW > BN_BWD32_FUSED_MAX) return 0;
  if (clip32_fwd_ok(d)) {
    // the clip-resident launch: whole pixel workgroups per half, every workgroup resident at once (else the kernels below)
    const int npx = d->SW == 2 ? 32 : 64, npw = d->B * d->OW / npx, cus = current_device_cus();
    if (npw % 2 == 0 && cus > 0 && cdiv(d->Cout, 32) * npw <= cus) return 1;
  }
  return stat_pair_fwd_epilogue_ok(d) ? 1 : 0;
}

int ms_dgrad_fuses_prev_bn(const ms_conv_desc* d) {
  if (!d || validate(d, "ms_dgrad_fuses_prev_bn") || (d->dtype & MS_DT_STAT_PAIR)) return 0;
  return (dt_of(d) == DT_F32 && g_precision == 0 && clip32_dgrad_bn_ok(d)) ? 1 : 0;
}

int ms_dgrad_takes_accum(const ms_conv_desc* d) {
  if (!d || validate(d, "ms_dgrad_takes_accum")) return 0;
  return (dt_of(d) == DT_F32 && g_precision == 0 && d->in_mode == MS_IN_PLAIN && clip32_dgrad_ok(d)) ? 1 : 0;
}

int ms_conv_block_bwd(const ms_conv_desc* d, const float* x, const float* x2, const float* w, const float* gamma,
                      const float* running_mean, const float* running_var, const float* y_raw, const float* y,
                      const float* save, const float* dy, float* dyr, float* dx, float* dx2, float* dw, float* dbias,
                      float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream) {
  ms_bwd_options o = {};
  return ms_conv_block_bwd_ex(d, x, x2, w, gamma, running_mean, running_var, y_raw, y, save, dy, dyr, dx, dx2, dw, dbias,
                              dgamma, dbeta, workspace, workspace_bytes, stream, &o);
}

static hipEvent_t fork_event() {
  static thread_local hipEvent_t ev = nullptr;
  if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) ev = nullptr;
  return ev;
}

int ms_conv_block_bwd_overlap(const ms_conv_desc* d, const float* x, const float* x2, const float* w, const float* gamma,
                              const float* running_mean, const float* running_var, const float* y_raw, const float* y,
                              const float* save, const float* dy, float* dyr, float* dx, float* dx2, float* dw,
                              float* dbias, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                              void* stream, void* side_stream, void* side_workspace, size_t side_workspace_bytes) {
  ms_bwd_options o = {};
  o.side_stream = side_stream; o.side_workspace = side_workspace; o.side_workspace_bytes = side_workspace_bytes;
  return ms_conv_block_bwd_ex(d, x, x2, w, gamma, running_mean, running_var, y_raw, y, save, dy, dyr, dx, dx2, dw, dbias,
                              dgamma, dbeta, workspace, workspace_bytes, stream, &o);
}

int ms_conv_block_bwd_ex(const ms_conv_desc* d, const float* x, const float* x2, const float* w, const float* gamma,
                         const float* running_mean, const float* running_var, const float* y_raw, const float* y,
                         const float* save, const float* dy, float* dyr, float* dx, float* dx2, float* dw, float* dbias,
                         float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream,
                         const ms_bwd_options* opt) {
  (void)running_mean; (void)running_var;
  ms_bwd_options none = {};
  if (!opt) opt = &none;
  void* side_stream = opt->side_stream;
  void* side_workspace = opt->side_workspace;
  const size_t side_workspace_bytes = opt->side_workspace_bytes;
  const float* wt_prepared = opt->wt_prepared;
  const bool defer_wgrad = opt->wgrad_partials != nullptr;
  int rc = validate(d, "ms_conv_block_bwd");
  if (rc) return rc;
  if (d->mode == MS_BN_EVAL) return set_error("ms_conv_block_bwd: BN_EVAL blocks are never differentiated on the path");
  if (!dy || !w || !workspace) return set_error("ms_conv_block_bwd: null tensor");
  if (d->mode == MS_BN_TRAIN && !opt->dy_is_dyr && (!y_raw || !save || !gamma || !dyr)) return set_error("ms_conv_block_bwd: BN_TRAIN needs y_raw/save/gamma/dyr");
  if ((opt->dy_is_dyr || opt->prev_y) && (dt_of(d) != DT_F32 || side_stream)) return set_error("ms_conv_block_bwd: the fused producer-BatchNorm backward is an fp32, single-stream form");
  if (d->mode == MS_LRELU && (!y || !dyr)) return set_error("ms_conv_block_bwd: LRELU needs y/dyr");
  if (dw && !x) return set_error("ms_conv_block_bwd: dw needs x");
  if (opt->dx_accum && !(dx && ms_dgrad_takes_accum(d))) return set_error("ms_conv_block_bwd: dx_accum on a block ms_dgrad_takes_accum() declines");
  if (d->in_mode == MS_IN_UP2ADD && ((dw && !x2) || (dx && !dx2))) return set_error("ms_conv_block_bwd: UP2ADD needs x2/dx2");
  if (dt_of(d) != DT_F32) {
    if (side_stream) return set_error("ms_conv_block_bwd: no side-stream form in the 16-bit modes");
    if (d->mode == MS_BARE && out_f32_of(d) && !dyr) return set_error("ms_conv_block_bwd: fp32 dy needs the dyr scratch");
    return block_bwd16(d, x, x2, w, gamma, y_raw, y, save, dy, dyr, dx, dx2, dw, dbias, dgamma, dbeta, workspace, workspace_bytes,
                       (hipStream_t)stream, wt_prepared, opt->wgrad_partials, opt->defer_wgrad_launch);
  }
  if (workspace_bytes < ms_conv_block_bwd_workspace(d)) return set_error("ms_conv_block_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int C = ctot_of(d), npix = d->B * d->OH * d->OW, hw = d->OH * d->OW;
  const int khw = d->KH * d->KW;
  const bool up2 = d->in_mode == MS_IN_UP2ADD, bcast = d->in_mode == MS_IN_BCAST;

  int bpc;
  const int nchunk = bwd_chunks(d->B, C, &bpc);
  char* wsp = (char*)workspace;
  float* fuse_part = (float*)wsp; wsp += clip32_dgrad_bn_part_bytes(d);
  float* bn_part = (float*)wsp; wsp += align_up((size_t)C * nchunk * 2 * sizeof(float), 256);
  float* colpart = (float*)wsp; wsp += align_up((size_t)C * nchunk * sizeof(float), 256);
  float* wt = (float*)wsp; wsp += align_up(std::max(dgrad_weight_elems(d->groups, d->Cout, d->Cin, d->KH, d->KW, d->SH, d->SW), std::max(clip32_dgrad_weight_floats(d), gdgrad32_weight_floats(d))) * sizeof(float), 256);
  float* wg_part = (float*)wsp;
  {
    const int sp = wgrad_total_splits(d);
    wsp += align_up(wsize_of(d) * sizeof(float) * (sp > 1 ? sp : 0), 256);
  }
  float* dg_part = (float*)wsp;
  // weight gradient on the side stream (its own scratch): it only needs dy_raw and the saved input, so it runs
  // concurrently with this block's data gradient and the earlier blocks' backward on `stream`
  if (defer_wgrad) wg_part = opt->wgrad_partials;   // pixel-split partial slabs stay there; the caller reduces them later
  hipStream_t ws_stream = s;
  if (side_stream && side_stream != stream && dw) {
    if (defer_wgrad) return set_error("ms_conv_block_bwd_ex: wgrad_partials and side_stream are exclusive");
    if (!side_workspace || side_workspace_bytes < ms_conv_block_bwd_workspace(d))
      return set_error("ms_conv_block_bwd_overlap: side workspace too small");
    ws_stream = (hipStream_t)side_stream;
    wg_part = (float*)((char*)side_workspace + ((char*)wg_part - (char*)workspace));
  }

  // 1. gradient wrt the raw conv output (+ per-channel column sums = bias gradient)
  const float* g = dy;
  int bias_done = 0;
  const bool have_prev = opt->prev_y != nullptr;
  if (have_prev && !(dx && g_precision == 0 && clip32_dgrad_bn_ok(d)))
    return set_error("ms_conv_block_bwd: this block's data gradient cannot carry the producer's BatchNorm backward (ms_dgrad_fuses_prev_bn)");
  if (opt->dy_is_dyr) {
    // a consumer's fused data-gradient launch already applied this block's BatchNorm + activation backward (and wrote dgamma / dbeta / dbias)
    if (d->mode != MS_BN_TRAIN) return set_error("ms_conv_block_bwd: dy_is_dyr is the BN_TRAIN form");
    bias_done = 1;
  } else if (d->mode == MS_BN_TRAIN) {
    rc = launch_bn_bwd(dy, y_raw, y, save, gamma, bn_part, dyr, colpart, dbias, dgamma, dbeta, d->B, C, hw, d->slope, &bias_done, s, sg_of(d));
    g = dyr;
  } else if (d->mode == MS_LRELU) {
    rc = launch_act_bwd(dy, y, dyr, colpart, dbias, d->B, C, hw, 1, d->slope, &bias_done, s);
    g = dyr;
  } else if (dbias) {
    rc = launch_act_bwd(dy, nullptr, nullptr, colpart, dbias, d->B, C, hw, 0, 0.f, &bias_done, s);
  }
  if (rc) return rc;
  if (dbias && !bias_done && !opt->dy_is_dyr) {
    rc = launch_colsum_finalize(colpart, dbias, d->B, C, s);
    if (rc) return rc;
  }

  if (ws_stream != s) {
    hipEvent_t ev = fork_event();
    if (!ev || hipEventRecord(ev, s) != hipSuccess || hipStreamWaitEvent(ws_stream, ev, 0) != hipSuccess)
      return set_error("ms_conv_block_bwd_overlap: stream fork failed");
  }

  // 2. data gradient: transposed gather over dyr with wt[g][ci][co][khw]
  bool dx_done = false;
  if (dx && g_precision == 0 && clip32_dgrad_ok(d)) {
    // k3 s1 blocks whose reduction fits a workgroup: one launch of the clip-resident kernel (clip32.hip), no split-K slab
    const float* wp = wt_prepared;
    if (!wp) {
      rc = clip32_prep_queue(w, wt, d->Cin, d->Cout, d->KW, d->KW == 4 ? 2 : 1, d->Cin, s);
      if (!rc) rc = clip32_prep_flush(s);
      if (rc) return rc;
      wp = wt;
    }
    if (have_prev) {
      const Clip32PrevBN pv = {opt->prev_y, opt->prev_y_raw, opt->prev_save, opt->prev_gamma, opt->prev_dgamma, opt->prev_dbeta, opt->prev_dbias, opt->prev_slope};
      rc = clip32_block_dgrad(d, g, wp, dx, dx2, s, &pv, fuse_part, opt->bn_sync, opt->bn_sync_words, opt->dx_accum);
      if (rc == -2) return set_error("ms_conv_block_bwd: the fused data gradient is not resident at once on this device");
    } else {
      rc = clip32_block_dgrad(d, g, wp, dx, dx2, s, nullptr, nullptr, nullptr, 0, opt->dx_accum);
    }
    if (rc && rc != -2) return rc;
    dx_done = rc == 0;
    rc = 0;
  }
  if (dx && !dx_done && g_precision == 0 && g_clip32 && gdgrad32_ok(d)) {
    // grouped decoder blocks: one clip of one group per workgroup, all 256 rows, weights streamed into registers (chain32.hip)
    const float* wp = wt_prepared;
    if (!wp) {
      rc = gdgrad32_prepare(d, w, wt, s);
      if (!rc) rc = gdgrad32_prep_flush(s);
      if (rc) return rc;
      wp = wt;
    }
    rc = gdgrad32_launch(d, g, wp, dx, s);
    if (rc) return rc;
    dx_done = true;
  }
  if (dx && !dx_done) {
    const int tg = bcast ? 1 : d->groups;          // broadcast input: all groups sum into the same channels
    const int tcog = bcast ? C : d->Cout;
    const int jh = cdiv(d->KH, d->SH), jw = cdiv(d->KW, d->SW);
    const bool one_d = d->H == 1 && d->KH == 1;
    const int ncls = d->SH * d->SW;
    // patch-staged path: each output-parity class is a dense stride-1 forward conv of dy_raw with the class's taps
    // reversed (weights prepared by transpose_weight_kernel(flip=1)); outputs are scattered with stride (SH, SW)
    const PatchPlan pp0 = plan_patch(one_d ? 1 : 2, d->Cin, tg, tcog, jh, jw, 1, 1, d->B, cdiv(d->H, d->SH), cdiv(d->W, d->SW), ncls, d->OW);
    // stride-1 convs with whole 64-channel tiles: the patch kernel reads w in place, no transposed copy
    const bool direct = pp0.ok && !pp0.p6 && patch_dgrad_direct_ok(w, d->Cin, d->KH, d->KW, d->SH, d->SW, bcast != 0);
    if (wt_prepared) {
      wt = const_cast<float*>(wt_prepared);      // built by ms_dgrad_weights_prepare for this very descriptor
    } else if (!direct) {
      rc = launch_transpose_weight(w, wt, tg, tcog, d->Cin, d->KH, d->KW, d->SH, d->SW, d->PH, d->PW, pp0.ok ? 1 : 0, s);
      if (rc) return rc;
    }
    if (pp0.ok) {
      const int Kg2 = tcog * jh * jw;
      const int cin_tot = tg * d->Cin;
      // all output-parity classes in ONE launch (class 0 has the largest extent: its tiling serves the others)
      PatchArgs q = {};
      q.A = wt; q.src = g; q.out = dx; q.out2 = dx2;
      q.Mg = d->Cin; q.Kg = Kg2; q.groups = tg; q.Kc = tcog; q.bcast = 0; q.a_vec = (Kg2 % 4 == 0);
      if (direct) { q.A = w; q.a_vec = 2; }
      q.ep = up2 ? EP_DGRAD_UP2 : EP_BARE; q.is_dgrad = 1;
      q.ncls = ncls; q.cls_a_stride = (unsigned)((size_t)tg * d->Cin * Kg2);
      double flops = 0, bytes = 4.0 * ((double)ncls * tg * d->Cin * Kg2 + (double)d->B * C * hw);
      for (int cls = 0; cls < ncls; ++cls) {
        const int ry = cls / d->SW, rx = cls - ry * d->SW;
        const int kh0 = (ry + d->PH) % d->SH, kw0 = (rx + d->PW) % d->SW;
        const int cy = (ry + d->PH - kh0) / d->SH, cx = (rx + d->PW - kw0) / d->SW;
        const int QH = std::max(0, (d->H - ry + d->SH - 1) / d->SH), QW = std::max(0, (d->W - rx + d->SW - 1) / d->SW);
        q.cls_PH[cls] = one_d ? 0 : (jh - 1) - cy; q.cls_PW[cls] = (jw - 1) - cx;
        q.cls_OUTH[cls] = one_d ? d->B : QH; q.cls_OUTW[cls] = QW;
        q.cls_ry[cls] = one_d ? 0 : ry; q.cls_rx[cls] = rx;
        flops += 2.0 * d->Cin * Kg2 * (double)d->B * (one_d ? 1 : QH) * QW * tg;
        bytes += 4.0 * (double)d->B * cin_tot * (one_d ? 1 : QH) * QW;
      }
      if (one_d) {
        q.SRCH = d->B; q.SRCW = d->OW; q.s_img = 0; q.s_chan = d->OW; q.s_row = C * d->OW;
        q.o_img = 0; q.o_chan = d->W; q.o_row = cin_tot * d->W; q.o_sh = 1;
      } else {
        q.SRCH = d->OH; q.SRCW = d->OW; q.s_img = C * hw; q.s_chan = hw; q.s_row = d->OW;
        q.o_img = cin_tot * d->H * d->W; q.o_chan = d->H * d->W; q.o_row = d->W; q.o_sh = d->SH;
      }
      q.o_sw = d->SW;
      q.PH = q.cls_PH[0]; q.PW = q.cls_PW[0]; q.OUTH = q.cls_OUTH[0]; q.OUTW = q.cls_OUTW[0]; q.o_ry = q.cls_ry[0]; q.o_rx = q.cls_rx[0];
      q.tiles_x = pp0.tiles_x; q.tiles_y = pp0.tiles_y;
      q.splitk = pp0.splitk; q.chunks_per_split = pp0.chunks_per_split;
      q.src_elems = (size_t)d->B * C * hw; q.a_elems = (size_t)ncls * tg * d->Cin * Kg2;
      if (pp0.splitk > 1) {
        q.part = dg_part; q.part_stride = (size_t)d->B * cin_tot * d->H * d->W;
        q.ep = EP_BARE;               // partial tiles: plain full-resolution layout, the reduce kernel splits UP2
      }
      if (pp0.p6) {
        // bf16x6: planes of the transposed class slabs -- behind the prepared fp32 copy when the trainer built them
        // (ms_dgrad_weights_prepare), else split here
        const int re = patch6_row_elems(tcog, jh, jw), rows = ncls * tg * d->Cin;
        const unsigned short* planes;
        if (wt_prepared) {
          planes = (const unsigned short*)(wt_prepared + align_up(dgrad_weight_elems(d->groups, d->Cout, d->Cin, d->KH, d->KW, d->SH, d->SW), 64));
        } else {
          unsigned short* mine = (unsigned short*)((char*)dg_part + (pp0.splitk > 1 ? align_up((size_t)pp0.splitk * d->B * cin_tot *
                                                                                                  d->H * d->W * sizeof(float), 256) : 0));
          rc = launch_split_weights(wt, mine, rows, tcog, jh, jw, s);
          if (rc) return rc;
          planes = mine;
        }
        q.Aplanes = planes; q.plane_stride = (unsigned)((size_t)rows * re); q.a_row_elems = re;
        q.cls_a_stride = (unsigned)(tg * d->Cin);
        rc = launch_patch6(q, pp0, jh, jw, 1, false, flops, bytes, s);
      } else {
        rc = launch_patch(q, pp0, jh, jw, 1, false, flops, bytes, s);
      }
      if (rc) return rc;
      if (pp0.splitk > 1) {
        const size_t n = (size_t)d->B * cin_tot * d->H * d->W;
        rc = launch_splitk_dgrad_epilogue(dg_part, pp0.splitk, n, dx, dx2, n, d->W, up2 ? 1 : 0, s);
        if (rc) return rc;
      }
    } else {
    GatherArgs a = {};
    a.A = wt; a.src = g; a.out = dx; a.out2 = dx2;
    a.Mg = d->Cin; a.Kg = tcog * jh * jw; a.groups = tg; a.Kc = tcog; a.src_ctotal = C;
    a.SRCH = d->OH; a.SRCW = d->OW; a.OUTH = d->H; a.OUTW = d->W; a.Npix = d->B * d->H * d->W; a.batch = d->B;
    a.KH = jh; a.KW = jw; a.SH = d->SH; a.SW = d->SW; a.PH = d->PH; a.PW = d->PW;
    a.bcast = 0;
    a.a_vec = (a.Kg % 4 == 0);
    a.ep = up2 ? EP_DGRAD_UP2 : EP_DGRAD;
    const GatherPlan pl = dgrad_plan(d);
    if (pl.splitk > 1) {
      a.part = dg_part;
      a.part_stride = (size_t)d->B * tg * d->Cin * d->H * d->W;
      a.ep = EP_DGRAD;               // partial tiles use the plain full-resolution layout
    }
    rc = launch_gather(a, true, up2, pl, s);
    if (rc) return rc;
    if (pl.splitk > 1) {
      rc = launch_splitk_dgrad_epilogue(dg_part, pl.splitk, a.part_stride, dx, dx2, a.part_stride, d->W, up2 ? 1 : 0, s);
      if (rc) return rc;
    }
    }
  }

  // 3. weight gradient
  if (dw && wgrad_c1_of(d)) {
    // the single-input-channel 3x3 block: a stream over dy_raw on the vector unit (conv_c1.hip), slabs reduced like any split
    rc = launch_wgrad_c1(g, x, wg_part, d->B, d->H, d->W, ws_stream);
    if (!rc && !defer_wgrad) rc = launch_reduce_splits(wg_part, dw, C * d->Cin * khw, wgrad_c1_splits(d->B, d->H), ws_stream);
    return rc;
  }
  if (dw) {
    const bool one_d = d->H == 1 && d->KH == 1;
    const WgradPatchPlan wp = plan_wgrad_patch(one_d ? 1 : 2, d->Cout, d->Cin * khw, d->groups, d->KH, d->KW, d->SH, d->SW,
                                               d->B, d->OH, d->OW, d->W, up2);
    if (wp.ok) {
      const int cin_tot = bcast ? d->Cin : d->groups * d->Cin;
      WgradPatchArgs q = {};
      q.dyr = g; q.src = x; q.src2 = x2;
      q.out = wp.splits > 1 ? wg_part : dw;
      q.Cog = d->Cout; q.Cig = d->Cin; q.Kg = d->Cin * khw; q.groups = d->groups; q.bcast = bcast;
      if (one_d) {
        q.SRCH = d->B; q.SRCW = d->W; q.s_img = 0; q.s_chan = d->W; q.s_row = cin_tot * d->W; q.PH = 0;
        q.OUTH = d->B; q.OUTW = d->OW; q.o_img = 0; q.o_chan = d->OW; q.o_row = C * d->OW;
      } else {
        q.SRCH = d->H; q.SRCW = d->W; q.s_img = cin_tot * d->H * d->W; q.s_chan = d->H * d->W; q.s_row = d->W; q.PH = d->PH;
        q.OUTH = d->OH; q.OUTW = d->OW; q.o_img = C * hw; q.o_chan = hw; q.o_row = d->OW;
      }
      q.PW = d->PW; q.tiles_x = wp.tiles_x; q.tiles_y = wp.tiles_y; q.n_tiles = wp.n_tiles;
      q.tiles_per_split = wp.tiles_per_split; q.splits = wp.splits;
      const double flops = 2.0 * d->Cout * q.Kg * (double)npix * d->groups;
      const double bytes = 4.0 * ((double)npix * C + (double)d->B * cin_tot * d->H * d->W + (double)C * q.Kg);
      if (wp.splits > 1 && !defer_wgrad && !wp.p6) {
        q.counters = counter_region(CNT_WGRAD, cdiv(q.Kg, 64) * cdiv(d->Cout, 64) * d->groups);
        q.final_out = dw;
      }
      // queued form (ms_wgrad_flush): only when nothing of this call reads the result -- dw written in place, or slabs left
      // for the caller's ms_wgrad_reduce_multi
      if (opt->defer_wgrad_launch && !wp.p6 && ws_stream == s && (wp.splits == 1 || defer_wgrad))
        return queue_wgrad_patch(q, wp, d->KH, d->KW, d->SW, up2, flops, bytes);
      rc = wp.p6 ? launch_wgrad_patch6(q, wp, d->KH, d->KW, d->SW, up2, flops, bytes, ws_stream)
                 : launch_wgrad_patch(q, wp, d->KH, d->KW, d->SW, up2, flops, bytes, ws_stream);
      if (rc) return rc;
      if (wp.splits > 1 && !q.counters && !defer_wgrad) rc = launch_reduce_splits(wg_part, dw, C * q.Kg, wp.splits, ws_stream);
    } else {
      WgradArgs a = {};
      a.dyr = g; a.src = x; a.src2 = x2;
      a.Cog = d->Cout; a.Cig = d->Cin; a.Kg = d->Cin * khw; a.groups = d->groups;
      a.src_ctotal = bcast ? d->Cin : d->groups * d->Cin;
      a.H = d->H; a.W = d->W; a.OH = d->OH; a.OW = d->OW; a.Npix = npix;
      a.KH = d->KH; a.KW = d->KW; a.SH = d->SH; a.SW = d->SW; a.PH = d->PH; a.PW = d->PW;
      a.bcast = bcast;
      rc = launch_wgrad(a, up2, dw, wg_part, defer_wgrad, ws_stream, opt->defer_wgrad_launch && ws_stream == s);
    }
  }
  return rc;
}

}  // extern "C"
