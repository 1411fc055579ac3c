// 16-bit arithmetic mode (desc.dtype = MS_BF16 / MS_F16): one conv block forward / backward on cb8 tensors, reached from the
// ms_conv_block_* entry points of api.hip, plus the batched weight preparation (ms_weights16_prepare).
#include <algorithm>

#include "conv16.h"

namespace ms {

namespace {

struct Geo16 {
  int dt, nd, C, C8, cin_tot, cin8_tot, npix, hw, up2, bcast, one_d;
};
Geo16 geo_of(const ms_conv_desc* d) {
  Geo16 g;
  g.dt = dt_of(d);
  g.one_d = d->H == 1 && d->KH == 1;
  g.nd = g.one_d ? 1 : 2;
  g.C = d->groups * d->Cout; g.C8 = c8_of(g.C);
  g.bcast = d->in_mode == MS_IN_BCAST; g.up2 = d->in_mode == MS_IN_UP2ADD;
  g.cin_tot = g.bcast ? d->Cin : d->groups * d->Cin; g.cin8_tot = c8_of(g.cin_tot);
  g.npix = d->B * d->OH * d->OW; g.hw = d->OH * d->OW;
  return g;
}

Conv16Plan fwd_plan16(const ms_conv_desc* d) {
  const Geo16 g = geo_of(d);
  return plan_conv16(g.nd, d->Cout, d->groups, d->Cin, d->KH, d->KW, d->SH, d->SW, d->B, d->OH, d->OW, 1, g.up2 != 0);
}
struct Dgrad16 { int tg, tcog, jh, jw, ncls; Conv16Plan pl; };
Dgrad16 dgrad_plan16(const ms_conv_desc* d) {
  const Geo16 g = geo_of(d);
  Dgrad16 r;
  r.tg = g.bcast ? 1 : d->groups;
  r.tcog = g.bcast ? g.C : d->Cout;
  r.jh = cdiv(d->KH, d->SH); r.jw = cdiv(d->KW, d->SW);
  r.ncls = d->SH * d->SW;
  r.pl = plan_conv16(g.nd, d->Cin, r.tg, r.tcog, r.jh, r.jw, 1, 1, d->B, cdiv(d->H, d->SH), cdiv(d->W, d->SW), r.ncls, false);
  return r;
}
Wgrad16Plan wgrad_plan16(const ms_conv_desc* d) {
  const Geo16 g = geo_of(d);
  return plan_wgrad16(g.nd, d->Cout, d->Cin, d->groups, d->KH, d->KW, d->SH, d->SW, d->B, d->OH, d->OW, g.up2 != 0);
}
size_t fwd_a_bytes(const ms_conv_desc* d, const Conv16Plan& pl) {
  return conv16_weight_bytes(pl, d->Cout, d->groups, d->Cin, d->KH, d->KW, 1);
}
size_t dgrad_a_bytes(const ms_conv_desc* d, const Dgrad16& dg) {
  return conv16_weight_bytes(dg.pl, d->Cin, dg.tg, dg.tcog, dg.jh, dg.jw, dg.ncls);
}

Prep16Job fwd_job(const ms_conv_desc* d, const Conv16Plan& pl, const float* w, void* out, const float* scale) {
  Prep16Job jb = {};
  jb.w = w; jb.out = out; jb.scale = scale; jb.dgrad = 0;
  jb.groups = d->groups; jb.Cog = d->Cout; jb.Cig = d->Cin; jb.KH = d->KH; jb.KW = d->KW; jb.SH = d->SH; jb.SW = d->SW;
  jb.PH = d->PH; jb.PW = d->PW; jb.bcast = d->in_mode == MS_IN_BCAST;
  jb.BM = 64 * pl.wm; jb.CK8 = pl.ck8; jb.nchunks = pl.nchunks; jb.n_mt = cdiv(d->Cout, jb.BM); jb.dt = dt_of(d);
  return jb;
}
Prep16Job dgrad_job(const ms_conv_desc* d, const Dgrad16& dg, const float* w, void* out) {
  Prep16Job jb = fwd_job(d, dg.pl, w, out, nullptr);
  jb.dgrad = 1;
  jb.n_mt = cdiv(d->Cin, jb.BM);
  return jb;
}

// partial statistics of the in-launch BatchNorm: one (sum, M2, count, 0) per tile and (padded) channel
size_t bn_part_bytes(const ms_conv_desc* d, const Conv16Plan& pl) {
  const int bm = 64 * pl.wm;
  return align_up((size_t)pl.n_tiles * cdiv(d->Cout, bm) * d->groups * bm * 16, 256);
}

}  // namespace

size_t weights16_bytes(const ms_conv_desc* d, int which) {
  if (which == 0) {
    const Conv16Plan pl = fwd_plan16(d);
    if (!pl.ok) return 0;
    size_t b = align_up(fwd_a_bytes(d, pl), 256);
    if (bn_folded_of(d)) b += align_up((size_t)2 * d->groups * d->Cout * sizeof(float), 256);   // folded scale | bias
    return b;
  }
  const Dgrad16 dg = dgrad_plan16(d);
  return dg.pl.ok ? align_up(dgrad_a_bytes(d, dg), 256) : 0;
}

int wgrad16_splits(const ms_conv_desc* d) {
  const Wgrad16Plan wp = wgrad_plan16(d);
  return wp.tp ? wp.splits : 1;
}

size_t block_fwd16_workspace(const ms_conv_desc* d) {
  const Geo16 g = geo_of(d);
  const Conv16Plan pl = fwd_plan16(d);
  size_t bytes = 256;
  if (pl.ok) {
    bytes += align_up((size_t)pl.n_tiles * g.C * 2 * sizeof(float), 256) + align_up((size_t)pl.n_tiles * sizeof(float), 256);
    if (d->mode == MS_BN_TRAIN) bytes += bn_part_bytes(d, pl);
    bytes += weights16_bytes(d, 0) + align_up((size_t)2 * g.C * sizeof(float), 256);   // per-call weight preparation
  }
  return bytes;
}

size_t block_bwd16_workspace(const ms_conv_desc* d) {
  const Geo16 g = geo_of(d);
  int bpc;
  const int nchunk = bwd16_chunks(d->B, g.C8, g.hw, &bpc);
  size_t bytes = 256;
  bytes += align_up((size_t)g.C8 * 8 * nchunk * 2 * sizeof(float), 256);
  bytes += align_up((size_t)g.C8 * 8 * nchunk * sizeof(float), 256);
  bytes += weights16_bytes(d, 1);
  const int sp = wgrad16_splits(d);
  if (sp > 1) bytes += align_up((size_t)sp * g.C * d->Cin * d->KH * d->KW * sizeof(float), 256);
  return bytes;
}

// MS_DT_STAT_PAIR in the 16-bit modes: the conv leaves its tile statistics (EP_RAW_STATS) and the normalising launch combines them
// per half of the batch (bn_finalize_apply16_kernel) -- needs whole statistics tiles per half, at most 32 of them, a cb8 output,
// and the one-launch BatchNorm backward
bool stat_pair16_ok(const ms_conv_desc* d) {
  if (d->mode != MS_BN_TRAIN) return true;
  if ((d->B & 1) || out_f32_of(d)) return false;
  const Geo16 g = geo_of(d);
  const Conv16Plan pl = fwd_plan16(d);
  if (!pl.ok || (pl.n_tiles & 1) || pl.n_tiles > 64) return false;
  if (g.one_d && ((d->B / 2) % pl.th)) return false;          // (1-D: the batch is the row axis of one image; 2-D: tiles per image)
  return (long)d->B / 2 * g.hw <= BN_BWD16_FUSED_MAX;
}

int block_fwd16(const ms_conv_desc* d, const void* x, const void* x2, const float* w, const float* bias, const float* gamma,
                const float* beta, float* running_mean, float* running_var, void* y_raw, void* y, float* save, void* workspace,
                size_t workspace_bytes, hipStream_t s, const void* w_prepared, int32_t* bn_sync, int bn_sync_words) {
  const Geo16 g = geo_of(d);
  if (g.dt != DT_BF16 && g.dt != DT_F16) return set_error("ms_conv_block_fwd: dtype %d", d->dtype);
  if (d->mode == MS_BN_TRAIN && sg_of(d) > 1 && !stat_pair16_ok(d))
    return set_error("ms_conv_block_fwd: MS_DT_STAT_PAIR is not implemented for this 16-bit block (ms_stat_pair_ok)");
  if (!bn_sync) { bn_sync = g_bn_sync; bn_sync_words = g_bn_sync_n; }      // (the process-wide buffer of ms_set_bn_sync_buffer)
  const Conv16Plan pl = fwd_plan16(d);
  if (!pl.ok) return set_error("ms_conv_block_fwd: no 16-bit kernel for a %dx%d stride (%d,%d) block", d->KH, d->KW, d->SH, d->SW);
  if (d->groups > 1 && ((d->Cout & 7) || (!g.bcast && (d->Cin & 7))))
    return set_error("ms_conv_block_fwd: 16-bit grouped blocks need channels per group %% 8 == 0");
  if (workspace_bytes < block_fwd16_workspace(d)) return set_error("ms_conv_block_fwd: workspace too small");
  const bool outf32 = out_f32_of(d);
  char* wsp = (char*)workspace;
  float* stats = (float*)wsp; wsp += align_up((size_t)pl.n_tiles * g.C * 2 * sizeof(float), 256);
  float* counts = (float*)wsp; wsp += align_up((size_t)pl.n_tiles * sizeof(float), 256);
  float* bn_part = nullptr;
  if (d->mode == MS_BN_TRAIN) { bn_part = (float*)wsp; wsp += bn_part_bytes(d, pl); }
  // the single-input-channel 3x3 block (AudioEncoder conv.0): vector-unit kernel, no operand preparation
  // (train mode stays on the matrix-pipe kernel: with the tile statistics the vector-unit kernel is the slower one, 46 vs 35 us;
  // eval mode -- the D-step's generator pass -- 41 -> 28.5 us)
  if (conv16_c1_ok(d) && d->mode != MS_BN_TRAIN && !outf32 && conv16_c1_tiles(d) <= pl.n_tiles && w && (!bn_folded_of(d) || w_prepared)) {
    // inference with eval BatchNorm folded: the prepared buffer carries scale | bias' behind the operand stages (the same values the
    // matrix-pipe path folds into its weights and adds in its epilogue); what is left of the block is the activation
    const bool fold = bn_folded_of(d);
    const float* fscale = fold ? (const float*)((const char*)w_prepared + align_up(fwd_a_bytes(d, pl), 256)) : nullptr;
    const int ep = d->mode == MS_BARE ? EP_BARE : (d->mode == MS_LRELU || fold) ? EP_LRELU : EP_BN_EVAL;
    return launch_conv16_c1(g.dt, x, w, fscale, fold ? fscale + g.C : bias, y, gamma, beta, running_mean, running_var, stats, counts,
                            d->B, d->H, d->W, ep, d->slope, d->eps, s);
  }
  const size_t a_bytes = align_up(fwd_a_bytes(d, pl), 256);
  const void* A = w_prepared;
  const float* bias_use = bias;
  int rc;
  const bool folded = bn_folded_of(d);
  if (w_prepared) {
    if (folded) bias_use = (const float*)((const char*)w_prepared + a_bytes) + g.C;
  } else {
    float* fold = (float*)(wsp + weights16_bytes(d, 0));        // scale | bias'
    const float* scale = nullptr;
    if (folded) {
      rc = launch_bn_fold(bias, gamma, beta, running_mean, running_var, fold, fold + g.C, g.C, d->eps, s);
      if (rc) return rc;
      scale = fold; bias_use = fold + g.C;
    }
    Prep16Batch pb;
    pb.n = 1; pb.job[0] = fwd_job(d, pl, w, wsp, scale);
    rc = launch_prep16_multi(pb, s);
    if (rc) return rc;
    A = wsp;
  }
  Conv16Args a = {};
  a.A = A; a.src = x; a.src2 = x2;
  a.out = d->mode == MS_BN_TRAIN ? y_raw : y;
  a.bias = bias_use; a.stats = stats; a.counts = counts;
  a.Mg = d->Cout; a.groups = d->groups; a.Kc8g = c8_of(d->Cin); a.bcast = g.bcast;
  // eval BatchNorm: either folded into the prepared weights and bias (inference: what is left is the activation), or applied
  // from the running statistics in the epilogue (eval passes inside training, whose statistics move between calls)
  a.ep = d->mode == MS_BARE ? EP_BARE : d->mode == MS_BN_TRAIN ? EP_RAW_STATS : (d->mode == MS_LRELU || folded) ? EP_LRELU : EP_BN_EVAL;
  a.bn_g = gamma; a.bn_b = beta; a.bn_m = running_mean; a.bn_v = running_var;
  a.KH = d->KH; a.S = d->SW; a.SV = g.one_d ? 1 : d->SH;
  if (g.one_d) {
    a.SRCH = d->B; a.SRCW = d->W; a.s_img = 0; a.s_cblk = d->W; a.s_row = g.cin8_tot * d->W;
    a.OUTH = d->B; a.OUTW = d->OW; a.o_img = 0; a.o_cblk = d->OW; a.o_row = g.C8 * d->OW;
    a.of_img = 0; a.of_chan = d->OW; a.of_row = g.C * d->OW;
    a.PH = 0;
  } else {
    a.SRCH = d->H; a.SRCW = d->W; a.s_img = g.cin8_tot * d->H * d->W; a.s_cblk = d->H * d->W; a.s_row = d->W;
    a.OUTH = d->OH; a.OUTW = d->OW; a.o_img = g.C8 * g.hw; a.o_cblk = g.hw; a.o_row = d->OW;
    a.of_img = g.C * g.hw; a.of_chan = g.hw; a.of_row = d->OW;
    a.PH = d->PH;
  }
  a.PW = d->PW; a.o_sh = 1; a.o_sw = 1;
  a.slope = d->slope; a.eps = d->eps;
  if (outf32 && d->mode != MS_BN_TRAIN) a.out_f32 = (float*)y;
  // train-mode BatchNorm inside the conv launch (conv16_kernel.h, EP_BN_FUSED): the workgroups of a channel tile exchange
  // their partial statistics and normalise from registers -- taken when the whole grid is resident at once
  bool fused = false;
  const int sg = d->mode == MS_BN_TRAIN ? sg_of(d) : 1;
  if (d->mode == MS_BN_TRAIN && !outf32 && g_bn_fused && bn_sync && sg == 1) {
    const int bm = 64 * pl.wm, gy = cdiv(d->Cout, bm), nwg = pl.n_tiles * gy * d->groups;
    const int scratch = (pl.nwn * bm * 4 + (128 * pl.nwn / bm) * bm * 3 * 2 + bm * 2 + bm / 8 + pl.nwn + 1) * 4;      // red | dred | scsh | rawflag | wcnt
    // (every workgroup reads its group's n_tiles partials: beyond 64 tiles per channel tile that traffic -- n_tiles^2 x 1 KB per
    // channel tile -- costs more than the normalising launch it replaces; measured on the 512-tile audio-encoder layers)
    fused = pl.n_tiles <= 64 && nwg >= g_bn_fused_min_wgs && (1 + gy * d->groups) * BNF_SYNC_WORDS_PER_GROUP <= bn_sync_words &&
            pl.lds_bytes >= scratch &&
            conv16_coresident(g.dt, pl, d->KW, g.up2 != 0, nwg);
  }
  if (fused) {
    a.ep = EP_BN_FUSED; a.out = y; a.out_raw = y_raw; a.bn_part = bn_part; a.bn_sync = bn_sync; a.save = save;
    a.momentum = d->momentum;
    a.raw_all = (long)d->B * g.hw > BN_BWD16_FUSED_MAX;      // its two-pass backward reads y_raw, not y
  }
  const double esz = 2.0;
  const double flops = 2.0 * d->Cout * d->Cin * d->KH * d->KW * (double)g.npix * d->groups;
  const double bytes = esz * ((double)g.C * d->Cin * d->KH * d->KW + (double)d->B * g.cin_tot * d->H * d->W) +
                       (a.out_f32 ? 4.0 : esz) * (double)g.npix * g.C;
  rc = launch_conv16(g.dt, a, pl, d->KW, g.up2 != 0, flops, bytes, s);          // (bytes: algorithmic |x| + |w| + |y| of the block)
  if (rc || fused) return rc;
  if (d->mode == MS_BN_TRAIN && pl.n_tiles <= 64)      // few statistics tiles: finalize inside the normalising launch
    return launch_bn_finalize_apply16(g.dt, stats, counts, pl.n_tiles, g.npix, gamma, beta, running_mean, running_var, save, d->eps,
                                      d->momentum, y_raw, outf32 ? nullptr : y, outf32 ? (float*)y : nullptr, d->B, g.C, g.hw,
                                      d->slope, s, sg);
  if (d->mode == MS_BN_TRAIN) {
    rc = launch_bn_finalize(stats, counts, pl.n_tiles, 0, g.npix, g.C, gamma, beta, running_mean, running_var, save, d->eps,
                            d->momentum, s);
    if (rc) return rc;
    rc = launch_bn_apply16(g.dt, y_raw, outf32 ? nullptr : y, outf32 ? (float*)y : nullptr, save, d->B, g.C, g.hw, d->slope, s);
  }
  return rc;
}

int block_bwd16(const ms_conv_desc* d, const void* x, const void* x2, const float* w, const float* gamma, const void* y_raw,
                const void* y, const float* save, const void* dy, void* dyr, void* dx, void* dx2, float* dw, float* dbias,
                float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, hipStream_t s, const void* wt_prepared,
                float* wgrad_partials, int defer_wgrad_launch) {
  const Geo16 g = geo_of(d);
  if (g.dt != DT_BF16 && g.dt != DT_F16) return set_error("ms_conv_block_bwd: dtype %d", d->dtype);
  if (workspace_bytes < block_bwd16_workspace(d)) return set_error("ms_conv_block_bwd: workspace too small");
  const bool outf32 = out_f32_of(d);
  if (outf32 && d->mode == MS_LRELU) return set_error("ms_conv_block_bwd: fp32 outputs are for BARE and BatchNorm blocks");
  int bpc;
  const int nchunk = bwd16_chunks(d->B, g.C8, g.hw, &bpc);
  char* wsp = (char*)workspace;
  float* bn_part = (float*)wsp; wsp += align_up((size_t)g.C8 * 8 * nchunk * 2 * sizeof(float), 256);
  float* colpart = (float*)wsp; wsp += align_up((size_t)g.C8 * 8 * nchunk * sizeof(float), 256);
  char* wt_ws = wsp; wsp += weights16_bytes(d, 1);
  float* wg_part = (float*)wsp;
  int rc = 0;

  // 1. gradient wrt the raw conv output (+ per-channel column sums = bias gradient)
  const void* gsrc = dy;
  const float* dyf = outf32 ? (const float*)dy : nullptr;
  int bias_done = 0;
  if (d->mode == MS_BN_TRAIN) {
    // (y: blocks with a cb8 output read x_hat and the activation mask from it wherever the map inverts safely -- conv16.h)
    rc = launch_bn_bwd16(g.dt, outf32 ? nullptr : dy, dyf, y_raw, outf32 ? nullptr : y, save, gamma, bn_part, dyr, colpart, dbias, dgamma, dbeta, d->B, g.C,
                         g.hw, d->slope, &bias_done, s, sg_of(d));
    gsrc = dyr;
  } else if (d->mode == MS_LRELU) {
    rc = launch_act_bwd16(g.dt, dy, nullptr, y, dyr, colpart, d->B, g.C, g.hw, 1, d->slope, s);
    gsrc = dyr;
  } else if (dbias || outf32) {
    rc = launch_act_bwd16(g.dt, outf32 ? nullptr : dy, dyf, nullptr, dyr, colpart, d->B, g.C, g.hw, 0, 0.f, s);
    if (outf32) gsrc = dyr;
  }
  if (rc) return rc;
  if (dbias && !bias_done) {
    rc = launch_colsum16(colpart, dbias, g.C, nchunk, s);
    if (rc) return rc;
  }

  // 2. data gradient: each output-parity class is a dense stride-1 forward conv of dy_raw with the class's taps reversed
  if (dx) {
    const Dgrad16 dg = dgrad_plan16(d);
    if (!dg.pl.ok) return set_error("ms_conv_block_bwd: no 16-bit data-gradient kernel for this geometry");
    const void* A = wt_prepared;
    if (!A) {
      Prep16Batch pb;
      pb.n = 1; pb.job[0] = dgrad_job(d, dg, w, wt_ws);
      rc = launch_prep16_multi(pb, s);
      if (rc) return rc;
      A = wt_ws;
    }
    Conv16Args q = {};
    q.A = A; q.src = gsrc; q.out = dx; q.out2 = dx2;
    q.Mg = d->Cin; q.groups = dg.tg; q.Kc8g = c8_of(dg.tcog); q.bcast = 0;
    q.ep = g.up2 ? EP_DGRAD_UP2 : EP_BARE; q.is_dgrad = 1;
    q.KH = dg.jh; q.S = 1; q.SV = 1;
    q.ncls = dg.ncls;
    double flops = 0;
    for (int cls = 0; cls < dg.ncls; ++cls) {
      const int ry = cls / d->SW, rx = cls - ry * d->SW;
      const int kh0 = (ry + d->PH) % d->SH, kw0 = (rx + d->PW) % d->SW;
      const int cy = (ry + d->PH - kh0) / d->SH, cx = (rx + d->PW - kw0) / d->SW;
      const int QH = std::max(0, (d->H - ry + d->SH - 1) / d->SH), QW = std::max(0, (d->W - rx + d->SW - 1) / d->SW);
      q.cls_PH[cls] = g.one_d ? 0 : (dg.jh - 1) - cy; q.cls_PW[cls] = (dg.jw - 1) - cx;
      q.cls_OUTH[cls] = g.one_d ? d->B : QH; q.cls_OUTW[cls] = QW;
      q.cls_ry[cls] = g.one_d ? 0 : ry; q.cls_rx[cls] = rx;
      flops += 2.0 * d->Cin * dg.tcog * dg.jh * dg.jw * (double)d->B * (g.one_d ? 1 : QH) * QW * dg.tg;
    }
    if (g.one_d) {
      q.SRCH = d->B; q.SRCW = d->OW; q.s_img = 0; q.s_cblk = d->OW; q.s_row = g.C8 * d->OW;
      q.o_img = 0; q.o_cblk = d->W; q.o_row = g.cin8_tot * d->W; q.o_sh = 1;
    } else {
      q.SRCH = d->OH; q.SRCW = d->OW; q.s_img = g.C8 * g.hw; q.s_cblk = g.hw; q.s_row = d->OW;
      q.o_img = g.cin8_tot * d->H * d->W; q.o_cblk = d->H * d->W; q.o_row = d->W; q.o_sh = d->SH;
    }
    q.o_sw = d->SW;
    q.PH = q.cls_PH[0]; q.PW = q.cls_PW[0]; q.OUTH = q.cls_OUTH[0]; q.OUTW = q.cls_OUTW[0]; q.o_ry = q.cls_ry[0]; q.o_rx = q.cls_rx[0];
    q.slope = d->slope; q.eps = d->eps;
    const double bytes = 2.0 * ((double)g.C * d->Cin * d->KH * d->KW + (double)d->B * g.C * g.hw + (double)d->B * g.cin_tot * d->H * d->W);
    rc = launch_conv16(g.dt, q, dg.pl, dg.jw, false, flops, bytes, s);
    if (rc) return rc;
  }

  // 3. weight gradient
  if (dw) {
    const Wgrad16Plan wp = wgrad_plan16(d);
    if (!wp.tp) return set_error("ms_conv_block_bwd: no 16-bit weight-gradient kernel for this geometry");
    const size_t wsize = (size_t)g.C * d->Cin * d->KH * d->KW;
    Wgrad16Args a = {};
    a.dyr = gsrc; a.src = x; a.src2 = x2;
    const bool defer = wgrad_partials != nullptr && wp.splits > 1;
    a.out = wp.splits > 1 ? (defer ? wgrad_partials : wg_part) : dw;
    a.out_split_stride = wsize;
    a.Cog = d->Cout; a.Cig = d->Cin; a.groups = d->groups; a.bcast = g.bcast;
    a.KH = d->KH; a.KW = d->KW; a.S = d->SW; a.SV = g.one_d ? 1 : d->SH; a.PW = d->PW;
    if (g.one_d) {
      a.SRCH = d->B; a.SRCW = d->W; a.s_img = 0; a.s_cblk = d->W; a.s_row = g.cin8_tot * d->W; a.PH = 0;
      a.OUTH = d->B; a.OUTW = d->OW; a.o_img = 0; a.o_cblk = d->OW; a.o_row = g.C8 * d->OW;
    } else {
      a.SRCH = d->H; a.SRCW = d->W; a.s_img = g.cin8_tot * d->H * d->W; a.s_cblk = d->H * d->W; a.s_row = d->W; a.PH = d->PH;
      a.OUTH = d->OH; a.OUTW = d->OW; a.o_img = g.C8 * g.hw; a.o_cblk = g.hw; a.o_row = d->OW;
    }
    const double flops = 2.0 * d->Cout * d->Cin * d->KH * d->KW * (double)g.npix * d->groups;
    const double bytes = 2.0 * ((double)g.npix * g.C + (double)d->B * g.cin_tot * d->H * d->W) + 4.0 * (double)wsize;
    // queued form: only when nothing of this call reads the result (dw written in place, or slabs left for the caller)
    if (defer_wgrad_launch && (wp.splits == 1 || defer) && !wgrad16_c1_ok(a, g.up2 != 0)) return queue_wgrad16(g.dt, a, wp, g.up2 != 0, flops, bytes);
    rc = launch_wgrad16(g.dt, a, wp, g.up2 != 0, flops, bytes, s);
    if (rc) return rc;
    if (wp.splits > 1 && !defer) rc = launch_reduce_splits(wg_part, dw, (int)wsize, wp.splits, s);
  }
  return rc;
}

}  // namespace ms

using namespace ms;

extern "C" {

int ms_wgrad_flush(void* stream) {
  const int rc = wgrad_patch_flush((hipStream_t)stream);
  const int rcg = wgrad_gather_flush((hipStream_t)stream);
  const int rc16 = wgrad16_flush((hipStream_t)stream);
  return rc ? rc : rcg ? rcg : rc16;
}
int ms_wgrad_discard(void) { wgrad_patch_discard(); wgrad_gather_discard(); wgrad16_discard(); return 0; }

int ms_set_bn_sync_buffer(int32_t* zeroed_words, int n) {
  g_bn_sync = zeroed_words;
  g_bn_sync_n = zeroed_words ? n : 0;
  return 0;
}
int ms_debug_set_bn_fused(int on) {
  const int prev = g_bn_fused;
  g_bn_fused = on ? 1 : 0;
  return prev;
}
int ms_debug_set_bn_fused_min_workgroups(int n) {
  const int prev = g_bn_fused_min_wgs;
  g_bn_fused_min_wgs = n < 0 ? 0 : n;
  return prev;
}

size_t ms_weights16_bytes(const ms_conv_desc* d, int which) {
  if (!d || (dt_of(d) != DT_BF16 && dt_of(d) != DT_F16) || (which != 0 && which != 1)) return 0;
  return weights16_bytes(d, which);
}

int ms_weights16_prepare(int n, const ms_prep16_item* items, void* stream) {
  if (n < 0 || (n && !items)) return set_error("ms_weights16_prepare: null argument");
  hipStream_t s = (hipStream_t)stream;
  Prep16Batch pb;
  pb.n = 0;
  auto flush = [&]() -> int {
    const int rc = pb.n ? launch_prep16_multi(pb, s) : 0;
    pb.n = 0;
    return rc;
  };
  for (int i = 0; i < n; ++i) {
    const ms_prep16_item& it = items[i];
    const ms_conv_desc* d = it.desc;
    if (!d || !it.w) return set_error("ms_weights16_prepare: item %d: null descriptor or weights", i);
    if (dt_of(d) != DT_BF16 && dt_of(d) != DT_F16) return set_error("ms_weights16_prepare: item %d is not a 16-bit block", i);
    int rc;
    if (it.fwd) {
      const Conv16Plan pl = plan_conv16((d->H == 1 && d->KH == 1) ? 1 : 2, d->Cout, d->groups, d->Cin, d->KH, d->KW, d->SH, d->SW, d->B,
                                        d->OH, d->OW, 1, d->in_mode == MS_IN_UP2ADD);
      if (!pl.ok) return set_error("ms_weights16_prepare: item %d: geometry not supported", i);
      const float* scale = nullptr;
      if (bn_folded_of(d)) {
        if (!it.gamma || !it.beta || !it.running_mean || !it.running_var) return set_error("ms_weights16_prepare: item %d: BN tensors missing", i);
        const int C = d->groups * d->Cout;
        float* fold = (float*)((char*)it.fwd + align_up(conv16_weight_bytes(pl, d->Cout, d->groups, d->Cin, d->KH, d->KW, 1), 256));
        rc = launch_bn_fold(it.bias, it.gamma, it.beta, it.running_mean, it.running_var, fold, fold + C, C, d->eps, s);
        if (rc) return rc;
        scale = fold;
      }
      Prep16Job jb = {};
      jb.w = it.w; jb.out = it.fwd; jb.scale = scale;
      jb.groups = d->groups; jb.Cog = d->Cout; jb.Cig = d->Cin; jb.KH = d->KH; jb.KW = d->KW; jb.SH = d->SH; jb.SW = d->SW;
      jb.PH = d->PH; jb.PW = d->PW; jb.bcast = d->in_mode == MS_IN_BCAST;
      jb.BM = 64 * pl.wm; jb.CK8 = pl.ck8; jb.nchunks = pl.nchunks; jb.n_mt = cdiv(d->Cout, jb.BM); jb.dt = dt_of(d);
      pb.job[pb.n++] = jb;
      if (pb.n == PREP16_BATCH_MAX && (rc = flush())) return rc;
    }
    if (it.dgrad) {
      const bool bc = d->in_mode == MS_IN_BCAST;
      const int tg = bc ? 1 : d->groups, tcog = bc ? d->groups * d->Cout : d->Cout;
      const int jh = cdiv(d->KH, d->SH), jw = cdiv(d->KW, d->SW);
      const Conv16Plan pl = plan_conv16((d->H == 1 && d->KH == 1) ? 1 : 2, d->Cin, tg, tcog, jh, jw, 1, 1, d->B, cdiv(d->H, d->SH),
                                        cdiv(d->W, d->SW), d->SH * d->SW, false);
      if (!pl.ok) return set_error("ms_weights16_prepare: item %d: data-gradient geometry not supported", i);
      Prep16Job jb = {};
      jb.w = it.w; jb.out = it.dgrad; jb.dgrad = 1;
      jb.groups = d->groups; jb.Cog = d->Cout; jb.Cig = d->Cin; jb.KH = d->KH; jb.KW = d->KW; jb.SH = d->SH; jb.SW = d->SW;
      jb.PH = d->PH; jb.PW = d->PW; jb.bcast = bc;
      jb.BM = 64 * pl.wm; jb.CK8 = pl.ck8; jb.nchunks = pl.nchunks; jb.n_mt = cdiv(d->Cin, jb.BM); jb.dt = dt_of(d);
      pb.job[pb.n++] = jb;
      if (pb.n == PREP16_BATCH_MAX && (rc = flush())) return rc;
    }
  }
  return flush();
}

}  // extern "C"
