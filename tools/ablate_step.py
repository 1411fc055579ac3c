"""GPU: what each kernel family costs INSIDE the captured train step.  Per family the launches whose timing label matches are
dropped (ms_debug_set_skip: results become meaningless, only the replay time is read), the step is captured again and the
G-step / D-step replay time is compared with the complete step.  Eager event timings overstate small kernels; this does not.

  python tools/ablate_step.py [fp32|bf16x6|bf16] [reps]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from mix_stage_amd import _lib  # noqa: E402
from mix_stage_amd.train_step import MixStageTrainStep  # noqa: E402
from oracle import mixstage_oracle as O  # noqa: E402

dev = torch.device('cuda:0')
precision = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
audio, pose, labels, style = O.synthetic_batch(32, M=8, S=8)
batch = [t.to(dev) for t in (audio, labels, pose, style)]

FAMILIES = {
    'fp32': [('chained decoder (decoder.0-3 + logits + mixture, one launch)', 'decoder_chain'),
             ('clip-resident 1-D blocks, forward (conv + BN + LeakyReLU in one launch)', 'conv_fwd_clip'),
             ('clip-resident data gradients (1-D blocks and grouped decoder)', 'conv_dgrad_clip;conv_dgrad_gclip'),
             ('conv fwd (all, chain excluded)', 'conv_fwd'), ('conv dgrad (all)', 'conv_dgrad'), ('conv wgrad', 'conv_wgrad'),
             ('split-K epilogues', 'splitk_'), ('wgrad slab reduce', 'reduce_splits;wgrad_reduce'), ('bn fwd', 'bn_finalize;bn_apply'),
             ('bn bwd', 'bn_bwd'), ('activation bwd of the blocks without BN', 'act_bwd'), ('weight prep', 'transpose_weight;split_weights;chain_prep;clip_prep;gdgrad_prep'),
             ('losses, mixing, Adam, converters', 'ew_'), ('softmax mixture alone (fwd + bwd)', 'ew_softmax_mix'),
             ('everything labelled', 'conv_;decoder_chain;chain_prep;clip_prep;gdgrad_prep;reduce_splits;wgrad_reduce;bn_;act_bwd;transpose_weight;split_weights;splitk_;ew_')],
    'bf16': [('chained decoder (decoder.0-3 + logits + mixture, one launch)', 'decoder_chain'), ('conv fwd', 'conv_fwd'), ('conv dgrad', 'conv_dgrad'), ('conv wgrad', 'conv_wgrad'),
             ('wgrad slab reduce', 'reduce_splits;wgrad_reduce'), ('bn fwd', 'bn_finalize;bn_apply'), ('bn bwd', 'bn_bwd'), ('activation bwd of the blocks without BN', 'act_bwd'),
             ('weight prep', 'prep16;chain_prep'), ('layout converters', 'cb8_'),
             ('losses, mixing, Adam', 'ew_'), ('softmax mixture alone (fwd + bwd)', 'ew_softmax_mix'),
             ('everything labelled', 'conv_;decoder_chain;chain_prep;reduce_splits;wgrad_reduce;bn_;act_bwd;prep16;cb8_;splitk_;ew_')],
}
FAMILIES['bf16x6'] = FAMILIES['fp32']


def measure(skip):
  _lib.lib().ms_debug_set_skip(skip.encode() if skip else None)
  model = bench.build_model(dev, precision)
  ts = MixStageTrainStep(model, use_graphs=True)
  ts._post_health = lambda: None      # (a dropped family leaves garbage behind: the optimizer refuses such steps on the device; nobody needs to hear of it here)
  out = {}
  for kind in 'GD':
    for _ in range(4):
      ts.step(*batch, kind=kind)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
      ts.step(*batch, kind=kind)
    torch.cuda.synchronize()
    out[kind] = (time.perf_counter() - t0) / reps * 1e3
  _lib.lib().ms_debug_set_skip(None)
  del ts, model
  return out


if os.environ.get('MS_RING'):          # cap the conv16 LDS-DMA ring depth (experiments)
  _lib.lib().ms_debug_set_conv16_ring(int(os.environ['MS_RING']), 0)
if os.environ.get('MS_WG_TARGET'):     # the same for the fp32 patch-staged weight gradient
  _lib.lib().ms_debug_set_wgrad_target(int(os.environ['MS_WG_TARGET']))
if os.environ.get('MS_WG16_TARGET'):   # workgroups a 16-bit weight-gradient launch aims for (pixel splits)
  _lib.lib().ms_debug_set_wgrad16_target(int(os.environ['MS_WG16_TARGET']))
base = measure('')
print('%-24s G %.3f ms   D %.3f ms' % ('complete step', base['G'], base['D']))
for name, pat in ([] if os.environ.get('ABL_BASE_ONLY') else FAMILIES[precision]):
  m = measure(pat)
  print('%-24s G %.3f ms (-%.3f)   D %.3f ms (-%.3f)   [%s]' % ('without ' + name, m['G'], base['G'] - m['G'], m['D'],
                                                               base['D'] - m['D'], pat))
base2 = measure('')
print('%-24s G %.3f ms   D %.3f ms' % ('complete step (again)', base2['G'], base2['D']))
