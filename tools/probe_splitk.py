import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mix_stage_amd import ops, _lib
from mix_stage_amd._lib import MS_BARE
from tools.probe_conv import time_conv
for sk in (1, 2, 3, 4, 6, 8):
  _lib.lib().ms_debug_set_patch_tuning(0, sk)
  ops.timing_enable(True)
  us, tf = time_conv(32, 256, 256, 64, iters=30)
  torch.cuda.synchronize()
  rows = ops.timing_report(); ops.timing_enable(False)
  det = ', '.join('%s %.1fus' % (r['label'].split('|')[-1].split()[0], r['total_ms'] / r['count'] * 1e3) for r in rows)
  print('splitk %d: op %.1f us | %s' % (sk, us, det))
