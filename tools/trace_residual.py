"""GPU: what is left of a captured step when every labelled launch of the library is dropped (ms_debug_set_skip, the
"without everything labelled" row of tools/ablate_step.py): replays the remaining graph under `rocprofv3 --kernel-trace`
(run it under the profiler; tools/residual.sh) so that every remaining launch -- torch kernels, fills, copies -- is listed.
  python3 tools/trace_residual.py PRECISION KIND N   -> prints wall ms per replayed step as well"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mix_stage_amd import _lib
from mix_stage_amd.train_step import MixStageTrainStep
from oracle import mixstage_oracle as O
precision, kind, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
ALL = 'conv_;decoder_chain;chain_prep;clip_prep;gdgrad_prep;reduce_splits;wgrad_reduce;bn_;act_bwd;transpose_weight;split_weights;splitk_;ew_;prep16;cb8_'
dev = torch.device('cuda:0')
audio, pose, labels, style = O.synthetic_batch(32, M=8, S=8)
batch = [t.to(dev) for t in (audio, labels, pose, style)]
_lib.lib().ms_debug_set_skip(ALL.encode())
model = bench.build_model(dev, precision)
ts = MixStageTrainStep(model, use_graphs=True)
ts.on_bad_step = 'skip'
import warnings; warnings.simplefilter('ignore')
for _ in range(4):
  ts.step(*batch, kind=kind)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
  ts.step(*batch, kind=kind)
torch.cuda.synchronize()
print('residual %s %s-step: %.4f ms per replay (wall, %d replays)' % (precision, kind, (time.perf_counter() - t0) / n * 1e3, n))
# the same without the per-step input copies (inputs_unchanged) and host-side health polling: what the graph alone costs
t0 = time.perf_counter()
for _ in range(n):
  ts.step(*batch, kind=kind, inputs_unchanged=True)
torch.cuda.synchronize()
print('residual %s %s-step, inputs_unchanged: %.4f ms per replay' % (precision, kind, (time.perf_counter() - t0) / n * 1e3))
g = ts._graphs[next(iter(ts._graphs))]['fwd_bwd']
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
  g.replay()
torch.cuda.synchronize()
print('residual %s %s-step, bare graph replay: %.4f ms per replay' % (precision, kind, (time.perf_counter() - t0) / n * 1e3))
