"""GPU: where the HOST time of MixStageTrainStep.step() goes (cProfile over N replayed steps with every labelled launch of the
library dropped, so that the device never holds the host back).   python3 tools/prof_host.py PRECISION KIND [N]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mix_stage_amd import _lib
from mix_stage_amd.train_step import MixStageTrainStep
from oracle import mixstage_oracle as O
precision, kind = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 300
ALL = 'conv_;decoder_chain;chain_prep;clip_prep;gdgrad_prep;reduce_splits;wgrad_reduce;bn_;act_bwd;transpose_weight;split_weights;splitk_;ew_;prep16;cb8_'
dev = torch.device('cuda:0')
audio, pose, labels, style = O.synthetic_batch(32, M=8, S=8)
batch = [t.to(dev) for t in (audio, labels, pose, style)]
_lib.lib().ms_debug_set_skip(ALL.encode())
model = bench.build_model(dev, precision)
ts = MixStageTrainStep(model, use_graphs=True)
ts.on_bad_step = 'skip'
import warnings; warnings.simplefilter('ignore')
for _ in range(6):
  ts.step(*batch, kind=kind)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
  ts.step(*batch, kind=kind)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host %s %s-step: %.4f ms per call issued, %.4f ms incl. final sync (%d calls)' % (precision, kind, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, n))
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
  ts.step(*batch, kind=kind)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(28)
