import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from mix_stage_amd import ops
from mix_stage_amd.train_step import MixStageTrainStep
from oracle import mixstage_oracle as O
from test_gpu_model import build_hip_gan
DEV = 'cuda:0'
M = S = 4
batches = [O.synthetic_batch(4, M=M, S=S, seed=70 + i) for i in range(3)]
kinds = ['G', 'D', 'G']
def run(prepared, graphs=False):
  torch.manual_seed(5)
  model = build_hip_gan(M, S)
  ts = MixStageTrainStep(model, use_graphs=graphs)
  ops.enable_prepared_weights(prepared)
  snaps = []
  for (audio, pose, labels, style), k in zip(batches, kinds):
    ts.step(audio.to(DEV), labels.to(DEV), pose.to(DEV), style.to(DEV), kind=k)
    torch.cuda.synchronize()
    snaps.append((ts.optim_G.flat_p.clone(), ts.optim_D.flat_p.clone(), [float(l) for l in ts.losses]))
    if prepared:
      es = list(ops._prepared['entries'].values())
      st = {}
      for e in es: st.setdefault(e['w'].untyped_storage().data_ptr(), []).append(e['n'])
      print('  step', k, 'entries', len(es), 'with n>0', sum(1 for e in es if e['n']), 'storages', {hex(a): len(v) for a, v in st.items()},
            'G storage', hex(ts.optim_G.flat_p.untyped_storage().data_ptr()), 'D', hex(ts.optim_D.flat_p.untyped_storage().data_ptr()))
      # check every entry against a fresh transposition
      bad = 0
      for e in es:
        if not e['n']: continue
        ref = torch.empty_like(e['wt'])
        import ctypes
        from mix_stage_amd._lib import lib, ConvDesc
        d = (ConvDesc * 1)(e['d']); w = (ctypes.c_void_p * 1)(e['w'].data_ptr()); wt = (ctypes.c_void_p * 1)(ref.data_ptr())
        lib().ms_dgrad_weights_prepare(1, d, w, wt, None)
        torch.cuda.synchronize()
        if not torch.equal(ref, e['wt']): bad += 1
      print('   stale entries:', bad)
  return snaps
a = run(True, False); b = run(True, True)
for i in range(3):
  print('step', i, 'G equal', torch.equal(a[i][0], b[i][0]), 'D equal', torch.equal(a[i][1], b[i][1]), a[i][2][:3], b[i][2][:3])
