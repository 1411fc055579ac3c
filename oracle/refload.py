"""Container-only helper (test infrastructure): import the REFERENCE's own model files
read-only from /root/reference so the oracle restatement can be validated against them and
golden vectors can be generated (tests/golden/make_golden.py).

Nothing here copies reference source: the files are executed where they lie.  The reference
imports two symbols from the un-vendored `pycasper` package (README.md:54-63 tells the user to
clone master; no version pin).  They are provided by an in-memory stand-in whose semantics are
INFERRED from the call sites (PARITY UNPINNED, see oracle/mixstage_oracle.py header).

/root/reference does not exist on the GPU box; `available()` gates every caller.
"""
import contextlib
import importlib.util
import os
import sys
import types

REF_MODEL_DIR = '/root/reference/src/model'


def available():
  return os.path.isfile(os.path.join(REF_MODEL_DIR, 'layers.py'))


def _install_pycasper_standin():
  if 'pycasper.torchUtils' in sys.modules:
    return
  pkg = types.ModuleType('pycasper')
  pkg.__path__ = []
  tu = types.ModuleType('pycasper.torchUtils')

  @contextlib.contextmanager
  def some_grad(model):
    flags = [(p, p.requires_grad) for p in model.parameters()]
    for p, _ in flags:
      p.requires_grad_(False)
    try:
      yield
    finally:
      for p, f in flags:
        p.requires_grad_(f)

  class LambdaScheduler:
    def __init__(self, lmbdas, **kwargs):
      self.lmbdas = list(lmbdas)

    def step(self):
      return list(self.lmbdas)

  tu.some_grad = some_grad
  tu.LambdaScheduler = LambdaScheduler
  pkg.torchUtils = tu
  sys.modules['pycasper'] = pkg
  sys.modules['pycasper.torchUtils'] = tu


_cache = {}


def load():
  """Returns a namespace with the reference classes (layers, G, D, GAN)."""
  if 'ns' in _cache:
    return _cache['ns']
  assert available(), 'reference tree not present'
  _install_pycasper_standin()
  pkg = types.ModuleType('refmodel')
  pkg.__path__ = [REF_MODEL_DIR]
  sys.modules['refmodel'] = pkg
  mods = {}
  for name in ('layers', 'speech2gesture', 'joint_late_cluster_soft_style', 'gan'):
    spec = importlib.util.spec_from_file_location('refmodel.' + name,
                                                  os.path.join(REF_MODEL_DIR, name + '.py'))
    m = importlib.util.module_from_spec(spec)
    sys.modules['refmodel.' + name] = m
    spec.loader.exec_module(m)
    mods[name] = m
  ns = types.SimpleNamespace(
      layers=mods['layers'],
      JointLateClusterSoftStyle4_G=mods['joint_late_cluster_soft_style'].JointLateClusterSoftStyle4_G,
      Speech2Gesture_D=mods['speech2gesture'].Speech2Gesture_D,
      GAN=mods['gan'].GAN)
  _cache['ns'] = ns
  return ns


def load_transform_and_metrics():
  """The reference's src/data/transform.py and src/evaluation/metrics.py, executed where they lie, behind in-memory stand-ins
  for the modules of the dataset / trainer stack they import at module level (h5py, pycasper, the trainer): none of those
  is touched by the functions pinned here -- ZNorm.znorm (transform.py:221-226), KMeans.get_feats / predict
  (transform.py:352-410), L1 / VelL1 / PCK (metrics.py:94-131,247-303).  RemoveJoints still rests on the un-vendored
  pycasper.torchUtils.remove_slices (stays unpinned)."""
  if 'tm' in _cache:
    return _cache['tm']
  assert available(), 'reference tree not present'
  _install_pycasper_standin()

  class _Absent:
    def __init__(self, *a, **k):
      raise RuntimeError('stand-in for a module of the reference dataset stack')

  def stub(name, **attrs):
    if name not in sys.modules:
      m = types.ModuleType(name)
      m.__dict__.update(attrs)
      sys.modules[name] = m
    return sys.modules[name]
  stub('dataUtils', DummyData=_Absent)
  stub('text', POStagging=_Absent)
  stub('common', HDF5=_Absent)
  stub('skeleton', Skeleton2D=_Absent)
  stub('argsUtils', get_args_perm=lambda *a, **k: None)
  stub('trainer_chooser')
  bk = stub('pycasper.BookKeeper', BookKeeper=_Absent)
  sys.modules['pycasper'].BookKeeper = bk
  mods = {}
  for name, rel in (('ref_transform', 'data/transform.py'), ('ref_metrics', 'evaluation/metrics.py')):
    spec = importlib.util.spec_from_file_location(name, os.path.join(os.path.dirname(REF_MODEL_DIR), rel))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    mods[name] = m
  _cache['tm'] = types.SimpleNamespace(transform=mods['ref_transform'], metrics=mods['ref_metrics'])
  return _cache['tm']


def build_ref_gan(M=8, S=8, T=64, P=104, dtype=None, state=None, no_grad=0):
  """Reference GAN(G, D) in the job-script configuration (src/jobs/mix-stage.py:3)."""
  import torch
  ns = load()
  G = ns.JointLateClusterSoftStyle4_G(time_steps=T, out_feats=P, num_clusters=M,
                                      style_dict={i: i for i in range(S)}, style_dim=10,
                                      lambda_id=0.1, argmax=1, some_grad_flag=1, train_only=1,
                                      shape={})
  D = ns.Speech2Gesture_D(in_channels=P)
  model = ns.GAN(G, D, criterion='L1Loss', input_modalities=['audio/log_mel_400'],
                 update_D_prob_flag=0, no_grad=no_grad)
  if state is not None:
    model.load_state_dict(state)
  if dtype is not None:
    model.to(dtype)
  model.G.thresh.value, model.G.thresh.iters = 1, 10 ** 9
  return model
