"""Loading reference-format weights ("next" row N4 of SURVEY.md section 8f).

The reference's experiments store `model.state_dict()` of `GAN(G, D)` as a pickle (`*_weights.p`, README.md:124-141; written by
pycasper's BookKeeper, which is not part of the reference tree -- so the exact container is PARITY UNPINNED: handled here are a
pickled or torch.save'd mapping name -> tensor / numpy array, optionally under a 'model' / 'state_dict' key, optionally with a
DataParallel 'module.' prefix, in fp32 or fp64 (`.double()` models, trainer.py).  The module tree of this package mirrors the
reference's state_dict schema (tests/test_boundary_cpu.py), so such a file loads with strict=True.
"""
import io
import os
import pickle

import torch


def _as_mapping(obj):
  for key in ('model', 'state_dict', 'model_state_dict'):
    if isinstance(obj, dict) and key in obj and isinstance(obj[key], dict):
      obj = obj[key]
  if not isinstance(obj, dict):
    raise TypeError('expected a mapping name -> tensor, got %s' % type(obj).__name__)
  return obj


class _TensorsOnlyUnpickler(pickle.Unpickler):
  """A plain pickle of a state_dict needs only containers, torch tensors (rebuilt through torch._utils) and numpy arrays;
  anything else in the stream is refused instead of executed."""
  _ALLOWED = {
      ('collections', 'OrderedDict'), ('builtins', 'dict'), ('builtins', 'list'), ('builtins', 'tuple'), ('builtins', 'set'),
      ('torch._utils', '_rebuild_tensor_v2'), ('torch._utils', '_rebuild_parameter'), ('torch._utils', '_rebuild_tensor'),
      ('torch', 'Size'), ('torch.storage', '_load_from_bytes'), ('torch', 'device'),
      ('numpy.core.multiarray', '_reconstruct'), ('numpy._core.multiarray', '_reconstruct'), ('numpy', 'ndarray'),
      ('numpy', 'dtype'), ('_codecs', 'encode'), ('numpy.core.multiarray', 'scalar'), ('numpy._core.multiarray', 'scalar'),
  }

  @staticmethod
  def _load_from_bytes_weights_only(b):
    # torch.storage._load_from_bytes is torch.load(..., weights_only=False) in torch >= 2.x: a nested torch.save payload
    # would be unpickled without any restriction.  Same call, restricted.
    return torch.load(io.BytesIO(b), map_location='cpu', weights_only=True)

  def find_class(self, module, name):
    if (module, name) == ('torch.storage', '_load_from_bytes'):
      return self._load_from_bytes_weights_only
    if (module, name) in self._ALLOWED or (module == 'torch' and name.endswith(('Storage', 'Tensor'))) or \
        (module == 'torch' and name in ('float32', 'float64', 'float16', 'bfloat16', 'int64', 'int32', 'uint8', 'bool')):
      return super().find_class(module, name)
    raise pickle.UnpicklingError('refusing to load %s.%s from a weights file (pass trusted=True for files you produced yourself)'
                                 % (module, name))


def read_weights(source, trusted=False):
  """source: a path (str / os.PathLike; pickle or torch.save file) or an already loaded mapping.  Returns an ordered
  name -> tensor dict.  Files are read with torch.load(weights_only=True) first, then as a plain pickle through an
  unpickler restricted to containers, tensors and numpy arrays; trusted=True allows an unrestricted pickle.load (arbitrary
  code execution: only for files you produced yourself)."""
  if isinstance(source, (str, bytes, os.PathLike)) and not isinstance(source, dict):
    path = os.fspath(source)
    if not os.path.isfile(path):
      raise FileNotFoundError(path)
    try:
      obj = torch.load(path, map_location='cpu', weights_only=True)
    except (pickle.UnpicklingError, RuntimeError, ValueError, EOFError, KeyError, AttributeError):
      with open(path, 'rb') as f:
        obj = pickle.load(f) if trusted else _TensorsOnlyUnpickler(io.BytesIO(f.read())).load()
  else:
    obj = source
  out = {}
  for name, v in _as_mapping(obj).items():
    if name.startswith('module.'):
      name = name[len('module.'):]
    out[name] = v if torch.is_tensor(v) else torch.as_tensor(v)
  return out


def load_weights(model, source, strict=True, trusted=False):
  """Load reference-format weights into a mix_stage_amd GAN (or any sub-module).  Floating-point tensors are cast to the
  module's dtype (fp64 checkpoints -> fp32); parameters owned by a MixStageTrainStep keep living in its flat buffers (the copy
  is in place) and the trainer notices the edit through the parameters' version counters."""
  weights = read_weights(source, trusted=trusted)
  own = model.state_dict()
  cast = {}
  for name, v in weights.items():
    ref = own.get(name)
    if ref is not None and v.is_floating_point() and ref.is_floating_point():
      v = v.to(ref.dtype)
    if ref is not None and tuple(v.shape) != tuple(ref.shape) and v.numel() == ref.numel():
      v = v.reshape(ref.shape)          # e.g. 0-dim vs 1-element num_batches_tracked
    cast[name] = v
  return model.load_state_dict(cast, strict=strict)


def save_weights(model, path, train_step=None):
  """Write `model.state_dict()` in the reference's container (a torch.save'd name -> CPU tensor mapping: what BookKeeper's
  `*_weights.p` holds, README.md:124-141).  With a MixStageTrainStep the device is checked first (check_health: a refused step or
  an expired in-launch meeting raises there), and a state that holds a non-finite value is never written."""
  if train_step is not None:
    train_step.check_health()
  state = {k: v.detach().to('cpu') for k, v in model.state_dict().items()}
  bad = [k for k, v in state.items() if v.is_floating_point() and not bool(torch.isfinite(v).all())]
  if bad:
    raise RuntimeError('refusing to save: %d tensors hold non-finite values (first: %s)' % (len(bad), bad[0]))
  # written beside the target under a name of its own (process id + random suffix), then renamed over it: concurrent savers -- the
  # ranks of a data-parallel job pointed at one path; save from rank 0 only -- never see or replace each other's partial file
  import tempfile
  target = os.fspath(path)
  fd, tmp = tempfile.mkstemp(prefix=os.path.basename(target) + '.', suffix='.%d.tmp' % os.getpid(), dir=os.path.dirname(target) or '.')
  try:
    with os.fdopen(fd, 'wb') as f:
      torch.save(state, f)
    os.replace(tmp, target)
  except BaseException:
    if os.path.exists(tmp):
      os.unlink(tmp)
    raise
  return path
