"""Loading reference-format weights ("next" row N4 of SURVEY.md section 8f).

The reference's experiments store `model.state_dict()` of `GAN(G, D)` as a pickle (`*_weights.p`, README.md:124-141; written by
pycasper's BookKeeper, which is not part of the reference tree -- so the exact container is PARITY UNPINNED: handled here are a
pickled or torch.save'd mapping name -> tensor / numpy array, optionally under a 'model' / 'state_dict' key, optionally with a
DataParallel 'module.' prefix, in fp32 or fp64 (`.double()` models, trainer.py).  The module tree of this package mirrors the
reference's state_dict schema (tests/test_boundary_cpu.py), so such a file loads with strict=True.
"""
import pickle

import torch


def _as_mapping(obj):
  for key in ('model', 'state_dict', 'model_state_dict'):
    if isinstance(obj, dict) and key in obj and isinstance(obj[key], dict):
      obj = obj[key]
  if not isinstance(obj, dict):
    raise TypeError('expected a mapping name -> tensor, got %s' % type(obj).__name__)
  return obj


def read_weights(source):
  """source: a path (pickle or torch.save file) or an already loaded mapping.  Returns an ordered name -> fp-tensor dict."""
  if isinstance(source, (str, bytes)) and not isinstance(source, dict):
    try:
      obj = torch.load(source, map_location='cpu', weights_only=False)
    except Exception:
      with open(source, 'rb') as f:
        obj = pickle.load(f)
  else:
    obj = source
  out = {}
  for name, v in _as_mapping(obj).items():
    if name.startswith('module.'):
      name = name[len('module.'):]
    out[name] = v if torch.is_tensor(v) else torch.as_tensor(v)
  return out


def load_weights(model, source, strict=True):
  """Load reference-format weights into a mix_stage_amd GAN (or any sub-module).  Floating-point tensors are cast to the
  module's dtype (fp64 checkpoints -> fp32); parameters owned by a MixStageTrainStep keep living in its flat buffers (the copy
  is in place) and the trainer notices the edit through the parameters' version counters."""
  weights = read_weights(source)
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
