"""N4: reference-format weight files (pickled fp64 state_dict, DataParallel prefix, wrapper key) load into the package's modules."""
import io
import pickle

import torch

from oracle import mixstage_oracle as O


def _hip_gan_cpu(M, S):
  import mix_stage_amd as A
  G = A.JointLateClusterSoftStyle4_G(time_steps=64, out_feats=104, num_clusters=M, style_dict={i: i for i in range(S)},
                                     style_dim=10, lambda_id=0.1, argmax=1, some_grad_flag=1, train_only=1, shape={})
  D = A.Speech2Gesture_D(in_channels=104)
  return A.GAN(G, D, criterion='L1Loss', input_modalities=['audio/log_mel_400'], update_D_prob_flag=0, no_grad=0)


def test_pickled_double_checkpoint_loads_strictly(tmp_path):
  from mix_stage_amd.checkpoint import load_weights, read_weights
  ref = O.build_gan(M=4, S=4).double()                      # the reference trains some models with .double()
  sd = {'module.' + k: v for k, v in ref.state_dict().items()}
  path = tmp_path / 'exp_1_weights.p'
  with open(path, 'wb') as f:
    pickle.dump({'model': sd}, f)
  model = _hip_gan_cpu(4, 4)
  res = load_weights(model, str(path))
  assert not res.missing_keys and not res.unexpected_keys
  own = model.state_dict()
  for k, v in ref.state_dict().items():
    assert own[k].dtype == (torch.float32 if v.is_floating_point() else v.dtype)
    assert torch.equal(own[k].double() if v.is_floating_point() else own[k], v.double().float().double() if v.is_floating_point() else v), k
  # torch.save container and a plain mapping work too
  buf = tmp_path / 'w.pt'
  torch.save(ref.state_dict(), buf)
  assert list(read_weights(str(buf))) == list(ref.state_dict())
  assert not load_weights(_hip_gan_cpu(4, 4), ref.state_dict()).missing_keys


def test_weight_files_cannot_execute_code(tmp_path):
  """A weights file is data: a pickle that references anything but containers / tensors / numpy arrays is refused."""
  import os
  import pathlib
  import pytest
  from mix_stage_amd.checkpoint import read_weights

  class Evil:
    def __reduce__(self):
      return (os.system, ('echo pwned > %s' % (tmp_path / 'pwned'),))
  bad = tmp_path / 'evil_weights.p'
  with open(bad, 'wb') as f:
    pickle.dump({'model': {'w': Evil()}}, f)
  with pytest.raises(Exception):
    read_weights(bad)
  assert not (tmp_path / 'pwned').exists()
  # nested form: an allowed outer pickle whose payload goes through torch.storage._load_from_bytes, which is an
  # unrestricted torch.load in torch 2.x -- the inner payload must be held to the same rules
  class EvilInner:
    def __reduce__(self):
      return (os.system, ('echo pwned > %s' % (tmp_path / 'pwned2'),))
  inner = io.BytesIO()
  torch.save({'w': EvilInner()}, inner)

  class Outer:
    def __reduce__(self):
      return (torch.storage._load_from_bytes, (inner.getvalue(),))
  nested = tmp_path / 'nested_weights.p'
  with open(nested, 'wb') as f:
    pickle.dump({'model': {'w': Outer()}}, f)
  with pytest.raises(Exception):
    read_weights(nested)
  assert not (tmp_path / 'pwned2').exists()
  # the legitimate use of that hook (a tensor pickled by value) still loads
  legit = tmp_path / 'bytes_weights.p'
  with open(legit, 'wb') as f:
    pickle.dump({'w': torch.arange(6.).reshape(2, 3)}, f, protocol=2)
  assert torch.equal(read_weights(legit)['w'], torch.arange(6.).reshape(2, 3))
  good = tmp_path / 'np_weights.p'
  with open(good, 'wb') as f:
    pickle.dump({'G.eye': torch.eye(2).numpy(), 'n': torch.ones(3)}, f)
  got = read_weights(pathlib.Path(good))                        # os.PathLike is accepted
  assert torch.equal(got['G.eye'], torch.eye(2)) and torch.equal(got['n'], torch.ones(3))
  with pytest.raises(FileNotFoundError):
    read_weights(str(tmp_path / 'missing.p'))


def test_save_weights_round_trip_and_refusal(tmp_path):
  """save_weights -> load_weights gives the state back; a state with a non-finite value is refused and leaves neither the target
  nor a temporary file behind; savers aimed at one path do not share a temporary name."""
  import os
  import pytest
  from mix_stage_amd.checkpoint import load_weights, save_weights
  ref = O.build_gan(M=2, S=2)
  model = _hip_gan_cpu(2, 2)
  load_weights(model, ref.state_dict())
  path = tmp_path / 'exp_2_weights.p'
  assert save_weights(model, str(path)) == str(path)
  assert sorted(os.listdir(tmp_path)) == ['exp_2_weights.p']
  back = _hip_gan_cpu(2, 2)
  res = load_weights(back, str(path))
  assert not res.missing_keys and not res.unexpected_keys
  for (k, a), (_, b) in zip(back.state_dict().items(), model.state_dict().items()):
    assert torch.equal(a, b), k
  # an existing file survives a refused save untouched
  before = path.read_bytes()
  with torch.no_grad():
    next(model.G.parameters()).view(-1)[0] = float('nan')
  with pytest.raises(RuntimeError, match='non-finite'):
    save_weights(model, str(path))
  assert path.read_bytes() == before
  assert sorted(os.listdir(tmp_path)) == ['exp_2_weights.p']
