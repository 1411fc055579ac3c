"""CPU: the drop-in boundary -- the C-ABI library loads and exports every symbol include/mixstage.h declares (no
compute calls without a GPU), the host modules mirror the reference's state_dict schema, and the product path fails
loudly instead of falling back to CPU arithmetic."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
  from mix_stage_amd import _lib
  # include/mixstage.h = the drop-in boundary, include/mixstage_aux.h = measurement / test / tuning aids
  hdr = ''.join(open(os.path.join(ROOT, 'include', f)).read() for f in ('mixstage.h', 'mixstage_aux.h'))
  declared = set(re.findall(r'\b(ms_[a-z0-9_]+)\s*\(', hdr))
  public = open(os.path.join(ROOT, 'include', 'mixstage.h')).read()
  assert 'ms_debug_' not in public and 'ms_conv_block_bwd_overlap' not in public and 'ms_set_counter_buffer' not in public
  assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
  L = _lib.lib()                                   # raises if the .so is missing or a symbol is absent
  assert L.ms_abi_version() == 4
  d = _lib.ConvDesc(32, 256, 1, 64, 256, 8, 1, 3, 1, 1, 0, 1, 1, 64, _lib.MS_BN_TRAIN, _lib.MS_IN_PLAIN, 0.2, 1e-5,
                    0.1, 0)
  import ctypes
  assert L.ms_conv_block_fwd_workspace(ctypes.byref(d)) >= 256      # pure host arithmetic
  assert L.ms_conv_block_bwd_workspace(ctypes.byref(d)) >= 256
  assert L.ms_reduce_partials_count(1 << 20) >= 1


def test_state_dict_schema_matches_oracle_and_survey():
  import mix_stage_amd as A
  from oracle import mixstage_oracle as O
  G = A.JointLateClusterSoftStyle4_G(num_clusters=8, style_dict={i: i for i in range(8)}, shape={})
  D = A.Speech2Gesture_D()
  m = A.GAN(G, D, criterion='L1Loss', input_modalities=['audio/log_mel_400'])
  sd, ref = m.state_dict(), O.build_gan().state_dict()
  assert list(sd.keys()) == list(ref.keys()) and len(sd) == 409
  assert all(sd[k].shape == ref[k].shape and sd[k].dtype == ref[k].dtype for k in ref)
  m.load_state_dict(ref)                           # a reference-shaped checkpoint loads unchanged
  assert m.D_prob == 0.5 and m.G_flag is True and G.labels_cap_soft is None
  assert G.thresh.step(True) == 0 and abs(G.thresh.value - 1e-3) < 1e-12


def test_same_seed_gives_reference_initialisation():
  """Sub-modules are constructed in the reference's order with torch.nn containers, so a seeded construction
  consumes the global RNG identically (checked against the oracle, which mirrors the reference's order)."""
  import mix_stage_amd as A
  from oracle import mixstage_oracle as O
  torch.manual_seed(11212)
  a = A.Speech2Gesture_D()
  torch.manual_seed(11212)
  b = O.Speech2Gesture_D()
  for (k, v), (_, w) in zip(a.state_dict().items(), b.state_dict().items()):
    assert torch.equal(v, w), k


def test_cpu_tensors_raise_instead_of_falling_back():
  import mix_stage_amd as A
  from mix_stage_amd import _lib
  blk = A.ConvNormRelu(4, 4)
  with pytest.raises(_lib.MixStageLibError):
    blk(torch.zeros(1, 4, 8))
  with pytest.raises(NotImplementedError):
    A.GAN(torch.nn.Identity(), torch.nn.Identity(), criterion='SmoothL1Loss', input_modalities=[])


def test_product_package_never_imports_the_oracle():
  pkg = os.path.join(ROOT, 'mix_stage_amd')
  for fn in os.listdir(pkg):
    if fn.endswith('.py'):
      src = open(os.path.join(pkg, fn)).read()
      assert 'oracle' not in src.replace('"""', ''), fn


def test_group_module_matches_the_reference_semantics():
  """layers.py:593-650 (off the audio path, kept for API parity): joined inputs -> models -> per-frame soft mixture of the
  groups, or one tensor per group; JL:106-115 index_select_outputs likewise."""
  import torch
  import torch.nn as nn
  import mix_stage_amd as A
  from mix_stage_amd.layers import Group
  torch.manual_seed(0)
  g = Group([nn.Conv1d(8, 12, 1, groups=2)], groups=2, dim=1)
  xs = [torch.randn(3, 4, 5), torch.randn(3, 4, 5)]                  # (B, C, T) each, joined along the channels (dim=1)
  labels = torch.softmax(torch.randn(3, 5, 2), -1)
  out = g(xs, labels=labels, transpose=False)
  z = g.models[0](torch.cat(xs, 1))                                  # (B, 12, T)
  zz = z.transpose(2, 1).reshape(3, z.shape[2], 2, -1)
  ref = (zz * labels.reshape(3, z.shape[2], 2).unsqueeze(-1)).sum(-2).transpose(-1, -2)
  assert torch.allclose(out, ref, atol=1e-6)
  parts = g(xs, transpose=False)
  assert len(parts) == 2 and torch.equal(torch.cat(parts, 1), z)
  G = A.JointLateClusterSoftStyle4_G(time_steps=64, out_feats=6, num_clusters=2, style_dict={0: 0, 1: 1}, style_dim=10, shape={})
  x = torch.randn(3, 12, 5); lab = torch.softmax(torch.randn(3, 5, 2), -1)
  got = G.index_select_outputs(x, lab, 2)
  exp = (x.transpose(2, 1).reshape(3, 5, 2, 6) * lab.unsqueeze(-1)).sum(-2)
  assert torch.allclose(got, exp, atol=1e-6)
