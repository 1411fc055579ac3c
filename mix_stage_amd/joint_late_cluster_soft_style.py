"""JointLateClusterSoftStyle4_G -- the Mix-StAGE generator (reference:
src/model/joint_late_cluster_soft_style.py:17-209) on the HIP kernels.

Same constructor (kwargs['shape'] required), forward(x, y, time_steps=None, **kwargs) ->
(pose (B,T,P), [cluster CE, lambda_id*id_in, lambda_id*id_out]), attribute names (incl. labels_cap_soft,
thresh) and state_dict keys.  Internally the tensors stay channel-major (B,C,T):
  * torch.cat([x]*M, 1) of JL:190 is never materialised (ConvNormRelu.forward_broadcast),
  * the grouped 1x1 `logits` output is mixed by the softmax of the cluster scores in one kernel
    (ops.softmax_mix, JL:186-187,194),
  * the (B,T,C) <-> (B,C,T) transposes of JL:151,180,183 disappear.
"""
import contextlib

import torch
import torch.nn as nn

from . import ops, ops16
from .layers import (AudioEncoder, ClusterClassify, ConvNormRelu, Curriculum, EmbLin, Group, PoseEncoder,
                     PoseStyleEncoder, TextEncoder1D, UNet1D, bare_conv)
from .speech2gesture import Speech2Gesture_D

JointLateClusterSoftStyle4_D = Speech2Gesture_D


@contextlib.contextmanager
def some_grad(model):
  """pycasper.torchUtils.some_grad (not vendored by the reference, semantics inferred from JL:198):
  inside the context the module's parameters receive no gradient; activations still do."""
  flags = [(p, p.requires_grad) for p in model.parameters()]
  for p, _ in flags:
    p.requires_grad_(False)
  try:
    yield
  finally:
    for p, f in flags:
      p.requires_grad_(f)


class JointLateClusterSoftStyle4_G(nn.Module):
  '''
  input_shape audio:  (N, time, frequency)
  input_shape text:  (N, time, embedding_size)
  output_shape: (N, time, pose_feats)
  '''

  def __init__(self, time_steps=64, in_channels=256, out_feats=104, p=0, num_clusters=8, cluster=None,
               style_dict={}, style_dim=10, lambda_id=1, train_only=0, softmax=1, argmax=0, some_grad_flag=False,
               **kwargs):
    super().__init__()
    self.num_clusters = num_clusters
    self.audio_encoder = AudioEncoder(output_feats=time_steps, p=p)
    self.style_dict = style_dict
    self.style_dim = style_dim
    self.lambda_id = lambda_id
    self.train_only = train_only
    self.softmax = softmax
    self.argmax = argmax
    self.some_grad_flag = some_grad_flag
    self.out_feats = out_feats

    text_key = None
    for key in kwargs['shape']:
      if key in ['text/w2v', 'text/bert']:
        text_key = key
    if text_key:
      self.text_encoder = TextEncoder1D(output_feats=time_steps, input_channels=kwargs['shape'][text_key][-1], p=p)
    else:
      self.text_encoder = TextEncoder1D(output_feats=time_steps, p=p)
    self.pose_encoder = PoseEncoder(output_feats=time_steps, input_channels=out_feats, p=p)
    self.unet = UNet1D(input_channels=in_channels, output_channels=in_channels, p=p, groups=1)

    ## Style
    self.pose_style_encoder = PoseStyleEncoder(input_channels=out_feats, p=p, num_speakers=len(self.style_dict))
    self.style_emb = EmbLin(num_embeddings=len(self.style_dict), embedding_dim=self.style_dim)
    self.style_dec = nn.Sequential(*nn.ModuleList([ConvNormRelu(in_channels, in_channels, type='1d', leaky=True,
                                                                downsample=False, p=p, groups=self.style_dim)
                                                   for i in range(2)]))
    self.style_dec_gr = Group([self.style_dec], groups=self.style_dim)

    ## Content: the bank of M sub-generators as grouped blocks
    decoder_list = nn.ModuleList()
    decoder_list.append(ConvNormRelu(self.style_dim + in_channels, in_channels, type='1d', leaky=True,
                                     downsample=False, p=p, groups=self.num_clusters))
    decoder_list += nn.ModuleList([ConvNormRelu(in_channels, in_channels, type='1d', leaky=True, downsample=False,
                                                p=p, groups=self.num_clusters) for i in range(3)])
    self.decoder = nn.Sequential(*decoder_list)
    self.concat_encoder = nn.Sequential(*nn.ModuleList([ConvNormRelu(512, 256, type='1d', leaky=True,
                                                                     downsample=False, p=p)]))
    self.logits = nn.Conv1d(in_channels * self.num_clusters, out_feats * self.num_clusters, kernel_size=1, stride=1,
                            groups=self.num_clusters)
    self.classify_cluster = ClusterClassify(num_clusters=self.num_clusters, groups=1,
                                            input_channels=self.style_dim + in_channels)
    self.classify_loss = nn.CrossEntropyLoss()
    self.eye = nn.Parameter(torch.eye(self.num_clusters, self.num_clusters), requires_grad=False)
    self.smoothen = ConvNormRelu(out_feats, out_feats, type='1d', leaky=True, downsample=False, p=p)
    self.cluster = cluster

    self.thresh = Curriculum(0, 1, 1000)
    self.labels_cap_soft = None

  def index_select_outputs(self, x, labels, groups):
    """JL:106-115 (API parity: forward() uses the fused softmax + mixture kernel, ops.softmax_mix): x (B, M*P, T) mixed with
    the per-frame cluster weights labels (B, T, M) -> (B, T, P)."""
    from .layers import Group
    return Group.mix_groups(x, labels, groups)

  def forward(self, x, y, time_steps=None, **kwargs):
    internal_losses = []
    labels = x[-1]          # cluster labels ride along with the inputs (JL:119)
    x = list(x[:-1])

    # host RNG draw on every call, training or not (JL:127)
    if torch.rand(1).item() > self.thresh.step(self.training) and self.training:
      x = self.pose_encoder(y, time_steps)
    else:
      for i, modality in enumerate(kwargs['input_modalities']):
        if modality.split('/')[0] == 'text':
          x[i] = self.text_encoder(x[i], time_steps)
        if modality.split('/')[0] == 'audio':
          if x[i].dim() == 3:
            x[i] = x[i].unsqueeze(dim=1)
          x[i] = self.audio_encoder(x[i], time_steps)
      if len(x) >= 2:
        x = self.concat_encoder[0](torch.cat(tuple(x), dim=1))
      else:
        x = x[0]

    dt = getattr(self, '_ms_dt', 0)       # 16-bit modes: cb8 tensors between the conv blocks, fp32 at the boundaries
    if dt:
      x = ops16.to_cb8(x, dt)
    x = self.unet(x)                                            # (B, 256, T) channel-major throughout
    x = ops.backward_marker(x)       # (data-parallel trainers: the decoder / classifier gradients are complete here)
    B, T = x.shape[0], x.shape[2]

    ## Pose Style
    style = kwargs['style']
    pose_style_encoder_flag = not kwargs['sample_flag'] and (kwargs['description'] == 'train' or not self.train_only)
    if pose_style_encoder_flag:
      mode = 'lin'
      pose_style_score = self.pose_style_encoder(y)             # (B, S)
      style_id = style[:, 0].contiguous()                       # one copy of the strided column for both cross-entropy terms
      id_in_loss = ops.cross_entropy(pose_style_score, style_id, scale=self.lambda_id)
      if self.softmax:
        pose_style = torch.softmax(pose_style_score, dim=-1)    # per clip; the reference expands over T first,
        if self.argmax:                                         # which repeats identical rows (JL:160-165)
          pose_style = torch.argmax(pose_style, dim=-1).unsqueeze(1).expand(B, T)
          mode = 'emb'
        else:
          pose_style = pose_style.unsqueeze(1).expand(B, T, pose_style.shape[-1])
      else:
        pose_style = pose_style_score.unsqueeze(1).expand(B, T, pose_style_score.shape[-1])
    else:
      pose_style = style
      if len(style.shape) == 2:
        mode = 'emb'
      elif len(style.shape) == 3:
        mode = 'lin'
      id_in_loss = torch.zeros(1)[0]
    if mode == 'emb' and pose_style.dim() == 2 and pose_style.shape != (B, T) and pose_style.numel() == B * T:
      pose_style = pose_style.reshape(B, T)                  # windows concatenated into one long sequence (TR:779-786)
    if dt:
      x = ops16.from_cb8(x, self.unet.conv2[-1].conv.weight.shape[0])     # the style concat runs on the fp32 kernel
    if mode == 'emb' and pose_style.dim() == 2 and pose_style.shape[1] == T:
      ## content || style embedding, channel-major, one kernel (JL:175-180)
      self.pose_style_ids = pose_style
      x = ops.concat_style(x, self.style_emb.emb.weight, pose_style)      # (B, 256+style_dim, T)
    else:
      labels_style = self.style_emb(pose_style, mode=mode)     # (B, T, style_dim)
      if labels_style.shape[1] != T:
        labels_style = labels_style.view(B, -1, labels_style.shape[-1])
      x = torch.cat([x, labels_style.transpose(2, 1)], dim=1)  # (B, 256+style_dim, T)

    if dt:
      x = ops16.to_cb8(x, dt)
    ## cluster scores from content+style (JL:183-187)
    labels_score = self.classify_cluster(x)                     # (B, M, T)
    internal_losses.append(ops.cross_entropy(labels_score, labels, layout='bct'))

    ## M sub-generators on the same input, mixed by softmax(labels_score) (JL:190-194)
    chained = (ops16.decoder_chain16 if dt else ops.decoder_chain)(x, list(self.decoder), self.logits, labels_score, self.out_feats)
    if chained is not None:
      # decoder.0-3 + logits + softmax mixture as ONE launch (a workgroup carries a clip of a sub-generator through all blocks)
      x, self.labels_cap_soft = chained
    else:
      z = self.decoder[0].forward_broadcast(x)
      for m in list(self.decoder)[1:]:
        z = m(z)
      z = bare_conv(self.logits, z, out_f32=True)                 # (B, M*P, T), fp32 in every mode
      x, self.labels_cap_soft = ops.softmax_mix(z, labels_score, self.out_feats)   # (B,T,P), (B,T,M)

    if pose_style_encoder_flag:
      if self.some_grad_flag:
        with some_grad(self.pose_style_encoder):
          pose_style_score_out = self.pose_style_encoder(x)
      else:
        pose_style_score_out = self.pose_style_encoder(x)
      id_out_loss = ops.cross_entropy(pose_style_score_out, style_id, scale=self.lambda_id)
    else:
      id_out_loss = torch.zeros(1)[0]

    internal_losses.append(id_in_loss if pose_style_encoder_flag else id_in_loss * self.lambda_id)
    internal_losses.append(id_out_loss if pose_style_encoder_flag else id_out_loss * self.lambda_id)
    return x, internal_losses
