#!/usr/bin/env python3
"""Times the chained decoder launch (ms_decoder_chain_fwd) against the blocks one by one at the headline size, train and eval:
N launches captured in a HIP graph, replayed.  usage: python tools/probe_chain.py [B] [M]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch

from test_gpu_chain import _build, _inputs


def timed(fn, n=20):
  cs = torch.cuda.Stream()
  cs.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(cs), torch.no_grad():
    fn()
  torch.cuda.current_stream().wait_stream(cs)
  torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g, stream=cs), torch.no_grad():
    for _ in range(n):
      fn()
  for _ in range(3):
    g.replay()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(5):
    g.replay()
  b.record()
  torch.cuda.synchronize()
  return a.elapsed_time(b) * 1e3 / (5 * n)


def main():
  from mix_stage_amd import ops
  from mix_stage_amd.layers import bare_conv
  B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
  M = int(sys.argv[2]) if len(sys.argv) > 2 else 8
  dtn = sys.argv[3] if len(sys.argv) > 3 else 'fp32'
  P = 104
  blocks, logits = _build(M, P, 10)
  x, score = _inputs(B, M, 266)
  if dtn != 'fp32':
    import torch.nn as nn
    import mix_stage_amd as A
    from mix_stage_amd import ops16
    A.set_compute_dtype(nn.ModuleList(list(blocks) + [logits]), dtn)
    x = ops16.to_cb8(x, ops16.NAME_DT[dtn])
  gflop = 2.0 * B * 64 * M * (256 * 3 * (266 + 3 * 256) + P * 256) / 1e9
  for train in (True, False):
    for m in blocks:
      m.train(train)

    def chain():
      return (ops.decoder_chain if dtn == 'fp32' else ops16.decoder_chain16)(x, blocks, logits, score, P)

    def one_by_one():
      z = blocks[0].forward_broadcast(x)
      for m in blocks[1:]:
        z = m(z)
      z = bare_conv(logits, z, out_f32=True)
      return ops.softmax_mix(z, score, P)

    t_c = timed(chain)
    ops.USE_DECODER_CHAIN = False
    t_b = timed(one_by_one)
    ops.USE_DECODER_CHAIN = True
    peak = 157.3 if dtn == 'fp32' else 2500.0
    print('%s %s B=%d M=%d: chain %.1f us = %.1f TF (%.3f of peak); blocks one by one %.1f us = %.1f TF' %
          (dtn, 'train' if train else 'eval', B, M, t_c, gflop / t_c * 1e3, gflop / t_c * 1e3 / peak, t_b, gflop / t_b * 1e3))


if __name__ == '__main__':
  main()
