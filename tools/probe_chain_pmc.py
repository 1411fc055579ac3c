#!/usr/bin/env python3
"""GPU: the north-star launch alone -- the chained decoder (ms_decoder_chain_fwd: decoder.0-3 + logits + softmax mixture, headline
size B=32, M=8) in train mode WITH the stores a G-step needs (y_raw / y / z for the backward pass) -- repeated, for rocprofv3
--pmc passes (tools/pmc_chain.sh).  usage: probe_chain_pmc.py [fp32|bf16] [iters] [M]      (M = 4: BASELINE configs[1])"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import torch.nn as nn
from test_gpu_chain import _build, _inputs
precision = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
M = int(sys.argv[3]) if len(sys.argv) > 3 else 8
import mix_stage_amd as A
from mix_stage_amd import ops, ops16
blocks, logits = _build(M, 104, 10)
x, score = _inputs(32, M, 266)
if precision != 'fp32':
  A.set_compute_dtype(nn.ModuleList(list(blocks) + [logits]), precision)
  x = ops16.to_cb8(x, ops16.NAME_DT[precision])
for m in blocks:
  m.train(True)
x.requires_grad_(True)           # grad mode: the launch keeps what the backward pass reads
fn = ops.decoder_chain if precision == 'fp32' else ops16.decoder_chain16
for _ in range(iters):
  out = fn(x, blocks, logits, score, 104)
  assert out is not None
torch.cuda.synchronize()
print('ok', precision, iters)
