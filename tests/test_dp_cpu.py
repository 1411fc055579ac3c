"""CPU, world_size 2 over gloo: the data-parallel pieces of the train step that do not need a GPU --
the flat-gradient all-reduce (mean over ranks) and the rank-consistent host decisions."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def _worker(rank, world, port, out):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    from mix_stage_amd.train_step import average_flat_gradients, broadcast_from_rank0, peek_step_decisions
    w = torch.full((1000,), float(rank + 1))
    broadcast_from_rank0([w])
    assert torch.equal(w, torch.ones(1000))              # every rank starts from rank 0's weights
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(100003, generator=g)               # each rank: gradients of its own shard of clips
    mine = flat.clone()
    average_flat_gradients(flat)
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    expect = sum(gathered) / world
    ok_mean = torch.allclose(flat, expect, atol=1e-6)
    # the 16-bit exchange: the mean of the bf16-rounded gradients, identical on every rank
    flat16 = mine.clone()
    wire = torch.empty(flat16.numel(), dtype=torch.bfloat16)
    average_flat_gradients(flat16, wire=wire)
    expect16 = sum(t.to(torch.bfloat16).float() for t in gathered) / world
    both = [torch.zeros_like(flat16) for _ in range(world)]
    dist.all_gather(both, flat16)
    ok_mean = ok_mean and torch.allclose(flat16, expect16, rtol=2 ** -7, atol=1e-6) and torch.equal(both[0], both[1])
    # identical host seeds -> identical D/G and curriculum decisions on every rank, RNG left untouched by the peek
    torch.manual_seed(4321)
    kinds = []
    for i in range(32):
      before = torch.get_rng_state()
      k, pose_branch = peek_step_decisions(0.5, min(i, 10) / 10.0, i, 10, 1.0)
      assert torch.equal(before, torch.get_rng_state())
      torch.rand(1); torch.rand(1)
      kinds.append((k, pose_branch))
    enc = torch.tensor([(1 if k == 'G' else 0) * 2 + int(pb) for k, pb in kinds])
    allk = [torch.zeros_like(enc) for _ in range(world)]
    dist.all_gather(allk, enc)
    ok_kinds = all(torch.equal(allk[0], a) for a in allk)
    if rank == 0:
      out.put((ok_mean, ok_kinds, len(set(k for k, _ in kinds))))
  finally:
    dist.destroy_process_group()


def test_flat_gradient_allreduce_and_decisions_world2():
  ctx = mp.get_context('spawn')
  out = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
  for p in procs:
    p.start()
  for p in procs:
    p.join(120)
    assert p.exitcode == 0
  ok_mean, ok_kinds, n_kinds = out.get(timeout=10)
  assert ok_mean and ok_kinds and n_kinds == 2


def test_single_process_is_identity():
  from mix_stage_amd.train_step import average_flat_gradients
  t = torch.arange(8.0)
  assert average_flat_gradients(t.clone()).equal(t)
