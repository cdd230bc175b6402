"""N>1 path on CPU: two gloo ranks, each with its shard of the global batch, reproduce the
reference's data-parallel semantics (clip locally -> SUM all-reduce -> identical update on every
replica) using the oracle as the per-replica model and the product's
train_lib reduce / sharding helpers."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from automl_amd import hparams_config, netspec, train_lib


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _labels(config, batch, size, seed):
  rng = np.random.default_rng(seed)
  spec = netspec.NetSpec(config)
  fs = spec.feat_sizes(size)
  out = {}
  for level in range(config.min_level, config.max_level + 1):
    h, w = fs[level]['height'], fs[level]['width']
    ct = np.full((batch, h, w, 9), -1, np.int32)
    ct[rng.random((batch, h, w, 9)) < 0.1] = 4
    bt = (rng.standard_normal((batch, h, w, 36)) * 0.1 * (np.repeat(ct, 4, -1) >= 0)).astype(np.float32)
    out['cls_targets_%d' % level] = torch.from_numpy(ct)
    out['box_targets_%d' % level] = torch.from_numpy(bt)
  out['mean_num_positives'] = torch.full((batch,), 3.0)
  return out


def _worker(rank, world, port, tmp):
  from oracle import efficientdet_oracle as orc
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  torch.set_num_threads(2)
  config = hparams_config.get_efficientdet_config('efficientdet-d0')
  size, global_batch = 64, 2
  vals = netspec.init_params(netspec.NetSpec(config), 0)
  images = torch.from_numpy(np.random.default_rng(1).standard_normal((global_batch, size, size, 3)).astype(np.float32))
  labels = _labels(config, global_batch, size, 2)
  b, e = train_lib.split_global_batch(global_batch, world, rank)
  shard = {k: v[b:e] for k, v in labels.items()}
  oracle = orc.Oracle(config=config, params={k: torch.from_numpy(v.copy()) for k, v in vals.items()})
  with torch.no_grad():
    oracle.forward(images[:1], False)
  reduce_flat = train_lib.make_grad_all_reduce()

  pre = {}

  def grad_reduce(grads):
    pre.update({k: v.detach().clone() for k, v in grads.items()})   # locally clipped, before the reduce
    names = sorted(grads)
    flat = torch.cat([grads[n].reshape(-1) for n in names])
    reduce_flat(flat)
    out, off = {}, 0
    for n in names:
      k = grads[n].numel()
      out[n] = flat[off:off + k].reshape(grads[n].shape)
      off += k
    return out

  orc.train_step(oracle, images[b:e], shard, {}, 0.05, None, grad_reduce=grad_reduce)
  torch.save({'params': {k: v.detach().clone() for k, v in oracle.params().items()}, 'local_grads': pre},
             os.path.join(tmp, 'rank%d.pt' % rank))
  dist.barrier()
  dist.destroy_process_group()


def test_two_rank_gloo_data_parallel(tmp_path):
  world = 2
  mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
  r0 = torch.load(os.path.join(str(tmp_path), 'rank0.pt'))
  r1 = torch.load(os.path.join(str(tmp_path), 'rank1.pt'))
  config = hparams_config.get_efficientdet_config('efficientdet-d0')
  vals = netspec.init_params(netspec.NetSpec(config), 0)
  moved = 0
  for name, p0 in r0['params'].items():
    if name.endswith('moving_mean') or name.endswith('moving_variance'):
      continue    # local BatchNorm statistics differ per replica by design (strategy=None semantics)
    assert torch.equal(p0, r1['params'][name]), 'replicas diverged on %s' % name
    if name in r0['local_grads']:
      # the applied update is lr * (g_rank0 + g_rank1): SUM, not MEAN
      want = torch.from_numpy(vals[name]) - 0.05 * (r0['local_grads'][name] + r1['local_grads'][name])
      assert torch.allclose(p0, want, rtol=1e-5, atol=1e-6), name
      moved += int(not torch.equal(p0, torch.from_numpy(vals[name])))
  assert moved > 400


def test_split_requires_divisibility():
  assert train_lib.split_global_batch(1024, 8, 3) == (384, 512)
  with pytest.raises(ValueError):
    train_lib.split_global_batch(10, 4, 0)


def test_lr_schedules_match_reference_formulas():
  """train_lib.py:37-173 (values re-derived by hand for batch 64, 100 steps/epoch)."""
  p = hparams_config.get_efficientdet_config('efficientdet-d0').as_dict()
  p.update(batch_size=128, steps_per_epoch=100)
  f = train_lib.learning_rate_schedule(dict(p))
  assert abs(f(0) - 0.008) < 1e-12
  assert abs(f(50) - (0.008 + 0.5 * (0.16 - 0.008))) < 1e-12
  total, warm = 300 * 100, 100
  assert abs(f(1000) - 0.5 * 0.16 * (1 + np.cos(np.pi * 1000 / (total - warm)))) < 1e-12
  p['lr_decay_method'] = 'stepwise'
  g = train_lib.learning_rate_schedule(dict(p))
  assert g(150) == 0.16 and abs(g(20000) - 0.016) < 1e-12 and abs(g(25000) - 0.0016) < 1e-12
  p['lr_decay_method'] = 'polynomial'
  h = train_lib.learning_rate_schedule(dict(p))
  assert abs(h(15000) - 0.16 * (1 - 15000 / 30000.0)**0.9) < 1e-12
  assert train_lib.ema_decay_dynamic(0.9998, 0) == 0.1 and train_lib.ema_decay_dynamic(0.9998, 10**7) == 0.9998


def _normalizer_worker(rank, world, port, tmp):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  config = hparams_config.get_efficientdet_config('efficientdet-d0')
  config.override('positives_momentum=-1.0')
  net = train_lib.EfficientDetNetTrain(config=config, use_dist=True)
  got = net._host_normalizer(10.0 + 30.0 * rank)                # rank 0: 10, rank 1: 40 -> cross-replica mean 25
  s = torch.tensor(got)
  eng = type('E', (), {'hyper': torch.zeros(4), 'set_normalizer': None})()
  net._device_normalizer(eng, torch.full((4,), 2.0 + rank))     # sum + 1 = 9 / 13 -> mean 11 -> hyper[2] = 1 / 11
  torch.save({'host': s, 'inv': eng.hyper[2].clone()}, os.path.join(tmp, 'norm%d.pt' % rank))
  dist.barrier()
  dist.destroy_process_group()


def test_negative_positives_momentum_is_the_cross_replica_mean(tmp_path):
  """config.positives_momentum < 0 (tf2/train_lib.py:532-533): the normalizer sum(mean_num_positives) + 1 becomes its
  mean over the replicas (utils.cross_replica_mean) -- host path (a caller-supplied float) and device path (the tensor
  the graph step reads), two gloo ranks on the CPU."""
  world = 2
  mp.spawn(_normalizer_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
  for r in range(world):
    d = torch.load(os.path.join(str(tmp_path), 'norm%d.pt' % r))
    assert abs(float(d['host']) - 25.0) < 1e-6, d
    assert abs(float(d['inv']) - 1.0 / 11.0) < 1e-7, d


def test_overlapped_reduce_is_refused_with_the_global_norm_clip():
  """The bucketed all-reduce under the backward pass is only legal without the global-norm clip (the reference clips the
  LOCAL gradient by its global norm before apply_gradients reduces it, train_lib.py:675-683)."""
  config = hparams_config.get_efficientdet_config('efficientdet-d0')
  assert config.clip_gradients_norm == 10.0
  with pytest.raises(ValueError, match='clip_gradients_norm=0'):
    train_lib.EfficientDetNetTrain(config=config, overlap_grad_reduce=True)
  config.clip_gradients_norm = 0
  net = train_lib.EfficientDetNetTrain(config=config, overlap_grad_reduce=True)
  assert net.overlap_grad_reduce
