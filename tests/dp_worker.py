"""One data-parallel replica of tests/test_gpu_network.py::test_two_replicas_equal_one_big_batch.

Launched twice by the test (RANK 0/1) on the SAME GPU with the gloo backend (gloo all-reduces CUDA tensors
through the host; RCCL refuses two ranks on one device).  Writes the updated weights of its replica to
<out>.rank<r>.npz."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from automl_amd import hparams_config, train_lib  # noqa: E402
from tests.test_gpu_network import make_labels, perturbed_params  # noqa: E402


def problem(batch, size):
  config = hparams_config.get_efficientdet_config('efficientdet-d0')
  config.override('clip_gradients_norm=0.0')     # clipping acts on the local gradient: not additive
  vals = perturbed_params(config, 3)
  rng = np.random.default_rng(71)
  images = rng.standard_normal((batch, size, size, 3)).astype(np.float32)
  labels = make_labels(config, batch, size, 73)
  return config, vals, images, labels


def main():
  out, sync_bn = sys.argv[1], sys.argv[2] == '1'
  rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
  dist.init_process_group('gloo', rank=rank, world_size=world)
  batch, size = 4, 128
  config, vals, images, labels = problem(batch, size)
  lo, hi = train_lib.split_global_batch(batch, world, rank)
  labels = {k: v[lo:hi] for k, v in labels.items()}
  labels['normalizer'] = 4 * 7.0 + 1.0            # the single-process run's normalizer (sum over the global batch + 1)
  net = train_lib.EfficientDetNetTrain(config=config, dtype='f32', params=vals, steps_per_epoch=10,
                                       global_batch_size=64, use_dist=True, sync_bn=sync_bn)
  vals_out = net.train_step((images[lo:hi], labels))
  torch.cuda.synchronize()
  w = net.get_weights()
  np.savez(out + '.rank%d.npz' % rank, loss=np.float64(vals_out['det_loss']), **{k.replace('/', '|'): v for k, v in w.items()})
  dist.barrier()
  dist.destroy_process_group()


if __name__ == '__main__':
  main()
