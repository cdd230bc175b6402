"""Host-side timing of the CPU oracle under different torch thread counts (what the `-m gpu` suite spends most of its
time in).  usage: python scripts/oracle_threads.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from automl_amd import hparams_config  # noqa: E402
from oracle import efficientdet_oracle as orc  # noqa: E402
from tests.test_gpu_network import perturbed_params  # noqa: E402


def problem(model, size, batch):
  config = hparams_config.get_efficientdet_config(model)
  config.override('image_size=%d' % size)
  vals = perturbed_params(config, 3)
  rng = np.random.default_rng(1)
  images = torch.from_numpy(rng.standard_normal((batch, size, size, 3)).astype(np.float32))
  return config, vals, images


def run(config, vals, images, grad):
  o = orc.Oracle(config=config, params={k: torch.from_numpy(v.copy()) for k, v in vals.items()})
  t = time.time()
  if not grad:
    with torch.no_grad():
      o.forward(images, True)
    return time.time() - t
  with torch.no_grad():
    o.forward(images[:1, :64, :64], False)
  P = o.params()
  for n in o.trainable_names():
    P[n].requires_grad_(True)
  cls, box = o.forward(images, True)
  (sum((c**2).mean() for c in cls) + sum((b**2).mean() for b in box)).backward()
  return time.time() - t


print('cores', os.cpu_count(), 'default threads', torch.get_num_threads())
for model, size, batch, grad in (('efficientdet-d0', 128, 2, True), ('efficientdet-d0', 640, 2, True),
                                 ('efficientdet-d7x', 768, 1, False)):
  config, vals, images = problem(model, size, batch)
  for th in (0, 128, 64, 32, 16, 8):
    if th:
      torch.set_num_threads(th)
    dt = run(config, vals, images, grad)
    print('%s %d b%d %s threads=%s: %.2f s' % (model, size, batch, 'fwd+bwd' if grad else 'fwd', th or 'default', dt), flush=True)
