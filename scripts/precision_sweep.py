"""Error budget of the bf16 path -- CPU only, the storage-emulating oracle with SELECTIVE roundings.

Which of the roundings of the bf16 inference forward carry the 3e-3 of logit range that separates it from the fp32
oracle (north_star asks 1e-3)?  Classes of roundings (oracle/efficientdet_oracle.py):
  operand   the matrix-core operands (activations and weights rounded to bf16 on the way into every 1x1 / stem MFMA)
  exp / dw / proj / out      backbone: raw expand / depthwise / project convolution outputs, materialised block outputs
  fpn       BiFPN: fusion outputs, depthwise / pointwise outputs, resample convolutions, pooled extra levels
  tower     class / box towers: depthwise / pointwise outputs
  logits    the class / box predict outputs themselves
  image     the input image
usage: python scripts/precision_sweep.py [size] [batch]   (test infrastructure: imports oracle/)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from automl_amd import hparams_config  # noqa: E402
from oracle import efficientdet_oracle as orc  # noqa: E402
from oracle.problems import perturbed_params  # noqa: E402


def klass(key):
  if key is None:
    return 'other'
  if key == 'stem':
    return 'exp'
  if key.endswith('-predict:l3:pw') or '-predict:l' in key and key.endswith(':pw'):
    return 'logits'
  if key.startswith('class_net') or key.startswith('box_net'):
    return 'tower'
  if key.startswith('fpn_cells') or key.startswith('resample_p'):
    return 'fpn'
  for suf, k in ((':exp', 'exp'), (':dw', 'dw'), (':proj', 'proj'), (':out', 'out')):
    if key.endswith(suf):
      return k
  return 'other'


class Selective(orc.Oracle):
  """Oracle(storage='bf16') that rounds only the classes in `on`."""

  def __init__(self, on, **kw):
    orc.Oracle.__init__(self, storage='bf16', **kw)
    self.on = set(on)
    self.seen = {}

  def q(self, x, key=None, grad_key=''):
    k = klass(key)
    self.seen[k] = self.seen.get(k, 0) + 1
    return x.to(torch.bfloat16).to(torch.float32) if k in self.on else x

  def qg(self, x, grad_key=None):
    return x

  def qop(self, x):
    return x.to(torch.bfloat16).to(torch.float32) if 'operand' in self.on else x

  def forward(self, images_nhwc, training):
    self.emulate = 'image' in self.on      # (the base class rounds the image when emulating)
    x = images_nhwc
    if self.emulate:
      x = x.to(torch.bfloat16).to(torch.float32)
    self.emulate = False
    try:
      return orc.Oracle.forward(self, x, training)
    finally:
      self.emulate = True


def main():
  size = int(sys.argv[1]) if len(sys.argv) > 1 else 640
  batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
  torch.set_num_threads(min(16, os.cpu_count() or 8))
  config = hparams_config.get_efficientdet_config('efficientdet-d0')
  config.override('image_size=%d' % size)
  vals = perturbed_params(config, 7)
  rng = np.random.default_rng(31)
  images = torch.from_numpy(rng.standard_normal((batch, size, size, 3)).astype(np.float32))
  params = lambda: {k: torch.from_numpy(v.copy()) for k, v in vals.items()}
  with torch.no_grad():
    ref = orc.Oracle(config=config, params=params()).forward(images, False)
  ALL = ['image', 'operand', 'exp', 'dw', 'proj', 'out', 'fpn', 'tower', 'logits']
  rows = [('nothing rounded', [])] + [('only ' + c, [c]) for c in ALL] + [
      ('everything (the bf16 path)', ALL),
      ('all but logits', [c for c in ALL if c != 'logits']),
      ('all but logits, tower', [c for c in ALL if c not in ('logits', 'tower')]),
      ('all but logits, tower, fpn', [c for c in ALL if c not in ('logits', 'tower', 'fpn')]),
      ('all but logits, tower, fpn, out, proj', [c for c in ALL if c not in ('logits', 'tower', 'fpn', 'out', 'proj')]),
      ('operand + exp + dw (the expanded tensors only)', ['operand', 'exp', 'dw']),
      ('exp + dw', ['exp', 'dw']),
      ('operand only in the backbone = exp + dw + image, fp32 elsewhere', ['image', 'exp', 'dw']),
  ]
  print('%-70s %10s %10s' % ('rounded classes (d0 %dx%d, %d images, inference)' % (size, size, batch), 'class', 'box'))
  for name, on in rows:
    t0 = time.time()
    o = Selective(on, config=config, params=params())
    with torch.no_grad():
      got = o.forward(images, False)
    errs = []
    for g, r in zip(got, ref):
      e = [float((a - b).abs().max()) / max(float(b.abs().max()), 1e-20) for a, b in zip(g, r)]
      errs.append(max(e))
    print('%-70s %10.2e %10.2e   (%.0f s)' % (name, errs[0], errs[1], time.time() - t0), flush=True)


if __name__ == '__main__':
  main()
