"""Seeded test problems shared by tests/ and the checker legs of bench.py -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/ and bench.py's parity block import this module."""
import numpy as np

from automl_amd import netspec


def perturbed_params(config, seed):
  """Reference initialisers, then every BatchNorm vector / bias / fusion weight perturbed so that no path is trivial (with
  the initialisers alone every class logit is the -log(99) bias and every BatchNorm an identity)."""
  spec = netspec.NetSpec(config)
  vals = netspec.init_params(spec, seed)
  rng = np.random.default_rng(seed + 1)
  for p in spec.params:
    v = vals[p.name]
    if p.name.endswith('/gamma'):
      v += 0.2 * rng.standard_normal(v.shape).astype(np.float32)
    elif p.name.endswith('/beta') or p.name.endswith('/moving_mean'):
      v += 0.2 * rng.standard_normal(v.shape).astype(np.float32)
    elif p.name.endswith('/moving_variance'):
      v *= rng.uniform(0.5, 1.5, v.shape).astype(np.float32)
    elif p.name.endswith('/bias'):
      v += 0.1 * rng.standard_normal(v.shape).astype(np.float32)
    elif '/WSM' in p.name:
      v += 0.3 * rng.standard_normal(v.shape).astype(np.float32)
    vals[p.name] = v
  return vals
