"""Variable values as a deterministic function of the variable's full name (shared by tests/golden/mini_keras.py,
which executes the reference graph, and by the tests that rebuild the same weights for the oracle / device path)."""
import zlib

import numpy as np


def value_for(name, shape):
  """Deterministic fp32 value of the variable `name`: kind by suffix, seed by crc32 of the name."""
  rng = np.random.default_rng(zlib.crc32(name.encode()))
  shape = tuple(shape)
  leaf = name.rsplit('/', 1)[-1]
  if leaf in ('kernel', 'depthwise_kernel', 'pointwise_kernel'):
    fan_in = max(1, int(np.prod(shape[:-1])) if leaf != 'depthwise_kernel' else int(shape[0] * shape[1]))
    v = rng.standard_normal(shape) / np.sqrt(fan_in)
  elif leaf == 'gamma':
    v = 1.0 + 0.1 * rng.standard_normal(shape)
  elif leaf == 'moving_variance':
    v = rng.uniform(0.5, 1.5, shape)
  elif leaf.startswith('WSM'):
    v = 1.0 + 0.3 * rng.standard_normal(shape)
  else:                     # bias, beta, moving_mean
    v = 0.1 * rng.standard_normal(shape)
  return np.asarray(v, np.float32)
