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


def make_labels(min_level, max_level, num_classes, num_anchors, batch, image_size, seed):
  """Dense label maps in the dataloader's layout (dataloader.py:365-394): cls_targets_l int32 [B,H,W,A] with -1
  background / -2 ignore, box_targets_l [B,H,W,4A] (0 where unmatched), mean_num_positives [B]."""
  rng = np.random.default_rng(seed)
  labels = {}
  h = image_size
  for level in range(1, max_level + 1):
    h = (h - 1) // 2 + 1
    if level < min_level:
      continue
    ct = np.full((batch, h, h, num_anchors), -1, np.int32)
    r = rng.random((batch, h, h, num_anchors))
    ct[r < 0.05] = rng.integers(0, num_classes, int((r < 0.05).sum()))
    ct[(r >= 0.05) & (r < 0.08)] = -2
    bt = np.zeros((batch, h, h, num_anchors, 4), np.float32)
    pos = ct >= 0
    bt[pos] = rng.standard_normal((int(pos.sum()), 4)).astype(np.float32) * 0.2
    labels['cls_targets_%d' % level] = ct
    labels['box_targets_%d' % level] = bt.reshape(batch, h, h, num_anchors * 4)
  labels['mean_num_positives'] = np.full((batch,), 7.0, np.float32)
  return labels
