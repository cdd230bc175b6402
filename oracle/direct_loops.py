"""Independent direct-loop numpy twin of the oracle's TensorFlow-semantics ops -- TEST INFRASTRUCTURE.

Written from the op definitions (not from the torch calls in efficientdet_oracle.py) so that the
fast oracle can be validated on tiny shapes: TF 'SAME' conv / depthwise conv (asymmetric padding),
3x3/s2 max-pool that ignores padding, nearest-neighbour resize, training BatchNorm, fast-attention
fusion.  NHWC float64 throughout.  Reference semantics: SURVEY.md section 8(c) items (1)-(6).
"""
import math

import numpy as np


def _same(in_size, k, s):
  out = int(math.ceil(in_size / s))
  total = max((out - 1) * s + k - in_size, 0)
  return out, total // 2


def conv2d_same(x, w, stride):
  """x [n,h,w,cin], w [kh,kw,cin,cout] -> [n,oh,ow,cout]."""
  n, h, wd, cin = x.shape
  kh, kw, _, cout = w.shape
  oh, pt = _same(h, kh, stride)
  ow, pl = _same(wd, kw, stride)
  y = np.zeros((n, oh, ow, cout), np.float64)
  for b in range(n):
    for i in range(oh):
      for j in range(ow):
        for a in range(kh):
          for c in range(kw):
            yy, xx = i * stride - pt + a, j * stride - pl + c
            if 0 <= yy < h and 0 <= xx < wd:
              y[b, i, j] += x[b, yy, xx].astype(np.float64) @ w[a, c].astype(np.float64)
  return y


def depthwise_same(x, w, stride):
  """x [n,h,w,c], w [kh,kw,c] -> [n,oh,ow,c]."""
  n, h, wd, ch = x.shape
  kh, kw, _ = w.shape
  oh, pt = _same(h, kh, stride)
  ow, pl = _same(wd, kw, stride)
  y = np.zeros((n, oh, ow, ch), np.float64)
  for i in range(oh):
    for j in range(ow):
      for a in range(kh):
        for c in range(kw):
          yy, xx = i * stride - pt + a, j * stride - pl + c
          if 0 <= yy < h and 0 <= xx < wd:
            y[:, i, j] += x[:, yy, xx].astype(np.float64) * w[a, c].astype(np.float64)
  return y


def max_pool_3x3_s2_same(x):
  n, h, wd, ch = x.shape
  oh, pt = _same(h, 3, 2)
  ow, pl = _same(wd, 3, 2)
  y = np.full((n, oh, ow, ch), -np.inf)
  for i in range(oh):
    for j in range(ow):
      for a in range(3):
        for c in range(3):
          yy, xx = i * 2 - pt + a, j * 2 - pl + c
          if 0 <= yy < h and 0 <= xx < wd:
            y[:, i, j] = np.maximum(y[:, i, j], x[:, yy, xx])
  return y


def resize_nearest(x, oh, ow):
  n, h, wd, ch = x.shape
  y = np.zeros((n, oh, ow, ch), x.dtype)
  for i in range(oh):
    for j in range(ow):
      y[:, i, j] = x[:, min(int(math.floor(i * (h / oh))), h - 1), min(int(math.floor(j * (wd / ow))), wd - 1)]
  return y


def batch_norm_train(x, gamma, beta, eps=1e-3):
  x = x.astype(np.float64)
  mean = x.mean((0, 1, 2))
  var = ((x - mean)**2).mean((0, 1, 2))
  return (x - mean) / np.sqrt(var + eps) * gamma + beta, mean, var


def swish(x):
  return x / (1.0 + np.exp(-x))


def fast_attention(nodes, weights):
  r = [max(w, 0.0) for w in weights]
  s = sum(r) + 0.0001
  return sum(n * ri / s for n, ri in zip(nodes, r))
