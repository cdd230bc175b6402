"""Generates tests/golden/reference_labels.npz by EXECUTING the reference's anchor labelling code:
tf2/anchors.AnchorLabeler (label_anchors, _unpack_labels) with object_detection/{argmax_matcher, matcher,
target_assigner, region_similarity_calculator, faster_rcnn_box_coder, box_coder, box_list, shape_utils}.py -- all
unmodified -- on the torch-backed `tf` of tests/golden/mini_keras.py plus the element-wise tensor ops below.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden_labels.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import mini_keras   # noqa
from mini_keras import T   # noqa
from make_golden_anchors import REF   # noqa


class _Ctx(object):
  def __init__(self, *a, **k):
    pass

  def __enter__(self):
    return self

  def __exit__(self, *a):
    return False


def add_tensor_ops(tf):
  def tt(x, like=None):
    if torch.is_tensor(x):
      return T(x)
    return T(torch.as_tensor(x, dtype=like.dtype if like is not None and torch.is_tensor(like) else None))

  def where(cond, x=None, y=None):
    if x is None:
      return torch.nonzero(cond)
    cond = T(cond)
    x, y = tt(x, y), tt(y, x)
    while cond.dim() < x.dim():              # TF1 tf.where: a vector condition selects rows
      cond = cond.unsqueeze(-1)
    return torch.where(cond, x, y)

  def cast(x, dtype=None):
    x = tt(x)
    return x.to(dtype) if isinstance(dtype, torch.dtype) else x

  tf.float32, tf.int32, tf.int64, tf.bool = torch.float32, torch.int32, torch.int64, torch.bool
  tf.name_scope = _Ctx
  tf.control_dependencies = _Ctx
  tf.no_op = lambda *a, **k: None
  tf.assert_equal = lambda a, b, **k: None
  tf.maximum = lambda a, b: torch.maximum(tt(a, b), tt(b, a))
  tf.minimum = lambda a, b: torch.minimum(tt(a, b), tt(b, a))
  tf.transpose = lambda x, perm=None: T(x).permute(*perm) if perm is not None else T(x).t()
  tf.where = where
  tf.shape = lambda x: list(T(x).shape)
  tf.greater = lambda a, b: tt(a, b) > tt(b, a)
  tf.greater_equal = lambda a, b: tt(a, b) >= tt(b, a)
  tf.equal = lambda a, b: tt(a, b) == tt(b, a)
  tf.not_equal = lambda a, b: tt(a, b) != tt(b, a)
  tf.logical_and = torch.logical_and
  tf.zeros = lambda shape, dtype=torch.float32: T(torch.zeros(shape if isinstance(shape, (list, tuple)) else [shape], dtype=dtype))
  tf.ones = lambda shape, dtype=torch.float32: T(torch.ones(list(shape), dtype=dtype))
  tf.zeros_like = torch.zeros_like
  tf.stack = lambda xs, axis=0, name=None: torch.stack([tt(x) for x in xs], dim=axis)
  tf.unstack = lambda x, num=None, axis=0: list(torch.unbind(T(x), dim=axis))
  tf.split = lambda value, num_or_size_splits, axis=0: list(torch.chunk(T(value), num_or_size_splits, dim=axis))
  tf.expand_dims = lambda x, axis=-1: tt(x).unsqueeze(axis)
  tf.squeeze = lambda x, axis=None: T(x).squeeze(axis[0] if isinstance(axis, (list, tuple)) else axis)
  tf.cast = cast
  tf.argmax = lambda x, axis=0, output_type=torch.int64: torch.argmax(T(x), dim=axis).to(output_type)
  tf.reduce_max = lambda x, axis=None: T(x).max(dim=axis).values if axis is not None else T(x).max()
  tf.reduce_sum = lambda x, axis=None, keepdims=False: T(x).sum() if axis is None else T(x).sum(dim=axis, keepdim=keepdims)
  tf.log = torch.log
  tf.exp = torch.exp
  tf.truediv = lambda a, b: a / b
  tf.gather = lambda params, indices, **kw: T(params)[T(indices).long()]
  tf.constant = lambda v, dtype=None: T(torch.as_tensor(v, dtype=dtype))
  tf.concat = lambda xs, axis=0: torch.cat([tt(x) for x in xs], dim=axis)
  tf.tile = lambda x, reps: T(x).repeat(*[int(r) for r in reps])
  tf.reshape = lambda x, shape, name=None: T(x).reshape([int(v) for v in shape])
  tf.range = lambda a, b=None: T(torch.arange(int(a), int(b)) if b is not None else torch.arange(int(a)))
  tf.one_hot = lambda idx, depth, **kw: T(torch.nn.functional.one_hot(T(idx).long(), int(depth)).float())
  tf.cond = lambda pred, a, b: a() if bool(pred) else b()
  tf.convert_to_tensor = lambda x, dtype=None: T(torch.as_tensor(np.asarray(x), dtype=dtype if isinstance(dtype, torch.dtype) else None))


def make_boxes(rng, n, size):
  """n groundtruth boxes (ymin, xmin, ymax, xmax) inside a size x size image, a mix of scales; labels 1..90."""
  ctr = rng.uniform(0.1, 0.9, (n, 2)) * size
  hw = np.exp(rng.uniform(np.log(0.03), np.log(0.6), (n, 2))) * size
  b = np.concatenate([ctr - hw / 2, ctr + hw / 2], 1)
  return np.clip(b, 0, size).astype(np.float32), rng.integers(1, 91, (n, 1)).astype(np.int32)


CASES = {   # name: (image_size, min_level, max_level, number of boxes, seed)
    'd0_256_8': (256, 3, 7, 8, 11),
    'd0_256_0': (256, 3, 7, 0, 12),          # an image without objects
    'd0_384_40': (384, 3, 7, 40, 13),
    'l8_320_5': (320, 3, 8, 5, 14),
    'dup_192': (192, 3, 7, 6, 15),           # duplicated and degenerate boxes: ties of the two argmax steps
}


def main():
  tf = mini_keras.build_tf()
  add_tensor_ops(tf)
  mini_keras.install(tf)
  sys.path.insert(0, REF)
  from tf2 import anchors as ref_anchors   # noqa: the reference modules
  out = {}
  for name, (size, lo, hi, nbox, seed) in CASES.items():
    rng = np.random.default_rng(seed)
    boxes, labels = make_boxes(rng, nbox, size)
    if name.startswith('dup'):
      boxes[1] = boxes[0]                       # identical boxes, different classes
      boxes[3] = [50.0, 60.0, 50.0, 90.0]       # zero height
      boxes[4] = [0.0, 0.0, float(size), float(size)]   # the whole image
    a = ref_anchors.Anchors(lo, hi, 3, [1.0, 2.0, 0.5], 4.0, size)
    labeler = ref_anchors.AnchorLabeler(a, 90)
    cls, box, npos = labeler.label_anchors(T(torch.from_numpy(boxes)), T(torch.from_numpy(labels)))
    out[name + '/gt_boxes'], out[name + '/gt_labels'] = boxes, labels
    for level in range(lo, hi + 1):
      out['%s/cls_%d' % (name, level)] = cls[level].numpy().astype(np.int32)
      out['%s/box_%d' % (name, level)] = box[level].numpy().astype(np.float32)
    out[name + '/num_positives'] = np.float32(float(npos))
    print(name, 'positives', float(npos), [tuple(cls[l].shape) for l in range(lo, hi + 1)])
  np.savez_compressed(os.path.join(HERE, 'reference_labels.npz'), **out)


if __name__ == '__main__':
  main()
