"""Generates tests/golden/reference_losses.npz by EXECUTING the reference's own loss / schedule code.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden_losses.py
efficientdet/tf2/train_lib.py is imported under the permissive TensorFlow stub of make_golden_anchors.py, with the
handful of elementary TensorFlow functions its loss and learning-rate code calls replaced by their documented numpy
equivalents (the table `NUMPY_TF` below: sigmoid, sigmoid_cross_entropy_with_logits = max(x,0) - x*z +
log(1+exp(-|x|)), one_hot with all-zero rows for negative ids, Keras Huber = 0.5 e^2 if |e| <= d else d|e| - 0.5 d^2
averaged over the last axis, cast / reshape / reduce_sum / where / cos / pow ...).  What runs unmodified is the
reference's COMPOSITION: FocalLoss.call (:380-406), BoxLoss.call (:423-437), EfficientDetNetTrain._detection_loss
(:493-604: per-level one-hot targets, the -2 ignore mask, the sum(mean_num_positives)+1 normalizer, box_loss_weight)
and the three learning-rate schedules (:50-173).  Stored: the seeded inputs and the resulting losses / rates.
"""
import math
import os
import sys
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden_anchors import REF, stub_module   # noqa


class Arr(np.ndarray):
  """ndarray with the two TensorFlow tensor methods the reference calls."""

  def get_shape(self):
    shape = self.shape
    return types.SimpleNamespace(as_list=lambda: list(shape))


def A(x, dtype=None):
  return np.asarray(x, dtype=dtype).view(Arr)


def one_hot(ids, depth, dtype=np.float32):
  ids = np.asarray(ids)
  out = np.zeros(ids.shape + (depth,), dtype)
  ok = (ids >= 0) & (ids < depth)
  out[ok, ids[ok]] = 1
  return A(out)


class Huber(object):
  def __init__(self, delta, reduction=None):
    self.delta = delta

  def __call__(self, y_true, y_pred):
    e = np.asarray(y_pred, np.float64) - np.asarray(y_true, np.float64)
    a = np.abs(e)
    v = np.where(a <= self.delta, 0.5 * e * e, self.delta * a - 0.5 * self.delta**2)
    return A(v.mean(-1).astype(np.float32))


class Loss(object):
  def __init__(self, **kwargs):
    self.kwargs = kwargs

  def __call__(self, y_true, y_pred):
    return self.call(y_true, y_pred)          # the reference builds its losses with Reduction.NONE


def build_tf():
  tf = stub_module('tensorflow')
  f32 = np.float32
  tf.float32 = f32
  tf.convert_to_tensor = lambda x, dtype=None: A(x, dtype)
  tf.cast = lambda x, dtype=None: A(np.asarray(x).astype(dtype))
  tf.sigmoid = lambda x: A(1.0 / (1.0 + np.exp(-np.asarray(x, np.float64)))).astype(np.asarray(x).dtype)
  def reduce_sum(x, axis=None, **k):
    return A(np.sum(np.asarray(x, np.float64), axis=axis)).astype(f32)
  tf.reduce_sum = reduce_sum
  tf.stack = lambda xs, axis=0: A(np.stack([np.asarray(x) for x in xs], axis=axis))

  class _Scope(object):
    def __enter__(self):
      return self

    def __exit__(self, *a):
      return False
  tf.name_scope = lambda *a, **k: _Scope()
  tf.reshape = lambda x, shape: A(np.reshape(x, shape))
  tf.expand_dims = lambda x, axis=-1: A(np.expand_dims(x, axis))
  tf.not_equal = lambda a, b: A(np.not_equal(a, b))
  tf.one_hot = lambda ids, depth, dtype=f32: one_hot(ids, depth, dtype)
  tf.add_n = lambda xs: A(sum(np.asarray(x, np.float64) for x in xs)).astype(f32)
  tf.where = lambda c, a, b: A(np.where(c, a, b))
  tf.cos = lambda x: A(np.cos(x))
  tf.pow = lambda x, p: A(np.power(x, p))
  nn = types.SimpleNamespace()
  nn.sigmoid_cross_entropy_with_logits = lambda labels, logits: A(
      (np.maximum(np.asarray(logits, np.float64), 0) - np.asarray(logits, np.float64) * np.asarray(labels, np.float64) +
       np.log1p(np.exp(-np.abs(np.asarray(logits, np.float64))))).astype(np.asarray(logits).dtype))
  nn.relu = lambda x: A(np.maximum(np.asarray(x), 0))

  def softmax(x, axis=-1):
    x = np.asarray(x, np.float64)
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    return A((e / e.sum(axis=axis, keepdims=True)).astype(np.float32))
  nn.softmax = softmax
  tf.nn = nn
  tf.unstack = lambda x, num=None, axis=-1: [A(v) for v in np.moveaxis(np.asarray(x), axis, 0)]
  tf.math = types.SimpleNamespace(exp=lambda x: A(np.exp(np.asarray(x))))
  tf.floor = lambda x: A(np.floor(np.asarray(x)))
  tf.shape = lambda x: np.asarray(np.asarray(x).shape)
  tf.autograph = types.SimpleNamespace(experimental=types.SimpleNamespace(do_not_convert=lambda f: f))
  keras = stub_module('tensorflow.keras')
  keras.losses = types.SimpleNamespace(Loss=Loss, Huber=Huber, Reduction=types.SimpleNamespace(NONE='none'))
  tf.keras = keras
  tf.compat = types.SimpleNamespace(v1=tf, v2=tf)      # `import tensorflow.compat.v1 as tf` in efficientdet/utils.py
  return tf


def main():
  tf = build_tf()
  names = ['tensorflow', 'tensorflow.compat', 'tensorflow.compat.v1', 'tensorflow.compat.v2', 'absl', 'absl.logging',
           'absl.flags', 'tensorflow.python', 'tensorflow.python.eager', 'tensorflow.python.tpu',
           'tensorflow.python.eager.tape', 'tensorflow.python.tpu.tpu_function', 'tensorflow_addons',
           'tensorflow.python.framework', 'tensorflow.python.ops', 'neural_structured_learning', 'tensorflow_hub',
           'coco_metric', 'inference', 'PIL', 'PIL.Image', 'pycocotools', 'tensorflow_model_optimization']
  for n in names:
    # efficientdet/utils.py does `import tensorflow.compat.v1 as tf`: the same numpy-backed module
    sys.modules[n] = tf if n in ('tensorflow', 'tensorflow.compat.v1', 'tensorflow.compat.v2') else stub_module(n)
  sys.path.insert(0, REF)
  from tf2 import train_lib as ref          # noqa: the reference module

  out = {}
  # ---- learning-rate schedules on a grid of steps
  steps = np.array([0, 1, 5, 9, 10, 11, 50, 99, 100, 150, 199, 200, 249, 250, 299], np.int64)
  out['lr_steps'] = steps
  for method in ('stepwise', 'cosine', 'polynomial'):
    params = dict(learning_rate=0.08, batch_size=128, steps_per_epoch=1, lr_warmup_epoch=10.0, lr_warmup_init=0.008,
                  first_lr_drop_epoch=200.0, second_lr_drop_epoch=250.0, num_epochs=300, poly_lr_power=0.9,
                  lr_decay_method=method)
    sched = ref.learning_rate_schedule(params)
    out['lr_' + method] = np.array([float(np.asarray(sched(int(s)))) for s in steps], np.float64)
  # ---- detection loss on seeded inputs (3 levels, 5 classes, 2 anchors, with ignore and background labels)
  rng = np.random.default_rng(2024)
  cfg = types.SimpleNamespace(min_level=3, num_classes=5, data_format='channels_last', box_loss_weight=50.0,
                              iou_loss_type=None, iou_loss_weight=1.0, positives_momentum=None)
  fake = types.SimpleNamespace(config=cfg, loss={ref.FocalLoss.__name__: ref.FocalLoss(0.25, 1.5, label_smoothing=0.0),
                                                   ref.BoxLoss.__name__: ref.BoxLoss(0.1)})
  na, nc, batch = 2, 5, 3
  cls_outputs, box_outputs, labels = [], [], {}
  for li, hw in enumerate((4, 2, 1)):
    level = 3 + li
    logits = (rng.standard_normal((batch, hw, hw, na * nc)) * 2.0 - 1.0).astype(np.float32)
    boxes = (rng.standard_normal((batch, hw, hw, na * 4)) * 0.3).astype(np.float32)
    ct = rng.integers(-2, nc, (batch, hw, hw, na)).astype(np.int32)
    bt = (rng.standard_normal((batch, hw, hw, na * 4)) * 0.2).astype(np.float32)
    bt[np.repeat(ct < 0, 4, axis=-1)] = 0.0
    cls_outputs.append(A(logits)); box_outputs.append(A(boxes))
    labels['cls_targets_%d' % level] = A(ct)
    labels['box_targets_%d' % level] = A(bt)
    out['logits_%d' % level], out['boxes_%d' % level] = logits, boxes
    out['cls_targets_%d' % level], out['box_targets_%d' % level] = ct, bt
  labels['mean_num_positives'] = A(np.array([[2.5], [4.0], [1.5]], np.float32))
  out['mean_num_positives'] = np.asarray(labels['mean_num_positives'])
  loss_vals = {}
  total = ref.EfficientDetNetTrain._detection_loss(fake, cls_outputs, box_outputs, labels, loss_vals)
  out['det_loss'] = np.float64(total)
  out['cls_loss'] = np.float64(loss_vals['cls_loss'])
  out['box_loss'] = np.float64(loss_vals['box_loss'])
  # focal loss with label smoothing, elementwise
  fl = ref.FocalLoss(0.25, 2.0, label_smoothing=0.1)
  yt = one_hot(rng.integers(-1, 4, (6,)), 4)
  yp = A((rng.standard_normal((6, 4)) * 3).astype(np.float32))
  out['fl_targets'], out['fl_logits'] = np.asarray(yt), np.asarray(yp)
  out['fl_values'] = np.asarray(fl.call([np.float32(7.0), yt], yp), np.float64)
  # ---- BiFPN fusion: FNode.fuse_features (efficientdet_keras.py:75-121), all five weight methods, NHWC nodes
  from tf2 import efficientdet_keras as ref_keras   # noqa
  nodes = [A(rng.standard_normal((2, 3, 4, 8)).astype(np.float32)) for _ in range(3)]
  out['fuse_nodes'] = np.stack([np.asarray(n) for n in nodes])
  scal = [A(np.float32(v)) for v in (0.7, -0.2, 1.3)]
  vec = [A((rng.standard_normal(8) * 0.8 + 0.5).astype(np.float32)) for _ in range(3)]
  out['fuse_scalars'] = np.array([float(v) for v in scal], np.float32)
  out['fuse_vectors'] = np.stack([np.asarray(v) for v in vec])
  for method in ('attn', 'fastattn', 'channel_attn', 'channel_fastattn', 'sum'):
    fake_node = types.SimpleNamespace(weight_method=method, vars=vec if method.startswith('channel_') else scal)
    out['fuse_' + method] = np.asarray(ref_keras.FNode.fuse_features(fake_node, list(nodes)), np.float32)
  # ---- box decoding (tf2/anchors.py:30-58) on the first anchors of d0-512 and random codes
  from tf2 import anchors as ref_anchors   # noqa
  anc = np.asarray(ref_anchors.Anchors(3, 7, 3, [1.0, 2.0, 0.5], 4.0, 512).boxes)[::97][:64]
  codes = (rng.standard_normal((2, anc.shape[0], 4)) * 0.4).astype(np.float32)
  out['decode_anchors'], out['decode_codes'] = anc, codes
  out['decode_boxes'] = np.asarray(ref_anchors.decode_box_outputs(A(codes), A(anc)), np.float32)
  # ---- stochastic depth (utils.drop_connect, utils.py:329-344) with the uniform draws supplied
  import utils as ref_utils   # noqa
  u = rng.random((5, 1, 1, 1)).astype(np.float32)
  x = rng.standard_normal((5, 2, 2, 3)).astype(np.float32)
  tf.random = types.SimpleNamespace(uniform=lambda shape, dtype=None: A(u))
  out['drop_u'], out['drop_x'] = u, x
  out['drop_out'] = np.asarray(ref_utils.drop_connect(A(x), True, 0.8), np.float32)
  assert np.array_equal(np.asarray(ref_utils.drop_connect(A(x), False, 0.8)), x)
  here = os.path.dirname(os.path.abspath(__file__))
  np.savez_compressed(os.path.join(here, 'reference_losses.npz'), **out)
  print({k: out[k] for k in ('det_loss', 'cls_loss', 'box_loss')}, out['lr_cosine'][:4])


if __name__ == '__main__':
  main()
