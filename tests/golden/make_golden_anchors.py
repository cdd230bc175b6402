"""Generates tests/golden/reference_anchors.npz by executing the REFERENCE's own anchor code.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden_anchors.py
efficientdet/tf2/anchors.py builds the anchor grid in numpy (float64) and only converts the result with
tf.convert_to_tensor(dtype=float32) (:117-165); efficientdet/utils.py supplies get_feat_sizes /
parse_image_size (:484-526).  Both modules import TensorFlow at module scope, so a permissive stub stands in
for it: every attribute is a dummy class (good enough as a Keras base class or decorator), and
convert_to_tensor / float32 are numpy's.  No TensorFlow arithmetic runs: the boxes come out of the
reference's own numpy statements.  Stored per configuration: the full [N,4] float32 anchor array.
"""
import os
import sys
import types

import numpy as np

REF = '/root/reference/efficientdet'


class _Meta(type):
  def __getattr__(cls, name):
    if name.startswith('__'):
      raise AttributeError(name)
    return _Meta(name, (_Stub,), {})


class _Stub(metaclass=_Meta):
  def __init__(self, *a, **k):
    pass

  def __call__(self, *a, **k):
    return a[0] if a else self

  def __getattr__(self, name):
    if name.startswith('__'):
      raise AttributeError(name)
    return _Meta(name, (_Stub,), {})


def stub_module(name):
  m = types.ModuleType(name)
  m.__getattr__ = lambda attr: _Meta(attr, (_Stub,), {})
  m.__path__ = []
  return m


CONFIGS = {   # name: (min_level, max_level, num_scales, aspect_ratios, anchor_scale, image_size)
    'd0_512': (3, 7, 3, [1.0, 2.0, 0.5], 4.0, 512),
    'd0_640': (3, 7, 3, [1.0, 2.0, 0.5], 4.0, 640),
    'd7x_1536': (3, 8, 3, [1.0, 2.0, 0.5], 4.0, 1536),
    'odd_1920x1280': (3, 7, 3, [1.0, 2.0, 0.5], 4.0, '1920x1280'),
    'odd_333': (2, 6, 2, [1.0, 1.5], 3.0, 333),
}


def main():
  tf = stub_module('tensorflow')
  tf.convert_to_tensor = lambda x, dtype=None: np.asarray(x, dtype=np.float32)
  tf.float32 = np.float32
  for name in ('tensorflow', 'tensorflow.compat', 'tensorflow.compat.v1', 'tensorflow.compat.v2', 'absl',
               'absl.logging', 'tensorflow.python', 'tensorflow.python.eager', 'tensorflow.python.tpu',
               'tensorflow.python.eager.tape', 'tensorflow.python.tpu.tpu_function', 'tensorflow_addons',
               'tensorflow.python.framework', 'tensorflow.python.ops'):
    sys.modules[name] = tf if name == 'tensorflow' else stub_module(name)
  sys.path.insert(0, REF)
  sys.path.insert(0, os.path.join(REF, 'tf2'))
  import anchors as ref_anchors        # noqa: the reference module
  out = {}
  for key, (lo, hi, ns, ar, scale, size) in CONFIGS.items():
    a = ref_anchors.Anchors(lo, hi, ns, ar, scale, size)
    boxes = np.asarray(a.boxes)
    assert boxes.dtype == np.float32 and boxes.shape[1] == 4
    out[key] = boxes
    print(key, boxes.shape, boxes[0], boxes[-1])
  here = os.path.dirname(os.path.abspath(__file__))
  np.savez_compressed(os.path.join(here, 'reference_anchors.npz'), **out)


if __name__ == '__main__':
  main()
