"""A minimal stand-in for the TensorFlow / Keras API surface that the reference's GRAPH code touches, so that
efficientdet/tf2/efficientdet_keras.py and backbone/efficientnet_model.py can be EXECUTED unmodified
(tests/golden/make_golden_graph.py).  TEST INFRASTRUCTURE for generating golden tensors in the build container.

What is real: every line of the reference's layer wiring (which conv feeds which BatchNorm, where activations,
squeeze-excite, residuals, resampling, fusion and the shared / per-level head layers sit, and the names the
variables get through Keras-style name scopes).  What is stood in: the elementary layers (Conv2D,
DepthwiseConv2D, SeparableConv2D, BatchNormalization, MaxPooling2D, nearest resize), implemented with the oracle's
TensorFlow-semantics helpers (oracle/efficientdet_oracle.py: 'SAME' padding, pooling that ignores padding, nearest
resize, biased batch variance), which tests/test_oracle_twin.py checks against direct loops.

Variables are not drawn from the reference initialisers: each one is a deterministic function of its FULL NAME and
shape (`value_for`), so a consumer that reproduces the names reproduces the weights without any file.
"""
import sys
import types

import numpy as np
import torch

from make_golden_anchors import stub_module   # noqa (same directory)


from name_values import value_for   # noqa (same directory)


SCOPE = []          # current Keras name-scope stack
TRAINABLE = []      # trainable variables in creation order (one model per process)
GRAD = [False]      # True: trainable variables require grad (make_golden_trainstep.py)
DRAWS = []          # tf.random.uniform results, in call order
DRAW_RNG = [np.random.default_rng(1234)]
TRAINING = []       # `training` of the enclosing layer calls
VARIABLES = {}      # full name -> torch tensor, in creation order


class Shape(tuple):
  """torch.Size with TensorFlow's as_list()."""

  def as_list(self):
    return list(self)

  def is_fully_defined(self):
    return True

  @property
  def ndims(self):
    return len(self)


class KT(torch.Tensor):
  """torch.Tensor whose .shape answers as_list() (the reference reads static shapes that way); results of torch
  operations on a KT are KTs again (default __torch_function__)."""

  @property
  def shape(self):
    return Shape(super().shape)

  def get_shape(self):
    return self.shape

  def set_shape(self, shape):
    assert list(shape) == list(self.shape), (shape, self.shape)

  @property
  def name(self):            # variables: '<scope path>/<leaf>:0'
    return getattr(self, '_vname', None)

  @name.setter
  def name(self, v):
    self._vname = v


def T(x):
  t = x if torch.is_tensor(x) else torch.as_tensor(np.asarray(x))
  return t if isinstance(t, KT) else t.as_subclass(KT)


class Layer(object):
  def __init__(self, name=None, **kwargs):
    self.name = name if name is not None else type(self).__name__.lower()
    self.built = False
    self._weights = []

  def add_weight(self, name=None, shape=None, initializer=None, trainable=True, dtype=None, **kwargs):
    shape = () if shape is None else tuple(np.atleast_1d(shape).tolist()) if not isinstance(shape, tuple) else shape
    full = '/'.join([s for s in SCOPE if s] + [name])
    assert full not in VARIABLES, 'variable created twice: ' + full
    t = T(torch.from_numpy(value_for(full, shape)))
    t.name = full + ':0'
    if trainable:
      TRAINABLE.append(t)
      t.requires_grad_(bool(GRAD[0]))
    VARIABLES[full] = t
    self._weights.append(t)
    return t

  def build(self, input_shape):
    pass

  def __call__(self, inputs, *args, **kwargs):
    # Keras propagates `training` from the enclosing layer call when a sub-layer is called without it
    import inspect
    params = list(inspect.signature(self.call).parameters)
    explicit = None
    if 'training' in params:
      pos = params.index('training') - 1          # position among *args (inputs is the first parameter)
      if 'training' in kwargs:
        explicit = kwargs['training']
      elif 0 <= pos < len(args):
        explicit = args[pos]
      else:
        kwargs['training'] = TRAINING[-1] if TRAINING else None
        explicit = kwargs['training']
    TRAINING.append(explicit if explicit is not None else (TRAINING[-1] if TRAINING else None))
    SCOPE.append(self.name)
    try:
      if torch.is_tensor(inputs):
        inputs = T(inputs)
      if not self.built:
        shapes = [Shape(i.shape) for i in inputs] if isinstance(inputs, (list, tuple)) else Shape(inputs.shape)
        self.build(shapes)
        self.built = True
      return self.call(inputs, *args, **kwargs)
    finally:
      SCOPE.pop()
      TRAINING.pop()


class Model(Layer):
  @property
  def trainable_variables(self):
    return list(TRAINABLE)


def _orc():
  from oracle import efficientdet_oracle as orc
  return orc


def _nchw(x):
  return x.permute(0, 3, 1, 2)


def _nhwc(x):
  return x.permute(0, 2, 3, 1).contiguous()


def _pair(v):
  return (v, v) if isinstance(v, int) else tuple(v)


class Conv2D(Layer):
  def __init__(self, filters, kernel_size, strides=1, padding='valid', data_format='channels_last', use_bias=True,
               name=None, **kwargs):
    super().__init__(name=name or 'conv2d')
    assert padding.lower() == 'same' and data_format == 'channels_last'
    self.filters, self.k, self.s, self.use_bias = filters, _pair(kernel_size), _pair(strides), use_bias

  def build(self, shape):
    self.kernel = self.add_weight('kernel', (self.k[0], self.k[1], shape[-1], self.filters))
    self.bias = self.add_weight('bias', (self.filters,)) if self.use_bias else None

  def call(self, x):
    assert self.k[0] == self.k[1] and self.s[0] == self.s[1]
    return _nhwc(_orc().conv2d_same(_nchw(x), self.kernel, self.s[0], self.bias))


class DepthwiseConv2D(Layer):
  def __init__(self, kernel_size, strides=1, padding='valid', data_format='channels_last', use_bias=True, name=None,
               **kwargs):
    super().__init__(name=name or 'depthwise_conv2d')
    assert padding.lower() == 'same' and not use_bias
    self.k, self.s = _pair(kernel_size), _pair(strides)

  def build(self, shape):
    self.kernel = self.add_weight('depthwise_kernel', (self.k[0], self.k[1], shape[-1], 1))

  def call(self, x):
    return _nhwc(_orc().depthwise_same(_nchw(x), self.kernel, self.s[0]))


class SeparableConv2D(Layer):
  def __init__(self, filters, kernel_size, padding='valid', data_format='channels_last', use_bias=True,
               depth_multiplier=1, name=None, **kwargs):
    super().__init__(name=name or 'separable_conv2d')
    assert padding.lower() == 'same' and depth_multiplier == 1
    self.filters, self.k, self.use_bias = filters, _pair(kernel_size), use_bias

  def build(self, shape):
    self.dw = self.add_weight('depthwise_kernel', (self.k[0], self.k[1], shape[-1], 1))
    self.pw = self.add_weight('pointwise_kernel', (1, 1, shape[-1], self.filters))
    self.bias = self.add_weight('bias', (self.filters,)) if self.use_bias else None

  def call(self, x):
    orc = _orc()
    return _nhwc(orc.conv2d_same(orc.depthwise_same(_nchw(x), self.dw, 1), self.pw, 1, self.bias))


class BatchNormalization(Layer):
  def __init__(self, axis=-1, momentum=0.99, epsilon=1e-3, center=True, scale=True, name=None, **kwargs):
    super().__init__(name=name or 'batch_normalization')
    assert axis in (-1, 3) and center and scale
    self.momentum, self.epsilon = momentum, epsilon
    self.updates = []

  def build(self, shape):
    c = shape[-1]
    self.gamma = self.add_weight('gamma', (c,))
    self.beta = self.add_weight('beta', (c,))
    self.moving_mean = self.add_weight('moving_mean', (c,), trainable=False)
    self.moving_variance = self.add_weight('moving_variance', (c,), trainable=False)

  def call(self, x, training=None):
    if training:
      mean = x.mean(dim=(0, 1, 2))
      var = x.var(dim=(0, 1, 2), unbiased=False)
    else:
      mean, var = self.moving_mean, self.moving_variance
    inv = torch.rsqrt(var + self.epsilon) * self.gamma
    return x * inv + (self.beta - mean * inv)


class Dense(Layer):
  def __init__(self, units, use_bias=True, name=None, **kwargs):
    super().__init__(name=name or 'dense')
    self.units, self.use_bias = units, use_bias

  def build(self, shape):
    self.kernel = self.add_weight('kernel', (shape[-1], self.units))
    self.bias = self.add_weight('bias', (self.units,)) if self.use_bias else None

  def call(self, x):
    y = x @ self.kernel
    return y + self.bias if self.use_bias else y


class GlobalAveragePooling2D(Layer):
  def __init__(self, data_format='channels_last', name=None, **kwargs):
    super().__init__(name=name or 'global_average_pooling2d')
    assert data_format == 'channels_last'

  def call(self, x):
    return x.mean(dim=(1, 2))


class Dropout(Layer):
  def __init__(self, rate, name=None, **kwargs):
    super().__init__(name=name or 'dropout')
    self.rate = rate

  def call(self, x, training=None):
    assert not (training and self.rate), 'random dropout is not reproducible: generate with dropout_rate=0'
    return x


class MaxPooling2D(Layer):
  def __init__(self, pool_size, strides, padding='valid', data_format='channels_last', name=None, **kwargs):
    super().__init__(name=name or 'max_pooling2d')
    assert padding.upper() == 'SAME' and _pair(pool_size) == (3, 3) and _pair(strides) == (2, 2), (pool_size, strides)

  def call(self, x):
    return _nhwc(_orc().max_pool_same_3x3_s2(_nchw(x)))


def ns(_name, **attrs):
  """Permissive namespace: the given attributes are real, anything else is a dummy class."""
  m = stub_module(_name)
  for k, v in attrs.items():
    setattr(m, k, v)
  return m


def build_tf():
  """The numpy/torch-backed `tensorflow` module."""
  tf = stub_module('tensorflow')
  tf.float32, tf.bfloat16, tf.float16 = torch.float32, torch.bfloat16, torch.float16
  tf.Tensor = torch.Tensor
  tf.is_tensor = torch.is_tensor
  tf.cast = lambda x, dtype=None: T(x).to(dtype) if isinstance(dtype, torch.dtype) else T(x)
  tf.identity = lambda x, name=None: x
  tf.add = lambda a, b, name=None: a + b
  tf.add_n = lambda xs: sum(xs[1:], xs[0])
  tf.stack = lambda xs, axis=0: torch.stack(list(xs), dim=axis)
  tf.concat = lambda xs, axis=0: torch.cat(list(xs), dim=axis)
  tf.reduce_sum = lambda x, axis=None, keepdims=False: x.sum() if axis is None else x.sum(dim=axis, keepdim=keepdims)
  tf.reduce_mean = lambda x, axis=None, keepdims=False: x.mean() if axis is None else x.mean(dim=tuple(axis) if isinstance(axis, (list, tuple)) else axis, keepdim=keepdims)
  tf.sigmoid = torch.sigmoid
  tf.floor = torch.floor
  tf.shape = lambda x: list(x.shape)

  def uniform(shape, dtype=None, **kwargs):
    # stochastic depth draws (utils.drop_connect): recorded in call order so the consumer can replay them
    u = torch.from_numpy(DRAW_RNG[0].random(tuple(int(s) for s in shape)).astype(np.float32))
    DRAWS.append(u.reshape(-1).numpy().copy())
    return T(u)
  tf.random = ns('random', uniform=uniform)

  # -- the pieces tf2/train_lib.py's losses and train_step touch ---------------------------------------------
  def one_hot(indices, depth, dtype=None, **kwargs):
    idx = T(indices).long()
    out = torch.zeros(tuple(idx.shape) + (int(depth),), dtype=dtype or torch.float32)
    ok = (idx >= 0) & (idx < depth)                   # tf.one_hot: out-of-range indices give all-zero rows
    out.scatter_(-1, idx.clamp(0, depth - 1).unsqueeze(-1), ok.unsqueeze(-1).to(out.dtype))
    return T(out)
  tf.one_hot = one_hot
  tf.reshape = lambda x, shape, name=None: T(x).reshape([int(v) for v in shape])
  tf.expand_dims = lambda x, axis=-1: T(x).unsqueeze(axis)
  tf.not_equal = lambda a, b: T(a) != b
  tf.cast = lambda x, dtype=None: T(x).to(dtype) if isinstance(dtype, torch.dtype) else T(x)
  tf.convert_to_tensor = lambda x, dtype=None: T(torch.as_tensor(x, dtype=dtype) if isinstance(dtype, torch.dtype)
                                                 else x)

  def clip_by_norm(t, clip_norm):
    return t * clip_norm / torch.clamp(t.norm(), min=clip_norm)

  def global_norm(ts):
    return torch.sqrt(sum((t * t).sum() for t in ts if t is not None))

  def clip_by_global_norm(ts, clip_norm):
    g = global_norm(ts)
    scale = clip_norm * torch.minimum(1.0 / g, torch.tensor(1.0 / clip_norm))
    return [t * scale if t is not None else None for t in ts], g
  tf.clip_by_norm, tf.clip_by_global_norm = clip_by_norm, clip_by_global_norm
  tf.linalg = ns('linalg', global_norm=global_norm)

  class GradientTape(object):
    def __enter__(self):
      return self

    def __exit__(self, *a):
      return False

    def gradient(self, loss, variables):
      return list(torch.autograd.grad(loss, variables, allow_unused=True))
  tf.GradientTape = GradientTape
  tf.convert_to_tensor = lambda x, dtype=None: T(x)
  tf.stop_gradient = lambda x: x
  tf.zeros_initializer = lambda *a, **k: 'zeros'
  tf.ones = 'ones'
  tf.constant_initializer = lambda *a, **k: 'const'
  tf.random_normal_initializer = lambda *a, **k: 'normal'
  tf.variance_scaling_initializer = lambda *a, **k: 'vs'
  tf.initializers = ns('initializers', variance_scaling=lambda *a, **k: 'vs')

  class _Scope(object):
    def __init__(self, name=None):
      self.name = name

    def __enter__(self):
      SCOPE.append(self.name)       # TF2: variables created inside a name scope carry its prefix
      return self

    def __exit__(self, *a):
      SCOPE.pop()
      return False
  tf.name_scope = lambda name=None, *a, **k: _Scope(name)
  nn = ns('nn')
  nn.swish = lambda x: x * torch.sigmoid(x)
  nn.silu = nn.swish
  nn.relu = torch.relu
  nn.relu6 = lambda x: torch.clamp(x, 0, 6)
  nn.sigmoid = torch.sigmoid
  nn.softmax = lambda x, axis=-1: torch.softmax(x, dim=axis)
  tf.nn = nn
  tf.nn.l2_loss = lambda v: (v * v).sum() / 2
  tf.nn.sigmoid_cross_entropy_with_logits = lambda labels=None, logits=None: (
      torch.clamp(logits, min=0) - logits * labels + torch.log1p(torch.exp(-logits.abs())))   # TF's stable form
  tf.autograph = ns('autograph', experimental=ns('experimental', do_not_convert=lambda f: f))

  def resize_nearest(images, size, **kwargs):
    return _nhwc(_orc().resize_nearest(_nchw(images), int(size[0]), int(size[1])))
  tf.image = ns('image', resize_nearest_neighbor=resize_nearest)
  tf.compat = types.SimpleNamespace(v1=tf, v2=tf)     # `import tensorflow.compat.v1 as tf` resolves to this module
  layers = ns('layers', Layer=Layer, Conv2D=Conv2D, DepthwiseConv2D=DepthwiseConv2D,
              SeparableConv2D=SeparableConv2D, BatchNormalization=BatchNormalization, MaxPooling2D=MaxPooling2D,
              Dense=Dense, GlobalAveragePooling2D=GlobalAveragePooling2D, Dropout=Dropout)
  layers.experimental = ns('experimental', SyncBatchNormalization=BatchNormalization)
  keras = stub_module('tensorflow.keras')
  keras.layers = layers
  keras.Model = Model
  tf.keras = keras

  class Reduction(object):
    NONE, AUTO, SUM, SUM_OVER_BATCH_SIZE = 'none', 'auto', 'sum', 'sum_over_batch_size'

  class Loss(object):
    """keras.losses.Loss.__call__: call() then the reduction; the reference builds its losses with Reduction.NONE
    (tf2/train.py:115-140), which returns call()'s result as is."""

    def __init__(self, reduction='auto', name=None, **kwargs):
      self.reduction = reduction

    def __call__(self, y_true, y_pred, sample_weight=None):
      assert sample_weight is None
      out = self.call(y_true, y_pred)
      if self.reduction == Reduction.NONE:
        return out
      return out.sum() if self.reduction == Reduction.SUM else out.mean()

  class Huber(Loss):
    """keras.losses.Huber: 0.5 e^2 for |e| <= delta else delta |e| - 0.5 delta^2, mean over the LAST axis."""

    def __init__(self, delta=1.0, **kwargs):
      super().__init__(**kwargs)
      self.delta = delta

    def call(self, y_true, y_pred):
      err = (y_pred - y_true).abs()
      quad = torch.clamp(err, max=self.delta)
      return (0.5 * quad * quad + self.delta * (err - quad)).mean(dim=-1)
  keras.losses = ns('losses', Loss=Loss, Huber=Huber, Reduction=Reduction)
  keras.mixed_precision = ns('mixed_precision', LossScaleOptimizer=type('LossScaleOptimizer', (), {}))
  tf.io = ns('io', gfile=ns('gfile', GFile=open, exists=lambda p: False))
  return tf


def install(tf):
  names = ['tensorflow', 'tensorflow.compat', 'absl', 'absl.logging', 'absl.flags', 'tensorflow.python',
           'tensorflow.python.eager', 'tensorflow.python.tpu', 'tensorflow.python.eager.tape',
           'tensorflow.python.tpu.tpu_function', 'tensorflow_addons', 'tensorflow_addons.layers', 'tensorflow.python.framework',
           'tensorflow.python.ops', 'neural_structured_learning', 'tensorflow_hub', 'coco_metric', 'inference', 'PIL',
           'PIL.Image', 'pycocotools', 'tensorflow_model_optimization', 'dataloader', 'tf2.postprocess', 'tf2.label_util',
           'nms_np', 'det_model_fn']
  for n in names:
    sys.modules[n] = tf if n == 'tensorflow' else stub_module(n)
  sys.modules['tensorflow.compat.v1'] = tf
  sys.modules['tensorflow.compat.v2'] = tf
