"""EfficientDetNet: the reference's network interface on MI355X kernels.

Mirror of ``efficientdet/tf2/efficientdet_keras.py:787-915``:
``EfficientDetNet(model_name=None, config=None)(images[B,H,W,3], training) ->
(cls_outputs, box_outputs)``, each a list over levels min_level..max_level of
``[B,H_l,W_l,A*num_classes]`` / ``[B,H_l,W_l,A*4]``.  Buffers are sized at the first
call for that batch / image size (static shapes), extra keyword arguments choose
the storage dtype and device.
"""
import numpy as np
import torch

from automl_amd import engine as engine_lib
from automl_amd import hparams_config
from automl_amd import utils


class EfficientDetNet(object):
  """EfficientDet network without pre/post-processing."""

  def __init__(self, model_name=None, config=None, name='', feature_only=False, dtype='bf16',
               device='cuda:0', seed=0, params=None, stochastic_depth=True):
    if feature_only:
      raise ValueError('feature_only=True is out of scope')
    config = config or hparams_config.get_efficientdet_config(model_name)
    if 'object_detection' not in config.heads or len(config.heads) != 1:
      raise ValueError('No valid head found: {}'.format(config.heads))
    # (config.survival_prob -- efficientdet_keras.py:434-436 / :612-614: residual connections with stochastic depth inside
    # the class / box towers; no d0..d7x configuration sets it, the backbone's own survival_prob 0.8 is independent of this
    # key -- is built since r04: Engine._head_level)
    self.config = config
    self.name = name
    self._dtype, self._device, self._seed = dtype, device, seed
    self._init_params = params
    self._init_ema = None
    self._stochastic_depth = stochastic_depth   # False: survival_prob off for every backbone (deterministic)
    self.engine = None
    self._engines = {}

  MAX_ENGINES = 2      # activation buffers of that many (batch, image size) shapes stay allocated

  def _ensure_engine(self, batch, height, width):
    """The executor for this batch / image shape.  Buffers are per shape (static shapes); the variables, the
    momentum and EMA slots and the iteration count live in ONE arena shared by every executor of the model, so an
    evaluation pass at another batch size or a partial last batch does not disturb the training state (the
    reference's Keras variables and optimizer slots are shape independent)."""
    key = (batch, height, width)
    e = self._engines.get(key)
    if e is None:
      arena = self.engine.arena if self.engine is not None else None
      e = engine_lib.Engine(self.config, batch, (height, width), dtype=self._dtype, device=self._device,
                            seed=self._seed, params=self._init_params, stochastic_depth=self._stochastic_depth,
                            arena=arena)
      if self._init_ema and arena is None:
        e.arena.set_ema_params(self._init_ema)      # shadows restored before the model was built (util_keras.restore_ckpt)
        self._init_ema = None
      while len(self._engines) >= self.MAX_ENGINES:
        self._engines.pop(next(iter(self._engines)))          # least recently used shape
      self._engines[key] = e
    else:
      self._engines[key] = self._engines.pop(key)             # most recently used last
    self.engine = e
    return e

  def _to_device_images(self, images, eng):
    if isinstance(images, np.ndarray):
      images = torch.from_numpy(images)
    if images.dim() != 4 or images.shape[-1] != 3:
      raise ValueError('images must be [batch, height, width, 3], got %s' % (tuple(images.shape),))
    return images.to(device=eng.device, dtype=eng.tdtype).contiguous()

  def __call__(self, inputs, training=False):
    b, h, w = int(inputs.shape[0]), int(inputs.shape[1]), int(inputs.shape[2])
    eng = self._ensure_engine(b, h, w)
    eng.forward(self._to_device_images(inputs, eng), training=training)
    return eng.outputs()

  call = __call__

  # ---- variables, addressed by the reference names ------------------------------------------
  def set_weights(self, values):
    if self.engine is None:
      self._init_params = dict(values) if self._init_params is None else {**self._init_params, **values}
    else:
      self.engine.set_params(values)

  def get_weights(self):
    if self.engine is None:
      raise RuntimeError('the network has not been built yet (call it once)')
    return self.engine.get_params()

  def get_ema_weights(self):
    """Variables for an EMA evaluation: MovingAverage shadows of the trainable variables (train_lib.py:193-197),
    BatchNorm moving statistics as they are."""
    if self.engine is None:
      raise RuntimeError('the network has not been built yet (call it once)')
    return self.engine.arena.get_ema_params()

  def set_ema_weights(self, values):
    """EMA shadows of trainable variables by name (the 'average' slots util_keras.restore_ckpt fills, :166-178)."""
    if self.engine is None:
      self._init_ema = dict(values) if self._init_ema is None else {**self._init_ema, **values}
    else:
      self.engine.arena.set_ema_params(values)

  def get_optimizer_state(self):
    """Momentum, EMA shadows and iteration count (host copies), for checkpointing."""
    if self.engine is None:
      raise RuntimeError('the network has not been built yet (call it once)')
    return self.engine.arena.get_optimizer_state()

  def set_optimizer_state(self, state):
    if self.engine is None:
      raise RuntimeError('the network has not been built yet (call it once)')
    self.engine.arena.set_optimizer_state(state)

  def anchors(self, image_size=None):
    from automl_amd import anchors as anchors_lib
    c = self.config
    size = image_size if image_size is not None else c.image_size
    return anchors_lib.Anchors(c.min_level, c.max_level, c.num_scales, c.aspect_ratios, c.anchor_scale,
                               size)


class EfficientDetModel(EfficientDetNet):
  """EfficientDet full model with post-processing: ``efficientdet_keras.EfficientDetModel`` (:917-1000).

  ``model(inputs, training=False, pre_mode='infer', post_mode='global')``.  ``pre_mode='infer'`` (the reference's
  default): ``inputs`` are raw [B, H, W, 3] images (uint8 or float, 0..255) of one size; they are normalised with
  config.mean_rgb / stddev_rgb, resized with the aspect ratio kept into the top-left corner of config.image_size
  and zero padded on the GPU (automl_amd/preprocess.py), and the detections come back in the pixels of the raw
  images.  ``pre_mode`` None / '': ``inputs`` is the normalised batch at config.image_size.  ``post_mode``: 'global' /
  'per_class' run on the GPU (automl_amd/postprocess.py) and return (boxes, scores, classes, valid_len); None returns
  the raw level outputs.
  """

  def _preprocessing(self, raw_images, image_size, mean_rgb, stddev_rgb, mode=None):
    """Preprocess images before feeding to the network (efficientdet_keras.py:920-951)."""
    if not mode:
      return raw_images, None
    if mode != 'infer':
      raise ValueError('preprocessing must be infer or empty')
    from automl_amd import preprocess
    if isinstance(raw_images, np.ndarray):
      raw_images = torch.from_numpy(raw_images)
    tdt = torch.bfloat16 if self._dtype in ('bf16', 1) else torch.float32
    return preprocess.preprocess_infer(raw_images, image_size, mean_rgb, stddev_rgb, dtype=tdt)

  def _postprocess(self, cls_outputs, box_outputs, scales, mode='global'):
    """Postprocess class and box predictions (efficientdet_keras.py:953-976)."""
    if not mode:
      return cls_outputs, box_outputs
    from automl_amd import postprocess
    if mode == 'global':
      return postprocess.postprocess_global(self.config.as_dict(), cls_outputs, box_outputs, scales)
    if mode == 'per_class':
      return postprocess.postprocess_per_class(self.config.as_dict(), cls_outputs, box_outputs, scales)
    if mode in ('combined', 'tflite'):
      raise ValueError('postprocess mode {} is not built'.format(mode))
    raise ValueError('Unsupported postprocess mode {}'.format(mode))

  def __call__(self, inputs, training=False, pre_mode='infer', post_mode='global'):
    config = self.config
    inputs, scales = self._preprocessing(inputs, config.image_size, config.mean_rgb, config.stddev_rgb, pre_mode)
    cls_outputs, box_outputs = EfficientDetNet.__call__(self, inputs, training)
    if post_mode:
      return self._postprocess(cls_outputs, box_outputs, scales, post_mode)
    return cls_outputs, box_outputs

  call = __call__
