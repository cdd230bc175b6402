"""Variable inventory of EfficientDetNet: names, shapes, initialisers (host side, no GPU).

Names follow the reference's checkpoint contract (SURVEY.md Appendix B;
efficientdet/tf2/efficientdet_keras.py:123-131,160,406,415 and
efficientdet/backbone/efficientnet_model.py:272-277): e.g.
``efficientnet-b0/blocks_1/conv2d/kernel``,
``fpn_cells/cell_0/fnode3/op_after_combine8/conv/depthwise_kernel``,
``class_net/class-0-bn-3/gamma``, ``fpn_cells/cell_0/fnode0/WSM_1``.
Kernel layouts are the reference's: conv [kh,kw,cin,cout], depthwise [kh,kw,c,1].
"""
import collections
import math

import numpy as np

from automl_amd import efficientnet_builder as eb
from automl_amd import fpn_configs
from automl_amd import utils

ParamSpec = collections.namedtuple('ParamSpec', ['name', 'shape', 'init', 'trainable'])

BN_MOMENTUM = 0.99
BN_EPSILON = 1e-3


class NetSpec(object):
  """Structure + variable list for one detection config."""

  def __init__(self, config):
    self.config = config
    self.params = []           # ParamSpec, creation order == reference variable order per layer
    self._names = set()
    c = config
    if not c.backbone_name.startswith('efficientnet-b'):
      raise ValueError('backbone %r is out of scope (efficientnet-b0..b7 are built)' % c.backbone_name)
    # utils.activation_fn (utils.py:36-53): codes of include/edet_hip.h (every type the reference knows)
    codes = {'swish': 1, 'silu': 1, 'swish_native': 1, 'relu': 2, 'relu6': 3, 'hswish': 4, 'mish': 5, 'srelu': 6}
    if c.act_type not in codes:
      raise ValueError('Unsupported act_type {}'.format(c.act_type))
    self.act_code = codes[c.act_type]
    if not c.separable_conv or c.conv_bn_act_pattern:
      raise ValueError('only the default separable_conv / conv-bn ordering of the d0..d7x configs is built')
    self.stem_filters, self.blocks = eb.backbone_blocks(
        c.backbone_name, c.backbone_config.blocks if c.backbone_config is not None else None)
    self.reductions = eb.reduction_indices(self.blocks)
    # stochastic depth (efficientdet_keras.py:811-812; efficientnet_builder.py:174; efficientnet_model.py:751-756):
    # the builder's survival_prob 0.8, forced to 0 (= off) for the b0 backbone, per block 1 - 0.2 * idx / n.
    sp = 0.0 if 'b0' in c.backbone_name else 0.8
    self.survival_probs = [(1.0 - (1.0 - sp) * float(b.index) / len(self.blocks)) if sp else None
                           for b in self.blocks]
    self.fpn = c.fpn_config or fpn_configs.get_fpn_config(c.fpn_name, c.min_level, c.max_level,
                                                         c.fpn_weight_method)
    if self.fpn.weight_method not in ('fastattn', 'sum', 'attn', 'channel_fastattn', 'channel_attn'):
      raise ValueError('unknown weight_method %s' % self.fpn.weight_method)
    self.num_anchors = len(c.aspect_ratios) * c.num_scales
    self._build()

  # -- helpers ---------------------------------------------------------------------------------
  def _add(self, name, shape, init, trainable=True):
    assert name not in self._names, name
    self._names.add(name)
    self.params.append(ParamSpec(name, tuple(shape), init, trainable))

  def _bn(self, scope, c):
    self._add(scope + '/gamma', (c,), 'ones')
    self._add(scope + '/beta', (c,), 'zeros')
    self._add(scope + '/moving_mean', (c,), 'zeros', False)
    self._add(scope + '/moving_variance', (c,), 'ones', False)

  def _build(self):
    c = self.config
    bb = c.backbone_name
    self._add(bb + '/stem/conv2d/kernel', (3, 3, 3, self.stem_filters), 'conv')
    self._bn(bb + '/stem/tpu_batch_normalization', self.stem_filters)
    for b in self.blocks:
      s = '%s/blocks_%d' % (bb, b.index)
      cexp = b.input_filters * b.expand_ratio
      bn_i = conv_i = 0
      bn_names = ['tpu_batch_normalization', 'tpu_batch_normalization_1', 'tpu_batch_normalization_2']
      conv_names = ['conv2d', 'conv2d_1']
      if b.expand_ratio != 1:
        self._add('%s/%s/kernel' % (s, conv_names[conv_i]), (1, 1, b.input_filters, cexp), 'conv')
        conv_i += 1
        self._bn('%s/%s' % (s, bn_names[bn_i]), cexp)
        bn_i += 1
      self._add(s + '/depthwise_conv2d/depthwise_kernel', (b.kernel_size, b.kernel_size, cexp, 1), 'conv')
      self._bn('%s/%s' % (s, bn_names[bn_i]), cexp)
      bn_i += 1
      if b.se_filters:
        self._add(s + '/se/conv2d/kernel', (1, 1, cexp, b.se_filters), 'conv')
        self._add(s + '/se/conv2d/bias', (b.se_filters,), 'zeros')
        self._add(s + '/se/conv2d_1/kernel', (1, 1, b.se_filters, cexp), 'conv')
        self._add(s + '/se/conv2d_1/bias', (cexp,), 'zeros')
      self._add('%s/%s/kernel' % (s, conv_names[conv_i]), (1, 1, cexp, b.output_filters), 'conv')
      self._bn('%s/%s' % (s, bn_names[bn_i]), b.output_filters)
    # channels of backbone reductions 1..5
    red_ch = [self.blocks[i].output_filters for i in self.reductions]
    wf = c.fpn_num_filters
    feat_ch = [None] + red_ch
    level_ch = list(feat_ch[c.min_level:c.max_level + 1])
    for level in range(6, c.max_level + 1):
      if level_ch[-1] != wf:
        s = 'resample_p%d' % level
        self._add(s + '/conv2d/kernel', (1, 1, level_ch[-1], wf), 'glorot')
        self._add(s + '/conv2d/bias', (wf,), 'zeros')
        if c.apply_bn_for_resampling:      # ResampleFeatureMap creates its BatchNorm only then (efficientdet_keras.py:290-296)
          self._bn(s + '/bn', wf)
      level_ch.append(wf)
    self.level_channels = level_ch
    num_levels = c.max_level - c.min_level + 1
    for rep in range(c.fpn_cell_repeats):
      ch = list(level_ch) if rep == 0 else [wf] * num_levels
      for n, node in enumerate(self.fpn.nodes):
        s = 'fpn_cells/cell_%d/fnode%d' % (rep, n)
        for i, off in enumerate(node['inputs_offsets']):
          if ch[off] != wf:
            rs = '%s/resample_%d_%d_%d' % (s, i, off, len(ch))
            self._add(rs + '/conv2d/kernel', (1, 1, ch[off], wf), 'glorot')
            self._add(rs + '/conv2d/bias', (wf,), 'zeros')
            if c.apply_bn_for_resampling:
              self._bn(rs + '/bn', wf)
        if self.fpn.weight_method != 'sum':
          # one scalar per input, or one weight per channel for the channel_* methods
          # (efficientdet_keras.py:123-127,142-151)
          wshape = (wf,) if self.fpn.weight_method.startswith('channel_') else ()
          for i in range(len(node['inputs_offsets'])):
            self._add(s + '/WSM' + ('' if i == 0 else '_%d' % i), wshape, 'ones')
        oc = '%s/op_after_combine%d' % (s, len(ch))
        self._add(oc + '/conv/depthwise_kernel', (3, 3, wf, 1), 'glorot')
        self._add(oc + '/conv/pointwise_kernel', (1, 1, wf, wf), 'glorot')
        self._add(oc + '/conv/bias', (wf,), 'zeros')
        self._bn(oc + '/bn', wf)
        ch.append(wf)
    for net, prefix, out_ch, binit in (('class_net', 'class', c.num_classes * self.num_anchors, 'class_bias'),
                                       ('box_net', 'box', 4 * self.num_anchors, 'zeros')):
      for i in range(c.box_class_repeats):
        s = '%s/%s-%d' % (net, prefix, i)
        self._add(s + '/depthwise_kernel', (3, 3, wf, 1), 'varscale')
        self._add(s + '/pointwise_kernel', (1, 1, wf, wf), 'varscale')
        self._add(s + '/bias', (wf,), 'zeros')
        for level in range(c.min_level, c.max_level + 1):
          self._bn('%s/%s-%d-bn-%d' % (net, prefix, i, level), wf)
      s = '%s/%s-predict' % (net, prefix)
      self._add(s + '/depthwise_kernel', (3, 3, wf, 1), 'varscale')
      self._add(s + '/pointwise_kernel', (1, 1, wf, out_ch), 'varscale')
      self._add(s + '/bias', (out_ch,), binit)

  # -- queries ------------------------------------------------------------------------------------
  def trainable(self):
    return [p for p in self.params if p.trainable]

  def num_trainable_elements(self):
    return sum(int(np.prod(p.shape)) for p in self.trainable())

  def feat_sizes(self, image_size):
    return utils.get_feat_sizes(image_size, self.config.max_level)


def is_l2_regularised(name):
  """reference train_lib.py:486: regex r'.*(kernel|weight):0$'."""
  return name.endswith('kernel') or name.endswith('weight')


def init_value(spec, rng):
  """Reference initialisers (efficientnet_model.py:52-73; efficientdet_keras.py:459-478)."""
  shape, kind = spec.shape, spec.init
  if kind == 'zeros':
    return np.zeros(shape, np.float32)
  if kind == 'ones':
    return np.ones(shape, np.float32)
  if kind == 'class_bias':
    return np.full(shape, -math.log((1 - 0.01) / 0.01), np.float32)
  if kind == 'conv':
    kh, kw, _, cout = shape
    return (rng.standard_normal(shape) * math.sqrt(2.0 / (kh * kw * cout))).astype(np.float32)
  if kind == 'glorot':
    receptive = shape[0] * shape[1]
    lim = math.sqrt(6.0 / (shape[2] * receptive + shape[3] * receptive))
    return rng.uniform(-lim, lim, shape).astype(np.float32)
  if kind == 'varscale':
    fan_in = shape[0] * shape[1] * shape[2]
    return (rng.standard_normal(shape) * math.sqrt(1.0 / fan_in)).astype(np.float32)
  raise ValueError(kind)


def init_params(netspec, seed=0):
  """name -> numpy fp32 array, drawn in variable order from numpy default_rng(seed)."""
  rng = np.random.default_rng(seed)
  return collections.OrderedDict((p.name, init_value(p, rng)) for p in netspec.params)
