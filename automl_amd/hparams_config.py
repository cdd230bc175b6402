"""Hyper-parameter surface of the EfficientDet hot path (host side, pure Python).

Mirrors the public names of the reference's ``efficientdet/hparams_config.py``
(Config :35-167, default_detection_configs :170-298, model tables :301-467,
get_efficientdet_config :470-487) so that a user of the reference finds the
same model names (``efficientdet-d0`` .. ``efficientdet-d7x``) and the same
``k=v,k2.sub=v`` override syntax.  No TensorFlow: yaml I/O uses plain ``open``.
"""
import ast
import copy
from collections import abc

import yaml


def _scalar_from_text(text):
  """Value of one override token: true / false, a Python literal (int, float, tuple ...), else the text itself
  (so that '640x640' or 'ss' stay strings)."""
  lowered = {'true': True, 'false': False}
  if text in lowered:
    return lowered[text]
  try:
    return ast.literal_eval(text)
  except (ValueError, SyntaxError):
    return text


eval_str_fn = _scalar_from_text      # the reference's public name for the same conversion


def parse_override_string(text):
  """'a=1,b.c=x,d=1*2*3' -> {'a': 1, 'b': {'c': 'x'}, 'd': [1, 2, 3]}.

  Grammar of the reference's command-line overrides (hparams_config.py:127-157): comma-separated ``key=value``
  pairs, empty pairs ignored, dots in the key open nested tables, ``*`` separates list elements.  Anything that is
  not exactly one ``key=value`` raises ValueError('Invalid config_str: ...').
  """
  tree = {}
  for item in (text or '').split(','):
    if not item:
      continue
    if item.count('=') != 1:
      raise ValueError('Invalid config_str: {}'.format(text))
    dotted, raw = item.split('=')
    path = dotted.strip().split('.')
    value = [_scalar_from_text(piece) for piece in raw.split('*')] if '*' in raw else _scalar_from_text(raw)
    node = tree
    for part in path[:-1]:
      child = node.get(part)
      if not isinstance(child, dict):
        child = node[part] = {}
      node = child
    leaf = path[-1]
    if isinstance(value, dict) and isinstance(node.get(leaf), dict):
      node[leaf].update(value)
    else:
      node[leaf] = value
  return tree


class Config(object):
  """Attribute-style table of hyper-parameters; nested tables are Configs themselves.

  The fields live in one ordered mapping (``_fields``) instead of the instance ``__dict__``; values are stored by
  copy, mappings are wrapped on the way in.  Public surface = the reference's: attribute get / set, ``update``
  (new keys allowed), ``override`` (dict, ``k=v`` string or ``*.yaml`` path; unknown keys raise KeyError unless
  ``allow_new_keys``), ``get``, ``keys``, ``as_dict``, yaml load / save.
  """

  __slots__ = ('_fields',)

  def __init__(self, config_dict=None):
    object.__setattr__(self, '_fields', {})
    self.update(config_dict)

  # ---- element access
  @staticmethod
  def _wrap(value):
    if isinstance(value, Config):
      return Config(value.as_dict())
    if isinstance(value, abc.Mapping):
      return Config(dict(value))
    return copy.deepcopy(value)

  def __setattr__(self, key, value):
    self._fields[key] = self._wrap(value)

  def __getattr__(self, key):
    fields = object.__getattribute__(self, '_fields')
    if key in fields:
      return fields[key]
    raise AttributeError(key)

  def __getitem__(self, key):
    return self._fields[key]

  def __contains__(self, key):
    return key in self._fields

  def __iter__(self):
    return iter(self._fields)

  def get(self, key, default_value=None):
    return self._fields.get(key, default_value)

  def keys(self):
    return self._fields.keys()

  # ---- conversion
  def as_dict(self):
    return {k: (v.as_dict() if isinstance(v, Config) else copy.deepcopy(v)) for k, v in self._fields.items()}

  def __deepcopy__(self, memo):
    return type(self)(self.as_dict())

  def __copy__(self):
    return type(self)(self.as_dict())

  def __reduce__(self):
    # pickle / copy protocol: the default slot-state path would setattr('_fields') through the redirecting
    # __setattr__ before the mapping exists (torch.multiprocessing spawn arguments, DataLoader workers)
    return (type(self), (self.as_dict(),))

  def __repr__(self):
    return 'Config(%r)' % (self.as_dict(),)

  def __str__(self):
    plain = self.as_dict()
    try:
      return yaml.dump(plain, indent=4)
    except TypeError:
      return str(plain)

  # ---- merging
  def _merge(self, mapping, strict):
    """Recursive merge of `mapping` into this table; strict: a key that does not exist yet is an error."""
    if isinstance(mapping, Config):
      mapping = mapping.as_dict()
    for key, value in (mapping or {}).items():
      present = key in self._fields
      if not present and strict:
        raise KeyError('Key `{}` does not exist for overriding. '.format(key))
      current = self._fields.get(key)
      if present and isinstance(current, Config) and isinstance(value, (abc.Mapping, Config)):
        current._merge(value, strict)
      else:
        setattr(self, key, value)

  def update(self, config_dict):
    """Merge, creating keys that do not exist yet."""
    self._merge(config_dict, strict=False)

  def override(self, config_dict_or_str, allow_new_keys=False):
    """Merge a dict, a 'k=v,k2.sub=v' string or the contents of a .yaml file."""
    source = config_dict_or_str
    if isinstance(source, str):
      if source == '':
        return
      if '=' in source:
        source = self.parse_from_str(source)
      elif source.endswith('.yaml'):
        source = self.parse_from_yaml(source)
      else:
        raise ValueError('Invalid string {}, must end with .yaml or contains "=".'.format(source))
    elif not isinstance(source, (dict, Config)):
      raise ValueError('Unknown value type: {}'.format(source))
    self._merge(source, strict=not allow_new_keys)

  # ---- text forms
  def parse_from_str(self, config_str):
    return parse_override_string(config_str)

  def parse_from_yaml(self, yaml_file_path):
    with open(yaml_file_path, 'r') as stream:
      return yaml.load(stream, Loader=yaml.FullLoader)

  def save_to_yaml(self, yaml_file_path):
    with open(yaml_file_path, 'w') as stream:
      yaml.dump(self.as_dict(), stream, default_flow_style=False)


# defaults of the detection path (values of reference hparams_config.py:170-298, pinned by
# tests/golden/reference_tables.json which is generated from the reference's own module), grouped by what
# consumes them here; insertion order = the reference's attribute order, which as_dict()/yaml output follows
_DEFAULT_GROUPS = (
    ('identity', dict(name='efficientdet-d1', act_type='swish')),
    # input pipeline: kept for interface completeness (the data pipeline is out of scope)
    ('input', dict(image_size=640, target_size=None, input_rand_hflip=True, jitter_min=0.1, jitter_max=2.0,
                   autoaugment_policy=None, grid_mask=False, sample_image=None, map_freq=5)),
    ('dataset', dict(num_classes=90, seg_num_classes=3, heads=['object_detection'], skip_crowd_during_training=True,
                     label_map=None, max_instances_per_image=100, regenerate_source_id=False)),
    ('anchors', dict(min_level=3, max_level=7, num_scales=3, aspect_ratios=[1.0, 2.0, 0.5], anchor_scale=4.0)),
    ('optimiser', dict(is_training_bn=True, momentum=0.9, optimizer='sgd', learning_rate=0.08, lr_warmup_init=0.008,
                       lr_warmup_epoch=1.0, first_lr_drop_epoch=200.0, second_lr_drop_epoch=250.0,
                       poly_lr_power=0.9, clip_gradients_norm=10.0, num_epochs=300, data_format='channels_last',
                       mean_rgb=[0.485 * 255, 0.456 * 255, 0.406 * 255],
                       stddev_rgb=[0.229 * 255, 0.224 * 255, 0.225 * 255], scale_range=False)),
    ('losses', dict(label_smoothing=0.0, alpha=0.25, gamma=1.5, delta=0.1, box_loss_weight=50.0, iou_loss_type=None,
                    iou_loss_weight=1.0, weight_decay=4e-5, strategy=None, mixed_precision=False, loss_scale=None)),
    ('structure', dict(box_class_repeats=3, fpn_cell_repeats=3, fpn_num_filters=88, separable_conv=True,
                       apply_bn_for_resampling=True, conv_after_downsample=False, conv_bn_act_pattern=False,
                       drop_remainder=True,
                       nms_configs=dict(method='gaussian', iou_thresh=None, score_thresh=0., sigma=None, pyfunc=False,
                                        max_nms_inputs=0, max_output_size=100),
                       tflite_max_detections=100, fpn_name=None, fpn_weight_method=None, fpn_config=None,
                       survival_prob=None, img_summary_steps=None, lr_decay_method='cosine',
                       moving_average_decay=0.9998, ckpt_var_scope=None, skip_mismatch=True,
                       backbone_name='efficientnet-b1', backbone_config=None, var_freeze_expr=None,
                       use_keras_model=True, dataset_type=None, positives_momentum=None, grad_checkpoint=False,
                       verbose=1, save_freq='epoch')),
)


def default_detection_configs():
  """A fresh Config holding the defaults of the detection path."""
  h = Config()
  for _, group in _DEFAULT_GROUPS:
    for key, value in group.items():
      setattr(h, key, value)
  return h


def _det(name, backbone, size, filters, cells, repeats, **extra):
  d = dict(name=name, backbone_name=backbone, image_size=size,
           fpn_num_filters=filters, fpn_cell_repeats=cells,
           box_class_repeats=repeats)
  d.update(extra)
  return d


# compound-scaling table (reference hparams_config.py:301-389)
efficientdet_model_param_dict = {
    'efficientdet-d0': _det('efficientdet-d0', 'efficientnet-b0', 512, 64, 3, 3),
    'efficientdet-d1': _det('efficientdet-d1', 'efficientnet-b1', 640, 88, 4, 3),
    'efficientdet-d2': _det('efficientdet-d2', 'efficientnet-b2', 768, 112, 5, 3),
    'efficientdet-d3': _det('efficientdet-d3', 'efficientnet-b3', 896, 160, 6, 4),
    'efficientdet-d4': _det('efficientdet-d4', 'efficientnet-b4', 1024, 224, 7, 4),
    'efficientdet-d5': _det('efficientdet-d5', 'efficientnet-b5', 1280, 288, 7, 4),
    'efficientdet-d6': _det('efficientdet-d6', 'efficientnet-b6', 1280, 384, 8, 5,
                            fpn_weight_method='sum'),
    'efficientdet-d7': _det('efficientdet-d7', 'efficientnet-b6', 1536, 384, 8, 5,
                            anchor_scale=5.0, fpn_weight_method='sum'),
    'efficientdet-d7x': _det('efficientdet-d7x', 'efficientnet-b7', 1536, 384, 8, 5,
                             anchor_scale=4.0, max_level=8,
                             fpn_weight_method='sum'),
}

# The lite family (relu6, efficientnet-lite backbones) is listed for name
# completeness; the MI355X path does not build efficientnet-lite backbones
# (out of scope per SURVEY.md section 8 row C2).
_lite_common = dict(mean_rgb=127.0, stddev_rgb=128.0, act_type='relu6',
                    fpn_weight_method='sum')
efficientdet_lite_param_dict = {
    'efficientdet-lite0': _det('efficientdet-lite0', 'efficientnet-lite0', 320, 64, 3, 3,
                               anchor_scale=3.0, **_lite_common),
    'efficientdet-lite1': _det('efficientdet-lite1', 'efficientnet-lite1', 384, 88, 4, 3,
                               anchor_scale=3.0, **_lite_common),
    'efficientdet-lite2': _det('efficientdet-lite2', 'efficientnet-lite2', 448, 112, 5, 3,
                               anchor_scale=3.0, **_lite_common),
    'efficientdet-lite3': _det('efficientdet-lite3', 'efficientnet-lite3', 512, 160, 6, 4,
                               **_lite_common),
    'efficientdet-lite3x': _det('efficientdet-lite3x', 'efficientnet-lite3', 640, 200, 6, 4,
                                anchor_scale=3.0, **_lite_common),
    'efficientdet-lite4': _det('efficientdet-lite4', 'efficientnet-lite4', 640, 224, 7, 4,
                               **_lite_common),
}


def get_efficientdet_config(model_name='efficientdet-d1'):
  """Default config of a named model (reference hparams_config.py:470-480)."""
  h = default_detection_configs()
  if model_name in efficientdet_model_param_dict:
    h.override(efficientdet_model_param_dict[model_name])
  elif model_name in efficientdet_lite_param_dict:
    h.override(efficientdet_lite_param_dict[model_name])
  else:
    raise ValueError('Unknown model name: {}'.format(model_name))
  return h


def get_detection_config(model_name):
  if model_name.startswith('efficientdet'):
    return get_efficientdet_config(model_name)
  raise ValueError('model name must start with efficientdet.')
