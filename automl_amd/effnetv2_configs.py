"""EfficientNetV2 (and V1-through-the-V2-codebase) model tables -- host side, integers only.

Mirrors the public names of the reference's ``efficientnetv2/effnetv2_configs.py``
(BlockDecoder :22-89, v1/v2 block strings :93-179, efficientnetv2_params :150-179,
efficientnetv2_config :190-211, get_model_config :214-241), ``efficientnetv2/hparams.py:221-243``
(base_config.model) and the rounding rules of ``efficientnetv2/effnetv2_model.py:84-102``.
Only the ``model`` sub-tree of the reference config matters to the compute path; the ``train`` /
``eval`` / ``data`` image sizes are carried along so that a user of the reference finds them.
"""
import collections
import copy
import math
import re

from automl_amd import hparams_config

Config = hparams_config.Config


_FIELDS = (('r', 'num_repeat', int), ('k', 'kernel_size', int), ('s', 'strides', int), ('e', 'expand_ratio', int),
           ('i', 'input_filters', int), ('o', 'output_filters', int), ('c', 'conv_type', int))


class BlockDecoder(object):
  """Block string notation <-> block Config, e.g. 'r2_k3_s1_e1_i24_o24_c1' or 'r6_k3_s2_e4_i64_o128_se0.25'
  (effnetv2_configs.py:22-89): r repeats, k kernel, s stride, e expand ratio, i/o filters, c conv type
  (1 = fused), se squeeze-excite ratio."""

  _TOKEN = re.compile(r'^([a-z]+)([0-9][0-9.]*)$')

  def _decode_block_string(self, block_string):
    if not isinstance(block_string, str):
      raise TypeError('block notation must be a string, got %r' % (block_string,))
    opts = {}
    for token in block_string.split('_'):
      m = self._TOKEN.match(token)
      if m:
        opts[m.group(1)] = m.group(2)
    block = {name: conv(opts[key]) for key, name, conv in _FIELDS if key in opts}
    block.setdefault('conv_type', 0)
    block['se_ratio'] = float(opts['se']) if 'se' in opts else None
    return Config(block)

  def _encode_block_string(self, block):
    parts = ['%s%s' % (key, block[name]) for key, name, _ in _FIELDS]
    if block.se_ratio is not None and 0 < block.se_ratio <= 1:
      parts.append('se%s' % block.se_ratio)
    return '_'.join(parts)

  def decode(self, string_list):
    if not isinstance(string_list, list):
      raise TypeError('expected a list of block strings')
    return [self._decode_block_string(s) for s in string_list]

  def encode(self, blocks_args):
    return [self._encode_block_string(b) for b in blocks_args]


def _stages(rows):
  """Stage rows (repeats, kernel, stride, expand, in, out, fused?, se) -> the reference's block strings."""
  out = []
  for r, k, st, e, i, o, fused, se in rows:
    text = 'r%d_k%d_s%d_e%d_i%d_o%d' % (r, k, st, e, i, o)
    out.append(text + ('_c1' if fused else '') + ('_se%s' % se if se else ''))
  return out


#################### EfficientNet V1 (as served by the V2 code base) ####################
# the seven MBConv stages of B0; wider / deeper variants scale them (effnetv2_configs.py:93-101)
v1_b0_block_str = _stages([
    (1, 3, 1, 1, 32, 16, 0, 0.25), (2, 3, 2, 6, 16, 24, 0, 0.25), (2, 5, 2, 6, 24, 40, 0, 0.25),
    (3, 3, 2, 6, 40, 80, 0, 0.25), (3, 5, 1, 6, 80, 112, 0, 0.25), (4, 5, 2, 6, 112, 192, 0, 0.25),
    (1, 3, 1, 6, 192, 320, 0, 0.25)])

# name: (width_coefficient, depth_coefficient, resolution, dropout_rate)   (effnetv2_configs.py:104-116)
efficientnetv1_params = dict(
    [('efficientnet-b%d' % i, v) for i, v in enumerate([
        (1.0, 1.0, 224, 0.2), (1.0, 1.1, 240, 0.2), (1.1, 1.2, 260, 0.3), (1.2, 1.4, 300, 0.3), (1.4, 1.8, 380, 0.4),
        (1.6, 2.2, 456, 0.4), (1.8, 2.6, 528, 0.5), (2.0, 3.1, 600, 0.5), (2.2, 3.6, 672, 0.5)])] +
    [('efficientnet-l2', (4.3, 5.3, 800, 0.5))])


def efficientnetv1_config(model_name='efficientnet-b0'):
  width, depth, isize, dropout = efficientnetv1_params[model_name]
  return Config(dict(
      model=dict(model_name=model_name, blocks_args=BlockDecoder().decode(v1_b0_block_str),
                 width_coefficient=width, depth_coefficient=depth, dropout_rate=dropout),
      eval=dict(isize=isize),
      train=dict(isize=0.8),
      data=dict(augname='effnetv1_autoaug'),
  ))


#################### EfficientNet V2 (effnetv2_configs.py:139-179) ####################
# fused stages first (conv type 1, no SE), then MBConv stages with SE 0.25
v2_base_block = _stages([
    (1, 3, 1, 1, 32, 16, 1, 0), (2, 3, 2, 4, 16, 32, 1, 0), (2, 3, 2, 4, 32, 48, 1, 0),
    (3, 3, 2, 4, 48, 96, 0, 0.25), (5, 3, 1, 6, 96, 112, 0, 0.25), (8, 3, 2, 6, 112, 192, 0, 0.25)])
v2_s_block = _stages([
    (2, 3, 1, 1, 24, 24, 1, 0), (4, 3, 2, 4, 24, 48, 1, 0), (4, 3, 2, 4, 48, 64, 1, 0),
    (6, 3, 2, 4, 64, 128, 0, 0.25), (9, 3, 1, 6, 128, 160, 0, 0.25), (15, 3, 2, 6, 160, 256, 0, 0.25)])
v2_m_block = _stages([
    (3, 3, 1, 1, 24, 24, 1, 0), (5, 3, 2, 4, 24, 48, 1, 0), (5, 3, 2, 4, 48, 80, 1, 0),
    (7, 3, 2, 4, 80, 160, 0, 0.25), (14, 3, 1, 6, 160, 176, 0, 0.25), (18, 3, 2, 6, 176, 304, 0, 0.25),
    (5, 3, 1, 6, 304, 512, 0, 0.25)])
v2_l_block = _stages([
    (4, 3, 1, 1, 32, 32, 1, 0), (7, 3, 2, 4, 32, 64, 1, 0), (7, 3, 2, 4, 64, 96, 1, 0),
    (10, 3, 2, 4, 96, 192, 0, 0.25), (19, 3, 1, 6, 192, 224, 0, 0.25), (25, 3, 2, 6, 224, 384, 0, 0.25),
    (7, 3, 1, 6, 384, 640, 0, 0.25)])
v2_xl_block = _stages([
    (4, 3, 1, 1, 32, 32, 1, 0), (8, 3, 2, 4, 32, 64, 1, 0), (8, 3, 2, 4, 64, 96, 1, 0),
    (16, 3, 2, 4, 96, 192, 0, 0.25), (24, 3, 1, 6, 192, 256, 0, 0.25), (32, 3, 2, 6, 256, 512, 0, 0.25),
    (8, 3, 1, 6, 512, 640, 0, 0.25)])

# name: (block, width, depth, train_size, eval_size, dropout, randaug, mixup, aug)
efficientnetv2_params = {}
for _name, _row in (('s', (v2_s_block, 1.0, 1.0, 300, 384, 0.2, 10, 0, 'randaug')),
                    ('m', (v2_m_block, 1.0, 1.0, 384, 480, 0.3, 15, 0.2, 'randaug')),
                    ('l', (v2_l_block, 1.0, 1.0, 384, 480, 0.4, 20, 0.5, 'randaug')),
                    ('xl', (v2_xl_block, 1.0, 1.0, 384, 512, 0.4, 20, 0.5, 'randaug')),
                    ('b0', (v2_base_block, 1.0, 1.0, 192, 224, 0.2, 0, 0, 'effnetv1_autoaug')),
                    ('b1', (v2_base_block, 1.0, 1.1, 192, 240, 0.2, 0, 0, 'effnetv1_autoaug')),
                    ('b2', (v2_base_block, 1.1, 1.2, 208, 260, 0.3, 0, 0, 'effnetv1_autoaug')),
                    ('b3', (v2_base_block, 1.2, 1.4, 240, 300, 0.3, 0, 0, 'effnetv1_autoaug'))):
  efficientnetv2_params['efficientnetv2-' + _name] = _row


def efficientnetv2_config(model_name='efficientnetv2-s'):
  block, width, depth, train_size, eval_size, dropout, randaug, mix, aug = efficientnetv2_params[model_name]
  return Config(dict(
      model=dict(model_name=model_name, blocks_args=BlockDecoder().decode(block),
                 width_coefficient=width, depth_coefficient=depth, dropout_rate=dropout),
      train=dict(isize=train_size, stages=4, sched=True),
      eval=dict(isize=eval_size),
      data=dict(augname=aug, ram=randaug, mixup_alpha=mix, cutmix_alpha=mix),
  ))


def get_model_config(model_name):
  """Main entry for model name to config (effnetv2_configs.py:214-241)."""
  if model_name.startswith('efficientnet-'):
    return efficientnetv1_config(model_name)
  if model_name.startswith('efficientnetv2-'):
    return efficientnetv2_config(model_name)
  raise ValueError('Unknown model_name {}'.format(model_name))


# hparams.base_config.model (efficientnetv2/hparams.py:221-243): defaults every model config overrides
base_model_config = dict(
    model_name='efficientnet-b0',
    data_format='channels_last',
    feature_size=1280,
    bn_type=None,
    bn_momentum=0.9,
    bn_epsilon=1e-3,
    gn_groups=8,
    depth_divisor=8,
    min_depth=8,
    act_fn='silu',
    survival_prob=0.8,
    local_pooling=False,
    headbias=None,
    conv_dropout=None,
    dropout_rate=None,
    depth_coefficient=None,
    width_coefficient=None,
    blocks_args=None,
    num_classes=1000,
)


def model_config(model_name, overrides=None):
  """base_config.model overridden by the named model and then by `overrides` (dict or 'k=v' string);
  what EffNetV2Model.__init__ computes (effnetv2_model.py:549-556)."""
  cfg = Config(copy.deepcopy(base_model_config))
  if model_name:
    cfg.override(get_model_config(model_name).model.as_dict())
  if overrides:
    cfg.override(overrides)
  return cfg


def round_filters(filters, mconfig, skip=False):
  """effnetv2_model.py:84-95 (no 10 % floor, unlike the V1 code base)."""
  multiplier = mconfig.width_coefficient
  divisor = mconfig.depth_divisor
  min_depth = mconfig.min_depth
  if skip or not multiplier:
    return filters
  filters *= multiplier
  min_depth = min_depth or divisor
  new_filters = max(min_depth, int(filters + divisor / 2) // divisor * divisor)
  return int(new_filters)


def round_repeats(repeats, multiplier, skip=False):
  """effnetv2_model.py:98-102."""
  if skip or not multiplier:
    return repeats
  return int(math.ceil(multiplier * repeats))


V2BlockSpec = collections.namedtuple('V2BlockSpec', [
    'index', 'conv_type', 'kernel_size', 'stride', 'input_filters', 'output_filters',
    'expand_ratio', 'se_filters', 'has_residual'])


def expand_blocks(mconfig):
  """(stem_filters, [V2BlockSpec]) -- the block list EffNetV2Model._build creates
  (effnetv2_model.py:558-590): the first block of a stage carries stride and filter change, the
  repeats run at stride 1 with input_filters = output_filters; se_filters =
  max(1, int(input_filters * se_ratio)) (:264-266,349-351)."""
  stages = mconfig.blocks_args
  stem = round_filters(stages[0].input_filters, mconfig)
  blocks = []
  for st in stages:
    assert st.num_repeat > 0
    cin = round_filters(st.input_filters, mconfig)
    cout = round_filters(st.output_filters, mconfig)
    reps = round_repeats(st.num_repeat, mconfig.depth_coefficient)
    for r in range(reps):
      b_in = cin if r == 0 else cout
      stride = st.strides if r == 0 else 1
      se = None
      if st.se_ratio is not None and 0 < st.se_ratio <= 1:
        se = max(1, int(b_in * st.se_ratio))
      blocks.append(V2BlockSpec(
          index=len(blocks), conv_type=st.conv_type, kernel_size=st.kernel_size, stride=stride,
          input_filters=b_in, output_filters=cout, expand_ratio=st.expand_ratio, se_filters=se,
          has_residual=bool(stride == 1 and b_in == cout)))
  return stem, blocks
