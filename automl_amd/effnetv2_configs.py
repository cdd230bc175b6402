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


class BlockDecoder(object):
  """'r2_k3_s1_e1_i24_o24_c1' <-> block Config (effnetv2_configs.py:22-89)."""

  def _decode_block_string(self, block_string):
    assert isinstance(block_string, str)
    options = {}
    for op in block_string.split('_'):
      splits = re.split(r'(\d.*)', op)
      if len(splits) >= 2:
        key, value = splits[:2]
        options[key] = value
    return Config(dict(
        kernel_size=int(options['k']),
        num_repeat=int(options['r']),
        input_filters=int(options['i']),
        output_filters=int(options['o']),
        expand_ratio=int(options['e']),
        se_ratio=float(options['se']) if 'se' in options else None,
        strides=int(options['s']),
        conv_type=int(options['c']) if 'c' in options else 0,
    ))

  def _encode_block_string(self, block):
    args = [
        'r%d' % block.num_repeat,
        'k%d' % block.kernel_size,
        's%d' % block.strides,
        'e%s' % block.expand_ratio,
        'i%d' % block.input_filters,
        'o%d' % block.output_filters,
        'c%d' % block.conv_type,
    ]
    if block.se_ratio is not None and 0 < block.se_ratio <= 1:
      args.append('se%s' % block.se_ratio)
    return '_'.join(args)

  def decode(self, string_list):
    assert isinstance(string_list, list)
    return [self._decode_block_string(s) for s in string_list]

  def encode(self, blocks_args):
    return [self._encode_block_string(b) for b in blocks_args]


#################### EfficientNet V1 configs (as served by the V2 codebase) ####################
v1_b0_block_str = [
    'r1_k3_s1_e1_i32_o16_se0.25',
    'r2_k3_s2_e6_i16_o24_se0.25',
    'r2_k5_s2_e6_i24_o40_se0.25',
    'r3_k3_s2_e6_i40_o80_se0.25',
    'r3_k5_s1_e6_i80_o112_se0.25',
    'r4_k5_s2_e6_i112_o192_se0.25',
    'r1_k3_s1_e6_i192_o320_se0.25',
]

efficientnetv1_params = {
    # (width_coefficient, depth_coefficient, resolution, dropout_rate)
    'efficientnet-b0': (1.0, 1.0, 224, 0.2),
    'efficientnet-b1': (1.0, 1.1, 240, 0.2),
    'efficientnet-b2': (1.1, 1.2, 260, 0.3),
    'efficientnet-b3': (1.2, 1.4, 300, 0.3),
    'efficientnet-b4': (1.4, 1.8, 380, 0.4),
    'efficientnet-b5': (1.6, 2.2, 456, 0.4),
    'efficientnet-b6': (1.8, 2.6, 528, 0.5),
    'efficientnet-b7': (2.0, 3.1, 600, 0.5),
    'efficientnet-b8': (2.2, 3.6, 672, 0.5),
    'efficientnet-l2': (4.3, 5.3, 800, 0.5),
}


def efficientnetv1_config(model_name='efficientnet-b0'):
  width, depth, isize, dropout = efficientnetv1_params[model_name]
  return Config(dict(
      model=dict(model_name=model_name, blocks_args=BlockDecoder().decode(v1_b0_block_str),
                 width_coefficient=width, depth_coefficient=depth, dropout_rate=dropout),
      eval=dict(isize=isize),
      train=dict(isize=0.8),
      data=dict(augname='effnetv1_autoaug'),
  ))


#################### EfficientNet V2 configs ####################
v2_base_block = [
    'r1_k3_s1_e1_i32_o16_c1',
    'r2_k3_s2_e4_i16_o32_c1',
    'r2_k3_s2_e4_i32_o48_c1',
    'r3_k3_s2_e4_i48_o96_se0.25',
    'r5_k3_s1_e6_i96_o112_se0.25',
    'r8_k3_s2_e6_i112_o192_se0.25',
]

v2_s_block = [
    'r2_k3_s1_e1_i24_o24_c1',
    'r4_k3_s2_e4_i24_o48_c1',
    'r4_k3_s2_e4_i48_o64_c1',
    'r6_k3_s2_e4_i64_o128_se0.25',
    'r9_k3_s1_e6_i128_o160_se0.25',
    'r15_k3_s2_e6_i160_o256_se0.25',
]

v2_m_block = [
    'r3_k3_s1_e1_i24_o24_c1',
    'r5_k3_s2_e4_i24_o48_c1',
    'r5_k3_s2_e4_i48_o80_c1',
    'r7_k3_s2_e4_i80_o160_se0.25',
    'r14_k3_s1_e6_i160_o176_se0.25',
    'r18_k3_s2_e6_i176_o304_se0.25',
    'r5_k3_s1_e6_i304_o512_se0.25',
]

v2_l_block = [
    'r4_k3_s1_e1_i32_o32_c1',
    'r7_k3_s2_e4_i32_o64_c1',
    'r7_k3_s2_e4_i64_o96_c1',
    'r10_k3_s2_e4_i96_o192_se0.25',
    'r19_k3_s1_e6_i192_o224_se0.25',
    'r25_k3_s2_e6_i224_o384_se0.25',
    'r7_k3_s1_e6_i384_o640_se0.25',
]

v2_xl_block = [
    'r4_k3_s1_e1_i32_o32_c1',
    'r8_k3_s2_e4_i32_o64_c1',
    'r8_k3_s2_e4_i64_o96_c1',
    'r16_k3_s2_e4_i96_o192_se0.25',
    'r24_k3_s1_e6_i192_o256_se0.25',
    'r32_k3_s2_e6_i256_o512_se0.25',
    'r8_k3_s1_e6_i512_o640_se0.25',
]

efficientnetv2_params = {
    # (block, width, depth, train_size, eval_size, dropout, randaug, mixup, aug)
    'efficientnetv2-s': (v2_s_block, 1.0, 1.0, 300, 384, 0.2, 10, 0, 'randaug'),
    'efficientnetv2-m': (v2_m_block, 1.0, 1.0, 384, 480, 0.3, 15, 0.2, 'randaug'),
    'efficientnetv2-l': (v2_l_block, 1.0, 1.0, 384, 480, 0.4, 20, 0.5, 'randaug'),
    'efficientnetv2-xl': (v2_xl_block, 1.0, 1.0, 384, 512, 0.4, 20, 0.5, 'randaug'),
    'efficientnetv2-b0': (v2_base_block, 1.0, 1.0, 192, 224, 0.2, 0, 0, 'effnetv1_autoaug'),
    'efficientnetv2-b1': (v2_base_block, 1.0, 1.1, 192, 240, 0.2, 0, 0, 'effnetv1_autoaug'),
    'efficientnetv2-b2': (v2_base_block, 1.1, 1.2, 208, 260, 0.3, 0, 0, 'effnetv1_autoaug'),
    'efficientnetv2-b3': (v2_base_block, 1.2, 1.4, 240, 300, 0.3, 0, 0, 'effnetv1_autoaug'),
}


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
