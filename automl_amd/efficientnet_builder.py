"""EfficientNet-B0..B7 backbone specification (host side, integers only).

Restates the block tables and rounding rules of the reference
(``efficientdet/backbone/efficientnet_builder.py:31-46,49-128,163-168`` and
``efficientnet_model.py:128-150,639-708``): width/depth coefficients, the
``r1_k3_s11_e1_i32_o16_se0.25`` block strings, filter rounding to multiples of
8 with the 10 % floor, ``ceil`` repeat rounding and the per-stage expansion in
which only the first block of a stage carries the stride / filter change.
"""
import collections
import math
import re

BlockArgs = collections.namedtuple('BlockArgs', [
    'kernel_size', 'num_repeat', 'input_filters', 'output_filters',
    'expand_ratio', 'id_skip', 'strides', 'se_ratio'])

# (width_coefficient, depth_coefficient, resolution, dropout_rate)
_PARAMS = {
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

_DEFAULT_BLOCKS_ARGS = [
    'r1_k3_s11_e1_i32_o16_se0.25', 'r2_k3_s22_e6_i16_o24_se0.25',
    'r2_k5_s22_e6_i24_o40_se0.25', 'r3_k3_s22_e6_i40_o80_se0.25',
    'r3_k5_s11_e6_i80_o112_se0.25', 'r4_k5_s22_e6_i112_o192_se0.25',
    'r1_k3_s11_e6_i192_o320_se0.25',
]

BN_MOMENTUM = 0.99
BN_EPSILON = 1e-3
DEPTH_DIVISOR = 8
STEM_FILTERS = 32


def efficientnet_params(model_name):
  return _PARAMS[model_name]


def decode_block_string(s):
  """'r2_k5_s22_e6_i24_o40_se0.25' -> BlockArgs."""
  opts = {}
  for op in s.split('_'):
    m = re.match(r'([a-z]+)(\d.*)', op)
    if m:
      opts[m.group(1)] = m.group(2)
  if 's' not in opts or len(opts['s']) != 2:
    raise ValueError('Strides options should be a pair of integers.')
  for unsupported in ('c', 'f', 'p'):
    if unsupported in opts and int(opts[unsupported]) != 0:
      raise ValueError('conv_type / fused_conv / super_pixel blocks are out of scope')
  if 'cc' in s.split('_'):
    raise ValueError('Condconv is not supported.')
  return BlockArgs(
      kernel_size=int(opts['k']), num_repeat=int(opts['r']),
      input_filters=int(opts['i']), output_filters=int(opts['o']),
      expand_ratio=int(opts['e']), id_skip=('noskip' not in s),
      strides=(int(opts['s'][0]), int(opts['s'][1])),
      se_ratio=float(opts['se']) if 'se' in opts else None)


def encode_block_string(b):
  args = ['r%d' % b.num_repeat, 'k%d' % b.kernel_size,
          's%d%d' % (b.strides[0], b.strides[1]), 'e%s' % b.expand_ratio,
          'i%d' % b.input_filters, 'o%d' % b.output_filters]
  if b.se_ratio is not None and 0 < b.se_ratio <= 1:
    args.append('se%s' % b.se_ratio)
  if b.id_skip is False:
    args.append('noskip')
  return '_'.join(args)


def round_filters(filters, width_coefficient, divisor=DEPTH_DIVISOR, min_depth=None):
  """Scale and round to a multiple of `divisor`, never dropping more than 10 %."""
  if not width_coefficient:
    return filters
  filters *= width_coefficient
  min_depth = min_depth or divisor
  new_filters = max(min_depth, int(filters + divisor / 2) // divisor * divisor)
  if new_filters < 0.9 * filters:
    new_filters += divisor
  return int(new_filters)


def round_repeats(repeats, depth_coefficient):
  if not depth_coefficient:
    return repeats
  return int(math.ceil(depth_coefficient * repeats))


BlockSpec = collections.namedtuple('BlockSpec', [
    'index', 'kernel_size', 'stride', 'input_filters', 'output_filters',
    'expand_ratio', 'se_filters', 'has_residual'])


def backbone_blocks(model_name, blocks_args=None):
  """Expanded list of MBConv blocks for `model_name` -> (stem_filters, [BlockSpec]).

  Only the first block of a stage carries stride and filter change
  (efficientnet_model.py:650-702).  se_filters = max(1, int(Cin_block * 0.25))
  (efficientnet_model.py:329-331).
  """
  width, depth, _, _ = efficientnet_params(model_name)
  strings = blocks_args or _DEFAULT_BLOCKS_ARGS
  stages = [decode_block_string(s) for s in strings]
  stem = round_filters(STEM_FILTERS, width)
  blocks = []
  for st in stages:
    cin = round_filters(st.input_filters, width)
    cout = round_filters(st.output_filters, width)
    reps = round_repeats(st.num_repeat, depth)
    if st.strides[0] != st.strides[1]:
      raise ValueError('non-square strides are not supported')
    for r in range(reps):
      b_in = cin if r == 0 else cout
      stride = st.strides[0] if r == 0 else 1
      se = None
      if st.se_ratio is not None and 0 < st.se_ratio <= 1:
        se = max(1, int(b_in * st.se_ratio))
      blocks.append(BlockSpec(
          index=len(blocks), kernel_size=st.kernel_size, stride=stride,
          input_filters=b_in, output_filters=cout,
          expand_ratio=st.expand_ratio, se_filters=se,
          has_residual=bool(st.id_skip and stride == 1 and b_in == cout)))
  return stem, blocks


def reduction_indices(blocks):
  """Block indices whose outputs are reduction_1..5 (efficientnet_model.py:741-765):
  the last block before each stride-2 block, plus the final block."""
  out = []
  for i in range(len(blocks)):
    if i == len(blocks) - 1 or blocks[i + 1].stride > 1:
      out.append(i)
  return out
