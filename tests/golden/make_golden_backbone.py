"""Generates tests/golden/reference_backbones.json by executing the REFERENCE's own EfficientNet builder code.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden_backbone.py
backbone/efficientnet_builder.py (efficientnet_params, BlockDecoder, efficientnet(), get_model_params :31-216) and
backbone/efficientnet_model.py (round_filters / round_repeats :128-150) are plain Python apart from their
module-scope TensorFlow imports, which the permissive stub of make_golden_anchors.py satisfies.  Stored per
backbone: stem filters and, per stage, the rounded (input_filters, output_filters, num_repeat) next to the
decoded block arguments -- the numbers Model._build expands into blocks (efficientnet_model.py:650-702).
"""
import json
import os
import sys

import numpy as np

from make_golden_anchors import REF, stub_module   # noqa


def main():
  tf = stub_module('tensorflow')
  tf.float32 = np.float32
  for name in ('tensorflow', 'tensorflow.compat', 'tensorflow.compat.v1', 'tensorflow.compat.v2', 'absl',
               'absl.logging', 'tensorflow.python', 'tensorflow.python.eager', 'tensorflow.python.tpu',
               'tensorflow.python.eager.tape', 'tensorflow.python.tpu.tpu_function', 'tensorflow_addons',
               'tensorflow.python.framework', 'tensorflow.python.ops', 'six', 'six.moves'):
    sys.modules.setdefault(name, tf if name == 'tensorflow' else stub_module(name))
  sys.path.insert(0, REF)
  from backbone import efficientnet_builder as ref_b    # noqa: the reference modules
  from backbone import efficientnet_model as ref_m      # noqa
  out = {}
  for i in range(8):
    name = 'efficientnet-b%d' % i
    blocks_args, gp = ref_b.get_model_params(name, None)
    stages = []
    for b in blocks_args:
      stages.append({
          'kernel_size': b.kernel_size, 'strides': list(b.strides), 'expand_ratio': b.expand_ratio,
          'se_ratio': b.se_ratio, 'id_skip': b.id_skip,
          'input_filters': ref_m.round_filters(b.input_filters, gp),
          'output_filters': ref_m.round_filters(b.output_filters, gp),
          'num_repeat': ref_m.round_repeats(b.num_repeat, gp)})
    out[name] = {'stem_filters': ref_m.round_filters(32, gp), 'stages': stages,
                 'survival_prob': gp.survival_prob, 'bn_momentum': gp.batch_norm_momentum,
                 'bn_epsilon': gp.batch_norm_epsilon, 'depth_divisor': gp.depth_divisor,
                 'width': gp.width_coefficient, 'depth': gp.depth_coefficient}
  here = os.path.dirname(os.path.abspath(__file__))
  with open(os.path.join(here, 'reference_backbones.json'), 'w') as f:
    json.dump(out, f, indent=1, sort_keys=True)
  print(json.dumps(out['efficientnet-b7']['stages'][-1]), out['efficientnet-b7']['stem_filters'])


if __name__ == '__main__':
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  main()
