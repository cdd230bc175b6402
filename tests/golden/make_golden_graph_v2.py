"""Generates tests/golden/reference_graph_v2_*.npz by EXECUTING the reference's own EfficientNetV2 model code:
efficientnetv2/effnetv2_model.EffNetV2Model (Stem, MBConvBlock, FusedMBConvBlock, SE, Head, utils.drop_connect) runs
unmodified on top of tests/golden/mini_keras.py (see make_golden_graph.py for the EfficientDet counterpart).

Run in the build container only (needs /root/reference):  python tests/golden/make_golden_graph_v2.py
Stored: images, logits / pooled features / reduction endpoints in both BatchNorm modes, the drop_connect draws of the
training pass (call order) and the variable inventory; variable values are a function of the variable name
(name_values.value_for).  dropout_rate is overridden to 0: head dropout is a random mask of the reference's RNG.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import mini_keras   # noqa

REF_V2 = '/root/reference/efficientnetv2'
ENDPOINTS = ['reduction_1', 'reduction_2', 'reduction_3', 'reduction_4', 'reduction_5', 'pooled_features']


def run(model_name, size, batch, seed, out_name):
  tf = mini_keras.build_tf()
  mini_keras.install(tf)
  sys.path.insert(0, REF_V2)
  import effnetv2_model as ref_model     # noqa: the reference module
  mini_keras.VARIABLES.clear()
  net = ref_model.EffNetV2Model(model_name, {'dropout_rate': 0.0})
  rng = np.random.default_rng(seed)
  images = rng.standard_normal((batch, size, size, 3)).astype(np.float32)
  out = {'images': images}
  for training in (False, True):
    del mini_keras.DRAWS[:]
    with torch.no_grad():
      logits = net(torch.from_numpy(images), training=training)
    out['logits_%d' % training] = logits.numpy()
    for e in ENDPOINTS:
      out['%s_%d' % (e, training)] = net.endpoints[e].numpy()
  out['drop_draws'] = np.stack(mini_keras.DRAWS) if mini_keras.DRAWS else np.zeros((0, batch), np.float32)
  names = sorted(mini_keras.VARIABLES)
  out['var_names'] = np.array(names)
  out['var_shapes'] = np.array([','.join(map(str, mini_keras.VARIABLES[n].shape)) for n in names])
  np.savez_compressed(os.path.join(HERE, out_name), **out)
  print(out_name, len(names), 'variables; logits', out['logits_0'].shape, 'draw rows', len(out['drop_draws']))


if __name__ == '__main__':
  run(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
