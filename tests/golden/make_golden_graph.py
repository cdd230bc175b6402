"""Generates tests/golden/reference_graph_d0.npz by EXECUTING the reference's own network code:
efficientdet/tf2/efficientdet_keras.EfficientDetNet (backbone efficientnet_model.Model, ResampleFeatureMap, FNode /
FPNCells, ClassNet, BoxNet) runs unmodified on top of tests/golden/mini_keras.py.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden_graph.py
Stored: the input image batch, the class / box outputs of every level in both BatchNorm modes, and the sorted list
of variable names the reference graph created (with shapes).  Variable VALUES are a function of the variable name
(mini_keras.value_for), so the consumer rebuilds them from the names.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))      # repo root: the oracle's TF-semantics helpers
import mini_keras   # noqa
from make_golden_anchors import REF   # noqa


def run(model_name, override, size, batch, seed, out_name):
  tf = mini_keras.build_tf()
  mini_keras.install(tf)
  sys.path.insert(0, REF)
  import hparams_config as ref_hp                 # noqa: the reference modules
  from tf2 import efficientdet_keras as ref_keras   # noqa
  config = ref_hp.get_efficientdet_config(model_name)
  config.override(override)
  mini_keras.VARIABLES.clear()
  net = ref_keras.EfficientDetNet(config=config)
  rng = np.random.default_rng(seed)
  images = rng.standard_normal((batch, size, size, 3)).astype(np.float32)
  out = {'images': images}
  for training in (False, True):
    del mini_keras.DRAWS[:]
    with torch.no_grad():
      cls, box = net(torch.from_numpy(images), training=training)
    for i, (c, b) in enumerate(zip(cls, box)):
      out['cls_%d_%d' % (training, i)] = c.numpy()
      out['box_%d_%d' % (training, i)] = b.numpy()
  # stochastic depth: the uniform draws of the training pass, one row per utils.drop_connect call in call order
  out['drop_draws'] = np.stack(mini_keras.DRAWS) if mini_keras.DRAWS else np.zeros((0, batch), np.float32)
  names = sorted(mini_keras.VARIABLES)
  out['var_names'] = np.array(names)
  out['var_shapes'] = np.array([','.join(map(str, mini_keras.VARIABLES[n].shape)) for n in names])
  np.savez_compressed(os.path.join(HERE, out_name), **out)
  print(out_name, len(names), 'variables;', [tuple(v.shape) for k, v in out.items() if k.startswith('cls_0')])


CASES = [   # (model, override, image size, batch, seed, file)
    ('efficientdet-d0', 'image_size=64', 64, 2, 5, 'reference_graph_d0.npz'),
    ('efficientdet-d1', 'image_size=64', 64, 2, 6, 'reference_graph_d1.npz'),
    ('efficientdet-d0', 'image_size=128,max_level=8,fpn_weight_method=sum', 128, 1, 7, 'reference_graph_d0_l8sum.npz'),
    ('efficientdet-d0', 'image_size=64,act_type=hswish', 64, 2, 8, 'reference_graph_d0_hswish.npz'),
]

if __name__ == '__main__':
  only = sys.argv[1:]
  for case in CASES:
    if not only or case[5] in only:
      run(*case)
