"""Generates tests/golden/reference_trainstep_d0.npz by EXECUTING the reference's own training step:
tf2/train_lib.EfficientDetNetTrain.train_step (:606-684) with _detection_loss (:493-604), _reg_l2_loss (:486-491),
FocalLoss.call (:357-406) and BoxLoss.call (:409-437), on the network of tf2/efficientdet_keras.py -- all unmodified,
running on tests/golden/mini_keras.py (torch-backed tf.keras stand-in; tf.GradientTape = torch.autograd).  The losses
are built as tf2/train.py:115-140 builds them (Reduction.NONE).  The optimizer is a recorder: the step's products are
the loss values and the clipped gradients the reference hands to optimizer.apply_gradients.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden_trainstep.py
Stored: images, labels, loss values, and per trainable variable the L2 norm of its clipped gradient plus its dot
product with a name-derived probe vector (name_values.value_for('probe/' + name)); small tensors (<= 1024 elements)
are stored whole.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import mini_keras   # noqa
from make_golden_anchors import REF   # noqa
from name_values import value_for, make_labels   # noqa


class RecordingOptimizer(object):
  learning_rate = 0.02
  iterations = 0

  def __init__(self):
    self.applied = []

  def apply_gradients(self, pairs):
    self.applied = [(v.name, g) for g, v in pairs]


def run(model_name, override, size, batch, seed, out_name):
  tf = mini_keras.build_tf()
  mini_keras.install(tf)
  sys.path.insert(0, REF)
  import hparams_config as ref_hp                 # noqa: the reference modules
  from tf2 import train_lib as ref_train          # noqa
  config = ref_hp.get_efficientdet_config(model_name)
  config.override(override)
  config.model_dir = '/tmp/unused'      # train.py sets it from the flags; only names a summary directory
  mini_keras.VARIABLES.clear()
  del mini_keras.TRAINABLE[:]
  mini_keras.GRAD[0] = True
  net = ref_train.EfficientDetNetTrain(config=config)
  Reduction = tf.keras.losses.Reduction
  net.loss = {     # tf2/train.py:115-140
      ref_train.BoxLoss.__name__: ref_train.BoxLoss(config.delta, reduction=Reduction.NONE),
      ref_train.FocalLoss.__name__: ref_train.FocalLoss(config.alpha, config.gamma,
                                                        label_smoothing=config.label_smoothing,
                                                        reduction=Reduction.NONE),
  }
  net.optimizer = RecordingOptimizer()
  rng = np.random.default_rng(seed)
  images = rng.standard_normal((batch, size, size, 3)).astype(np.float32)
  labels = make_labels(config.min_level, config.max_level, config.num_classes,
                       len(config.aspect_ratios) * config.num_scales, batch, size, seed + 1)
  tl = {k: torch.from_numpy(v) for k, v in labels.items()}
  vals = net.train_step((torch.from_numpy(images), tl))
  out = {'images': images}
  out.update({'label/' + k: v for k, v in labels.items()})
  for k in ('loss', 'det_loss', 'cls_loss', 'box_loss', 'reg_l2_loss', 'gradient_norm', 'learning_rate'):
    out['val/' + k] = np.float64(float(vals[k]))
  names, norms, dots = [], [], []
  for name, g in net.optimizer.applied:
    name = name[:-2]      # ':0'
    g = g.detach().numpy() if g is not None else np.zeros(tuple(mini_keras.VARIABLES[name].shape), np.float32)
    names.append(name)
    norms.append(float(np.sqrt((g.astype(np.float64)**2).sum())))
    dots.append(float((g.astype(np.float64) * value_for('probe/' + name, g.shape)).sum()))
    if g.size <= 1024:
      out['grad/' + name] = g
  out['grad_names'] = np.array(names)
  out['grad_norms'] = np.array(norms)
  out['grad_dots'] = np.array(dots)
  allv = sorted(mini_keras.VARIABLES)
  out['var_names'] = np.array(allv)
  out['var_shapes'] = np.array([','.join(map(str, mini_keras.VARIABLES[n].shape)) for n in allv])
  np.savez_compressed(os.path.join(HERE, out_name), **out)
  print(out_name, {k[4:]: float(v) for k, v in out.items() if k.startswith('val/')}, len(names), 'gradients')


if __name__ == '__main__':
  run('efficientdet-d0', 'image_size=192', 192, 4, 41, 'reference_trainstep_d0.npz')
