"""Generates tests/golden/reference_preprocess.npz by EXECUTING the reference's input-pipeline classes:
dataloader.InputProcessor / DetectionInputProcessor (normalize_image, set_scale_factors_to_output_size,
set_training_random_scale_factors, resize_and_crop_image, random_horizontal_flip, clip_boxes, resize_and_crop_boxes)
with object_detection/preprocessor.py and box_list.py -- unmodified -- on the torch-backed `tf` stand-in.
tf.image.resize is torch.nn.functional.interpolate(mode='bilinear', align_corners=False, antialias=False) (the same
half-pixel sampling rule as TF2's bilinear resize, implemented independently of oracle/preprocess_oracle.py);
tf.random.uniform / tf.random_uniform return values queued by this script, which are stored with the outputs.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden_preprocess.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import mini_keras   # noqa
from mini_keras import T, ns   # noqa
from make_golden_anchors import REF   # noqa
from make_golden_labels import add_tensor_ops   # noqa

DRAWS = []     # values handed out by tf.random.uniform, in call order


def add_image_ops(tf):
  def resize(image, size, method=None, **kw):
    x = T(image).permute(2, 0, 1)[None]
    y = torch.nn.functional.interpolate(x, size=(int(size[0]), int(size[1])), mode='bilinear', align_corners=False,
                                        antialias=False)
    return T(y[0].permute(1, 2, 0).contiguous())

  def pad_to_bounding_box(image, oy, ox, th, tw):
    image = T(image)
    out = torch.zeros((int(th), int(tw), image.shape[2]), dtype=image.dtype)
    out[int(oy):int(oy) + image.shape[0], int(ox):int(ox) + image.shape[1]] = image
    return T(out)

  def uniform(shape=(), minval=0.0, maxval=1.0, seed=None, **kw):
    u = DRAWS.pop(0)
    return T(torch.tensor(np.float32(minval) + np.float32(u) * (np.float32(maxval) - np.float32(minval))))

  tf.image = ns('image', resize=resize, pad_to_bounding_box=pad_to_bounding_box,
                flip_left_right=lambda im: T(im).flip(1),
                ResizeMethod=ns('ResizeMethod', BILINEAR='bilinear'))
  tf.random = ns('random', uniform=uniform)
  tf.random_uniform = uniform
  tf.subtract = lambda a, b: torch.as_tensor(a) - b
  tf.clip_by_value = lambda x, lo, hi: torch.clamp(T(x), float(lo), float(hi))
  tf.gather_nd = lambda params, indices: T(params)[T(indices).long()[:, 0]]
  base_constant = tf.constant
  tf.constant = lambda v, dtype=None, shape=None: (base_constant(v, dtype).reshape(shape) if shape is not None
                                                   else base_constant(v, dtype))
  tf.cast = lambda x, dtype=None: (torch.as_tensor(x).to(dtype) if isinstance(dtype, torch.dtype) else T(x))


MEAN = [0.485 * 255, 0.456 * 255, 0.406 * 255]
STD = [0.229 * 255, 0.224 * 255, 0.225 * 255]
CASES = {   # name: (raw height, raw width, output size, target size, boxes, seed, draws (flip, scale, u_y, u_x) or None)
    'infer_wide': (97, 150, (128, 128), None, 5, 1, None),
    'infer_tall': (211, 120, (128, 160), None, 4, 2, None),
    'train_up_noflip': (120, 90, (128, 128), None, 6, 3, (0.3, 0.85, 0.4, 0.7)),
    'train_down_flip': (200, 260, (128, 128), None, 8, 4, (0.9, 0.05, 0.5, 0.5)),
    'train_crop_flip': (150, 150, (96, 128), (128, 128), 7, 5, (0.7, 0.99, 0.35, 0.8)),
}
JITTER = (0.1, 2.0)


def main():
  tf = mini_keras.build_tf()
  add_tensor_ops(tf)
  add_image_ops(tf)
  mini_keras.install(tf)
  sys.modules.pop('dataloader', None)          # the real module, not the import stub
  sys.path.insert(0, REF)
  import dataloader as ref_dl     # noqa: the reference module
  out = {}
  for name, (h, w, osize, tsize, nbox, seed, draws) in CASES.items():
    rng = np.random.default_rng(seed)
    raw = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    ctr = rng.uniform(0.1, 0.9, (nbox, 2))
    hw = rng.uniform(0.05, 0.5, (nbox, 2))
    boxes = np.clip(np.concatenate([ctr - hw / 2, ctr + hw / 2], 1), 0, 1).astype(np.float32)
    boxes[0] = [0.2, 0.2, 0.2, 0.6]          # zero height: must be filtered out
    classes = rng.integers(1, 91, (nbox, 1)).astype(np.float32)
    p = ref_dl.DetectionInputProcessor(T(torch.from_numpy(raw)), osize, T(torch.from_numpy(boxes)),
                                       T(torch.from_numpy(classes)))
    p.normalize_image(MEAN, STD)
    if draws is not None:
      flip_u, scale_u, uy, ux = draws
      DRAWS[:] = [flip_u]
      p.random_horizontal_flip()
      DRAWS[:] = [scale_u, uy, ux]
      p.set_training_random_scale_factors(JITTER[0], JITTER[1], tsize)
      out[name + '/draws'] = np.asarray(draws, np.float32)
    else:
      p.set_scale_factors_to_output_size()
    image = p.resize_and_crop_image()
    b, c = p.resize_and_crop_boxes()
    out[name + '/raw'], out[name + '/boxes_in'], out[name + '/classes_in'] = raw, boxes, classes
    out[name + '/image'] = image.numpy().astype(np.float32)
    out[name + '/boxes'], out[name + '/classes'] = b.numpy().astype(np.float32), c.numpy().astype(np.float32)
    out[name + '/image_scale'] = np.float32(float(p.image_scale))
    out[name + '/scaled'] = np.asarray([int(p._scaled_height), int(p._scaled_width), int(p._crop_offset_y),
                                        int(p._crop_offset_x)], np.int32)
    print(name, out[name + '/scaled'], float(p.image_scale), image.shape, b.shape)
  np.savez_compressed(os.path.join(HERE, 'reference_preprocess.npz'), **out)


if __name__ == '__main__':
  main()
