"""Anchor grid and box decoding (reference efficientdet/tf2/anchors.py:30-58,117-165).

The grid is generated in float64 numpy and cast to float32, exactly like the
reference; ordering is level-major, then y, x, then a = octave*len(aspects)+aspect.
"""
import numpy as np

from automl_amd import utils

MAX_DETECTION_POINTS = 5000


class Anchors(object):
  """Multi-scale anchors."""

  def __init__(self, min_level, max_level, num_scales, aspect_ratios,
               anchor_scale, image_size):
    self.min_level = min_level
    self.max_level = max_level
    self.num_scales = num_scales
    self.aspect_ratios = aspect_ratios
    if isinstance(anchor_scale, (list, tuple)):
      assert len(anchor_scale) == max_level - min_level + 1
      self.anchor_scales = list(anchor_scale)
    else:
      self.anchor_scales = [anchor_scale] * (max_level - min_level + 1)
    self.image_size = utils.parse_image_size(image_size)
    self.feat_sizes = utils.get_feat_sizes(image_size, max_level)
    self.config = self._generate_configs()
    self.boxes = self._generate_boxes()

  def _generate_configs(self):
    cfg = {}
    fs = self.feat_sizes
    for level in range(self.min_level, self.max_level + 1):
      cfg[level] = []
      for octave in range(self.num_scales):
        for aspect in self.aspect_ratios:
          cfg[level].append(
              ((fs[0]['height'] / float(fs[level]['height']),
                fs[0]['width'] / float(fs[level]['width'])),
               octave / float(self.num_scales), aspect,
               self.anchor_scales[level - self.min_level]))
    return cfg

  def _generate_boxes(self):
    all_levels = []
    for _, configs in self.config.items():
      per_anchor = []
      for stride, octave_scale, aspect, anchor_scale in configs:
        base_x = anchor_scale * stride[1] * 2**octave_scale
        base_y = anchor_scale * stride[0] * 2**octave_scale
        if isinstance(aspect, (list, tuple)):
          ax, ay = aspect
        else:
          ax = np.sqrt(aspect)
          ay = 1.0 / ax
        half_x = base_x * ax / 2.0
        half_y = base_y * ay / 2.0
        x = np.arange(stride[1] / 2, self.image_size[1], stride[1])
        y = np.arange(stride[0] / 2, self.image_size[0], stride[0])
        xv, yv = np.meshgrid(x, y)
        xv = xv.reshape(-1)
        yv = yv.reshape(-1)
        boxes = np.stack([yv - half_y, xv - half_x, yv + half_y, xv + half_x], axis=1)
        per_anchor.append(boxes[:, None, :])
      lvl = np.concatenate(per_anchor, axis=1)  # [H*W, A, 4]
      all_levels.append(lvl.reshape(-1, 4))
    return np.vstack(all_levels).astype(np.float32)

  def get_anchors_per_location(self):
    return self.num_scales * len(self.aspect_ratios)


def decode_box_outputs(pred_boxes, anchor_boxes):
  """(ty, tx, th, tw) relative codes -> absolute (ymin, xmin, ymax, xmax); numpy."""
  anchor_boxes = np.asarray(anchor_boxes, dtype=pred_boxes.dtype)
  yc_a = (anchor_boxes[..., 0] + anchor_boxes[..., 2]) / 2
  xc_a = (anchor_boxes[..., 1] + anchor_boxes[..., 3]) / 2
  ha = anchor_boxes[..., 2] - anchor_boxes[..., 0]
  wa = anchor_boxes[..., 3] - anchor_boxes[..., 1]
  ty, tx, th, tw = (pred_boxes[..., i] for i in range(4))
  w = np.exp(tw) * wa
  h = np.exp(th) * ha
  yc = ty * ha + yc_a
  xc = tx * wa + xc_a
  return np.stack([yc - h / 2., xc - w / 2., yc + h / 2., xc + w / 2.], axis=-1)


def merge_class_box_level_outputs(num_classes, cls_outputs, box_outputs):
  """Per-level [B,H,W,A*C] / [B,H,W,A*4] -> [B,N,C] / [B,N,4] in anchor order
  (reference tf2/postprocess.py:67-79); numpy or torch tensors."""
  cls_all = [c.reshape(c.shape[0], -1, num_classes) for c in cls_outputs]
  box_all = [b.reshape(b.shape[0], -1, 4) for b in box_outputs]
  if hasattr(cls_all[0], 'numpy') and not isinstance(cls_all[0], np.ndarray):
    import torch
    return torch.cat(cls_all, 1), torch.cat(box_all, 1)
  return np.concatenate(cls_all, 1), np.concatenate(box_all, 1)
