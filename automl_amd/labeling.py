"""Anchor labelling on the GPU: the call surface of ``tf2/anchors.AnchorLabeler`` (tf2/anchors.py:170-250).

``AnchorLabeler(anchors, num_classes, match_threshold).label_anchors(gt_boxes [N,4], gt_labels [N,1])`` returns
``(cls_targets_dict, box_targets_dict, num_positives)`` keyed by level, as the reference does per image;
``label_anchors_batch`` does a padded batch in one call (the layout the train step consumes: dataloader.py:365-394).
Everything runs in the HIP library (edet_label_anchors); there is no CPU path.
"""
import collections
import ctypes

import numpy as np
import torch

from automl_amd import _lib


class AnchorLabeler(object):
  """Labeler for multiscale anchor boxes."""

  def __init__(self, anchors, num_classes, match_threshold=0.5, device='cuda:0'):
    self._anchors = anchors
    self._num_classes = num_classes
    self._match_threshold = match_threshold
    self._device = torch.device(device)
    self._boxes = torch.as_tensor(np.asarray(anchors.boxes, np.float32)).to(self._device).contiguous()
    a = anchors.get_anchors_per_location()
    self._levels = list(range(anchors.min_level, anchors.max_level + 1))
    self._hw = [(anchors.feat_sizes[l]['height'], anchors.feat_sizes[l]['width']) for l in self._levels]
    self._lanch = [h * w * a for h, w in self._hw]
    self._a = a

  def label_anchors_batch(self, gt_boxes, gt_labels, gt_count):
    """gt_boxes [B,M,4] float32, gt_labels [B,M] int (1-based), gt_count [B] -> (cls {level: [B,H,W,A] int32},
    box {level: [B,H,W,4A] float32}, num_positives [B] float32)."""
    dev = self._device
    gt_boxes = torch.as_tensor(gt_boxes, dtype=torch.float32).to(dev).contiguous()
    gt_labels = torch.as_tensor(gt_labels).to(dev).to(torch.int32).contiguous()
    gt_count = torch.as_tensor(gt_count).to(dev).to(torch.int32).contiguous()
    b, m = gt_boxes.shape[0], gt_boxes.shape[1]
    if m == 0:        # no image has an object: a single dummy row that gt_count masks out
      gt_boxes = torch.zeros((b, 1, 4), dtype=torch.float32, device=dev)
      gt_labels = torch.zeros((b, 1), dtype=torch.int32, device=dev)
      m = 1
    n = self._boxes.shape[0]
    cls = [torch.empty((b, h, w, self._a), dtype=torch.int32, device=dev) for h, w in self._hw]
    box = [torch.empty((b, h, w, self._a * 4), dtype=torch.float32, device=dev) for h, w in self._hw]
    npos = torch.empty((b,), dtype=torch.float32, device=dev)
    need = ctypes.c_size_t(0)
    _lib.call('edet_label_anchors_workspace_bytes', b, n, ctypes.byref(need))
    ws = torch.empty((need.value,), dtype=torch.uint8, device=dev)
    nlev = len(self._levels)
    la = (ctypes.c_int * nlev)(*self._lanch)
    cp = (ctypes.c_void_p * nlev)(*[c.data_ptr() for c in cls])
    bp = (ctypes.c_void_p * nlev)(*[x.data_ptr() for x in box])
    _lib.call('edet_label_anchors', self._boxes.data_ptr(), la, nlev, gt_boxes.data_ptr(), gt_labels.data_ptr(),
              gt_count.data_ptr(), b, m, float(self._match_threshold), ws.data_ptr(), need.value, cp, bp,
              npos.data_ptr(), torch.cuda.current_stream().cuda_stream)
    return (collections.OrderedDict(zip(self._levels, cls)), collections.OrderedDict(zip(self._levels, box)), npos)

  def label_anchors(self, gt_boxes, gt_labels):
    """One image, as the reference: gt_boxes [N,4], gt_labels [N,1] -> per-level [H,W,A] / [H,W,4A], scalar."""
    gt_boxes = torch.as_tensor(gt_boxes, dtype=torch.float32).reshape(1, -1, 4)
    gt_labels = torch.as_tensor(gt_labels).reshape(1, -1)
    cls, box, npos = self.label_anchors_batch(gt_boxes, gt_labels, [gt_boxes.shape[1]])
    return (collections.OrderedDict((k, v[0]) for k, v in cls.items()),
            collections.OrderedDict((k, v[0]) for k, v in box.items()), npos[0])
