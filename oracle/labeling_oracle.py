"""CPU oracle of anchor labelling -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/ (and the golden generators under tests/golden/) may import this module.  Plain numpy (float32)
restatement of (paths relative to the reference root):

  efficientdet/tf2/anchors.py                                   AnchorLabeler.__init__ :173-197, _unpack_labels :199-213,
                                                                label_anchors :215-250
  efficientdet/object_detection/region_similarity_calculator.py area :25-39, intersection :42-65, iou :68-88
  efficientdet/object_detection/argmax_matcher.py               ArgMaxMatcher._match :101-184 (matched = unmatched
                                                                threshold, negatives_lower_than_unmatched, force_match)
  efficientdet/object_detection/matcher.py                      Match.gather_based_on_match :170-196
  efficientdet/object_detection/target_assigner.py              assign :80-166, _create_regression_targets :184-219,
                                                                _create_classification_targets :233-254
  efficientdet/object_detection/faster_rcnn_box_coder.py        _encode :59-89 (EPSILON 1e-8, no scale factors)
  efficientdet/object_detection/box_list.py                     get_center_coordinates_and_sizes :157-173

PARITY STATUS: pinned against the reference's own modules executed on the torch-backed `tf` stand-in
(tests/golden/make_golden_labels.py -> tests/golden/reference_labels.npz).  The element-wise arithmetic underneath is
numpy / torch float32, not the TensorFlow binary.
"""
import numpy as np

EPSILON = np.float32(1e-8)      # faster_rcnn_box_coder.py:35


def iou_matrix(gt, anchors):
  """region_similarity_calculator.iou: [M,4] x [N,4] (ymin, xmin, ymax, xmax) -> [M,N] float32."""
  gt = np.asarray(gt, np.float32).reshape(-1, 4)
  an = np.asarray(anchors, np.float32).reshape(-1, 4)
  ih = np.maximum(np.float32(0), np.minimum(gt[:, None, 2], an[None, :, 2]) - np.maximum(gt[:, None, 0], an[None, :, 0]))
  iw = np.maximum(np.float32(0), np.minimum(gt[:, None, 3], an[None, :, 3]) - np.maximum(gt[:, None, 1], an[None, :, 1]))
  inter = ih * iw
  a1 = (gt[:, 2] - gt[:, 0]) * (gt[:, 3] - gt[:, 1])
  a2 = (an[:, 2] - an[:, 0]) * (an[:, 3] - an[:, 1])
  union = a1[:, None] + a2[None, :] - inter
  with np.errstate(divide='ignore', invalid='ignore'):
    return np.where(inter == 0, np.float32(0), inter / union).astype(np.float32)


def argmax_match(sim, matched_threshold=0.5, unmatched_threshold=None, negatives_lower_than_unmatched=True,
                 force_match_for_each_row=True):
  """ArgMaxMatcher._match -> int32 [N]: row index, -1 unmatched, -2 ignored."""
  if unmatched_threshold is None:
    unmatched_threshold = matched_threshold
  m, n = sim.shape
  if m == 0:
    return -np.ones((n,), np.int32)
  matches = np.argmax(sim, 0).astype(np.int32)           # first maximum
  vals = sim.max(0)
  below = np.float32(unmatched_threshold) > vals
  between = (vals >= np.float32(unmatched_threshold)) & (np.float32(matched_threshold) > vals)
  lo, mid = (-1, -2) if negatives_lower_than_unmatched else (-2, -1)
  matches = np.where(below, lo, matches)
  matches = np.where(between, mid, matches).astype(np.int32)
  if force_match_for_each_row:
    cols = np.argmax(sim, 1)                              # per groundtruth box: its best anchor (first maximum)
    indicators = np.zeros((m, n), np.float32)
    indicators[np.arange(m), cols] = 1
    rows = np.argmax(indicators, 0).astype(np.int32)      # first row that claims the column
    mask = indicators.max(0) > 0
    matches = np.where(mask, rows, matches).astype(np.int32)
  return matches


def center_size(boxes):
  ymin, xmin, ymax, xmax = (boxes[:, i] for i in range(4))
  w, h = xmax - xmin, ymax - ymin
  return ymin + h / np.float32(2), xmin + w / np.float32(2), h, w


def encode(boxes, anchors):
  """FasterRcnnBoxCoder._encode -> [N,4] = ty, tx, th, tw."""
  yc_a, xc_a, ha, wa = center_size(np.asarray(anchors, np.float32))
  yc, xc, h, w = center_size(np.asarray(boxes, np.float32))
  ha, wa, h, w = (np.maximum(EPSILON, v) for v in (ha, wa, h, w))
  return np.stack([(yc - yc_a) / ha, (xc - xc_a) / wa, np.log(h / ha), np.log(w / wa)], 1).astype(np.float32)


def label_anchors_flat(anchor_boxes, gt_boxes, gt_labels, match_threshold=0.5):
  """-> (cls_targets [N] int32: class - 1, -1 background; box_targets [N,4] float32; num_positives float32;
  match_results [N] int32)."""
  an = np.asarray(anchor_boxes, np.float32)
  gt = np.asarray(gt_boxes, np.float32).reshape(-1, 4)
  labels = np.asarray(gt_labels, np.float32).reshape(-1, 1)
  match = argmax_match(iou_matrix(gt, an), match_threshold, match_threshold, True, True)
  idx = np.maximum(match + 2, 0)
  # gather_based_on_match: rows [ignored, unmatched] are prepended
  boxes_tab = np.concatenate([np.zeros((2, 4), np.float32), gt], 0)
  labels_tab = np.concatenate([np.zeros((2, 1), np.float32), labels], 0)
  matched_boxes = boxes_tab[idx]
  reg = np.where((match >= 0)[:, None], encode(matched_boxes, an), np.float32(0)).astype(np.float32)
  cls = (labels_tab[idx] - 1).astype(np.int32)[:, 0]
  return cls, reg, np.float32((match != -1).sum()), match


def unpack_labels(flat, feat_sizes, anchors_per_location):
  """AnchorLabeler._unpack_labels: [N(,4)] -> per level [H, W, A(*4)]."""
  out, count = [], 0
  for h, w in feat_sizes:
    steps = h * w * anchors_per_location
    out.append(np.asarray(flat[count:count + steps]).reshape(h, w, -1))
    count += steps
  return out


def label_anchors(anchor_boxes, feat_sizes, anchors_per_location, gt_boxes, gt_labels, match_threshold=0.5):
  """AnchorLabeler.label_anchors -> (cls_targets per level [H,W,A] int32, box_targets per level [H,W,4A],
  num_positives)."""
  cls, reg, npos, _ = label_anchors_flat(anchor_boxes, gt_boxes, gt_labels, match_threshold)
  return unpack_labels(cls, feat_sizes, anchors_per_location), unpack_labels(reg, feat_sizes, anchors_per_location), npos
